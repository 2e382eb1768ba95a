// chain.hip -- host driver and C ABI (include/bmm_mcmc.h) of the allocation path.
//
// One bmm_chain = one MCMC chain resident on one GPU: the data matrix, the label
// rows, the integer sufficient statistics, the table image and every trace live in
// HBM; the host only enqueues kernels on the chain's stream.  The sweep loop mirrors
// the reference's (collapsed_gibbs.cpp:84-225, collapsed_gibbs_dp.cpp:98-283,
// stickbreaking.cpp:66-236) with the per-observation loop replaced by batches.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <rccl/rccl.h>  // types and prototypes only: the library is opened with dlopen when a run spans devices

#include <dlfcn.h>
#include <emmintrin.h>
#include <sched.h>
#include <sys/mman.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <exception>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif

#include "../../include/bmm_mcmc.h"
#include "kernels.hip.h"
#include "host_crew.h"

using namespace bmm;
using bmm_host::HostCrew;
using bmm_host::host_threads;

namespace {

thread_local char g_err[512] = "";

// Kernel-steering environment hooks exist only in the test variant of the library
// (-DBMM_DEBUG_HOOKS, lib/libbmmmcmc_hip_dbg.so, built next to the product by the tests):
// the product library never reads the environment.
#ifdef BMM_DEBUG_HOOKS
const char* dbg_env(const char* name) { return getenv(name); }
#else
const char* dbg_env(const char*) { return nullptr; }
#endif

// Test variant only (BMM_DEBUG_FAKE_DEVICES=n): the library then accepts device indices 0 .. n-1, all of them the one
// real device underneath, keeps them apart wherever it reasons about which chains share a device (bmm_multi_run's
// placement, bmm_chain_share_data, bmm_chains_broadcast_planes), and replaces the RCCL broadcast between them by
// device-to-device copies.  What a one-GPU box can run of the multi-device bookkeeping -- every line of it but the
// collective itself, which bmm_multi_selfcheck runs on the device there is.  The product library has no such mode.
int fake_devices() {
    const char* v = dbg_env("BMM_DEBUG_FAKE_DEVICES");
    const int n = v ? atoi(v) : 0;
    return n > 1 ? n : 0;
}

int set_err(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return set_err(BMM_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),   \
                           __FILE__, __LINE__);                                                \
    } while (0)

// device scratch that is released on every return path
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

// No C++ exception may cross the C ABI: the entry points that allocate on the host or start threads run
// their bodies through this.
template <class F>
int guarded(F&& body) noexcept {
    try {
        return body();
    } catch (const std::bad_alloc&) {
        return set_err(BMM_E_ARG, "out of host memory");
    } catch (const std::exception& e) {
        return set_err(BMM_E_STATE, "unexpected host error: %s", e.what());
    } catch (...) {
        return set_err(BMM_E_STATE, "unexpected host error");
    }
}
// host threads that are always joined, also when starting a later one throws
struct ThreadGroup {
    std::vector<std::thread> th;
    ~ThreadGroup() { join(); }
    void join() { for (std::thread& t : th) if (t.joinable()) t.join(); }
};

// Observations [a, b) of the N x P int32 column-major matrix R hands over, as bit planes: word w of
// observation i at plane[w][i - a0] (feature d at bit d % 32 of word d / 32, bits past P zero -- what
// k_pack_bits writes on the device).  Eight columns at a time into a block of output words that stays in L1:
// every cell of X is read once, in long contiguous runs, eight independent streams in flight per thread (one
// stream alone leaves a core at 6.5 GB/s, eight at 10).  Returns the OR of all cells read: the caller rejects
// the matrix when that has a bit other than bit 0 (data must be 0/1, as k_validate_binary checks).
uint32_t pack_rows_host(const int32_t* X, int64_t N, int P, int64_t a, int64_t b, uint32_t* const* plane, int64_t a0) {
    constexpr int64_t BL = 4096;
    constexpr int NS = 8;
    const int W = (P + 31) / 32;
    uint32_t seen = 0;
    for (int64_t r = a; r < b; r += BL) {
        const int64_t n = b - r < BL ? b - r : BL;
        for (int w = 0; w < W; ++w) {
            uint32_t* __restrict const o = plane[w] + (r - a0);
            const int d0 = 32 * w, nd = P - d0 < 32 ? P - d0 : 32;
            const uint32_t* const col = reinterpret_cast<const uint32_t*>(X) + (int64_t)d0 * N + r;
            int j = 0;
            for (; j + NS <= nd; j += NS) {
                const uint32_t* __restrict c[NS];
                for (int q = 0; q < NS; ++q) c[q] = col + (int64_t)(j + q) * N;
                for (int64_t t = 0; t < n; ++t) {
                    uint32_t acc = 0, s = 0;
                    for (int q = 0; q < NS; ++q) { const uint32_t v = c[q][t]; s |= v; acc |= v << q; }
                    seen |= s;
                    o[t] = j == 0 ? acc : (o[t] | (acc << j));
                }
            }
            for (; j < nd; ++j) {
                const uint32_t* __restrict const c0 = col + (int64_t)j * N;
                for (int64_t t = 0; t < n; ++t) { const uint32_t v = c0[t]; seen |= v; o[t] = j == 0 ? v : (o[t] | (v << j)); }
            }
        }
    }
    return seen;
}

// Pinned staging in pieces of 4 MiB that outlive a call: pinning costs about a millisecond per piece, a run
// needs two on the way in and two on the way out, and an R session calls the samplers again and again.  At
// most eight idle pieces are kept; the pool itself is never destroyed (no HIP call at process exit).
constexpr size_t kStageBytes = (size_t)4 << 20;
struct StagePool {
    std::mutex m;
    std::vector<void*> idle;
    void* get() {
        {
            std::lock_guard<std::mutex> g(m);
            if (!idle.empty()) { void* p = idle.back(); idle.pop_back(); return p; }
        }
        void* p = nullptr;
        if (hipHostMalloc(&p, kStageBytes, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        return p;
    }
    void put(void* p) noexcept {  // called from destructors: a piece that cannot be kept is freed
        if (!p) return;
        try {
            std::lock_guard<std::mutex> g(m);
            if (idle.size() < 8) { idle.push_back(p); return; }
        } catch (...) {
        }
        (void)hipHostFree(p);
    }
};
StagePool& stage_pool() { static StagePool* const pool = new StagePool(); return *pool; }
struct Stage {  // one piece, back to the pool on every return path
    void* p = stage_pool().get();
    ~Stage() { stage_pool().put(p); }
    Stage() = default;
    Stage(const Stage&) = delete;
    Stage& operator=(const Stage&) = delete;
    template <class T> T* as() const { return static_cast<T*>(p); }
};

// All of a host matrix packed by the crew while the calling thread goes on (a *_run call creates its chain
// and uploads the starting state meanwhile).  The planes land in pinned pieces of the staging pool, one per
// plane, when a plane fits a piece and there are at most four (N <= 2^20, P <= 128: the upload then runs at
// PCIe speed, 0.15 instead of 0.6 ms for the 8 MB of the north-star shape); in ordinary host memory, [w][N],
// otherwise.
struct AsyncPack {
    std::unique_ptr<uint32_t[]> words;
    std::unique_ptr<Stage[]> pinned;
    std::vector<uint32_t*> plane;
    std::atomic<uint32_t> seen{0};
    HostCrew* crew = nullptr;
    void start(HostCrew* cr, const int32_t* X, int64_t N, int P) {
        const int W = (P + 31) / 32;
        crew = cr;
        plane.assign((size_t)W, nullptr);
        if (W <= 4 && (size_t)N * sizeof(uint32_t) <= kStageBytes) {
            pinned.reset(new Stage[(size_t)W]);
            bool ok = true;
            for (int w = 0; w < W; ++w) { plane[(size_t)w] = pinned[(size_t)w].as<uint32_t>(); ok = ok && plane[(size_t)w]; }
            if (!ok) pinned.reset();
        }
        if (!pinned) {
            words.reset(new uint32_t[(size_t)W * (size_t)N]);
            for (int w = 0; w < W; ++w) plane[(size_t)w] = words.get() + (size_t)w * (size_t)N;
        }
        uint32_t* const* const pl = plane.data();
        std::atomic<uint32_t>* const sn = &seen;
        crew->begin(N, 32768, 64, [X, N, P, pl, sn](int64_t lo, int64_t hi) {
            sn->fetch_or(pack_rows_host(X, N, P, lo, hi, pl, 0), std::memory_order_relaxed);
        });
    }
    void join() { if (crew) crew->wait(); }
    void release() { join(); words.reset(); pinned.reset(); plane.clear(); }
    ~AsyncPack() { join(); }
};

// Device memory of a finished call, kept for the next one: a chain's arena, a run's arena and the bit planes
// come from here.  hipMalloc is usually free on this runtime (0.02 ms) but now and then takes 10 ms and more
// when the driver has to map fresh memory (a 220-sweep call at N = 1e6 lasts 37 ms in all), and hipFree costs
// 0.2 ms a piece.  Per device at most six idle blocks and 512 MiB in all are kept; a request takes the smallest
// idle block that fits and is at most twice as large.  bmm_release_pools frees them.
struct DevPool {
    struct Block { void* p; size_t bytes; };
    std::mutex m;
    std::vector<Block> idle[64];
    static constexpr size_t kMaxIdleBytes = (size_t)512 << 20;
    void* get(int device, size_t bytes, size_t* got) {
        if (device >= 0 && device < 64) {
            std::lock_guard<std::mutex> g(m);
            std::vector<Block>& v = idle[device];
            int best = -1;
            for (int i = 0; i < (int)v.size(); ++i)
                if (v[(size_t)i].bytes >= bytes && v[(size_t)i].bytes <= 2 * bytes + 4096 && (best < 0 || v[(size_t)i].bytes < v[(size_t)best].bytes)) best = i;
            if (best >= 0) {
                const Block b = v[(size_t)best];
                v.erase(v.begin() + best);
                *got = b.bytes;
                return b.p;
            }
        }
        void* p = nullptr;
        if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) return nullptr;  // the caller reports hipGetLastError
        *got = bytes;
        return p;
    }
    void put(int device, void* p, size_t bytes) noexcept {
        if (!p) return;
        try {
            if (device >= 0 && device < 64) {
                std::lock_guard<std::mutex> g(m);
                std::vector<Block>& v = idle[device];
                if (bytes <= kMaxIdleBytes) {  // the newest block stays, the oldest go: a session repeats its last shape
                    size_t held = bytes;
                    for (const Block& b : v) held += b.bytes;
                    while (!v.empty() && (v.size() >= 6 || held > kMaxIdleBytes)) {
                        held -= v.front().bytes;
                        (void)hipFree(v.front().p);
                        v.erase(v.begin());
                    }
                    v.push_back(Block{p, bytes});
                    return;
                }
            }
        } catch (...) {
        }
        (void)hipFree(p);
    }
};
DevPool& dev_pool() { static DevPool* const pool = new DevPool(); return *pool; }

// pinned host staging that is released on every return path
struct PinnedBuf {
    void* p = nullptr;
    ~PinnedBuf() { if (p) (void)hipHostFree(p); }
    hipError_t alloc(size_t bytes) { return hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault); }
    template <class T> T* as() const { return static_cast<T*>(p); }
};
struct EventPair {
    hipEvent_t e[2] = {nullptr, nullptr};
    ~EventPair() { for (hipEvent_t x : e) if (x) (void)hipEventDestroy(x); }
    hipError_t create() {
        hipError_t r = hipEventCreateWithFlags(&e[0], hipEventDisableTiming);
        return r == hipSuccess ? hipEventCreateWithFlags(&e[1], hipEventDisableTiming) : r;
    }
};

// ---- progress reports and phase times of the *_run entry points (per calling thread)
struct Progress { bmm_progress_fn fn = nullptr; void* user = nullptr; int every = 0; };
thread_local Progress g_progress;
thread_local double g_phase_ms[BMM_RUN_PHASES] = {0, 0, 0, 0, 0, 0};
struct PhaseClock {
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void lap(int phase) {
        const auto n = std::chrono::steady_clock::now();
        g_phase_ms[phase] += std::chrono::duration<double, std::milli>(n - t).count();
        t = n;
    }
};

// accumulator counts the resample kernel is instantiated for
const int kKT[] = {4, 8, 12, 16, 20, 24, 28, 32, 40, 48, 52, 56, 64};  // (52: maxK = 50, BASELINE config 4)
// workgroup size by accumulator count: the VGPR budget per lane is 512 / (waves per SIMD)
constexpr int kThreadsSmall = 1024;  // KT <= 12: 128 VGPRs
constexpr int kThreadsMid = 768;     // KT 16, 20: 168 VGPRs
constexpr int kThreadsLarge = 512;   // 256 VGPRs
#ifndef BMM_STREAM_MODE
#define BMM_STREAM_MODE 2  // chains that share a device: a hardware queue of its own each (chain_stream_create)
#endif
#ifndef BMM_STAGE_WIDE
#define BMM_STAGE_WIDE 32
#endif
#ifndef BMM_SMALL_SPLIT
#define BMM_SMALL_SPLIT 1  // short launches of 16-32 accumulators run two lanes per observation (pick_kernel)
#endif
#ifndef BMM_SELF_TABLES
#define BMM_SELF_TABLES 1  // small finite-sampler shapes: resample workgroups build their own tables (pick_kernel)
#endif
constexpr int kStageWide = BMM_STAGE_WIDE;  // features in flight per wave where registers allow

int pick_kt(int cats) {
    for (int kt : kKT)
        if (kt >= cats) return kt;
    return -1;
}
// the bit-plane kernel holds no feature loads in flight, which frees 20-odd VGPRs: one size up
int threads_for(int kt, bool bits) {
    if (bits) return kt <= 20 ? kThreadsSmall : (kt <= 32 ? kThreadsMid : kThreadsLarge);
    return kt <= 12 ? kThreadsSmall : (kt <= 20 ? kThreadsMid : kThreadsLarge);
}

typedef void (*resample_fn)(ChainParams, ResampleArgs);
// BITS: X streamed as bit planes (k_pack_bits) instead of the int32 matrix as handed over
// GW: the shape's group width (bmm_spec.h).  The default-sized kernels, their two-lane forms and their
// emitting twins exist for both widths; the stepped-down workgroup sizes for the preferred width only
// (shapes that fall back to the narrower groups are the ones with big tables).
template <int MINUS, bool BITS, int GW>
resample_fn resample_kernel_m(int kt) {
    constexpr int SW = BITS ? 16 : kStageWide;
    switch (kt) {
        case 4: return k_resample<4, kThreadsSmall, MINUS, SW, BITS, 1, false, GW>;
        case 8: return k_resample<8, kThreadsSmall, MINUS, SW, BITS, 1, false, GW>;
        case 12: return k_resample<12, kThreadsSmall, MINUS, SW, BITS, 1, false, GW>;
        case 16: return k_resample<16, BITS ? kThreadsSmall : kThreadsMid, MINUS, SW, BITS, 1, false, GW>;
        case 20: return k_resample<20, BITS ? kThreadsSmall : kThreadsMid, MINUS, SW, BITS, 1, false, GW>;
        case 24: return k_resample<24, BITS ? kThreadsMid : kThreadsLarge, MINUS, 16, BITS, 1, false, GW>;
        case 28: return k_resample<28, BITS ? kThreadsMid : kThreadsLarge, MINUS, 16, BITS, 1, false, GW>;
        case 32: return k_resample<32, BITS ? kThreadsMid : kThreadsLarge, MINUS, 16, BITS, 1, false, GW>;
        case 40: return k_resample<40, kThreadsLarge, MINUS, 16, BITS, 1, false, GW>;
        case 48: return k_resample<48, kThreadsLarge, MINUS, 16, BITS, 1, false, GW>;
        case 52: return k_resample<52, kThreadsLarge, MINUS, 16, BITS, 1, false, GW>;
        case 56: return k_resample<56, kThreadsLarge, MINUS, 16, BITS, 1, false, GW>;
        case 64: return k_resample<64, kThreadsLarge, MINUS, 16, BITS, 1, false, GW>;
    }
    return nullptr;
}
// 256-thread variants: used when a batch is too small to give every CU a workgroup otherwise
template <int MINUS, bool BITS>
resample_fn resample_kernel_small(int kt) {
    switch (kt) {
        case 4: return k_resample<4, 256, MINUS, 16, BITS>;
        case 8: return k_resample<8, 256, MINUS, 16, BITS>;
        case 12: return k_resample<12, 256, MINUS, 16, BITS>;
        case 16: return k_resample<16, 256, MINUS, 16, BITS>;
        case 20: return k_resample<20, 256, MINUS, 16, BITS>;
        case 24: return k_resample<24, 256, MINUS, 16, BITS>;
        case 28: return k_resample<28, 256, MINUS, 16, BITS>;
        case 32: return k_resample<32, 256, MINUS, 16, BITS>;
        case 40: return k_resample<40, 256, MINUS, 16, BITS>;
        case 48: return k_resample<48, 256, MINUS, 16, BITS>;
        case 52: return k_resample<52, 256, MINUS, 16, BITS>;
        case 56: return k_resample<56, 256, MINUS, 16, BITS>;
        case 64: return k_resample<64, 256, MINUS, 16, BITS>;
    }
    return nullptr;
}
// the same kernels at a chosen workgroup size (up to 32 categories): used when a batch is too
// small to give every CU one of the default-sized workgroups, and by BMM_DEBUG_THREADS
template <int NT, int MINUS, bool BITS>
resample_fn resample_kernel_nt(int kt) {
    constexpr int SW = BITS || NT == 1024 ? 16 : kStageWide;  // 1024 threads: 128 VGPRs, 16 loads in flight per wave
    switch (kt) {
        case 4: return k_resample<4, NT, MINUS, SW, BITS>;
        case 8: return k_resample<8, NT, MINUS, SW, BITS>;
        case 12: return k_resample<12, NT, MINUS, SW, BITS>;
        case 16: return k_resample<16, NT, MINUS, SW, BITS>;
        case 20: return k_resample<20, NT, MINUS, SW, BITS>;
        case 24: return k_resample<24, NT, MINUS, 16, BITS>;
        case 28: return k_resample<28, NT, MINUS, 16, BITS>;
        case 32: return k_resample<32, NT, MINUS, 16, BITS>;
    }
    return nullptr;
}
resample_fn resample_kernel_at(int kt, int nt, int minus, bool bits) {
    if (minus == 2) return nullptr;
    if (bits) {
        if (nt == 768) return minus ? resample_kernel_nt<768, 1, true>(kt) : resample_kernel_nt<768, 0, true>(kt);
        if (nt == 512) return minus ? resample_kernel_nt<512, 1, true>(kt) : resample_kernel_nt<512, 0, true>(kt);
        if (nt == 1024 && kt <= 20) return minus ? resample_kernel_nt<1024, 1, true>(kt) : resample_kernel_nt<1024, 0, true>(kt);
        return nullptr;
    }
    if (kt > 20 || minus != 1) return nullptr;
    if (nt == 1024) return resample_kernel_nt<1024, 1, false>(kt);
    if (nt == 768) return resample_kernel_nt<768, 1, false>(kt);
    if (nt == 512) return resample_kernel_nt<512, 1, false>(kt);
    return nullptr;
}
// more than 32 accumulators on bit planes: two lanes per observation (SPLIT = 2 in kernels.hip.h)
constexpr int kThreadsSplit = 1024;
template <int MINUS, int GW>
resample_fn resample_kernel_split_w(int kt) {
    switch (kt) {
        // 16 to 32 accumulators: for launches too short to give a wave more than a chunk or two (pick_kernel)
        case 16: return GW == kGroupW ? k_resample<16, kThreadsSplit, MINUS, 16, true, 2, false, kGroupW> : nullptr;
        case 20: return GW == kGroupW ? k_resample<20, kThreadsSplit, MINUS, 16, true, 2, false, kGroupW> : nullptr;
        case 24: return GW == kGroupW ? k_resample<24, kThreadsSplit, MINUS, 16, true, 2, false, kGroupW> : nullptr;
        case 28: return GW == kGroupW ? k_resample<28, kThreadsSplit, MINUS, 16, true, 2, false, kGroupW> : nullptr;
        case 32: return GW == kGroupW ? k_resample<32, kThreadsSplit, MINUS, 16, true, 2, false, kGroupW> : nullptr;
        case 40: return k_resample<40, kThreadsSplit, MINUS, 16, true, 2, false, GW>;
        case 48: return k_resample<48, kThreadsSplit, MINUS, 16, true, 2, false, GW>;
        case 52: return k_resample<52, kThreadsSplit, MINUS, 16, true, 2, false, GW>;
        case 56: return k_resample<56, kThreadsSplit, MINUS, 16, true, 2, false, GW>;
        case 64: return k_resample<64, kThreadsSplit, MINUS, 16, true, 2, false, GW>;
    }
    return nullptr;
}
template <int MINUS>
resample_fn resample_kernel_split(int kt, int gw) {
    return gw == kGroupW ? resample_kernel_split_w<MINUS, kGroupW>(kt) : resample_kernel_split_w<MINUS, kGroupWAlt>(kt);
}
// The weight-emitting twins (EMIT) of the default-sized bit-plane kernels: what a sweep runs on
// while its allocation probabilities go to the host (any workgroup size serves any batch).
template <int MINUS, int GW>
resample_fn resample_kernel_emit_m(int kt) {
    switch (kt) {
        case 4: return k_resample<4, kThreadsSmall, MINUS, 16, true, 1, true, GW>;
        case 8: return k_resample<8, kThreadsSmall, MINUS, 16, true, 1, true, GW>;
        case 12: return k_resample<12, kThreadsSmall, MINUS, 16, true, 1, true, GW>;
        case 16: return k_resample<16, kThreadsSmall, MINUS, 16, true, 1, true, GW>;
        case 20: return k_resample<20, kThreadsSmall, MINUS, 16, true, 1, true, GW>;
        case 24: return k_resample<24, kThreadsMid, MINUS, 16, true, 1, true, GW>;
        case 28: return k_resample<28, kThreadsMid, MINUS, 16, true, 1, true, GW>;
        case 32: return k_resample<32, kThreadsMid, MINUS, 16, true, 1, true, GW>;
        case 40: return k_resample<40, kThreadsLarge, MINUS, 16, true, 1, true, GW>;
        case 48: return k_resample<48, kThreadsLarge, MINUS, 16, true, 1, true, GW>;
        case 52: return k_resample<52, kThreadsLarge, MINUS, 16, true, 1, true, GW>;
        case 56: return k_resample<56, kThreadsLarge, MINUS, 16, true, 1, true, GW>;
        case 64: return k_resample<64, kThreadsLarge, MINUS, 16, true, 1, true, GW>;
    }
    return nullptr;
}
template <int GW>
resample_fn resample_kernel_emit_w(int kt, int minus) {
    return minus == 0 ? resample_kernel_emit_m<0, GW>(kt)
                      : (minus == 1 ? resample_kernel_emit_m<1, GW>(kt) : resample_kernel_emit_m<2, GW>(kt));
}
resample_fn resample_kernel_emit(int kt, int minus, int gw) {
    return gw == kGroupW ? resample_kernel_emit_w<kGroupW>(kt, minus) : resample_kernel_emit_w<kGroupWAlt>(kt, minus);
}

// minus: 0 no own-cluster tables (stick-breaking), 1 in LDS, 2 in global memory
template <int GW>
resample_fn resample_kernel_w(int kt, int minus, bool bits) {
    if (bits)
        return minus == 0 ? resample_kernel_m<0, true, GW>(kt)
                          : (minus == 1 ? resample_kernel_m<1, true, GW>(kt) : resample_kernel_m<2, true, GW>(kt));
    return minus == 0 ? resample_kernel_m<0, false, GW>(kt)
                      : (minus == 1 ? resample_kernel_m<1, false, GW>(kt) : resample_kernel_m<2, false, GW>(kt));
}
resample_fn resample_kernel(int kt, int minus, bool bits, int gw) {
    return gw == kGroupW ? resample_kernel_w<kGroupW>(kt, minus, bits) : resample_kernel_w<kGroupWAlt>(kt, minus, bits);
}
// SELF: 256-thread workgroups that build their own table image (finite sampler, small shapes: kernels.hip.h)
resample_fn resample_kernel_self(int kt) {
    switch (kt) {
        case 4: return k_resample<4, 256, 1, 16, true, 1, false, kGroupW, true>;
        case 8: return k_resample<8, 256, 1, 16, true, 1, false, kGroupW, true>;
        case 12: return k_resample<12, 256, 1, 16, true, 1, false, kGroupW, true>;
    }
    return nullptr;
}
resample_fn resample_kernel_small_of(int kt, int minus, bool bits) {
    if (bits) return minus == 0 ? resample_kernel_small<0, true>(kt) : resample_kernel_small<1, true>(kt);
    return minus == 0 ? resample_kernel_small<0, false>(kt) : resample_kernel_small<1, false>(kt);
}

}  // namespace

struct bmm_chain {
    ChainParams p{};
    int device = 0;  // the HIP device
    HostCrew* crew = nullptr;  // the host threads of the *_run call this chain belongs to (not owned), if any
    int slot = 0;    // the device index the caller named (= device, except under BMM_DEBUG_FAKE_DEVICES)
    hipStream_t stream = nullptr;
    bool dedicated_queue = false;  // the stream has a hardware queue of its own (chains sharing a device)
    int stream_kind = 0;           // what the stream pool takes back: 1 an ordinary non-blocking stream, 2 one with its own queue, 0 neither
    bool shares_device = false;    // another chain runs beside this one on the device (bmm_chain_share_data)
    bool self_tables = false;      // the resample workgroups build their own table image: no k_count_tables per batch
    size_t lds_bytes_base = 0;     // lds_bytes without the SELF kernels' scratch
    int64_t batch = 1;
    double alpha0 = 1.0;
    int NT = 0, grid_max = 0, minus_in_lds = 1;
    int OT = 0;  // observations per tile (= NT, or NT / 2 for the two-lanes-per-observation kernels)
    bool sharded = false, shard_open = false;  // one chain over several ranks (explicit-parameter samplers)
    bool generic = false;         // shape beyond the resident kernel: tables from global memory
    double* dScratch = nullptr;   // generic path: per-thread score columns
    // allocation probabilities for the host's relabelling: while probs_dst is
    // set, every resample launch also emits its weights (dWts [Kc][N], dWtot [N]) and k_probs_finish
    // normalises them into probs_dst (N x K column-major, device)
    double *dProbs = nullptr, *dWts = nullptr, *dWtot = nullptr;
    double* probs_dst = nullptr;
    int64_t scratch_stride = 0;
    size_t lds_bytes = 0;
    resample_fn fn = nullptr;
    resample_fn fn_emit = nullptr;  // weight-emitting twin (bit planes), its workgroup size and grid limit
    int NT_emit = 0, grid_max_emit = 0;

    const int32_t* dX = nullptr;
    int32_t* dX_owned = nullptr;
    uint32_t* dXb = nullptr;      // bit planes of X (k_pack_bits), what the resident kernels stream by default
    // the planes are shared by reference count between the chains of a device that run over the same data
    // (bmm_chain_share_data): the last chain to go frees them, in whatever order chains are destroyed
    struct Planes { uint32_t* d = nullptr; size_t bytes = 0; std::atomic<int> refs{1}; };
    Planes* planes = nullptr;
    bool xb_borrowed = false;     // this chain took its planes from another one
    bool bits = false;
    int num_cus = 0;
    // the chain's fixed state is carved out of one allocation (arena), the buffers of a *_run call out of a
    // second one (run_arena): a malloc / free pair per buffer cost the drop-in call more than a millisecond
    char *arena = nullptr, *run_arena = nullptr;
    size_t arena_bytes = 0, run_arena_bytes = 0;  // as handed out by the device pool
    int32_t* dZ[2] = {nullptr, nullptr};
    int32_t *dNk = nullptr, *dS = nullptr, *dDNk = nullptr, *dDS = nullptr;
    double *dAlpha = nullptr, *dTab = nullptr, *dPi = nullptr, *dTheta = nullptr;
    bool have_data = false, have_init = false, started = false;
    int sweep = 0;  // sweeps completed (= index j of the last one)

    // trace of the *_run entry points
    int burnin = 0, S = 0;
    int32_t* dTrace = nullptr;  // [S][N], 0-based
    double *dThetaTrace = nullptr, *dAlphaTrace = nullptr, *dPiTrace = nullptr;
    char* dOutBlk[2] = {nullptr, nullptr};  // staging of the label trace on its way out (trace_out)
    size_t out_blk_bytes = 0;

    int32_t* dNkTrace = nullptr;  // [n][K] cluster sizes per sweep of the current sweeps_counts call
    int nk_trace_base = 0;        // sweep index of its row 0
    unsigned long long* dDiag = nullptr;
    int* dViable = nullptr;       // stick-breaking / full: the cluster count the concentration's update uses (k_sb_params -> k_sb_theta_tables)
    int* dSelfDone = nullptr;     // SELF kernels: workgroups of the running launch that have read the statistics
    int32_t *dDNkAlt = nullptr, *dDSAlt = nullptr;  // ... and the second set of delta accumulators (self_fold_prev)
    int* dDbgFlag = nullptr;      // -DBMM_DEBUG_HOOKS: raised by a kernel that meets a label out of range
    int prof = 0;             // > 0: HIP events around the resample launches of every prof-th sweep
    std::vector<hipEvent_t> ev;
    size_t ev_used = 0;
    double prof_ms = 0.0;
    int64_t prof_n = 0;
};

namespace {

int planes_alloc(bmm_chain* c, size_t words) {
    size_t got = 0;
    uint32_t* d = static_cast<uint32_t*>(dev_pool().get(c->device, words * sizeof(uint32_t), &got));
    if (!d) return set_err(BMM_E_HIP, "allocating the bit planes failed: %s", hipGetErrorString(hipGetLastError()));
    c->planes = new (std::nothrow) bmm_chain::Planes();
    if (!c->planes) { dev_pool().put(c->device, d, got); return set_err(BMM_E_ARG, "out of host memory"); }
    c->planes->d = d;
    c->planes->bytes = got;
    c->dXb = d;
    return BMM_OK;
}

// every cell of X must be 0 or 1: one streaming pass when the matrix is handed over
int validate_binary(bmm_chain* c, const int32_t* dX, int64_t n) {
    DevBuf flagbuf;
    HIP_TRY(flagbuf.alloc(sizeof(int)));
    int* const dflag = flagbuf.as<int>();
    HIP_TRY(hipMemsetAsync(dflag, 0, sizeof(int), c->stream));
    const bool al16 = (reinterpret_cast<uintptr_t>(dX) & 15) == 0;
    const int64_t n16 = al16 ? n / 4 : 0;
    hipLaunchKernelGGL(k_validate_binary, dim3(2048), dim3(256), 0, c->stream,
                       reinterpret_cast<const uint4*>(dX), n16, reinterpret_cast<const uint32_t*>(dX), n, dflag);
    int flag = 0;
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(&flag, dflag, sizeof(int), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) return set_err(BMM_E_HIP, "validating X failed: %s", hipGetErrorString(e));
    if (flag) return set_err(BMM_E_ARG, "data must be binary: X holds a value other than 0 and 1");
    return BMM_OK;
}

// Default batch -- a pure function of (sampler, N): no device, occupancy or layout enters, so a defaulted batch
// names the same chain everywhere.  Below 2^16 observations: floor(N/8) for the finite sampler, floor(N/16) for the
// DP sampler (above that every "new" draw of a batch shares one label and spurious clusters open on small data).
// From 2^16 observations on: floor(N/4) for both.  The bias of a batch against the sequential scan shrinks
// with N -- the net flow into a cluster during a batch is O(sqrt N), its size O(N) -- and at those sizes it is not
// measurable: the committed batch-1 fixtures (tests/golden/tolerance_*.json; K = 20 at N = 2^16, 1e5, 2^18, 2^19, 1e6, 2e6
// and 1e7, K = 3 at 1e5, seven DP shapes from N = 2^16) are met to 2e-5 in the proportions and 1e-3 in theta-hat at N/4 as at
// N/8, and the DP's shares per generating component and cluster counts at N/4 as at N/16 (tests/test_gpu_tolerance_fixtures.py runs at
// whatever this function returns; profiles/r03/README.md has the numbers, including that a random start ends in the
// generating mode as often at N/4 as at N/8).  What a larger batch buys is launches: a batch costs about 20 us of
// fixed time whatever its size (k_count_tables + the resample launch's own set-up and flush).
// Never rounded: work is handed out per wave in chunks of 64 observations, so a launch that is not a whole number
// of rounds of the chip costs a fraction of a chunk per wave.
constexpr int64_t kLargeN = (int64_t)1 << 16;
int64_t default_batch(int sampler, int64_t N) {
    if (sampler == BMM_SAMPLER_SB || sampler == BMM_SAMPLER_FULL) return N;
    const int64_t div = N >= kLargeN ? 4 : (sampler == BMM_SAMPLER_DP ? 16 : 8);
    const int64_t b = N / div;
    return b < 1 ? 1 : b;
}

TableLayout layout_of(const bmm_chain* c) { return layout_of(c->p, !explicit_params(c->p.mode)); }

constexpr size_t kLdsMax = 163840;  // gfx950: 160 KiB per workgroup
size_t hist_bytes_of(int K, int P) { return ((size_t)K * P + K + 4) * sizeof(int32_t); }  // histogram + chunk counter

// The spec's rule for the group width of a shape (bmm_spec.h; the oracle restates it): groups of kGroupW
// features when the whole table image of the shape and the histogram fit in LDS that way, kGroupWAlt
// otherwise.  A pure function of (sampler, K, P).
int group_width_rule(int sampler, int K, int P) {
    const int cats = sampler == BMM_SAMPLER_DP ? K + 1 : K;
    const int kt = pick_kt(cats);
    if (kt < 0 || P > kMaxP) return kGroupWAlt;
    ChainParams q{};
    q.mode = sampler; q.P = P; q.K = K; q.KT = kt; q.W = kGroupW;
    q.G = (P + kGroupW - 1) / kGroupW; q.Gm = (P + kGroupWm - 1) / kGroupWm;
    const size_t bytes = (size_t)layout_of(q, !explicit_params(sampler)).doubles() * sizeof(double) + hist_bytes_of(K, P);
    return bytes <= kLdsMax ? kGroupW : kGroupWAlt;
}

int32_t* label_row(bmm_chain* c, int j) {
    if (c->dTrace && j >= c->burnin) return c->dTrace + (size_t)(j - c->burnin) * c->p.N;
    return c->dZ[j & 1];
}

// A chain's stream.  The runtime pools at most four hardware queues per priority level and lets later
// streams share them -- with whatever the host framework created before -- and two chains on one queue
// serialise: four chains of the north-star shape on one GPU ran at 10.5 k sweeps/s in all on plain
// streams in a fresh process and at 5.5 k (no overlap at all) at the end of bench.py's sequence of
// workloads; streams on the high-priority level, and GPU_MAX_HW_QUEUES=8, fixed the first case and not
// the second (profiles/r02/README.md).  A stream created with a CU mask -- here the full one -- gets a
// hardware queue of its own from the runtime, whatever came before: 14.9-15.7 k in both cases.  It costs
// about 6 ms more per chain created, and beyond four chains per device the queues start to thrash (8
// chains: 9.9 k against 12.5 k on shared queues).  mode 0: plain stream; 1: high priority; 2: CU mask.
// compute units of a device, asked once per process (hipGetDeviceProperties costs a good part of a
// millisecond, and a drop-in call creates a chain every time)
int device_cus(int device) {
    static std::mutex m;
    static int cus[64];
    if (device < 0 || device >= 64) return 0;
    std::lock_guard<std::mutex> g(m);
    if (cus[device] == 0) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) != hipSuccess) { (void)hipGetLastError(); return 0; }
        cus[device] = prop.multiProcessorCount;
    }
    return cus[device];
}

// Streams outlive their chain: creating and destroying a plain one costs about 2 ms each on this runtime
// (tools/hipcost_probe.hip), a fifth of what the rest of a 220-sweep drop-in call at the north-star shape
// spends outside its sweeps; one with a hardware queue of its own (chains sharing a device) costs 6 ms, and a
// process that created and destroyed four of those per call was seen to hang inside the runtime's queue
// creation after some hundred calls (a runtime thread stuck in the driver's queue ioctl, the caller waiting for
// its lock: profiles/r03/README.md).  A destroyed chain's (idle) stream goes back here, up to four of either
// kind per device.
struct StreamPool {
    std::mutex m;
    std::vector<hipStream_t> idle[2][64];  // [0] plain, [1] with a hardware queue of their own
    hipStream_t get(int device, bool dedicated) {
        if (device < 0 || device >= 64) return nullptr;
        std::lock_guard<std::mutex> g(m);
        std::vector<hipStream_t>& v = idle[dedicated ? 1 : 0][device];
        if (v.empty()) return nullptr;
        hipStream_t s = v.back();
        v.pop_back();
        return s;
    }
    bool put(int device, hipStream_t s, bool dedicated) noexcept {
        if (device < 0 || device >= 64) return false;
        try {
            std::lock_guard<std::mutex> g(m);
            std::vector<hipStream_t>& v = idle[dedicated ? 1 : 0][device];
            if (v.size() >= 4) return false;
            v.push_back(s);
            return true;
        } catch (...) {
            return false;
        }
    }
};
StreamPool& stream_pool() { static StreamPool* const pool = new StreamPool(); return *pool; }  // never destroyed: no HIP call at exit

int chain_stream_create(bmm_chain* c, bool dedicated) {
    int mode = dedicated ? BMM_STREAM_MODE : 0;
    if (const char* m = dbg_env("BMM_DEBUG_STREAM")) mode = atoi(m);
    hipError_t e = hipErrorUnknown;
    hipStream_t st = nullptr;
    if (mode == 1) {
        int least = 0, greatest = 0;
        e = hipDeviceGetStreamPriorityRange(&least, &greatest);
        if (e == hipSuccess) e = hipStreamCreateWithPriority(&st, hipStreamNonBlocking, greatest);
    } else if (mode == 2) {
        st = stream_pool().get(c->device, true);
        if (st) {
            e = hipSuccess;
        } else {
            uint32_t mask[16];
            for (uint32_t& w : mask) w = 0xffffffffu;
            const int cus = device_cus(c->device);
            e = cus > 0 ? hipExtStreamCreateWithCUMask(&st, (uint32_t)((cus + 31) / 32), mask) : hipErrorUnknown;
        }
    }
    int kind = mode == 2 ? 2 : 0;  // 0: not pooled, 1: a plain stream of the pool's kind, 2: one with its own queue
    if (e != hipSuccess) {  // mode 0, or the special stream could not be had
        (void)hipGetLastError();
        st = stream_pool().get(c->device, false);
        if (!st) HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        kind = 1;
    }
    c->stream = st;
    c->dedicated_queue = dedicated;
    c->stream_kind = kind;
    return BMM_OK;
}
// an idle stream goes back to the pool of its kind, anything else (and what the pool has no room for) is destroyed
void chain_stream_release(int device, hipStream_t st, int kind) {
    if (!st) return;
    if (kind != 0 && stream_pool().put(device, st, kind == 2)) return;
    (void)hipStreamDestroy(st);
}

// A chain that starts sharing its device with another one moves to a stream with a hardware queue of its
// own (a lone chain keeps a plain stream: the dedicated queue costs about 6 ms to create).  Only before
// the first sweep: nothing but the finished data hand-over has run on the old stream.
int chain_dedicated_queue(bmm_chain* c) {
    if (c->dedicated_queue || c->started) return BMM_OK;
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t old = c->stream;
    const int old_kind = c->stream_kind;
    if (old) HIP_TRY(hipStreamSynchronize(old));
    int rc = chain_stream_create(c, true);
    if (rc) { c->stream = old; c->stream_kind = old_kind; return rc; }
    chain_stream_release(c->device, old, old_kind);
    return BMM_OK;
}

// bump allocation out of one device buffer, 256-byte aligned pieces
struct Carver {
    char* base;
    size_t used = 0;
    template <class T> T* take(size_t n) {
        const size_t off = (used + 255) & ~(size_t)255;
        used = off + n * sizeof(T);
        return base ? reinterpret_cast<T*>(base + off) : nullptr;
    }
};

int chain_alloc(bmm_chain* c) {
    const ChainParams& p = c->p;
    HIP_TRY(hipSetDevice(c->device));
    // (chains that come to share a device are moved to streams with hardware queues of their own:
    // chain_dedicated_queue)
    {
        int rcs = chain_stream_create(c, false);
        if (rcs) return rcs;
    }
    const size_t nz = (size_t)p.N;
    const size_t ns = (size_t)p.K * p.P, nn = (size_t)p.K;
    const size_t ntab = (size_t)layout_of(c).doubles();
    auto carve = [&](Carver& a) {
        // the statistics, their delta replicas, the table image and the flags first: zeroed in one go
        c->dNk = a.take<int32_t>(nn);
        c->dDNk = a.take<int32_t>(nn * kDeltaReps);
        c->dS = a.take<int32_t>(ns);
        c->dDS = a.take<int32_t>(ns * kDeltaReps);
        c->dTab = a.take<double>(ntab);
#ifdef BMM_DIAG
        c->dDiag = a.take<unsigned long long>(16);
#endif
        c->dSelfDone = a.take<int>(1);
        c->dViable = a.take<int>(1);
        c->dDNkAlt = a.take<int32_t>(nn * kDeltaReps);
        c->dDSAlt = a.take<int32_t>(ns * kDeltaReps);
#ifdef BMM_DEBUG_HOOKS
        c->dDbgFlag = a.take<int>(1);
#endif
        const size_t zeroed = a.used;
        c->dAlpha = a.take<double>(1);
        c->dPi = a.take<double>(nn);
        c->dTheta = a.take<double>(ns);
        c->dZ[0] = a.take<int32_t>(nz);
        c->dZ[1] = a.take<int32_t>(nz);
        return zeroed;
    };
    Carver measure{nullptr};
    carve(measure);
    c->arena = static_cast<char*>(dev_pool().get(c->device, measure.used, &c->arena_bytes));
    if (!c->arena) return set_err(BMM_E_HIP, "allocating the chain's state failed: %s", hipGetErrorString(hipGetLastError()));
    Carver real{c->arena};
    const size_t zeroed = carve(real);
    if (c->generic) HIP_TRY(hipMalloc(&c->dScratch, (size_t)c->scratch_stride * p.Kc * sizeof(double)));
    HIP_TRY(hipMemsetAsync(c->arena, 0, zeroed, c->stream));
    HIP_TRY(hipMemsetAsync(c->dZ[0], 0xff, nz * sizeof(int32_t), c->stream));  // -1 = unassigned
    HIP_TRY(hipMemcpyAsync(c->dAlpha, &c->alpha0, sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return BMM_OK;
}

// test variant only: after a synchronisation, has a kernel met a label outside its range?
int dbg_labels_ok(bmm_chain* c) {
#ifdef BMM_DEBUG_HOOKS
    int flag = 0;
    HIP_TRY(hipMemcpy(&flag, c->dDbgFlag, sizeof(int), hipMemcpyDeviceToHost));
    if (flag) return set_err(BMM_E_STATE, "a resample kernel produced or read a label outside [0, K): kernel bug (debug-hooks build check)");
#else
    (void)c;
#endif
    return BMM_OK;
}

int launch_resample(bmm_chain* c, const int32_t* z_in, int32_t* z_out, int64_t lo, int64_t hi,
                    uint32_t sweep) {
    ResampleArgs a{};
    a.X = c->dX; a.Xb = c->dXb; a.z_in = z_in; a.z_out = z_out; a.tab = c->dTab; a.dNk = c->dDNk; a.dS = c->dDS;
    a.lo = lo; a.hi = hi; a.sweep = sweep; a.minus_in_lds = c->minus_in_lds; a.diag = c->dDiag;
    a.dbg_flag = c->dDbgFlag; a.dbg_inject = (dbg_env("BMM_DEBUG_BADLABEL") ? 1 : 0) | (dbg_env("BMM_DEBUG_STRAGGLER") ? 2 : 0);
    a.Nk = c->dNk; a.S = c->dS; a.alpha_ptr = c->dAlpha; a.self_done = c->dSelfDone;
    const bool emit = c->probs_dst != nullptr;
    const bool use_generic = c->generic || (emit && !c->fn_emit);  // the int32 layout has no emitting twin
    // a kernel that builds its own tables reads the pending deltas and flushes into the other (empty) set, which
    // is the pending one from then on (self_fold_prev, kernels.hip.h)
    const bool self_launch = c->self_tables && !emit && !use_generic;
    if (self_launch) { a.dNk_prev = c->dDNk; a.dS_prev = c->dDS; a.dNk = c->dDNkAlt; a.dS = c->dDSAlt; }
    const int OT = emit ? c->NT_emit : c->OT, gmax = emit ? c->grid_max_emit : c->grid_max;
    const int64_t ntiles = use_generic ? 1 : (hi - lo + OT - 1) / OT;
    int grid = (int)(ntiles < gmax ? ntiles : gmax);
    if (use_generic) {
        const int64_t nt256 = (hi - lo + 255) / 256;
        const int64_t maxb = c->scratch_stride / 256;
        grid = (int)(nt256 < maxb ? nt256 : maxb);
    }
    if (emit) { a.wts = c->dWts; a.wtot = c->dWtot; }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (c->prof > 0 && sweep % (uint32_t)c->prof == 0) {
        if (c->ev_used + 2 > c->ev.size()) {
            hipEvent_t a0, a1;
            HIP_TRY(hipEventCreate(&a0));
            HIP_TRY(hipEventCreate(&a1));
            c->ev.push_back(a0); c->ev.push_back(a1);
        }
        e0 = c->ev[c->ev_used]; e1 = c->ev[c->ev_used + 1];
        c->ev_used += 2;
    }
    // a profiled launch carries its own start / stop events (hipExtLaunchKernel): they take the kernel's
    // begin and end timestamps, as rocprofv3's kernel trace does -- a pair of hipEventRecord around the launch
    // would also span the dispatch of the kernel and of the second marker (4-8 us per launch here)
    if (use_generic) {
        if (e0) hipExtLaunchKernelGGL(k_resample_generic, dim3(grid), dim3(256), 0, c->stream, e0, e1, 0, c->p, a,
                                      c->dScratch, c->scratch_stride);
        else hipLaunchKernelGGL(k_resample_generic, dim3(grid), dim3(256), 0, c->stream, c->p, a, c->dScratch,
                                c->scratch_stride);
    } else {
        const resample_fn fn = emit ? c->fn_emit : c->fn;
        const int nt = emit ? c->NT_emit : c->NT;
        const size_t lds = emit ? c->lds_bytes_base : c->lds_bytes;  // the SELF kernels carry scratch behind the image
        if (e0) hipExtLaunchKernelGGL(fn, dim3(grid), dim3(nt), (uint32_t)lds, c->stream, e0, e1, 0, c->p, a);
        else hipLaunchKernelGGL(fn, dim3(grid), dim3(nt), lds, c->stream, c->p, a);
    }
    HIP_TRY(hipGetLastError());
    if (self_launch) { std::swap(c->dDNk, c->dDNkAlt); std::swap(c->dDS, c->dDSAlt); }
    if (emit) {  // before the next k_count_tables rewrites the image's cluster sizes
        const int64_t nb = (hi - lo + 255) / 256;
        hipLaunchKernelGGL(k_probs_finish, dim3((unsigned)(nb < 4096 ? nb : 4096)), dim3(256), 0, c->stream, c->p,
                           c->dTab, c->dWts, c->dWtot, z_in, lo, hi, c->probs_dst);
        HIP_TRY(hipGetLastError());
    }
    return BMM_OK;
}

// buffers and kernel of the probability hand-off, set up on first use
int probs_alloc(bmm_chain* c, bool with_matrix) {
    const size_t n = (size_t)c->p.N;
    if (c->bits && !c->generic && !c->fn_emit) {
        const int minus = explicit_params(c->p.mode) ? 0 : (c->minus_in_lds ? 1 : 2);
        c->fn_emit = resample_kernel_emit(c->p.KT, minus, c->p.W);
        c->NT_emit = threads_for(c->p.KT, true);
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(c->fn_emit), hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds_bytes_base);
        int pe = 0;
        if (e == hipSuccess)
            e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&pe, reinterpret_cast<const void*>(c->fn_emit), c->NT_emit, c->lds_bytes_base);
        if (e != hipSuccess) { c->fn_emit = nullptr; return set_err(BMM_E_HIP, "kernel set-up failed: %s", hipGetErrorString(e)); }
        c->grid_max_emit = (pe < 1 ? 1 : pe) * c->num_cus;
    }
    if (!c->dWts) HIP_TRY(hipMalloc(&c->dWts, n * c->p.Kc * sizeof(double)));
    if (!c->dWtot) HIP_TRY(hipMalloc(&c->dWtot, n * sizeof(double)));
    if (with_matrix && !c->dProbs) HIP_TRY(hipMalloc(&c->dProbs, n * c->p.K * sizeof(double)));
    if (!c->fn_emit && !c->dScratch) {  // int32 layout: the hand-off sweeps run on the generic kernel
        int64_t threads = (int64_t)256 * 1024;
        const int64_t cap = ((int64_t)256 << 20) / ((int64_t)c->p.Kc * 8);
        if (threads > cap) threads = cap / 256 * 256;
        if (threads < 256) threads = 256;
        c->scratch_stride = threads;
        HIP_TRY(hipMalloc(&c->dScratch, (size_t)threads * c->p.Kc * sizeof(double)));
    }
    return BMM_OK;
}

// replicas 1.. of the delta accumulators folded into replica 0: what the host-side accessors and the
// sharded chain's all-reduce read
int launch_reduce_deltas(bmm_chain* c) {
    const int n = c->p.K * c->p.P + c->p.K;
    hipLaunchKernelGGL(k_reduce_deltas, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->p, c->dDNk, c->dDS);
    HIP_TRY(hipGetLastError());
    return BMM_OK;
}

int launch_count_tables(bmm_chain* c) {
    hipLaunchKernelGGL(k_count_tables, dim3(c->p.KT), dim3(kCountTablesThreads), 0, c->stream, c->p, c->dNk, c->dS,
                       c->dDNk, c->dDS, c->dAlpha, c->dTab);
    HIP_TRY(hipGetLastError());
    return BMM_OK;
}

// one sweep (index j >= 1) enqueued on the stream
// phase 0: whole sweep; 1: z-resample only; 2: parameter draws and tables only (sharded chains)
int enqueue_sweep(bmm_chain* c, int j, int phase = 0) {
    const ChainParams& p = c->p;
    const int32_t* zin = (p.mode != MODE_COLLAPSED && j == 1) ? nullptr : label_row(c, j - 1);
    int32_t* zout = label_row(c, j);
    const bool rec = c->dTrace && j >= c->burnin;
    const int s = j - c->burnin;
    double* th_tr = rec ? c->dThetaTrace + (size_t)s * p.K * p.P : nullptr;
    double* al_tr = rec ? c->dAlphaTrace + s : nullptr;
    int32_t* nk_tr = c->dNkTrace ? c->dNkTrace + (size_t)(j - c->nk_trace_base) * p.K : nullptr;
    if (explicit_params(p.mode)) {
        if (phase != 2) {
            int rc = launch_resample(c, zin, zout, 0, p.N, (uint32_t)j);
            if (rc) return rc;
        }
        if (phase == 1) return launch_reduce_deltas(c);  // sharded chain: the caller all-reduces replica 0 now
        hipLaunchKernelGGL(k_sb_params, dim3(1), dim3(1024), 0, c->stream, p, c->dNk, c->dS, c->dDNk,
                           c->dDS, c->dAlpha, c->dPi, (uint32_t)j, rec ? c->dPiTrace + s : nullptr, c->S,
                           c->dViable, nk_tr);
        HIP_TRY(hipGetLastError());
        // (one workgroup more than clusters: the concentration's update runs beside the theta draws)
        hipLaunchKernelGGL(k_sb_theta_tables, dim3(p.KT + 1), dim3(256), 0, c->stream, p, c->dNk, c->dS,
                           c->dPi, c->dTheta, 1, (uint32_t)j, th_tr, c->dTab, c->dAlpha, al_tr, c->dViable);
        HIP_TRY(hipGetLastError());
        return BMM_OK;
    }
    int64_t lo = 0;
    while (lo < p.N) {
        int64_t len = c->batch;
        if (p.mode == MODE_DP && j == 1) {  // seat the first sweep 1,1,2,4,... at a time
            const int64_t dbl = lo < 1 ? 1 : lo;
            if (dbl < len) len = dbl;
        }
        const int64_t hi = lo + len > p.N ? p.N : lo + len;
        // (SELF kernels build their own image from Nk, S and the previous launch's deltas; a sweep that hands its
        // probabilities to the host runs the emitting twin, which reads the image k_count_tables writes)
        int rc = c->self_tables && !c->probs_dst ? BMM_OK : launch_count_tables(c);
        if (rc) return rc;
        rc = launch_resample(c, zin, zout, lo, hi, (uint32_t)j);
        if (rc) return rc;
        lo = hi;
    }
    hipLaunchKernelGGL(k_count_sweep_end, dim3(1), dim3(1024), 0, c->stream, p, c->dNk, c->dS, c->dDNk,
                       c->dDS, c->dAlpha, (uint32_t)j, th_tr, al_tr, nk_tr);
    HIP_TRY(hipGetLastError());
    return BMM_OK;
}

// The resident kernel for this chain's shape, workgroup size and X layout (c->bits).
int pick_kernel(bmm_chain* c) {
    const ChainParams& p = c->p;
    const size_t lds_max = kLdsMax;
    const int minus = explicit_params(p.mode) ? 0 : (c->minus_in_lds ? 1 : 2);
    c->NT = threads_for(p.KT, c->bits);
    c->lds_bytes = c->lds_bytes_base;  // (a previous choice may have been a SELF kernel, which carries scratch)
    c->self_tables = false;
    const bool alt = p.W != kGroupW;  // the narrower groups: default-sized kernels only
    c->fn = resample_kernel(p.KT, minus, c->bits, p.W);
    int split = 1;
    // Two lanes per observation: always above 32 accumulators (registers), and from 16 up when a launch is so
    // short that the default form would give a wave at most a chunk or two -- then a launch is all latency
    // (the north-star shape: 15 us per launch of which 4 are arithmetic, VALU busy 27 %), and the two-lane
    // form halves the chain of dependent work per wave (half the categories to score, to exponentiate and
    // to compare per lane) at the same number of workgroups.  Same draw, bit for bit.
    // "Short" = the batch cannot give every CU a one-lane workgroup of 768 threads (196 608 observations on 256
    // CUs): below that the one-lane form steps down to 512 threads and two waves per SIMD.  Same-box, north-star
    // shape by batch: 125 000 two-lane +11 %, 162 500 +11 %, 200 000 -2.5 %, 250 000 -8 % (profiles/r03/ab_smallsplit.log).
    // (not for chains that share their device: several chains' launches fill the chip between them, and then
    // the one-lane form's lower total work wins -- four north-star chains: 14.8 k against 12.1 k sweeps/s)
    const bool short_launch = BMM_SMALL_SPLIT && !c->shares_device && p.KT >= 16 && !alt && (c->batch < (int64_t)c->num_cus * kThreadsMid || dbg_env("BMM_DEBUG_SPLIT"));
    if (c->bits && (p.KT > 32 || short_launch) && minus != 2 && !dbg_env("BMM_DEBUG_NOSPLIT")) {
        if (resample_fn f = minus ? resample_kernel_split<1>(p.KT, p.W) : resample_kernel_split<0>(p.KT, p.W)) {
            c->fn = f;
            c->NT = kThreadsSplit;
            split = 2;
        }
    }
    if (const char* dbg = alt ? nullptr : dbg_env("BMM_DEBUG_THREADS")) {
        const int nt = atoi(dbg);
        if (resample_fn f = resample_kernel_at(p.KT, nt, minus, c->bits)) { c->fn = f; c->NT = nt; }
    } else if (c->bits && split == 1 && !alt) {
        // a batch that cannot give every CU a workgroup of the default size gets smaller ones -- as long as they
        // still hold the whole batch in one round (250 000 observations: 245 workgroups of 1024 threads, one chunk
        // per wave, beat 256 of 768 where a quarter of the waves takes a second chunk: north-star kernel -6 %)
        for (int nt : {768, 512}) {
            if ((c->batch + c->NT - 1) / c->NT >= c->num_cus || nt >= c->NT || (c->batch + nt - 1) / nt > c->num_cus) continue;
            if (resample_fn f = resample_kernel_at(p.KT, nt, minus, true)) { c->fn = f; c->NT = nt; }
        }
    }
    hipError_t e = hipSetDevice(c->device);
    c->fn_emit = nullptr;  // the weight-emitting twin is set up when a hand-off first asks for it (probs_alloc)
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(c->fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds_bytes);
    int per_cu = 0;
    if (e == hipSuccess)
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(c->fn), c->NT, c->lds_bytes);
    if (e != hipSuccess) return set_err(BMM_E_HIP, "kernel set-up failed: %s", hipGetErrorString(e));
    if (per_cu < 1) per_cu = 1;
    c->grid_max = per_cu * c->num_cus;
    // a batch that cannot give every CU a workgroup runs on 256-thread workgroups instead, when the
    // tables are small enough for several of them per CU (otherwise fewer waves per CU just hurts)
    c->OT = c->NT / split;
    const int64_t tiles = (c->batch + c->NT - 1) / c->NT;
    if (split == 1 && !alt && tiles < c->num_cus && c->NT > 256 && (c->lds_bytes * 4 <= lds_max || dbg_env("BMM_DEBUG_SMALL")) &&
        minus != 2 && !dbg_env("BMM_DEBUG_THREADS")) {
        resample_fn f = resample_kernel_small_of(p.KT, minus, c->bits);
        hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(f), hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds_bytes);
        int pc2 = 0;
        if (e2 == hipSuccess) e2 = hipOccupancyMaxActiveBlocksPerMultiprocessor(&pc2, reinterpret_cast<const void*>(f), 256, c->lds_bytes);
        if (e2 == hipSuccess && pc2 >= 1) { c->fn = f; c->NT = 256; c->OT = 256; c->grid_max = pc2 * c->num_cus; }
    }
    // ... and such workgroups build the table image themselves when that is at most two logs per thread: the
    // shape is bound by launches then, and this drops k_count_tables from every batch (BASELINE config 2)
    if (BMM_SELF_TABLES && c->NT == 256 && split == 1 && c->bits && !alt && minus == 1 && !c->shares_device &&
        self_tables_fit(p.mode, p.K, p.P, 256) && !dbg_env("BMM_DEBUG_NOSELF")) {
        if (resample_fn f = resample_kernel_self(p.KT)) {
            const size_t lds = (c->lds_bytes_base + 7) / 8 * 8 + self_scratch_doubles(p.K, p.KT, p.P) * sizeof(double);
            hipError_t e3 = lds <= lds_max ? hipFuncSetAttribute(reinterpret_cast<const void*>(f), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                                           : hipErrorInvalidValue;
            int pc3 = 0;
            if (e3 == hipSuccess) e3 = hipOccupancyMaxActiveBlocksPerMultiprocessor(&pc3, reinterpret_cast<const void*>(f), 256, lds);
            if (e3 == hipSuccess && pc3 >= 1) { c->fn = f; c->lds_bytes = lds; c->grid_max = pc3 * c->num_cus; c->self_tables = true; }
            else (void)hipGetLastError();
        }
    }
    return BMM_OK;
}

// things that must be in place before the first sweep
int chain_start(bmm_chain* c) {
    const ChainParams& p = c->p;
    if (!c->have_data) return set_err(BMM_E_STATE, "data matrix not set");
    if (p.mode != MODE_DP && !c->have_init)
        return set_err(BMM_E_STATE, p.mode == MODE_COLLAPSED ? "initial labels not set"
                                                             : "initial pi/theta not set");
    HIP_TRY(hipSetDevice(c->device));
    if (p.mode == MODE_COLLAPSED) {
        // statistics of the initial allocation (collapsed_gibbs.cpp:60-63)
        int32_t* row0 = label_row(c, 0);
        if (row0 != c->dZ[0])
            HIP_TRY(hipMemcpyAsync(row0, c->dZ[0], (size_t)p.N * sizeof(int32_t), hipMemcpyDeviceToDevice,
                                   c->stream));
        const size_t hb = ((size_t)p.K * p.P + p.K) * sizeof(int32_t);
        const int64_t nt = (p.N + 255) / 256;
        if (c->generic)
            hipLaunchKernelGGL(k_count_labels_generic, dim3((unsigned)(nt < 2048 ? nt : 2048)), dim3(256), 0,
                               c->stream, p, c->dX, c->dXb, row0, c->dDNk, c->dDS);
        else
            hipLaunchKernelGGL(k_count_labels, dim3((unsigned)(nt < 2048 ? nt : 2048)), dim3(256), hb, c->stream,
                               p, c->dX, c->dXb, row0, c->dDNk, c->dDS);
        HIP_TRY(hipGetLastError());
    } else if (explicit_params(p.mode)) {
        hipLaunchKernelGGL(k_sb_theta_tables, dim3(p.KT), dim3(256), 0, c->stream, p, c->dNk, c->dS,
                           c->dPi, c->dTheta, 0, 0u, (double*)nullptr, c->dTab, (double*)nullptr, (double*)nullptr,
                           (const int*)nullptr);
        HIP_TRY(hipGetLastError());
    }
    c->started = true;
    return BMM_OK;
}

int check_common(int64_t N, int P, int K, double beta, double gamma) {
    if (N < 1) return set_err(BMM_E_ARG, "N must be >= 1");
    if (P < 1) return set_err(BMM_E_ARG, "P must be >= 1");
    if (K < 1) return set_err(BMM_E_ARG, "K must be >= 1");
    if (!(beta > 0.0) || !(gamma > 0.0)) return set_err(BMM_E_ARG, "beta and gamma must be > 0");
    return BMM_OK;
}

}  // namespace

extern "C" {

const char* bmm_last_error(void) { return g_err; }
int bmm_spec_group_width(void) { return kGroupW; }
int bmm_spec_group_width_own(void) { return kGroupWm; }
int bmm_spec_group_width_for(int sampler, int K, int P) {
    if (sampler < 0 || sampler > 3 || K < 1 || P < 1) return -1;
    return group_width_rule(sampler, K, P);
}
int64_t bmm_default_batch(int sampler, int64_t N) { return default_batch(sampler, N); }

int bmm_set_progress(bmm_progress_fn fn, void* user, int every) {
    g_progress.fn = every > 0 ? fn : nullptr;
    g_progress.user = user;
    g_progress.every = fn && every > 0 ? every : 0;
    return BMM_OK;
}
int bmm_last_run_phases(double* ms) {
    if (!ms) return set_err(BMM_E_ARG, "null argument");
    for (int q = 0; q < BMM_RUN_PHASES; ++q) ms[q] = g_phase_ms[q];
    return BMM_OK;
}
int bmm_host_threads(void) { return host_threads(); }

// what the library keeps between calls -- up to eight 4 MiB pieces of pinned staging, and per device up to four
// idle streams of either kind and up to 512 MiB of device blocks -- released now (a long-lived host process that is done
// sampling)
int bmm_release_pools(void) {
    {
        StagePool& sp = stage_pool();
        std::vector<void*> idle;
        { std::lock_guard<std::mutex> g(sp.m); idle.swap(sp.idle); }
        for (void* p : idle) (void)hipHostFree(p);
    }
    StreamPool& st = stream_pool();
    DevPool& dp = dev_pool();
    for (int d = 0; d < 64; ++d) {
        std::vector<hipStream_t> idle, idle_q;
        std::vector<DevPool::Block> blocks;
        { std::lock_guard<std::mutex> g(st.m); idle.swap(st.idle[0][d]); idle_q.swap(st.idle[1][d]); }
        idle.insert(idle.end(), idle_q.begin(), idle_q.end());
        { std::lock_guard<std::mutex> g(dp.m); blocks.swap(dp.idle[d]); }
        if (idle.empty() && blocks.empty()) continue;
        if (hipSetDevice(d) != hipSuccess) { (void)hipGetLastError(); continue; }
        for (hipStream_t s : idle) (void)hipStreamDestroy(s);
        for (const DevPool::Block& b : blocks) (void)hipFree(b.p);
    }
    return BMM_OK;
}

int bmm_device_count(int* n) {
    int k = 0;
    hipError_t e = hipGetDeviceCount(&k);
    if (e != hipSuccess) { *n = 0; return set_err(BMM_E_NODEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    *n = k;
    return BMM_OK;
}

int bmm_chain_create(bmm_chain** out, int sampler, int64_t N, int P, int K, double alpha, double beta,
                     double gamma, double a, double b, int64_t batch, uint64_t seed, int device) {
    if (!out) return set_err(BMM_E_ARG, "out is null");
    *out = nullptr;
    if (sampler < 0 || sampler > 3) return set_err(BMM_E_ARG, "unknown sampler %d", sampler);
    int rc = check_common(N, P, K, beta, gamma);
    if (rc) return rc;
    if (sampler == BMM_SAMPLER_DP && beta != gamma)  // collapsed_gibbs_dp.cpp:48-50
        return set_err(BMM_E_ARG, "Error: sampler currently not implemented for non-symmetric priors on beta and gamma");
    if (sampler == BMM_SAMPLER_DP && K < 2) return set_err(BMM_E_ARG, "maxK must be >= 2");
    if (alpha < 0.0) return set_err(BMM_E_ARG, "alpha must be >= 0 (0 = sample it)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return set_err(BMM_E_NODEVICE, "no HIP device visible; this path has no CPU fallback");
    const int slot = device;
    if (const int fake = fake_devices()) {
        if (device < 0 || device >= fake) return set_err(BMM_E_ARG, "device %d out of range (have %d)", device, fake);
        device = device % ndev;
    }
    if (device < 0 || device >= ndev) return set_err(BMM_E_ARG, "device %d out of range (have %d)", device, ndev);

    bmm_chain* c = new (std::nothrow) bmm_chain();
    if (!c) return set_err(BMM_E_ARG, "out of host memory");
    ChainParams& p = c->p;
    p.mode = sampler; p.N = N; p.Ntot = N; p.obs0 = 0; p.P = P; p.K = K;
    p.W = group_width_rule(sampler, K, P);
    p.G = (P + p.W - 1) / p.W;
    p.Gm = (P + kGroupWm - 1) / kGroupWm;
    p.Kc = sampler == BMM_SAMPLER_DP ? K + 1 : K;
    p.KT = pick_kt(p.Kc);
    p.beta = beta; p.gamma = gamma; p.a = a; p.b = b; p.seed = seed;
    p.sample_alpha = alpha == 0.0;  // collapsed_gibbs.cpp:50-54
    c->alpha0 = p.sample_alpha ? 1.0 : alpha;
    c->device = device;
    c->slot = slot;
    c->batch = batch <= 0 ? default_batch(sampler, N) : (batch > N ? N : batch);
    if (explicit_params(sampler)) c->batch = N;
    if (p.Kc > kMaxCatsAny) {
        delete c;
        return set_err(BMM_E_UNSUPPORTED, "%d categories exceed the %d this build supports", p.Kc, kMaxCatsAny);
    }
    const int cus = device_cus(device);
    if (cus <= 0) { delete c; return set_err(BMM_E_HIP, "hipGetDeviceProperties failed"); }
    const size_t lds_max = kLdsMax;
    const size_t hist_bytes = hist_bytes_of(K, P);
    c->bits = dbg_env("BMM_X_LAYOUT_INT32") == nullptr;
    c->generic = p.KT < 0 || P > kMaxP || dbg_env("BMM_DEBUG_GENERIC") != nullptr;
    if (!c->generic) {
        c->NT = threads_for(p.KT, false);
        c->OT = c->NT;
        c->lds_bytes = (size_t)layout_of(c).doubles() * sizeof(double) + hist_bytes;
        if (c->lds_bytes > lds_max && !explicit_params(p.mode)) {  // second tier: own-cluster tables stay in L2
            c->minus_in_lds = 0;
            c->lds_bytes = (size_t)layout_of(c).head() * sizeof(double) + hist_bytes;
        }
        if (c->lds_bytes > lds_max) c->generic = true;  // third tier: nothing resident
        c->lds_bytes_base = c->lds_bytes;
    }
    if (c->generic) {
        // any shape: tables gathered from global memory, scores in a scratch column per thread
        p.KT = (p.Kc + 3) / 4 * 4;
        c->NT = 256;
        c->OT = 256;
        c->lds_bytes = 0;
        c->minus_in_lds = 0;
        int64_t threads = (int64_t)256 * 1024;
        const int64_t cap = ((int64_t)256 << 20) / ((int64_t)p.Kc * 8);  // <= 256 MiB of scratch
        if (threads > cap) threads = cap / 256 * 256;
        if (threads < 256) threads = 256;
        c->scratch_stride = threads;
        c->grid_max = (int)(threads / 256);
    } else {
        c->num_cus = cus;
        rc = pick_kernel(c);
        if (rc) { delete c; return rc; }
    }
    rc = chain_alloc(c);
    if (rc) { bmm_chain_destroy(c); return rc; }
    *out = c;
    return BMM_OK;
}

void bmm_chain_destroy(bmm_chain* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
#ifdef BMM_DIAG
    if (c->dDiag) {
        unsigned long long d[16];
        if (hipMemcpy(d, c->dDiag, sizeof d, hipMemcpyDeviceToHost) == hipSuccess && d[5]) {
            const double tot = (double)(d[0] + d[1] + d[2] + d[3] + d[4]);
            fprintf(stderr, "[bmm diag] waves=%llu cycles/wave: score %.0f (%.1f%%) pack %.0f (%.1f%%) draw %.0f (%.1f%%) movers %.0f (%.1f%%) prologue %.0f (%.1f%%) movers/wave-launch %.2f\n",
                    d[5], d[0] / (double)d[5], 100 * d[0] / tot, d[1] / (double)d[5], 100 * d[1] / tot, d[2] / (double)d[5],
                    100 * d[2] / tot, d[3] / (double)d[5], 100 * d[3] / tot, d[4] / (double)d[5], 100 * d[4] / tot, d[6] / (double)d[5]);
        }
        if (hipMemcpy(d, c->dDiag, sizeof d, hipMemcpyDeviceToHost) == hipSuccess && d[5])
            fprintf(stderr, "[bmm diag launch] ticks per wave and launch: table staging %.0f, tile loop %.0f, waiting for the workgroup %.0f, flush %.0f\n",
                    d[8] / (double)d[5], d[9] / (double)d[5], d[10] / (double)d[5], d[11] / (double)d[5]);
    }
#endif
    for (hipEvent_t e : c->ev) (void)hipEventDestroy(e);
    if (c->planes && c->planes->refs.fetch_sub(1) == 1) {  // the last chain over these planes
        dev_pool().put(c->device, c->planes->d, c->planes->bytes);
        delete c->planes;
    }
    dev_pool().put(c->device, c->arena, c->arena_bytes);  // the stream is idle (synchronised above)
    dev_pool().put(c->device, c->run_arena, c->run_arena_bytes);
    void* bufs[] = {c->dX_owned, c->dScratch, c->dProbs, c->dWts, c->dWtot};
    for (void* b : bufs)
        if (b) (void)hipFree(b);
    chain_stream_release(c->device, c->stream, c->stream_kind);  // synchronised above
    delete c;
}

// `rows` observations of an int32 matrix (feature d of observation i at X[i + d * ldx]) into the
// chain's bit planes, starting at observation i0
static int pack_rows(bmm_chain* c, const int32_t* dX, int64_t rows, int64_t ldx, int64_t i0) {
    const int64_t nb = (rows + 255) / 256;
    hipLaunchKernelGGL(k_pack_bits, dim3((unsigned)(nb < 16384 ? nb : 16384)), dim3(256), 0, c->stream, dX, rows,
                       ldx, c->p.P, c->dXb + i0, c->p.N);
    HIP_TRY(hipGetLastError());
    return BMM_OK;
}

int bmm_chain_set_x_layout(bmm_chain* c, int layout) {
    if (!c) return set_err(BMM_E_ARG, "null chain");
    if (layout != BMM_X_BITPLANES && layout != BMM_X_INT32) return set_err(BMM_E_ARG, "unknown X layout %d", layout);
    if (c->have_data || c->started) return set_err(BMM_E_STATE, "the X layout is chosen before the data are set");
    c->bits = layout == BMM_X_BITPLANES;
    if (c->generic) return BMM_OK;  // one kernel for both layouts there
    return pick_kernel(c);
}

int bmm_chain_get_x_layout(const bmm_chain* c, int* layout) {
    if (!c || !layout) return set_err(BMM_E_ARG, "null argument");
    *layout = c->bits ? BMM_X_BITPLANES : BMM_X_INT32;
    return BMM_OK;
}

int bmm_chain_set_data_host(bmm_chain* c, const int32_t* X) {
    if (!c || !X) return set_err(BMM_E_ARG, "null argument");
    if (c->started) return set_err(BMM_E_STATE, "chain already started");
    HIP_TRY(hipSetDevice(c->device));
    const int64_t N = c->p.N;
    const int P = c->p.P;
    if (!c->bits) {  // the matrix itself is what the sweeps stream: one copy, kept
        const size_t bytes = (size_t)N * P * sizeof(int32_t);
        if (!c->dX_owned) HIP_TRY(hipMalloc(&c->dX_owned, bytes));
        HIP_TRY(hipMemcpyAsync(c->dX_owned, X, bytes, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->dX = c->dX_owned;
        int rc = validate_binary(c, c->dX, N * P);
        if (rc) return rc;
        c->have_data = true;
        return BMM_OK;
    }
    // Bit planes from a host matrix: the host's own cores validate and pack it, slab by slab, into pinned
    // staging, and only the planes cross PCIe -- 4 * ceil(P/32) bytes per observation instead of 4 * P (8 MB
    // instead of 200 MB at N = 1e6, P = 50; a pageable 200 MB upload alone took 13.5 ms there, a third of
    // a 220-sweep run).  Slab s + 1 is packed while slab s is on its way.  Same planes, bit for bit, as
    // k_pack_bits makes from a matrix already on the device (tests/test_gpu_fullsize.py holds the two
    // hand-overs to the same chain).
    return guarded([&]() -> int {
        const int W = (P + 31) / 32;
        if (!c->dXb) { int rcp = planes_alloc(c, (size_t)W * N); if (rcp) return rcp; }
        int64_t slab = (int64_t)(kStageBytes / ((size_t)W * 4));  // one staging piece of packed words per slab
        slab = slab / 4096 * 4096;
        if (slab < 4096) slab = 4096;
        if (slab > N) slab = N;
        Stage pooled[2];
        PinnedBuf own[2];
        uint32_t* pin[2] = {pooled[0].as<uint32_t>(), pooled[1].as<uint32_t>()};
        if ((size_t)slab * W * 4 > kStageBytes || !pin[0] || !pin[1])  // more than 256 words per observation
            for (int q = 0; q < 2; ++q) { HIP_TRY(own[q].alloc((size_t)slab * W * 4)); pin[q] = own[q].as<uint32_t>(); }
        EventPair done;
        HIP_TRY(done.create());
        HostCrew crew;
        std::vector<uint32_t*> plane((size_t)W);
        int64_t s = 0;
        for (int64_t i0 = 0; i0 < N; i0 += slab, ++s) {
            const int64_t rows = N - i0 < slab ? N - i0 : slab;
            uint32_t* const buf = pin[s & 1];
            if (s >= 2) HIP_TRY(hipEventSynchronize(done.e[s & 1]));  // its previous copy has left the buffer
            std::atomic<uint32_t> seen{0};
            for (int w = 0; w < W; ++w) plane[(size_t)w] = buf + (int64_t)w * slab;
            crew.run(rows, 32768, 64, [&](int64_t lo, int64_t hi) {
                seen.fetch_or(pack_rows_host(X, N, P, i0 + lo, i0 + hi, plane.data(), i0), std::memory_order_relaxed);
            });
            if (seen.load() & ~1u) {
                (void)hipStreamSynchronize(c->stream);
                return set_err(BMM_E_ARG, "data must be binary: X holds a value other than 0 and 1");
            }
            HIP_TRY(hipMemcpy2DAsync(c->dXb + i0, (size_t)N * 4, buf, (size_t)slab * 4, (size_t)rows * 4, (size_t)W,
                                     hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipEventRecord(done.e[s & 1], c->stream));
        }
        HIP_TRY(hipStreamSynchronize(c->stream));  // the staging buffers go out of scope
        if (c->dX_owned) { (void)hipFree(c->dX_owned); c->dX_owned = nullptr; }
        c->dX = nullptr;
        c->have_data = true;
        return BMM_OK;
    });
}

int bmm_chain_set_data_device(bmm_chain* c, const void* dX) {
    if (!c || !dX) return set_err(BMM_E_ARG, "null argument");
    if (c->started) return set_err(BMM_E_STATE, "chain already started");
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, dX) != hipSuccess || at.type != hipMemoryTypeDevice) {
        (void)hipGetLastError();
        return set_err(BMM_E_ARG, "dX is not a device pointer");
    }
    if (at.device != c->device) return set_err(BMM_E_ARG, "dX lives on device %d, chain on %d", at.device, c->device);
    HIP_TRY(hipSetDevice(c->device));
    // The chain validates and packs on its own non-blocking stream, which nothing orders after the
    // stream that produced dX (a torch kernel, an RCCL broadcast still in flight): wait for the device.
    HIP_TRY(hipDeviceSynchronize());
    const int32_t* x = static_cast<const int32_t*>(dX);
    int rc = validate_binary(c, x, c->p.N * c->p.P);
    if (rc) return rc;
    if (c->bits) {  // packed here and now; the caller's matrix is not read again
        const int W = (c->p.P + 31) / 32;
        if (!c->dXb) { rc = planes_alloc(c, (size_t)W * c->p.N); if (rc) return rc; }
        rc = pack_rows(c, x, c->p.N, c->p.N, 0);
        if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->dX = nullptr;
    } else {
        c->dX = x;  // borrowed for the life of the chain
    }
    c->have_data = true;
    return BMM_OK;
}

// ---- bit planes shared between chains, or filled by the caller (a broadcast) ----------------
int bmm_chain_share_data(bmm_chain* c, bmm_chain* from) {
    if (!c || !from) return set_err(BMM_E_ARG, "null argument");
    if (c->started || c->have_data) return set_err(BMM_E_STATE, "the chain already has its data");
    if (!from->have_data || !from->bits || !from->dXb) return set_err(BMM_E_STATE, "the source chain holds no bit planes");
    if (from->slot != c->slot) return set_err(BMM_E_ARG, "chains on different devices cannot share planes");
    if (from->p.N != c->p.N || from->p.P != c->p.P) return set_err(BMM_E_ARG, "the chains differ in N or P");
    if (!c->bits) return set_err(BMM_E_STATE, "the int32 layout streams the caller's matrix: hand it over instead");
    int rcq = chain_dedicated_queue(from);  // two chains on one device: a hardware queue each
    if (rcq == BMM_OK) rcq = chain_dedicated_queue(c);
    if (rcq) return rcq;
    for (bmm_chain* q : {from, c}) {  // and the kernel form that suits a shared device (pick_kernel)
        if (q->shares_device || q->started || q->generic) continue;
        q->shares_device = true;
        rcq = pick_kernel(q);
        if (rcq) return rcq;
    }
    c->planes = from->planes;
    c->planes->refs.fetch_add(1);
    c->dXb = from->dXb;
    c->xb_borrowed = true;
    c->dX = nullptr;
    c->have_data = true;
    return BMM_OK;
}

int bmm_chain_planes(bmm_chain* c, void** dXb, int64_t* n_words) {
    if (!c || !dXb) return set_err(BMM_E_ARG, "null argument");
    if (!c->bits) return set_err(BMM_E_STATE, "the chain streams the int32 layout: it has no bit planes");
    HIP_TRY(hipSetDevice(c->device));
    const int64_t words = (int64_t)((c->p.P + 31) / 32) * c->p.N;
    if (!c->dXb) {
        if (c->started) return set_err(BMM_E_STATE, "chain already started");
        int rcp = planes_alloc(c, (size_t)words);
        if (rcp) return rcp;
    }
    *dXb = c->dXb;
    if (n_words) *n_words = words;
    return BMM_OK;
}

int bmm_chain_planes_filled(bmm_chain* c) {
    if (!c) return set_err(BMM_E_ARG, "null chain");
    if (c->started) return set_err(BMM_E_STATE, "chain already started");
    if (!c->bits || !c->dXb) return set_err(BMM_E_STATE, "no planes to declare filled (bmm_chain_planes first)");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipDeviceSynchronize());  // whatever filled them ran on a stream of the caller's
    c->dX = nullptr;
    c->have_data = true;
    return BMM_OK;
}

int bmm_chain_set_initial_labels(bmm_chain* c, const int32_t* z1) {
    if (!c || !z1) return set_err(BMM_E_ARG, "null argument");
    if (c->p.mode != MODE_COLLAPSED) return set_err(BMM_E_STATE, "only the finite collapsed sampler takes initial labels");
    if (c->started) return set_err(BMM_E_STATE, "chain already started");
    // uploaded as R holds them (1-based), checked and shifted on the device: no host pass over N labels
    HIP_TRY(hipSetDevice(c->device));
    const int64_t N = c->p.N;
    DevBuf badbuf;
    HIP_TRY(badbuf.alloc(sizeof(unsigned long long)));
    HIP_TRY(hipMemsetAsync(badbuf.p, 0xff, sizeof(unsigned long long), c->stream));
    HIP_TRY(hipMemcpyAsync(c->dZ[1], z1, (size_t)N * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    const int64_t nb = (N + 255) / 256;
    hipLaunchKernelGGL(k_labels_from_r, dim3((unsigned)(nb < 4096 ? nb : 4096)), dim3(256), 0, c->stream, c->dZ[1], N,
                       c->p.K, c->dZ[0], badbuf.as<unsigned long long>());
    HIP_TRY(hipGetLastError());
    unsigned long long bad = 0;
    HIP_TRY(hipMemcpyAsync(&bad, badbuf.p, sizeof bad, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (bad != ~0ull)
        return set_err(BMM_E_ARG, "initialK[%lld] = %d outside 1..%d", (long long)bad, z1[bad], c->p.K);
    c->have_init = true;
    return BMM_OK;
}

int bmm_chain_set_initial_params(bmm_chain* c, const double* pi, const double* theta) {
    if (!c || !pi || !theta) return set_err(BMM_E_ARG, "null argument");
    if (!explicit_params(c->p.mode)) return set_err(BMM_E_STATE, "only the stick-breaking and full samplers take initial pi/theta");
    if (c->started) return set_err(BMM_E_STATE, "chain already started");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpyAsync(c->dPi, pi, (size_t)c->p.K * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->dTheta, theta, (size_t)c->p.K * c->p.P * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->have_init = true;
    return BMM_OK;
}

int bmm_chain_sweeps(bmm_chain* c, int n) {
    return guarded([&]() -> int {
        if (!c) return set_err(BMM_E_ARG, "null chain");
        if (n < 0) return set_err(BMM_E_ARG, "n must be >= 0");
        if (c->sharded && n > 0) return set_err(BMM_E_STATE, "a sharded chain advances by bmm_chain_shard_resample / _finish");
        HIP_TRY(hipSetDevice(c->device));
        if (!c->started) {
            int rc = chain_start(c);
            if (rc) return rc;
        }
        for (int t = 0; t < n; ++t) {
            int rc = enqueue_sweep(c, c->sweep + 1);
            if (rc) return rc;
            c->sweep++;
        }
        return BMM_OK;
    });
}

int bmm_chain_set_shard(bmm_chain* c, int64_t N_total, int64_t first_row) {
    if (!c) return set_err(BMM_E_ARG, "null chain");
    if (c->started) return set_err(BMM_E_STATE, "chain already started");
    if (!explicit_params(c->p.mode))
        return set_err(BMM_E_UNSUPPORTED, "only the stick-breaking and full samplers shard exactly over observations");
    if (first_row < 0 || N_total < first_row + c->p.N) return set_err(BMM_E_ARG, "shard [first_row, first_row + N) lies outside N_total");
    c->p.Ntot = N_total;
    c->p.obs0 = first_row;
    c->sharded = true;
    return BMM_OK;
}

static int shard_resample(bmm_chain* c, bool wait) {
    if (!c) return set_err(BMM_E_ARG, "null chain");
    if (!explicit_params(c->p.mode)) return set_err(BMM_E_UNSUPPORTED, "not a shardable sampler");
    if (c->shard_open) return set_err(BMM_E_STATE, "previous sweep not finished (bmm_chain_shard_finish)");
    HIP_TRY(hipSetDevice(c->device));
    if (!c->started) {
        int rc = chain_start(c);
        if (rc) return rc;
    }
    int rc = enqueue_sweep(c, c->sweep + 1, 1);
    if (rc) return rc;
    c->shard_open = true;
    if (wait) HIP_TRY(hipStreamSynchronize(c->stream));  // the deltas are complete: safe to reduce on any stream
    return BMM_OK;
}
int bmm_chain_shard_resample(bmm_chain* c) { return shard_resample(c, true); }
int bmm_chain_shard_resample_async(bmm_chain* c) { return shard_resample(c, false); }

int bmm_chain_stream(bmm_chain* c, void** stream) {
    if (!c || !stream) return set_err(BMM_E_ARG, "null argument");
    *stream = c->stream;
    return BMM_OK;
}

int bmm_chain_shard_deltas(bmm_chain* c, void** dNk, void** dS) {
    if (!c || !dNk || !dS) return set_err(BMM_E_ARG, "null argument");
    *dNk = c->dDNk;
    *dS = c->dDS;
    return BMM_OK;
}

int bmm_chain_shard_finish(bmm_chain* c) {
    if (!c) return set_err(BMM_E_ARG, "null chain");
    if (!c->shard_open) return set_err(BMM_E_STATE, "no sweep open (bmm_chain_shard_resample)");
    HIP_TRY(hipSetDevice(c->device));
    int rc = enqueue_sweep(c, c->sweep + 1, 2);
    if (rc) return rc;
    c->sweep++;
    c->shard_open = false;
    return BMM_OK;
}

int bmm_chain_sweep_probs(bmm_chain* c, double* probs_out) {
    if (!c || !probs_out) return set_err(BMM_E_ARG, "null argument");
    if (c->sharded) return set_err(BMM_E_STATE, "not available on a sharded chain");
    HIP_TRY(hipSetDevice(c->device));
    int rc = probs_alloc(c, true);
    if (rc) return rc;
    c->probs_dst = c->dProbs;
    rc = bmm_chain_sweeps(c, 1);
    c->probs_dst = nullptr;
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(probs_out, c->dProbs, (size_t)c->p.N * c->p.K * sizeof(double), hipMemcpyDeviceToHost,
                           c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return BMM_OK;
}

int bmm_chain_sweeps_counts(bmm_chain* c, int n, int32_t* nk_out) {
    if (!c || !nk_out) return set_err(BMM_E_ARG, "null argument");
    if (n < 0) return set_err(BMM_E_ARG, "n must be >= 0");
    if (n == 0) return BMM_OK;
    HIP_TRY(hipSetDevice(c->device));
    const size_t bytes = (size_t)n * c->p.K * sizeof(int32_t);
    HIP_TRY(hipMalloc(&c->dNkTrace, bytes));
    c->nk_trace_base = c->sweep + 1;
    int rc = bmm_chain_sweeps(c, n);
    if (rc == BMM_OK) {
        hipError_t e = hipMemcpyAsync(nk_out, c->dNkTrace, bytes, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) rc = set_err(BMM_E_HIP, "copying the count trace failed: %s", hipGetErrorString(e));
    } else {
        (void)hipStreamSynchronize(c->stream);
    }
    (void)hipFree(c->dNkTrace);
    c->dNkTrace = nullptr;
    return rc;
}

int bmm_chain_sync(bmm_chain* c) {
    if (!c) return set_err(BMM_E_ARG, "null chain");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (size_t i = 0; i + 1 < c->ev_used; i += 2) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, c->ev[i], c->ev[i + 1]));
        c->prof_ms += ms;
        c->prof_n++;
    }
    c->ev_used = 0;
    return dbg_labels_ok(c);
}

int bmm_chain_sweep_index(const bmm_chain* c) { return c ? c->sweep : -1; }

int bmm_chain_get_labels(bmm_chain* c, int32_t* z1) {
    if (!c || !z1) return set_err(BMM_E_ARG, "null argument");
    int rc = bmm_chain_sync(c);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(z1, label_row(c, c->sweep), (size_t)c->p.N * sizeof(int32_t), hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < c->p.N; ++i) z1[i] = z1[i] < 0 ? BMM_NA_INTEGER : z1[i] + 1;
    return BMM_OK;
}

int bmm_chain_get_counts(bmm_chain* c, int32_t* Nk, int32_t* S) {
    return guarded([&]() -> int {
        if (!c || !Nk || !S) return set_err(BMM_E_ARG, "null argument");
        HIP_TRY(hipSetDevice(c->device));
        int rc = launch_reduce_deltas(c);
        if (rc) return rc;
        rc = bmm_chain_sync(c);
        if (rc) return rc;
        const size_t K = (size_t)c->p.K, KP = K * c->p.P;
        std::vector<int32_t> d(KP > K ? KP : K);
        HIP_TRY(hipMemcpy(Nk, c->dNk, K * sizeof(int32_t), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(d.data(), c->dDNk, K * sizeof(int32_t), hipMemcpyDeviceToHost));
        for (size_t k = 0; k < K; ++k) Nk[k] += d[k];
        HIP_TRY(hipMemcpy(S, c->dS, KP * sizeof(int32_t), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(d.data(), c->dDS, KP * sizeof(int32_t), hipMemcpyDeviceToHost));
        for (size_t q = 0; q < KP; ++q) S[q] += d[q];
        return BMM_OK;
    });
}

int bmm_chain_get_alpha(bmm_chain* c, double* alpha) {
    if (!c || !alpha) return set_err(BMM_E_ARG, "null argument");
    int rc = bmm_chain_sync(c);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(alpha, c->dAlpha, sizeof(double), hipMemcpyDeviceToHost));
    return BMM_OK;
}

int bmm_chain_get_params(bmm_chain* c, double* pi, double* theta) {
    if (!c || !pi || !theta) return set_err(BMM_E_ARG, "null argument");
    if (!explicit_params(c->p.mode)) return set_err(BMM_E_STATE, "only the stick-breaking and full samplers carry pi/theta");
    int rc = bmm_chain_sync(c);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(pi, c->dPi, (size_t)c->p.K * sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(theta, c->dTheta, (size_t)c->p.K * c->p.P * sizeof(double), hipMemcpyDeviceToHost));
    return BMM_OK;
}

int bmm_chain_profile(bmm_chain* c, int enable) {
    if (!c) return set_err(BMM_E_ARG, "null chain");
    int rc = bmm_chain_sync(c);
    if (rc) return rc;
    c->prof = enable > 0 ? enable : 0;
    c->prof_ms = 0.0;
    c->prof_n = 0;
    return BMM_OK;
}

int bmm_chain_profile_read(bmm_chain* c, double* resample_ms, int64_t* resample_launches) {
    if (!c) return set_err(BMM_E_ARG, "null chain");
    int rc = bmm_chain_sync(c);
    if (rc) return rc;
    if (resample_ms) *resample_ms = c->prof_ms;
    if (resample_launches) *resample_launches = c->prof_n;
    return BMM_OK;
}

int64_t bmm_chain_batch(const bmm_chain* c) { return c ? c->batch : -1; }

int bmm_chain_kernel_shape(const bmm_chain* c, int* lds_bytes, int* threads, int* grid_max) {
    if (!c) return set_err(BMM_E_ARG, "null chain");
    if (lds_bytes) *lds_bytes = (int)c->lds_bytes;
    if (threads) *threads = c->NT;
    if (grid_max) *grid_max = c->grid_max;
    return BMM_OK;
}

int bmm_chain_kernel_form(const bmm_chain* c, int* lanes_per_observation, int* builds_own_tables) {
    if (!c) return set_err(BMM_E_ARG, "null chain");
    if (lanes_per_observation) *lanes_per_observation = c->generic || c->OT <= 0 ? 1 : c->NT / c->OT;
    if (builds_own_tables) *builds_own_tables = c->self_tables ? 1 : 0;
    return BMM_OK;
}

}  // extern "C"

// ------------------------------------------------------------------ *_run entry points
namespace {

struct RunIO {
    const int32_t* z0 = nullptr;
    const double *pi0 = nullptr, *theta0 = nullptr;
    double* pi_out = nullptr;
    int32_t* z_out = nullptr;
    double *theta_out = nullptr, *alpha_out = nullptr;
};

// Blocks of the outgoing label trace (trace_out): a whole number of observations, at most a staging piece
size_t out_block_rows(int S, size_t el, int64_t N) {
    int64_t B = (int64_t)(kStageBytes / ((size_t)S * el));
    B = B / 32 * 32;
    if (B < 32) B = 32;
    return (size_t)(B > N ? N : B);
}

// buffers of a *_run call: the traces and the two device blocks of the outgoing label trace, one allocation
// (owned by the chain, released with it)
int run_prepare(bmm_chain* c, int nsamples, int burnin) {
    const int64_t N = c->p.N;
    const int K = c->p.K, P = c->p.P;
    const int S = nsamples - burnin;
    c->burnin = burnin; c->S = S;
    HIP_TRY(hipSetDevice(c->device));
    const size_t el = K <= 254 ? 1 : 4;
    c->out_blk_bytes = out_block_rows(S, el, N) * (size_t)S * el;
    auto carve = [&](Carver& a) {
        c->dThetaTrace = a.take<double>((size_t)S * K * P);
        c->dAlphaTrace = a.take<double>((size_t)S);
        c->dPiTrace = explicit_params(c->p.mode) ? a.take<double>((size_t)S * K) : nullptr;
        c->dOutBlk[0] = a.take<char>(c->out_blk_bytes);
        c->dOutBlk[1] = a.take<char>(c->out_blk_bytes);
        c->dTrace = a.take<int32_t>((size_t)S * N);
    };
    Carver measure{nullptr};
    carve(measure);
    c->run_arena = static_cast<char*>(dev_pool().get(c->device, measure.used, &c->run_arena_bytes));
    if (!c->run_arena) return set_err(BMM_E_HIP, "allocating the run's buffers failed: %s", hipGetErrorString(hipGetLastError()));
    Carver real{c->run_arena};
    carve(real);
    return BMM_OK;
}

// The sweeps of a run whose allocation probabilities go to the host's relabelling code
// (collapsed_gibbs.cpp:162-172, 187-201): sweeps burnin - burnrelabel .. burnin - 1 fill the batch
// cube (through a device ring when it fits, so that those sweeps need no host round trip), then
// every kept sweep hands its N x K matrix to on_sample -- two device and two pinned host buffers,
// so that sweep j + 1 runs while the host works on sweep j.
int run_sweeps_hooked(bmm_chain* c, int nsamples, const bmm_relabel_hooks* h) {
    const int64_t N = c->p.N;
    const int K = c->p.K, burnin = c->burnin;
    const size_t mat = (size_t)N * K, mat_bytes = mat * sizeof(double);
    // the cube has burnrelabel slices, sweep j in slice j - burnin + burnrelabel (collapsed_gibbs.cpp:163-166);
    // slices of sweeps that do not exist (j < 1: a window longer than the burn-in, which only the
    // stick-breaking wrapper lets through, R/utils.R:97-101) stay zero, as arma::fill::zeros leaves them
    const int W = h->burnrelabel < 0 ? 0 : h->burnrelabel;
    if (W > 0 && !h->probs_batch) return set_err(BMM_E_ARG, "relabel hooks: probs_batch is null");
    const int first_window = burnin - W;  // sweep of slice 0
    for (int q = 0; q < W && q + first_window < 1; ++q) std::memset(h->probs_batch + (size_t)q * mat, 0, mat_bytes);
    int rc = probs_alloc(c, false);
    if (rc) return rc;
    // device ring for the batch window when it takes at most half of the free memory
    DevBuf ring;
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    const bool use_ring = W > 0 && mat_bytes * (size_t)W <= free_b / 2;
    if (use_ring) {
        HIP_TRY(ring.alloc(mat_bytes * (size_t)W));
        HIP_TRY(hipMemsetAsync(ring.p, 0, mat_bytes * (size_t)W, c->stream));
    }
    DevBuf dbuf[2];
    double* hbuf[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    struct Pinned { double** h; hipEvent_t* e; ~Pinned() { for (int q = 0; q < 2; ++q) { if (h[q]) (void)hipHostFree(h[q]); if (e[q]) (void)hipEventDestroy(e[q]); } } } pin{hbuf, ev};
    for (int q = 0; q < 2; ++q) {
        HIP_TRY(dbuf[q].alloc(mat_bytes));
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&hbuf[q]), mat_bytes, hipHostMallocDefault));
        HIP_TRY(hipEventCreateWithFlags(&ev[q], hipEventDisableTiming));
    }
    int pending = -1;  // sweep whose matrix is on its way to hbuf[pending & 1]
    auto deliver = [&](int jj) -> int {
        HIP_TRY(hipEventSynchronize(ev[jj & 1]));
        if (jj < burnin) {  // a window sweep without the ring: straight into the cube
            std::memcpy(h->probs_batch + (size_t)(jj - first_window) * mat, hbuf[jj & 1], mat_bytes);
            return BMM_OK;
        }
        if (h->on_sample && h->on_sample(h->user, jj, hbuf[jj & 1]) != 0)
            return set_err(BMM_E_CALLBACK, "relabel hook on_sample stopped the run at sweep %d", jj);
        return BMM_OK;
    };
    for (int j = 1; j < nsamples; ++j) {
        const bool window = j >= first_window && j < burnin;
        const bool keep = j >= burnin && h->on_sample;
        if (window && use_ring) c->probs_dst = ring.as<double>() + (size_t)(j - first_window) * mat;
        else if (window || keep) c->probs_dst = dbuf[j & 1].as<double>();
        else c->probs_dst = nullptr;
        rc = bmm_chain_sweeps(c, 1);
        const bool staged = c->probs_dst && !(window && use_ring);
        c->probs_dst = nullptr;
        if (rc) return rc;
        // progress of a run that feeds the host's relabelling: as the sweeps are enqueued (the hand-off keeps the
        // host within a sweep of the device from the window on); the cluster count is not fetched here
        if (g_progress.fn && g_progress.every > 0 && (j % g_progress.every == 0 || j == nsamples - 1) &&
            g_progress.fn(g_progress.user, j + 1, nsamples, -1) != 0) {
            (void)hipStreamSynchronize(c->stream);
            return set_err(BMM_E_CALLBACK, "the progress hook stopped the run after sample %d", j + 1);
        }
        if (staged) {
            HIP_TRY(hipMemcpyAsync(hbuf[j & 1], dbuf[j & 1].p, mat_bytes, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipEventRecord(ev[j & 1], c->stream));
        }
        if (pending >= 0) { rc = deliver(pending); pending = -1; if (rc) return rc; }
        if (staged) pending = j;
        if (j == burnin - 1) {  // collapsed_gibbs.cpp:188-190
            if (pending >= 0) { rc = deliver(pending); pending = -1; if (rc) return rc; }
            if (use_ring) {  // slices of sweeps j < 1 arrive as the zeros the ring was filled with
                HIP_TRY(hipMemcpyAsync(h->probs_batch, ring.p, mat_bytes * (size_t)W, hipMemcpyDeviceToHost, c->stream));
                HIP_TRY(hipStreamSynchronize(c->stream));
            }
            if (h->batch_done && h->batch_done(h->user, j, h->probs_batch) != 0)
                return set_err(BMM_E_CALLBACK, "relabel hook batch_done stopped the run");
        }
    }
    if (pending >= 0) { rc = deliver(pending); if (rc) return rc; }
    return BMM_OK;
}

// One-byte labels widened into the caller's int32 trace (0 = unassigned -> NA_integer_), with streaming
// stores: the destination is written once and not read again by this call, so its lines need not be fetched
// first (that read-for-ownership doubles the memory traffic of a plain store loop).
void widen_labels(const uint8_t* src, int32_t* dst, int64_t n) {
    int64_t q = 0;
    for (; q < n && (reinterpret_cast<uintptr_t>(dst + q) & 15); ++q) dst[q] = src[q] ? (int32_t)src[q] : BMM_NA_INTEGER;
    const __m128i zero = _mm_setzero_si128(), na = _mm_set1_epi32(BMM_NA_INTEGER);
    for (; q + 16 <= n; q += 16) {
        const __m128i v = _mm_loadu_si128(reinterpret_cast<const __m128i*>(src + q));
        const __m128i lo = _mm_unpacklo_epi8(v, zero), hi = _mm_unpackhi_epi8(v, zero);
        __m128i w[4] = {_mm_unpacklo_epi16(lo, zero), _mm_unpackhi_epi16(lo, zero), _mm_unpacklo_epi16(hi, zero), _mm_unpackhi_epi16(hi, zero)};
        for (int k = 0; k < 4; ++k) {
            w[k] = _mm_or_si128(w[k], _mm_and_si128(_mm_cmpeq_epi32(w[k], zero), na));
            _mm_stream_si128(reinterpret_cast<__m128i*>(dst + q + 4 * k), w[k]);
        }
    }
    for (; q < n; ++q) dst[q] = src[q] ? (int32_t)src[q] : BMM_NA_INTEGER;
    _mm_sfence();
}
void copy_labels(const int32_t* src, int32_t* dst, int64_t n) {  // the same for labels that travel as int32
    int64_t q = 0;
    for (; q < n && (reinterpret_cast<uintptr_t>(dst + q) & 15); ++q) dst[q] = src[q];
    for (; q + 4 <= n; q += 4)
        _mm_stream_si128(reinterpret_cast<__m128i*>(dst + q), _mm_loadu_si128(reinterpret_cast<const __m128i*>(src + q)));
    for (; q < n; ++q) dst[q] = src[q];
    _mm_sfence();
}

// The label trace out to the caller's S x N column-major matrix (labels 1-based, NA where unassigned).
// In that layout the S labels of one observation are contiguous, so a block of observations is a contiguous
// run of the caller's buffer: the device transposes the [S][N] trace block by block (k_trace_block), each
// block goes to pinned staging, and the host's cores copy it into place while the next block is on its way --
// every byte of the (pageable) destination is written once, in order, at the speed of the slower of PCIe
// and the host's memory.  Up to 254 labels travel as one byte each and are widened by that host copy (a
// quarter of the PCIe bytes).  Nothing of it can start before the last sweep has finished: each
// observation's run is complete only then.
int trace_out(bmm_chain* c, int32_t* z_out, PhaseClock* clock) {
    const int64_t N = c->p.N;
    const int S = c->S;
    const bool narrow = c->p.K <= 254;
    const size_t el = narrow ? 1 : 4;
    const int64_t B = (int64_t)out_block_rows(S, el, N);
    const size_t blk_bytes = (size_t)B * S * el;  // = c->out_blk_bytes
    // host staging: two pieces of the pool; a trace with so many kept sweeps that 32 observations exceed a
    // piece gets buffers of its own
    Stage pooled[2];
    PinnedBuf own[2];
    void* hblk[2] = {pooled[0].p, pooled[1].p};
    if (blk_bytes > kStageBytes || !hblk[0] || !hblk[1])
        for (int q = 0; q < 2; ++q) { HIP_TRY(own[q].alloc(blk_bytes)); hblk[q] = own[q].p; }
    EventPair ev;
    HIP_TRY(ev.create());
    std::unique_ptr<HostCrew> own_crew;  // a resident chain's trace (no *_run call around it)
    HostCrew* crew = c->crew;
    if (!crew) { own_crew.reset(new HostCrew()); crew = own_crew.get(); }
    // The caller's S x N matrix is usually fresh (R's allocMatrix, numpy.empty): its pages are not there yet, and
    // taking the page faults inside the widening loop makes that loop fault-bound (1.08 GB at C5: 14-21 ms).  The
    // host is idle while the device still runs the sweeps enqueued ahead of this call's blocks, so the crew takes
    // the faults now (MADV_POPULATE_WRITE, Linux 5.14; anything else it answers is ignored: the widening then
    // faults as before).  Every cell of the matrix is overwritten below.
    {
        const uintptr_t page = 4096;
        const uintptr_t lo = (reinterpret_cast<uintptr_t>(z_out) + page - 1) & ~(page - 1);
        const uintptr_t hi = (reinterpret_cast<uintptr_t>(z_out) + (size_t)S * (size_t)N * sizeof(int32_t)) & ~(page - 1);
        if (hi > lo + (64 << 20) / 64) {  // from a megabyte up
            const int64_t pages = (int64_t)((hi - lo) / page);
            crew->begin(pages, 256, 1, [lo, page](int64_t a, int64_t b) {
                (void)madvise(reinterpret_cast<void*>(lo + (uintptr_t)a * page), (size_t)(b - a) * page, MADV_POPULATE_WRITE);
            });
        }
    }
    auto consume = [&](int64_t blk) {
        const int64_t i0 = blk * B, rows = N - i0 < B ? N - i0 : B;
        int32_t* const dst = z_out + (size_t)i0 * S;
        const int64_t n = rows * S;
        if (narrow) {
            const uint8_t* const src = static_cast<const uint8_t*>(hblk[blk & 1]);
            crew->run(n, 1 << 16, 64, [=](int64_t lo, int64_t hi) { widen_labels(src + lo, dst + lo, hi - lo); });
        } else {
            const int32_t* const src = static_cast<const int32_t*>(hblk[blk & 1]);
            crew->run(n, 1 << 16, 64, [=](int64_t lo, int64_t hi) { copy_labels(src + lo, dst + lo, hi - lo); });
        }
    };
    const int64_t nblk = (N + B - 1) / B;
    for (int64_t blk = 0; blk < nblk; ++blk) {
        const int64_t i0 = blk * B, rows = N - i0 < B ? N - i0 : B;
        const dim3 grid((unsigned)((rows + 31) / 32), (unsigned)((S + 31) / 32));
        char* const dblk = c->dOutBlk[blk & 1];
        // block blk - 2 has been consumed (below) before this iteration started: its staging is free
        if (narrow) hipLaunchKernelGGL(k_trace_block<uint8_t>, grid, dim3(256), 0, c->stream, c->dTrace, N, S, i0, rows, reinterpret_cast<uint8_t*>(dblk));
        else hipLaunchKernelGGL(k_trace_block<int32_t>, grid, dim3(256), 0, c->stream, c->dTrace, N, S, i0, rows, reinterpret_cast<int32_t*>(dblk));
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(hblk[blk & 1], dblk, (size_t)rows * S * el, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipEventRecord(ev.e[blk & 1], c->stream));
        if (blk >= 1) {
            HIP_TRY(hipEventSynchronize(ev.e[(blk - 1) & 1]));
            if (blk == 1 && clock) clock->lap(3);  // the first block has arrived: the sweeps are over
            consume(blk - 1);
        }
    }
    HIP_TRY(hipEventSynchronize(ev.e[(nblk - 1) & 1]));
    if (nblk == 1 && clock) clock->lap(3);
    consume(nblk - 1);
    HIP_TRY(hipStreamSynchronize(c->stream));  // the small traces queued ahead of the blocks
    return BMM_OK;
}

// The sweeps of a run that reports its progress (bmm_set_progress): everything is still enqueued ahead of
// the device, an event after every `every`-th sweep; the calling thread then follows the events and calls
// the hook as each one is reached.  For the DP sampler the cluster sizes at those sweeps ride along (K
// int32 into pinned memory) so that the hook gets the number of clusters in use, as the reference prints
// it (collapsed_gibbs_dp.cpp:99).
int run_sweeps_reported(bmm_chain* c, int nsamples) {
    const int every = g_progress.every, K = c->p.K;
    const int total = nsamples - 1, marks = (total + every - 1) / every;
    if (marks < 1) return BMM_OK;
    std::vector<hipEvent_t> evs((size_t)marks, nullptr);
    struct Free { std::vector<hipEvent_t>& v; ~Free() { for (hipEvent_t e : v) if (e) (void)hipEventDestroy(e); } } fr{evs};
    PinnedBuf sizes;
    const bool dp = c->p.mode == MODE_DP;
    if (dp) HIP_TRY(sizes.alloc((size_t)marks * K * sizeof(int32_t)));
    for (int m = 0; m < marks; ++m) {
        const int n = total - m * every < every ? total - m * every : every;
        int rc = bmm_chain_sweeps(c, n);
        if (rc) return rc;
        if (dp) HIP_TRY(hipMemcpyAsync(sizes.as<int32_t>() + (size_t)m * K, c->dNk, (size_t)K * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipEventCreateWithFlags(&evs[(size_t)m], hipEventDisableTiming));
        HIP_TRY(hipEventRecord(evs[(size_t)m], c->stream));
    }
    for (int m = 0; m < marks; ++m) {
        HIP_TRY(hipEventSynchronize(evs[(size_t)m]));
        const int done = (m + 1) * every < total ? (m + 1) * every : total;  // sweeps finished: j = 1 .. done
        int used = -1;
        if (dp) { used = 0; for (int k = 0; k < K; ++k) used += sizes.as<int32_t>()[(size_t)m * K + k] > 0; }
        if (g_progress.fn(g_progress.user, done + 1, nsamples, used) != 0) {
            (void)hipStreamSynchronize(c->stream);
            return set_err(BMM_E_CALLBACK, "the progress hook stopped the run after sample %d", done + 1);
        }
    }
    return BMM_OK;
}

// the starting state of a run: initial labels or parameters, and trace row 0 when burnin = 0
int run_start_state(bmm_chain* c, const RunIO& io) {
    const int sampler = c->p.mode, K = c->p.K, P = c->p.P, S = c->S, burnin = c->burnin;
    const int64_t N = c->p.N;
    HIP_TRY(hipSetDevice(c->device));
    int rc = BMM_OK;
    if (sampler == BMM_SAMPLER_COLLAPSED) rc = bmm_chain_set_initial_labels(c, io.z0);
    if (explicit_params(sampler)) rc = bmm_chain_set_initial_params(c, io.pi0, io.theta0);
    if (rc) return rc;
    if (burnin == 0) {
        // trace row 0 (DESIGN.md "Quirks"): labels = initial allocation or unassigned (-1 -> NA),
        // theta = NaN (collapsed, never written), 0 (dp, zero-filled) or the initial theta (sb)
        std::vector<double> t0((size_t)K * P, sampler == BMM_SAMPLER_DP ? 0.0 : std::nan(""));
        if (explicit_params(sampler)) std::memcpy(t0.data(), io.theta0, t0.size() * sizeof(double));
        HIP_TRY(hipMemcpy(c->dThetaTrace, t0.data(), t0.size() * sizeof(double), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(c->dAlphaTrace, &c->alpha0, sizeof(double), hipMemcpyHostToDevice));
        if (explicit_params(sampler))
            HIP_TRY(hipMemcpy2D(c->dPiTrace, (size_t)S * sizeof(double), io.pi0, sizeof(double), sizeof(double), K, hipMemcpyHostToDevice));
        if (sampler != BMM_SAMPLER_COLLAPSED) HIP_TRY(hipMemset(c->dTrace, 0xff, (size_t)N * sizeof(int32_t)));
    }
    return BMM_OK;
}

// the sweeps, then the traces out (data and starting state are in place)
int run_body(bmm_chain* c, int nsamples, const RunIO& io, const bmm_relabel_hooks* hooks) {
    const int sampler = c->p.mode, K = c->p.K, P = c->p.P, S = c->S;
    HIP_TRY(hipSetDevice(c->device));
    PhaseClock clock;
    int rc = hooks ? run_sweeps_hooked(c, nsamples, hooks)
                   : (g_progress.fn && g_progress.every > 0 ? run_sweeps_reported(c, nsamples) : bmm_chain_sweeps(c, nsamples - 1));
    if (rc) return rc;
    clock.lap(2);
    // theta, alpha, pi: small, queued behind the sweeps
    HIP_TRY(hipMemcpyAsync(io.theta_out, c->dThetaTrace, (size_t)S * K * P * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(io.alpha_out, c->dAlphaTrace, (size_t)S * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (explicit_params(sampler))
        HIP_TRY(hipMemcpyAsync(io.pi_out, c->dPiTrace, (size_t)S * K * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    rc = trace_out(c, io.z_out, &clock);
    if (rc) return rc;
    clock.lap(4);
    return dbg_labels_ok(c);
}

// planes packed on the host (AsyncPack: one pointer per plane, pinned pieces or ordinary memory) into the chain
int chain_set_planes_host(bmm_chain* c, uint32_t* const* plane) {
    const int W = (c->p.P + 31) / 32;
    HIP_TRY(hipSetDevice(c->device));
    if (!c->dXb) { int rcp = planes_alloc(c, (size_t)W * c->p.N); if (rcp) return rcp; }
    for (int w = 0; w < W; ++w)
        HIP_TRY(hipMemcpyAsync(c->dXb + (size_t)w * c->p.N, plane[w], (size_t)c->p.N * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->dX = nullptr;
    c->have_data = true;
    return BMM_OK;
}

#ifdef BMM_DEBUG_HOOKS
// Test variant only: the host-side ends of a run on their own, no device involved (tests/test_capi_cpu.py
// holds them to numpy): X packed into planes [w][N] by the crew, and a block of one-byte / int32 labels
// copied into an int32 trace.
extern "C" int bmm_dbg_host_pack(const int32_t* X, int64_t N, int P, uint32_t* out, uint32_t* seen_out) {
    return guarded([&]() -> int {
        const int W = (P + 31) / 32;
        std::vector<uint32_t*> plane((size_t)W);
        for (int w = 0; w < W; ++w) plane[(size_t)w] = out + (size_t)w * (size_t)N;
        std::atomic<uint32_t> seen{0};
        HostCrew crew;
        crew.run(N, 1024, 64, [&](int64_t lo, int64_t hi) {
            seen.fetch_or(pack_rows_host(X, N, P, lo, hi, plane.data(), 0), std::memory_order_relaxed);
        });
        *seen_out = seen.load();
        return BMM_OK;
    });
}
extern "C" int bmm_dbg_host_labels(const void* src, int narrow, int32_t* dst, int64_t n, int jobs) {
    return guarded([&]() -> int {
        HostCrew crew;
        for (int j = 0; j < jobs; ++j) {  // the same crew, job after job, as trace_out uses it
            if (narrow) crew.run(n, 256, 64, [=](int64_t lo, int64_t hi) { widen_labels(static_cast<const uint8_t*>(src) + lo, dst + lo, hi - lo); });
            else crew.run(n, 256, 64, [=](int64_t lo, int64_t hi) { copy_labels(static_cast<const int32_t*>(src) + lo, dst + lo, hi - lo); });
        }
        return BMM_OK;
    });
}
#endif

int check_run_args(const int32_t* X, int nsamples, int burnin, const RunIO& io, int sampler) {
    if (!X || !io.z_out || !io.theta_out || !io.alpha_out) return set_err(BMM_E_ARG, "null buffer");
    if (sampler == BMM_SAMPLER_COLLAPSED && !io.z0) return set_err(BMM_E_ARG, "initialK is null");
    if (explicit_params(sampler) && (!io.pi0 || !io.theta0 || !io.pi_out)) return set_err(BMM_E_ARG, "null buffer");
    if (nsamples < 1) return set_err(BMM_E_ARG, "nsamples must be >= 1");
    if (burnin < 0 || burnin >= nsamples) return set_err(BMM_E_ARG, "burnin must be in [0, nsamples)");
    return BMM_OK;
}

int run_chain(int sampler, const int32_t* X, int64_t N, int P, int nsamples, int K, double alpha, double beta,
              double gamma, double a, double b, int burnin, int64_t batch, uint64_t seed, int device,
              const RunIO& io, const bmm_relabel_hooks* hooks) {
    return guarded([&]() -> int {
        int rc = check_run_args(X, nsamples, burnin, io, sampler);
        if (rc) return rc;
        for (double& v : g_phase_ms) v = 0.0;
        PhaseClock clock;
        // The host's cores start validating and packing X at once (bit planes, the default layout) while this
        // thread creates the chain, allocates the run's buffers and uploads the starting state: the two take
        // about as long at the north-star shape, and only the planes -- 4 * ceil(P/32) bytes per observation
        // instead of 4 * P -- then cross PCIe.
        HostCrew crew;  // the host threads of this call: they pack X now and widen the label trace at the end
        AsyncPack pack;
        const bool host_pack = dbg_env("BMM_X_LAYOUT_INT32") == nullptr;
        if (host_pack) pack.start(&crew, X, N, P);
        bmm_chain* c = nullptr;
        rc = bmm_chain_create(&c, sampler, N, P, K, alpha, beta, gamma, a, b, batch, seed, device);
        if (rc) return rc;
        c->crew = &crew;
        {
            struct Guard { bmm_chain* c; ~Guard() { bmm_chain_destroy(c); } } guard{c};
            rc = run_prepare(c, nsamples, burnin);
            if (rc == BMM_OK) rc = run_start_state(c, io);
            if (rc) return rc;
            clock.lap(1);
            pack.join();
            if (host_pack && c->bits) {
                if (pack.seen.load() & ~1u) return set_err(BMM_E_ARG, "data must be binary: X holds a value other than 0 and 1");
                rc = chain_set_planes_host(c, pack.plane.data());
                pack.release();
            } else {  // the int32 layout (test variant): the resident API's way
                pack.release();
                rc = bmm_chain_set_data_host(c, X);
            }
            if (rc) return rc;
            clock.lap(0);
            rc = run_body(c, nsamples, io, hooks);
            clock.t = std::chrono::steady_clock::now();
        }
        clock.lap(5);  // releasing the chain
        return rc;
    });
}

// ---- RCCL, opened on demand: only a run that spans devices needs it, and a process that has
// torch loaded must end up with one copy of the library (dlopen by soname finds a loaded one)
struct Rccl {
    void* lib = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
int rccl_open(Rccl& r) {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (r.lib) break;
    }
    if (!r.lib) return set_err(BMM_E_RCCL, "RCCL is needed to broadcast the data across devices and could not be opened: %s", dlerror());
    r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(dlsym(r.lib, "ncclCommInitAll"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.lib, "ncclCommDestroy"));
    r.Broadcast = reinterpret_cast<decltype(r.Broadcast)>(dlsym(r.lib, "ncclBroadcast"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(dlsym(r.lib, "ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(dlsym(r.lib, "ncclGroupEnd"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.lib, "ncclGetErrorString"));
    if (!r.CommInitAll || !r.CommDestroy || !r.Broadcast || !r.GroupStart || !r.GroupEnd || !r.GetErrorString)
        return set_err(BMM_E_RCCL, "the RCCL library lacks an expected symbol");
    return BMM_OK;
}

// One broadcast of `count` 32-bit words from bufs[0] (on devs[0]) into bufs[r] on devs[r], over a
// communicator built in this process (ncclCommInitAll): the only collective of the chain path.
int rccl_broadcast_words(const std::vector<int>& devs, const std::vector<void*>& bufs, size_t count) {
    const int n = (int)devs.size();
    if (n < 2) return BMM_OK;
    if (fake_devices()) {  // test variant: the "devices" are one device; the broadcast is a copy per receiver
        for (int q = 1; q < n; ++q) HIP_TRY(hipMemcpy(bufs[(size_t)q], bufs[0], count * sizeof(uint32_t), hipMemcpyDeviceToDevice));
        HIP_TRY(hipDeviceSynchronize());
        return BMM_OK;
    }
    Rccl r;
    int rc = rccl_open(r);
    if (rc) return rc;
    std::vector<ncclComm_t> comms((size_t)n, nullptr);
    ncclResult_t e = r.CommInitAll(comms.data(), n, devs.data());
    if (e != ncclSuccess) return set_err(BMM_E_RCCL, "ncclCommInitAll failed: %s", r.GetErrorString(e));
    std::vector<hipStream_t> st((size_t)n, nullptr);
    hipError_t he = hipSuccess;
    for (int q = 0; q < n && he == hipSuccess; ++q) {
        he = hipSetDevice(devs[(size_t)q]);
        if (he == hipSuccess) he = hipStreamCreateWithFlags(&st[(size_t)q], hipStreamNonBlocking);
    }
    if (he == hipSuccess) {
        e = r.GroupStart();
        // every rank passes its own buffer, in place (the root's holds the data; a send buffer is read
        // on the root only, and a pointer of another device must not be handed to a rank's call)
        for (int q = 0; q < n && e == ncclSuccess; ++q)
            e = r.Broadcast(bufs[(size_t)q], bufs[(size_t)q], count, ncclUint32, 0, comms[(size_t)q], st[(size_t)q]);
        const ncclResult_t e2 = r.GroupEnd();
        if (e == ncclSuccess) e = e2;
        for (int q = 0; q < n && he == hipSuccess; ++q) {
            he = hipSetDevice(devs[(size_t)q]);
            if (he == hipSuccess) he = hipStreamSynchronize(st[(size_t)q]);
        }
    }
    for (int q = 0; q < n; ++q) {
        if (st[(size_t)q]) { (void)hipSetDevice(devs[(size_t)q]); (void)hipStreamDestroy(st[(size_t)q]); }
        if (comms[(size_t)q]) (void)r.CommDestroy(comms[(size_t)q]);
    }
    if (he != hipSuccess) return set_err(BMM_E_HIP, "broadcasting the data failed: %s", hipGetErrorString(he));
    if (e != ncclSuccess) return set_err(BMM_E_RCCL, "ncclBroadcast failed: %s", r.GetErrorString(e));
    return BMM_OK;
}

}  // namespace

extern "C" {

int bmm_collapsed_run(const int32_t* X, int64_t N, int P, const int32_t* initialK, int nsamples, int K,
                      double alpha, double beta, double gamma, double a, double b, int burnin,
                      int64_t batch, uint64_t seed, int device, int32_t* z_out, double* theta_out,
                      double* alpha_out) {
    return bmm_collapsed_run_probs(X, N, P, initialK, nsamples, K, alpha, beta, gamma, a, b, burnin, batch, seed,
                                   device, z_out, theta_out, alpha_out, nullptr);
}
int bmm_collapsed_run_probs(const int32_t* X, int64_t N, int P, const int32_t* initialK, int nsamples, int K,
                            double alpha, double beta, double gamma, double a, double b, int burnin,
                            int64_t batch, uint64_t seed, int device, int32_t* z_out, double* theta_out,
                            double* alpha_out, const bmm_relabel_hooks* hooks) {
    RunIO io; io.z0 = initialK; io.z_out = z_out; io.theta_out = theta_out; io.alpha_out = alpha_out;
    return run_chain(BMM_SAMPLER_COLLAPSED, X, N, P, nsamples, K, alpha, beta, gamma, a, b, burnin, batch, seed,
                     device, io, hooks);
}

int bmm_dp_run(const int32_t* X, int64_t N, int P, int nsamples, double alpha, double beta, double gamma,
               double a, double b, int burnin, int maxK, int64_t batch, uint64_t seed, int device,
               int32_t* z_out, double* theta_out, double* alpha_out) {
    return bmm_dp_run_probs(X, N, P, nsamples, alpha, beta, gamma, a, b, burnin, maxK, batch, seed, device, z_out,
                            theta_out, alpha_out, nullptr);
}
int bmm_dp_run_probs(const int32_t* X, int64_t N, int P, int nsamples, double alpha, double beta, double gamma,
                     double a, double b, int burnin, int maxK, int64_t batch, uint64_t seed, int device,
                     int32_t* z_out, double* theta_out, double* alpha_out, const bmm_relabel_hooks* hooks) {
    RunIO io; io.z_out = z_out; io.theta_out = theta_out; io.alpha_out = alpha_out;
    return run_chain(BMM_SAMPLER_DP, X, N, P, nsamples, maxK, alpha, beta, gamma, a, b, burnin, batch, seed, device,
                     io, hooks);
}

int bmm_sb_run(const int32_t* X, int64_t N, int P, const double* initialPi, const double* initialTheta,
               int nsamples, int maxK, double alpha, double beta, double gamma, double a, double b,
               int burnin, uint64_t seed, int device, double* pi_out, int32_t* z_out, double* theta_out,
               double* alpha_out) {
    return bmm_sb_run_probs(X, N, P, initialPi, initialTheta, nsamples, maxK, alpha, beta, gamma, a, b, burnin,
                            seed, device, pi_out, z_out, theta_out, alpha_out, nullptr);
}
int bmm_sb_run_probs(const int32_t* X, int64_t N, int P, const double* initialPi, const double* initialTheta,
                     int nsamples, int maxK, double alpha, double beta, double gamma, double a, double b,
                     int burnin, uint64_t seed, int device, double* pi_out, int32_t* z_out, double* theta_out,
                     double* alpha_out, const bmm_relabel_hooks* hooks) {
    RunIO io; io.pi0 = initialPi; io.theta0 = initialTheta; io.pi_out = pi_out; io.z_out = z_out;
    io.theta_out = theta_out; io.alpha_out = alpha_out;
    return run_chain(BMM_SAMPLER_SB, X, N, P, nsamples, maxK, alpha, beta, gamma, a, b, burnin, 0, seed, device, io,
                     hooks);
}

int bmm_full_run(const int32_t* X, int64_t N, int P, const double* initialPi, const double* initialTheta,
                 int nsamples, int K, double alpha, double beta, double gamma, double a, double b, int burnin,
                 uint64_t seed, int device, double* pi_out, int32_t* z_out, double* theta_out,
                 double* alpha_out) {
    return bmm_full_run_probs(X, N, P, initialPi, initialTheta, nsamples, K, alpha, beta, gamma, a, b, burnin, seed,
                              device, pi_out, z_out, theta_out, alpha_out, nullptr);
}
int bmm_full_run_probs(const int32_t* X, int64_t N, int P, const double* initialPi, const double* initialTheta,
                       int nsamples, int K, double alpha, double beta, double gamma, double a, double b,
                       int burnin, uint64_t seed, int device, double* pi_out, int32_t* z_out,
                       double* theta_out, double* alpha_out, const bmm_relabel_hooks* hooks) {
    RunIO io; io.pi0 = initialPi; io.theta0 = initialTheta; io.pi_out = pi_out; io.z_out = z_out;
    io.theta_out = theta_out; io.alpha_out = alpha_out;
    return run_chain(BMM_SAMPLER_FULL, X, N, P, nsamples, K, alpha, beta, gamma, a, b, burnin, 0, seed, device, io,
                     hooks);
}

// ---- several independent chains in one call (SURVEY.md section 8 rows b, e) ----------------
// Where bmm_multi_run puts things, as pure bookkeeping (no device is touched; tests/test_capi_cpu.py): the
// distinct devices in first-use order -- the RCCL broadcast list, root first -- and for every chain the chain
// that holds its device's copy of the bit planes (the first chain on that device; a holder names itself).
int bmm_multi_plan(int n_chains, const int* devices, int* n_devices_out, int* devices_out, int* holder_of_chain) {
    if (n_chains < 1 || !n_devices_out || !devices_out || !holder_of_chain) return set_err(BMM_E_ARG, "bad argument");
    int nd = 0;
    for (int c = 0; c < n_chains; ++c) {
        const int d = devices ? devices[c] : 0;
        if (d < 0) return set_err(BMM_E_ARG, "devices[%d] = %d is not a device index", c, d);
        int first = -1;
        for (int e = 0; e < c && first < 0; ++e)
            if ((devices ? devices[e] : 0) == d) first = e;
        holder_of_chain[c] = first < 0 ? c : holder_of_chain[first];
        if (first < 0) devices_out[nd++] = d;
    }
    *n_devices_out = nd;
    return BMM_OK;
}

int bmm_multi_run(int sampler, int n_chains, const int* devices, const int32_t* X, int64_t N, int P,
                  const int32_t* const* initialK, const double* const* initialPi,
                  const double* const* initialTheta, int nsamples, int K, double alpha, double beta,
                  double gamma, double a, double b, int burnin, int64_t batch, uint64_t seed,
                  double* const* pi_out, int32_t* const* z_out, double* const* theta_out,
                  double* const* alpha_out) {
    return guarded([&]() -> int {
        if (sampler < 0 || sampler > 3) return set_err(BMM_E_ARG, "unknown sampler %d", sampler);
        if (n_chains < 1) return set_err(BMM_E_ARG, "n_chains must be >= 1");
        if (!z_out || !theta_out || !alpha_out) return set_err(BMM_E_ARG, "null output table");
        if (sampler == BMM_SAMPLER_COLLAPSED && !initialK) return set_err(BMM_E_ARG, "initialK is null");
        if (explicit_params(sampler) && (!initialPi || !initialTheta || !pi_out)) return set_err(BMM_E_ARG, "null buffer table");
        std::vector<RunIO> io((size_t)n_chains);
        for (int c = 0; c < n_chains; ++c) {
            RunIO& q = io[(size_t)c];
            if (sampler == BMM_SAMPLER_COLLAPSED) q.z0 = initialK[c];
            if (explicit_params(sampler)) { q.pi0 = initialPi[c]; q.theta0 = initialTheta[c]; q.pi_out = pi_out[c]; }
            q.z_out = z_out[c]; q.theta_out = theta_out[c]; q.alpha_out = alpha_out[c];
            int rc = check_run_args(X, nsamples, burnin, q, sampler);
            if (rc) return rc;
        }
        // chain c lives on devices[c] (device 0 when the table is null); distinct devices in first-use order
        std::vector<int> dev_of((size_t)n_chains), devs((size_t)n_chains), holder_idx((size_t)n_chains);
        int ndev = 0;
        {
            int rcp = bmm_multi_plan(n_chains, devices, &ndev, devs.data(), holder_idx.data());
            if (rcp) return rcp;
            devs.resize((size_t)ndev);
            for (int c = 0; c < n_chains; ++c) dev_of[(size_t)c] = devices ? devices[c] : 0;
        }
        std::vector<bmm_chain*> chains((size_t)n_chains, nullptr);
        struct Guard {
            std::vector<bmm_chain*>& v;
            ~Guard() { for (bmm_chain* c : v) bmm_chain_destroy(c); }  // shared planes go with their last chain
        } guard{chains};
        int rc = BMM_OK;
        for (int c = 0; c < n_chains && rc == BMM_OK; ++c) {
            rc = bmm_chain_create(&chains[(size_t)c], sampler, N, P, K, alpha, beta, gamma, a, b, batch,
                                  seed + (uint64_t)c, dev_of[(size_t)c]);
            if (rc == BMM_OK) rc = run_prepare(chains[(size_t)c], nsamples, burnin);
        }
        if (rc) return rc;
        // the data: uploaded and packed once, on the first chain's device; the bit planes (not the int32
        // matrix: 160 MB instead of 4 GB at K=20, N=1e7, P=100) broadcast once to the other devices
        std::vector<bmm_chain*> holder(devs.size(), nullptr);  // first chain of each device: owns its planes
        for (size_t q = 0; q < devs.size(); ++q)
            for (int c = 0; c < n_chains && !holder[q]; ++c)
                if (holder_idx[(size_t)c] == c && dev_of[(size_t)c] == devs[q]) holder[q] = chains[(size_t)c];
        rc = bmm_chain_set_data_host(holder[0], X);
        if (rc) return rc;
        if (devs.size() > 1) {
            if (!holder[0]->bits) return set_err(BMM_E_UNSUPPORTED, "a run over several devices broadcasts bit planes");
            std::vector<void*> bufs(devs.size(), nullptr);
            int64_t words = 0;
            for (size_t q = 0; q < devs.size() && rc == BMM_OK; ++q) rc = bmm_chain_planes(holder[q], &bufs[q], &words);
            if (rc == BMM_OK) rc = rccl_broadcast_words(devs, bufs, (size_t)words);
            for (size_t q = 1; q < devs.size() && rc == BMM_OK; ++q) rc = bmm_chain_planes_filled(holder[q]);
            if (rc) return rc;
        }
        for (int c = 0; c < n_chains; ++c) {  // every other chain shares its device's copy
            if (holder_idx[(size_t)c] == c) continue;
            rc = bmm_chain_share_data(chains[(size_t)c], chains[(size_t)holder_idx[(size_t)c]]);
            if (rc) return rc;
        }
        // one host thread per chain; each enqueues on its own stream, so chains on one device overlap
        std::vector<int> status((size_t)n_chains, BMM_OK);
        std::vector<std::string> msg((size_t)n_chains);
        {
            ThreadGroup tg;  // joined also when starting a later thread throws: the workers hold references
            tg.th.reserve((size_t)n_chains);
            for (int c = 0; c < n_chains; ++c)
                tg.th.emplace_back([&, c]() {
                    // no exception may leave a thread: the trace's way out starts helper threads of its own
                    status[(size_t)c] = guarded([&]() -> int {
                        int rcw = run_start_state(chains[(size_t)c], io[(size_t)c]);
                        if (rcw == BMM_OK) rcw = run_body(chains[(size_t)c], nsamples, io[(size_t)c], nullptr);
                        return rcw;
                    });
                    try {
                        if (status[(size_t)c]) msg[(size_t)c] = bmm_last_error();
                    } catch (...) {  // the message could not be copied: the status stands
                    }
                });
        }
        for (int c = 0; c < n_chains; ++c)
            if (status[(size_t)c]) return set_err(status[(size_t)c], "chain %d: %s", c, msg[(size_t)c].c_str());
        return BMM_OK;
    });
}

// n resident chains advanced by `sweeps` sweeps each, one host thread per chain (launches of chains
// that share a device overlap on their streams).  Returns without waiting for the GPU, as bmm_chain_sweeps.
int bmm_chains_sweeps(bmm_chain* const* chains, int n_chains, int sweeps) {
    return guarded([&]() -> int {
        if (!chains || n_chains < 1) return set_err(BMM_E_ARG, "no chains");
        if (n_chains == 1) return bmm_chain_sweeps(chains[0], sweeps);
        std::vector<int> status((size_t)n_chains, BMM_OK);
        std::vector<std::string> msg((size_t)n_chains);
        {
            ThreadGroup tg;  // joined also when starting a later thread throws
            tg.th.reserve((size_t)n_chains);
            for (int c = 0; c < n_chains; ++c)
                tg.th.emplace_back([&, c]() {
                    status[(size_t)c] = bmm_chain_sweeps(chains[c], sweeps);  // guarded inside
                    try {
                        if (status[(size_t)c]) msg[(size_t)c] = bmm_last_error();
                    } catch (...) {
                    }
                });
        }
        for (int c = 0; c < n_chains; ++c)
            if (status[(size_t)c]) return set_err(status[(size_t)c], "chain %d: %s", c, msg[(size_t)c].c_str());
        return BMM_OK;
    });
}

// The data of resident chains on different devices from the first one's: chains[0] holds the bit planes
// (its data are set); every other chain -- one per further device -- receives them over RCCL, as in
// bmm_multi_run.  What a single-process multi-GPU driver of the resident API (bench.py without
// torch.distributed.run) calls once before the sweeps.
int bmm_chains_broadcast_planes(bmm_chain* const* chains, int n_chains) {
    return guarded([&]() -> int {
        if (!chains || n_chains < 1 || !chains[0]) return set_err(BMM_E_ARG, "no chains");
        if (!chains[0]->have_data || !chains[0]->bits) return set_err(BMM_E_STATE, "the first chain holds no bit planes");
        std::vector<int> devs;
        std::vector<void*> bufs((size_t)n_chains, nullptr);
        int64_t words = 0;
        for (int c = 0; c < n_chains; ++c) {
            bmm_chain* ch = chains[c];
            if (!ch) return set_err(BMM_E_ARG, "null chain");
            for (int d : devs)
                if (d == ch->slot) return set_err(BMM_E_ARG, "one chain per device here; chains on one device share (bmm_chain_share_data)");
            if (ch->p.N != chains[0]->p.N || ch->p.P != chains[0]->p.P) return set_err(BMM_E_ARG, "the chains differ in N or P");
            devs.push_back(ch->slot);
            int rc = bmm_chain_planes(ch, &bufs[(size_t)c], &words);
            if (rc) return rc;
        }
        int rc = rccl_broadcast_words(devs, bufs, (size_t)words);
        for (int c = 1; c < n_chains && rc == BMM_OK; ++c) rc = bmm_chain_planes_filled(chains[c]);
        return rc;
    });
}

// The broadcast of bmm_multi_run on a pattern: fills `words` 32-bit words on devices[0], broadcasts them
// to every listed device through RCCL (also with a single device: the library is opened, a
// communicator built and the collective run) and compares.  What a box without several GPUs can check.
int bmm_multi_selfcheck(int n_devices, const int* devices, int64_t words) {
    return guarded([&]() -> int {
        if (n_devices < 1 || !devices || words < 1) return set_err(BMM_E_ARG, "bad argument");
        std::vector<int> devs(devices, devices + n_devices);
        for (int q = 0; q < n_devices; ++q)
            for (int r2 = 0; r2 < q; ++r2)
                if (devs[(size_t)q] == devs[(size_t)r2]) return set_err(BMM_E_ARG, "devices must be distinct");
        std::vector<uint32_t> pat((size_t)words), got((size_t)words);
        for (int64_t i = 0; i < words; ++i) pat[(size_t)i] = (uint32_t)i * 2654435761u + 12345u;
        std::vector<DevBuf> bufs((size_t)n_devices);
        std::vector<void*> ptrs((size_t)n_devices);
        for (int q = 0; q < n_devices; ++q) {
            HIP_TRY(hipSetDevice(devs[(size_t)q]));
            HIP_TRY(bufs[(size_t)q].alloc((size_t)words * 4));
            ptrs[(size_t)q] = bufs[(size_t)q].p;
            if (q == 0) HIP_TRY(hipMemcpy(ptrs[0], pat.data(), (size_t)words * 4, hipMemcpyHostToDevice));
            else HIP_TRY(hipMemset(ptrs[(size_t)q], 0, (size_t)words * 4));
        }
        if (n_devices == 1) {  // rccl_broadcast_words skips a single device; here the collective itself is the point
            Rccl r;
            int rc = rccl_open(r);
            if (rc) return rc;
            ncclComm_t comm = nullptr;
            ncclResult_t e = r.CommInitAll(&comm, 1, devs.data());
            if (e != ncclSuccess) return set_err(BMM_E_RCCL, "ncclCommInitAll failed: %s", r.GetErrorString(e));
            hipStream_t st = nullptr;
            hipError_t he = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
            if (he == hipSuccess) {
                e = r.Broadcast(ptrs[0], ptrs[0], (size_t)words, ncclUint32, 0, comm, st);
                he = hipStreamSynchronize(st);
                (void)hipStreamDestroy(st);
            }
            (void)r.CommDestroy(comm);
            if (he != hipSuccess) return set_err(BMM_E_HIP, "self-check stream failed: %s", hipGetErrorString(he));
            if (e != ncclSuccess) return set_err(BMM_E_RCCL, "ncclBroadcast failed: %s", r.GetErrorString(e));
        } else {
            int rc = rccl_broadcast_words(devs, ptrs, (size_t)words);
            if (rc) return rc;
        }
        for (int q = 0; q < n_devices; ++q) {
            HIP_TRY(hipSetDevice(devs[(size_t)q]));
            HIP_TRY(hipMemcpy(got.data(), ptrs[(size_t)q], (size_t)words * 4, hipMemcpyDeviceToHost));
            if (std::memcmp(got.data(), pat.data(), (size_t)words * 4) != 0)
                return set_err(BMM_E_RCCL, "device %d holds different words after the broadcast", devs[(size_t)q]);
        }
        return BMM_OK;
    });
}

int bmm_device_math(int device, int op, const double* in, const double* in2, double* out, int64_t n) {
    if (!in || !out || n < 0 || op < 0 || op > 4) return set_err(BMM_E_ARG, "bad argument");
    if (op == 2 && !in2) return set_err(BMM_E_ARG, "division needs in2");
    HIP_TRY(hipSetDevice(device));
    DevBuf bi, bi2, bo;
    HIP_TRY(bi.alloc(n * sizeof(double)));
    HIP_TRY(bo.alloc(n * sizeof(double)));
    HIP_TRY(hipMemcpy(bi.p, in, n * sizeof(double), hipMemcpyHostToDevice));
    if (in2) {
        HIP_TRY(bi2.alloc(n * sizeof(double)));
        HIP_TRY(hipMemcpy(bi2.p, in2, n * sizeof(double), hipMemcpyHostToDevice));
    }
    hipLaunchKernelGGL(k_test_math, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, op, bi.as<double>(),
                       bi2.as<double>(), bo.as<double>(), n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, bo.p, n * sizeof(double), hipMemcpyDeviceToHost));
    return BMM_OK;
}

int bmm_device_variates(int device, int kind, double p, double q, uint64_t seed, uint32_t sweep, double* out,
                        int64_t n) {
    if (!out || n < 0 || kind < 0 || kind > 2) return set_err(BMM_E_ARG, "bad argument");
    HIP_TRY(hipSetDevice(device));
    DevBuf bo;
    HIP_TRY(bo.alloc(n * sizeof(double)));
    hipLaunchKernelGGL(k_test_variates, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, kind, p, q, seed, sweep,
                       bo.as<double>(), n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, bo.p, n * sizeof(double), hipMemcpyDeviceToHost));
    return BMM_OK;
}

}  // extern "C"
