// chain.hip -- host driver and C ABI (include/bmm_mcmc.h) of the allocation path.
//
// One bmm_chain = one MCMC chain resident on one GPU: the data matrix, the label
// rows, the integer sufficient statistics, the table image and every trace live in
// HBM; the host only enqueues kernels on the chain's stream.  The sweep loop mirrors
// the reference's (collapsed_gibbs.cpp:84-225, collapsed_gibbs_dp.cpp:98-283,
// stickbreaking.cpp:66-236) with the per-observation loop replaced by batches.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/bmm_mcmc.h"
#include "kernels.hip.h"

using namespace bmm;

namespace {

thread_local char g_err[512] = "";

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

// accumulator counts the resample kernel is instantiated for
const int kKT[] = {4, 8, 12, 16, 20, 24, 28, 32, 40, 48, 56, 64};
// workgroup size by accumulator count: the VGPR budget per lane is 512 / (waves per SIMD)
constexpr int kThreadsSmall = 1024;  // KT <= 12: 128 VGPRs
constexpr int kThreadsMid = 768;     // KT 16, 20: 168 VGPRs
constexpr int kThreadsLarge = 512;   // 256 VGPRs
#ifndef BMM_STAGE_WIDE
#define BMM_STAGE_WIDE 32
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
template <int MINUS, bool BITS>
resample_fn resample_kernel_m(int kt) {
    constexpr int SW = BITS ? 16 : kStageWide;
    switch (kt) {
        case 4: return k_resample<4, kThreadsSmall, MINUS, SW, BITS>;
        case 8: return k_resample<8, kThreadsSmall, MINUS, SW, BITS>;
        case 12: return k_resample<12, kThreadsSmall, MINUS, SW, BITS>;
        case 16: return k_resample<16, BITS ? kThreadsSmall : kThreadsMid, MINUS, SW, BITS>;
        case 20: return k_resample<20, BITS ? kThreadsSmall : kThreadsMid, MINUS, SW, BITS>;
        case 24: return k_resample<24, BITS ? kThreadsMid : kThreadsLarge, MINUS, 16, BITS>;
        case 28: return k_resample<28, BITS ? kThreadsMid : kThreadsLarge, MINUS, 16, BITS>;
        case 32: return k_resample<32, BITS ? kThreadsMid : kThreadsLarge, MINUS, 16, BITS>;
        case 40: return k_resample<40, kThreadsLarge, MINUS, 16, BITS>;
        case 48: return k_resample<48, kThreadsLarge, MINUS, 16, BITS>;
        case 56: return k_resample<56, kThreadsLarge, MINUS, 16, BITS>;
        case 64: return k_resample<64, kThreadsLarge, MINUS, 16, BITS>;
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
        case 56: return k_resample<56, 256, MINUS, 16, BITS>;
        case 64: return k_resample<64, 256, MINUS, 16, BITS>;
    }
    return nullptr;
}
// the same kernels at a chosen workgroup size (up to 32 categories): used when a batch is too
// small to give every CU one of the default-sized workgroups, and by BMM_DEBUG_THREADS
template <int NT, int MINUS, bool BITS>
resample_fn resample_kernel_nt(int kt) {
    constexpr int SW = BITS ? 16 : kStageWide;
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
    if (nt == 768) return resample_kernel_nt<768, 1, false>(kt);
    if (nt == 512) return resample_kernel_nt<512, 1, false>(kt);
    return nullptr;
}
// more than 32 accumulators on bit planes: two lanes per observation (SPLIT = 2 in kernels.hip.h)
constexpr int kThreadsSplit = 1024;
template <int MINUS>
resample_fn resample_kernel_split(int kt) {
    switch (kt) {
        case 40: return k_resample<40, kThreadsSplit, MINUS, 16, true, 2>;
        case 48: return k_resample<48, kThreadsSplit, MINUS, 16, true, 2>;
        case 56: return k_resample<56, kThreadsSplit, MINUS, 16, true, 2>;
        case 64: return k_resample<64, kThreadsSplit, MINUS, 16, true, 2>;
    }
    return nullptr;
}
// minus: 0 no own-cluster tables (stick-breaking), 1 in LDS, 2 in global memory
resample_fn resample_kernel(int kt, int minus, bool bits) {
    if (bits)
        return minus == 0 ? resample_kernel_m<0, true>(kt)
                          : (minus == 1 ? resample_kernel_m<1, true>(kt) : resample_kernel_m<2, true>(kt));
    return minus == 0 ? resample_kernel_m<0, false>(kt)
                      : (minus == 1 ? resample_kernel_m<1, false>(kt) : resample_kernel_m<2, false>(kt));
}
resample_fn resample_kernel_small_of(int kt, int minus, bool bits) {
    if (bits) return minus == 0 ? resample_kernel_small<0, true>(kt) : resample_kernel_small<1, true>(kt);
    return minus == 0 ? resample_kernel_small<0, false>(kt) : resample_kernel_small<1, false>(kt);
}

}  // namespace

struct bmm_chain {
    ChainParams p{};
    int device = 0;
    hipStream_t stream = nullptr;
    int64_t batch = 1;
    double alpha0 = 1.0;
    int NT = 0, grid_max = 0, minus_in_lds = 1;
    int OT = 0;  // observations per tile (= NT, or NT / 2 for the two-lanes-per-observation kernels)
    bool sharded = false, shard_open = false;  // one chain over several ranks (explicit-parameter samplers)
    bool generic = false;         // shape beyond the resident kernel: tables from global memory
    double* dScratch = nullptr;   // generic path: per-thread score columns
    double* dProbs = nullptr;     // sweep_probs: N x K probabilities of the sweep being run
    bool probs_sweep = false;     // route this sweep through the generic kernel and emit dProbs
    int64_t scratch_stride = 0;
    size_t lds_bytes = 0;
    resample_fn fn = nullptr;

    const int32_t* dX = nullptr;
    int32_t* dX_owned = nullptr;
    uint32_t* dXb = nullptr;      // bit planes of X (k_pack_bits), what the resident kernels stream by default
    bool bits = false;
    int num_cus = 0;
    bool batch_defaulted = false;
    int64_t batch_unrounded = 0;
    int32_t* dZ[2] = {nullptr, nullptr};
    int32_t *dNk = nullptr, *dS = nullptr, *dDNk = nullptr, *dDS = nullptr;
    double *dAlpha = nullptr, *dTab = nullptr, *dPi = nullptr, *dTheta = nullptr;
    bool have_data = false, have_init = false, started = false;
    int sweep = 0;  // sweeps completed (= index j of the last one)

    // trace of the *_run entry points
    int burnin = 0, S = 0;
    int32_t* dTrace = nullptr;  // [S][N], 0-based
    double *dThetaTrace = nullptr, *dAlphaTrace = nullptr, *dPiTrace = nullptr;

    int32_t* dNkTrace = nullptr;  // [n][K] cluster sizes per sweep of the current sweeps_counts call
    int nk_trace_base = 0;        // sweep index of its row 0
    unsigned long long* dDiag = nullptr;
    int prof = 0;             // > 0: HIP events around the resample launches of every prof-th sweep
    std::vector<hipEvent_t> ev;
    size_t ev_used = 0;
    double prof_ms = 0.0;
    int64_t prof_n = 0;
};

namespace {

// Default batch (profiles/r01/batch_bias_oracle.json): the finite sampler's posterior is
// within seed noise of the sequential scan up to about N/8 on the hardest bundled data set
// (no measurable shift at any batch on well-separated data); the DP sampler opens spurious
// clusters above about N/16, because every "new" draw of a batch shares one label.
// every cell of X must be 0 or 1: one streaming pass when the matrix is handed over
int validate_binary(bmm_chain* c, const int32_t* dX, int64_t n) {
    int* dflag = nullptr;
    HIP_TRY(hipMalloc(&dflag, sizeof(int)));
    HIP_TRY(hipMemsetAsync(dflag, 0, sizeof(int), c->stream));
    const bool al16 = (reinterpret_cast<uintptr_t>(dX) & 15) == 0;
    const int64_t n16 = al16 ? n / 4 : 0;
    hipLaunchKernelGGL(k_validate_binary, dim3(2048), dim3(256), 0, c->stream,
                       reinterpret_cast<const uint4*>(dX), n16, reinterpret_cast<const uint32_t*>(dX), n, dflag);
    int flag = 0;
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(&flag, dflag, sizeof(int), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(dflag);
    if (e != hipSuccess) return set_err(BMM_E_HIP, "validating X failed: %s", hipGetErrorString(e));
    if (flag) return set_err(BMM_E_ARG, "data must be binary: X holds a value other than 0 and 1");
    return BMM_OK;
}

int64_t default_batch(int sampler, int64_t N) {
    if (sampler == BMM_SAMPLER_SB || sampler == BMM_SAMPLER_FULL) return N;
    int64_t b = sampler == BMM_SAMPLER_DP ? N / 16 : N / 8;
    return b < 1 ? 1 : b;
}

TableLayout layout_of(const bmm_chain* c) {
    return TableLayout{c->p.G, c->p.KT, !explicit_params(c->p.mode) ? 1 : 0};
}

int32_t* label_row(bmm_chain* c, int j) {
    if (c->dTrace && j >= c->burnin) return c->dTrace + (size_t)(j - c->burnin) * c->p.N;
    return c->dZ[j & 1];
}

int chain_alloc(bmm_chain* c) {
    const ChainParams& p = c->p;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    const size_t nz = (size_t)p.N * sizeof(int32_t);
    HIP_TRY(hipMalloc(&c->dZ[0], nz));
    HIP_TRY(hipMalloc(&c->dZ[1], nz));
    const size_t ns = (size_t)p.K * p.P * sizeof(int32_t), nn = (size_t)p.K * sizeof(int32_t);
    HIP_TRY(hipMalloc(&c->dNk, nn));
    HIP_TRY(hipMalloc(&c->dDNk, nn * kDeltaReps));
    HIP_TRY(hipMalloc(&c->dS, ns));
    HIP_TRY(hipMalloc(&c->dDS, ns * kDeltaReps));
    HIP_TRY(hipMalloc(&c->dAlpha, sizeof(double)));
    HIP_TRY(hipMalloc(&c->dTab, (size_t)layout_of(c).doubles() * sizeof(double)));
    HIP_TRY(hipMalloc(&c->dPi, (size_t)p.K * sizeof(double)));
    HIP_TRY(hipMalloc(&c->dTheta, (size_t)p.K * p.P * sizeof(double)));
    if (c->generic) HIP_TRY(hipMalloc(&c->dScratch, (size_t)c->scratch_stride * p.Kc * sizeof(double)));
    HIP_TRY(hipMemsetAsync(c->dNk, 0, nn, c->stream));
    HIP_TRY(hipMemsetAsync(c->dDNk, 0, nn * kDeltaReps, c->stream));
    HIP_TRY(hipMemsetAsync(c->dS, 0, ns, c->stream));
    HIP_TRY(hipMemsetAsync(c->dDS, 0, ns * kDeltaReps, c->stream));
    HIP_TRY(hipMemsetAsync(c->dTab, 0, (size_t)layout_of(c).doubles() * sizeof(double), c->stream));
    HIP_TRY(hipMemsetAsync(c->dZ[0], 0xff, nz, c->stream));  // -1 = unassigned
    HIP_TRY(hipMemcpyAsync(c->dAlpha, &c->alpha0, sizeof(double), hipMemcpyHostToDevice, c->stream));
#ifdef BMM_DIAG
    HIP_TRY(hipMalloc(&c->dDiag, 16 * sizeof(unsigned long long)));
    HIP_TRY(hipMemsetAsync(c->dDiag, 0, 16 * sizeof(unsigned long long), c->stream));
#endif
    HIP_TRY(hipStreamSynchronize(c->stream));
    return BMM_OK;
}

int launch_resample(bmm_chain* c, const int32_t* z_in, int32_t* z_out, int64_t lo, int64_t hi,
                    uint32_t sweep) {
    ResampleArgs a{};
    a.X = c->dX; a.Xb = c->dXb; a.z_in = z_in; a.z_out = z_out; a.tab = c->dTab; a.dNk = c->dDNk; a.dS = c->dDS;
    a.lo = lo; a.hi = hi; a.sweep = sweep; a.minus_in_lds = c->minus_in_lds; a.diag = c->dDiag;
    const int64_t ntiles = (hi - lo + c->OT - 1) / c->OT;
    int grid = (int)(ntiles < c->grid_max ? ntiles : c->grid_max);
    const bool use_generic = c->generic || c->probs_sweep;
    if (use_generic) {
        const int64_t nt256 = (hi - lo + 255) / 256;
        const int64_t maxb = c->scratch_stride / 256;
        grid = (int)(nt256 < maxb ? nt256 : maxb);
        a.probs = c->probs_sweep ? c->dProbs : nullptr;
    }
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
        HIP_TRY(hipEventRecord(e0, c->stream));
    }
    if (use_generic)
        hipLaunchKernelGGL(k_resample_generic, dim3(grid), dim3(256), 0, c->stream, c->p, a, c->dScratch,
                           c->scratch_stride);
    else
        hipLaunchKernelGGL(c->fn, dim3(grid), dim3(c->NT), c->lds_bytes, c->stream, c->p, a);
    HIP_TRY(hipGetLastError());
    if (e1) HIP_TRY(hipEventRecord(e1, c->stream));
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
    hipLaunchKernelGGL(k_count_tables, dim3(c->p.KT), dim3(320), 0, c->stream, c->p, c->dNk, c->dS,
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
                           al_tr, nk_tr);
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(k_sb_theta_tables, dim3(p.KT), dim3(256), 0, c->stream, p, c->dNk, c->dS,
                           c->dPi, c->dTheta, 1, (uint32_t)j, th_tr, c->dTab);
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
        int rc = launch_count_tables(c);
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
    const size_t lds_max = 163840;
    const int minus = explicit_params(p.mode) ? 0 : (c->minus_in_lds ? 1 : 2);
    c->NT = threads_for(p.KT, c->bits);
    c->fn = resample_kernel(p.KT, minus, c->bits);
    int split = 1;
    if (c->bits && p.KT > 32 && minus != 2 && !getenv("BMM_DEBUG_NOSPLIT")) {
        c->fn = minus ? resample_kernel_split<1>(p.KT) : resample_kernel_split<0>(p.KT);
        c->NT = kThreadsSplit;
        split = 2;
    }
    if (const char* dbg = getenv("BMM_DEBUG_THREADS")) {
        const int nt = atoi(dbg);
        if (resample_fn f = resample_kernel_at(p.KT, nt, minus, c->bits)) { c->fn = f; c->NT = nt; }
    } else if (c->bits && split == 1) {
        // a batch that cannot give every CU a workgroup of the default size gets smaller ones
        for (int nt : {768, 512}) {
            if ((c->batch + c->NT - 1) / c->NT >= c->num_cus || nt >= c->NT) continue;
            if (resample_fn f = resample_kernel_at(p.KT, nt, minus, true)) { c->fn = f; c->NT = nt; }
        }
    }
    hipError_t e = hipSetDevice(c->device);
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
    if (split == 1 && tiles < c->num_cus && c->NT > 256 && c->lds_bytes * 4 <= lds_max && minus != 2 &&
        !getenv("BMM_DEBUG_THREADS")) {
        resample_fn f = resample_kernel_small_of(p.KT, minus, c->bits);
        hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(f), hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds_bytes);
        int pc2 = 0;
        if (e2 == hipSuccess) e2 = hipOccupancyMaxActiveBlocksPerMultiprocessor(&pc2, reinterpret_cast<const void*>(f), 256, c->lds_bytes);
        if (e2 == hipSuccess && pc2 >= 1) { c->fn = f; c->NT = 256; c->OT = 256; c->grid_max = pc2 * c->num_cus; }
    }
    return BMM_OK;
}

// a defaulted batch is rounded up to whole rounds of workgroups (no ragged last round)
void round_default_batch(bmm_chain* c) {
    if (!c->batch_defaulted) return;
    const int64_t round = (int64_t)c->grid_max * c->OT;
    c->batch = c->batch_unrounded;
    if (c->batch > round) c->batch = (c->batch + round - 1) / round * round;
    if (c->batch > c->p.N) c->batch = c->p.N;
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
                           c->dPi, c->dTheta, 0, 0u, (double*)nullptr, c->dTab);
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
int64_t bmm_default_batch(int sampler, int64_t N) { return default_batch(sampler, N); }

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
    if (device < 0 || device >= ndev) return set_err(BMM_E_ARG, "device %d out of range (have %d)", device, ndev);

    bmm_chain* c = new (std::nothrow) bmm_chain();
    if (!c) return set_err(BMM_E_ARG, "out of host memory");
    ChainParams& p = c->p;
    p.mode = sampler; p.N = N; p.Ntot = N; p.obs0 = 0; p.P = P; p.G = (P + kGroupW - 1) / kGroupW; p.K = K;
    p.Kc = sampler == BMM_SAMPLER_DP ? K + 1 : K;
    p.KT = pick_kt(p.Kc);
    p.beta = beta; p.gamma = gamma; p.a = a; p.b = b; p.seed = seed;
    p.sample_alpha = alpha == 0.0;  // collapsed_gibbs.cpp:50-54
    c->alpha0 = p.sample_alpha ? 1.0 : alpha;
    c->device = device;
    c->batch = batch <= 0 ? default_batch(sampler, N) : (batch > N ? N : batch);
    if (explicit_params(sampler)) c->batch = N;
    if (p.Kc > kMaxCatsAny) {
        delete c;
        return set_err(BMM_E_UNSUPPORTED, "%d categories exceed the %d this build supports", p.Kc, kMaxCatsAny);
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) { delete c; return set_err(BMM_E_HIP, "hipGetDeviceProperties failed"); }
    const size_t lds_max = 163840;  // gfx950: 160 KiB per workgroup
    const size_t hist_bytes = ((size_t)K * P + K) * sizeof(int32_t);
    c->bits = getenv("BMM_X_LAYOUT_INT32") == nullptr;
    c->generic = p.KT < 0 || P > kMaxP || getenv("BMM_DEBUG_GENERIC") != nullptr;
    if (!c->generic) {
        c->NT = threads_for(p.KT, false);
        c->OT = c->NT;
        c->lds_bytes = (size_t)layout_of(c).doubles() * sizeof(double) + hist_bytes;
        if (c->lds_bytes > lds_max && !explicit_params(p.mode)) {  // second tier: own-cluster tables stay in L2
            c->minus_in_lds = 0;
            c->lds_bytes = (size_t)layout_of(c).head() * sizeof(double) + hist_bytes;
        }
        if (c->lds_bytes > lds_max) c->generic = true;  // third tier: nothing resident
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
        c->num_cus = prop.multiProcessorCount;
        rc = pick_kernel(c);
        if (rc) { delete c; return rc; }
    }
    c->batch_defaulted = batch <= 0 && !explicit_params(sampler);
    c->batch_unrounded = c->batch;
    round_default_batch(c);
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
        (void)hipFree(c->dDiag);
    }
#endif
    for (hipEvent_t e : c->ev) (void)hipEventDestroy(e);
    void* bufs[] = {c->dX_owned, c->dXb, c->dZ[0], c->dZ[1], c->dNk, c->dS, c->dDNk, c->dDS, c->dAlpha, c->dTab,
                    c->dPi, c->dTheta, c->dTrace, c->dThetaTrace, c->dAlphaTrace, c->dPiTrace, c->dScratch, c->dProbs};
    for (void* b : bufs)
        if (b) (void)hipFree(b);
    if (c->stream) (void)hipStreamDestroy(c->stream);
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
    int rc = pick_kernel(c);
    if (rc) return rc;
    round_default_batch(c);
    return BMM_OK;
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
    // bit planes: the int32 matrix passes through a staging buffer in slabs of rows (<= 256 MiB) and
    // never exists whole on the device -- 16 bytes per observation stay instead of 4 P
    const int W = (P + 31) / 32;
    if (!c->dXb) HIP_TRY(hipMalloc(&c->dXb, (size_t)W * N * sizeof(uint32_t)));
    int64_t slab = ((int64_t)256 << 20) / ((int64_t)P * 4);
    slab = slab / 4 * 4;
    if (slab < 4) slab = 4;
    if (slab > N) slab = N;
    int32_t* stage = nullptr;
    HIP_TRY(hipMalloc(&stage, (size_t)slab * P * sizeof(int32_t)));
    int rc = BMM_OK;
    for (int64_t i0 = 0; i0 < N && rc == BMM_OK; i0 += slab) {
        const int64_t rows = N - i0 < slab ? N - i0 : slab;
        hipError_t e = hipMemcpy2DAsync(stage, (size_t)rows * 4, X + i0, (size_t)N * 4, (size_t)rows * 4, (size_t)P,
                                        hipMemcpyHostToDevice, c->stream);
        if (e != hipSuccess) { rc = set_err(BMM_E_HIP, "uploading X failed: %s", hipGetErrorString(e)); break; }
        rc = validate_binary(c, stage, rows * P);  // synchronises the stream
        if (rc == BMM_OK) rc = pack_rows(c, stage, rows, rows, i0);
        if (rc == BMM_OK && hipStreamSynchronize(c->stream) != hipSuccess) rc = set_err(BMM_E_HIP, "packing X failed");
    }
    (void)hipFree(stage);
    if (rc) return rc;
    if (c->dX_owned) { (void)hipFree(c->dX_owned); c->dX_owned = nullptr; }
    c->dX = nullptr;
    c->have_data = true;
    return BMM_OK;
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
    const int32_t* x = static_cast<const int32_t*>(dX);
    int rc = validate_binary(c, x, c->p.N * c->p.P);
    if (rc) return rc;
    if (c->bits) {  // packed here and now; the caller's matrix is not read again
        const int W = (c->p.P + 31) / 32;
        if (!c->dXb) HIP_TRY(hipMalloc(&c->dXb, (size_t)W * c->p.N * sizeof(uint32_t)));
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

int bmm_chain_set_initial_labels(bmm_chain* c, const int32_t* z1) {
    if (!c || !z1) return set_err(BMM_E_ARG, "null argument");
    if (c->p.mode != MODE_COLLAPSED) return set_err(BMM_E_STATE, "only the finite collapsed sampler takes initial labels");
    if (c->started) return set_err(BMM_E_STATE, "chain already started");
    std::vector<int32_t> z0((size_t)c->p.N);
    for (int64_t i = 0; i < c->p.N; ++i) {
        if (z1[i] < 1 || z1[i] > c->p.K) return set_err(BMM_E_ARG, "initialK[%lld] = %d outside 1..%d", (long long)i, z1[i], c->p.K);
        z0[(size_t)i] = z1[i] - 1;
    }
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpyAsync(c->dZ[0], z0.data(), z0.size() * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
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

int bmm_chain_shard_resample(bmm_chain* c) {
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
    HIP_TRY(hipStreamSynchronize(c->stream));  // the deltas are complete: safe to reduce on any stream
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
    const size_t bytes = (size_t)c->p.N * c->p.K * sizeof(double);
    if (!c->dProbs) HIP_TRY(hipMalloc(&c->dProbs, bytes));
    if (!c->dScratch) {  // a resident-kernel chain borrows the generic kernel for this sweep
        int64_t threads = (int64_t)256 * 1024;
        const int64_t cap = ((int64_t)256 << 20) / ((int64_t)c->p.Kc * 8);
        if (threads > cap) threads = cap / 256 * 256;
        if (threads < 256) threads = 256;
        c->scratch_stride = threads;
        HIP_TRY(hipMalloc(&c->dScratch, (size_t)threads * c->p.Kc * sizeof(double)));
    }
    HIP_TRY(hipMemsetAsync(c->dProbs, 0, bytes, c->stream));
    c->probs_sweep = true;
    int rc = bmm_chain_sweeps(c, 1);
    c->probs_sweep = false;
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(probs_out, c->dProbs, bytes, hipMemcpyDeviceToHost, c->stream));
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
    return BMM_OK;
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

}  // extern "C"

// ------------------------------------------------------------------ *_run entry points
namespace {

int run_chain(int sampler, const int32_t* X, int64_t N, int P, const int32_t* z0, const double* pi0,
              const double* theta0, int nsamples, int K, double alpha, double beta, double gamma, double a,
              double b, int burnin, int64_t batch, uint64_t seed, int device, double* pi_out,
              int32_t* z_out, double* theta_out, double* alpha_out) {
    if (!X || !z_out || !theta_out || !alpha_out) return set_err(BMM_E_ARG, "null buffer");
    if (nsamples < 1) return set_err(BMM_E_ARG, "nsamples must be >= 1");
    if (burnin < 0 || burnin >= nsamples) return set_err(BMM_E_ARG, "burnin must be in [0, nsamples)");
    bmm_chain* c = nullptr;
    int rc = bmm_chain_create(&c, sampler, N, P, K, alpha, beta, gamma, a, b, batch, seed, device);
    if (rc) return rc;
    struct Guard { bmm_chain* c; ~Guard() { bmm_chain_destroy(c); } } guard{c};
    const int S = nsamples - burnin;
    c->burnin = burnin; c->S = S;
    HIP_TRY(hipMalloc(&c->dTrace, (size_t)S * N * sizeof(int32_t)));
    HIP_TRY(hipMalloc(&c->dThetaTrace, (size_t)S * K * P * sizeof(double)));
    HIP_TRY(hipMalloc(&c->dAlphaTrace, (size_t)S * sizeof(double)));
    if (explicit_params(sampler)) HIP_TRY(hipMalloc(&c->dPiTrace, (size_t)S * K * sizeof(double)));
    rc = bmm_chain_set_data_host(c, X);
    if (rc) return rc;
    if (sampler == BMM_SAMPLER_COLLAPSED) rc = bmm_chain_set_initial_labels(c, z0);
    if (explicit_params(sampler)) rc = bmm_chain_set_initial_params(c, pi0, theta0);
    if (rc) return rc;
    if (burnin == 0) {
        // trace row 0 (DESIGN.md "Quirks"): labels = initial allocation or unassigned (-1 -> NA),
        // theta = NaN (collapsed, never written), 0 (dp, zero-filled) or the initial theta (sb)
        std::vector<double> t0((size_t)K * P, sampler == BMM_SAMPLER_DP ? 0.0 : std::nan(""));
        if (explicit_params(sampler)) std::memcpy(t0.data(), theta0, t0.size() * sizeof(double));
        HIP_TRY(hipMemcpy(c->dThetaTrace, t0.data(), t0.size() * sizeof(double), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(c->dAlphaTrace, &c->alpha0, sizeof(double), hipMemcpyHostToDevice));
        if (explicit_params(sampler))
            HIP_TRY(hipMemcpy2D(c->dPiTrace, (size_t)S * sizeof(double), pi0, sizeof(double), sizeof(double), K, hipMemcpyHostToDevice));
        if (sampler != BMM_SAMPLER_COLLAPSED) HIP_TRY(hipMemset(c->dTrace, 0xff, (size_t)N * sizeof(int32_t)));
    }
    rc = bmm_chain_sweeps(c, nsamples - 1);
    if (rc) return rc;
    // labels: [S][N] 0-based -> S x N column-major 1-based, on the device, then one copy out
    int32_t* dOut = nullptr;
    HIP_TRY(hipMalloc(&dOut, (size_t)S * N * sizeof(int32_t)));
    struct Free { void* p; ~Free() { (void)hipFree(p); } } fo{dOut};
    hipLaunchKernelGGL(k_trace_to_r, dim3((unsigned)((N + 31) / 32), (unsigned)((S + 31) / 32)), dim3(256), 0,
                       c->stream, c->dTrace, N, S, dOut);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(z_out, dOut, (size_t)S * N * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(theta_out, c->dThetaTrace, (size_t)S * K * P * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(alpha_out, c->dAlphaTrace, (size_t)S * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (explicit_params(sampler))
        HIP_TRY(hipMemcpyAsync(pi_out, c->dPiTrace, (size_t)S * K * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return BMM_OK;
}

}  // namespace

extern "C" {

int bmm_collapsed_run(const int32_t* X, int64_t N, int P, const int32_t* initialK, int nsamples, int K,
                      double alpha, double beta, double gamma, double a, double b, int burnin,
                      int64_t batch, uint64_t seed, int device, int32_t* z_out, double* theta_out,
                      double* alpha_out) {
    if (!initialK) return set_err(BMM_E_ARG, "initialK is null");
    return run_chain(BMM_SAMPLER_COLLAPSED, X, N, P, initialK, nullptr, nullptr, nsamples, K, alpha, beta,
                     gamma, a, b, burnin, batch, seed, device, nullptr, z_out, theta_out, alpha_out);
}

int bmm_dp_run(const int32_t* X, int64_t N, int P, int nsamples, double alpha, double beta, double gamma,
               double a, double b, int burnin, int maxK, int64_t batch, uint64_t seed, int device,
               int32_t* z_out, double* theta_out, double* alpha_out) {
    return run_chain(BMM_SAMPLER_DP, X, N, P, nullptr, nullptr, nullptr, nsamples, maxK, alpha, beta, gamma,
                     a, b, burnin, batch, seed, device, nullptr, z_out, theta_out, alpha_out);
}

int bmm_sb_run(const int32_t* X, int64_t N, int P, const double* initialPi, const double* initialTheta,
               int nsamples, int maxK, double alpha, double beta, double gamma, double a, double b,
               int burnin, uint64_t seed, int device, double* pi_out, int32_t* z_out, double* theta_out,
               double* alpha_out) {
    if (!initialPi || !initialTheta || !pi_out) return set_err(BMM_E_ARG, "null buffer");
    return run_chain(BMM_SAMPLER_SB, X, N, P, nullptr, initialPi, initialTheta, nsamples, maxK, alpha, beta,
                     gamma, a, b, burnin, 0, seed, device, pi_out, z_out, theta_out, alpha_out);
}

int bmm_full_run(const int32_t* X, int64_t N, int P, const double* initialPi, const double* initialTheta,
                 int nsamples, int K, double alpha, double beta, double gamma, double a, double b, int burnin,
                 uint64_t seed, int device, double* pi_out, int32_t* z_out, double* theta_out,
                 double* alpha_out) {
    if (!initialPi || !initialTheta || !pi_out) return set_err(BMM_E_ARG, "null buffer");
    return run_chain(BMM_SAMPLER_FULL, X, N, P, nullptr, initialPi, initialTheta, nsamples, K, alpha, beta,
                     gamma, a, b, burnin, 0, seed, device, pi_out, z_out, theta_out, alpha_out);
}

int bmm_device_math(int device, int op, const double* in, const double* in2, double* out, int64_t n) {
    if (!in || !out || n < 0 || op < 0 || op > 3) return set_err(BMM_E_ARG, "bad argument");
    if (op == 2 && !in2) return set_err(BMM_E_ARG, "division needs in2");
    HIP_TRY(hipSetDevice(device));
    double *di = nullptr, *di2 = nullptr, *dout = nullptr;
    HIP_TRY(hipMalloc(&di, n * sizeof(double)));
    HIP_TRY(hipMalloc(&dout, n * sizeof(double)));
    HIP_TRY(hipMemcpy(di, in, n * sizeof(double), hipMemcpyHostToDevice));
    if (in2) {
        HIP_TRY(hipMalloc(&di2, n * sizeof(double)));
        HIP_TRY(hipMemcpy(di2, in2, n * sizeof(double), hipMemcpyHostToDevice));
    }
    hipLaunchKernelGGL(k_test_math, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, op, di, di2, dout, n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, dout, n * sizeof(double), hipMemcpyDeviceToHost));
    (void)hipFree(di); (void)hipFree(dout);
    if (di2) (void)hipFree(di2);
    return BMM_OK;
}

int bmm_device_variates(int device, int kind, double p, double q, uint64_t seed, uint32_t sweep, double* out,
                        int64_t n) {
    if (!out || n < 0 || kind < 0 || kind > 2) return set_err(BMM_E_ARG, "bad argument");
    HIP_TRY(hipSetDevice(device));
    double* dout = nullptr;
    HIP_TRY(hipMalloc(&dout, n * sizeof(double)));
    hipLaunchKernelGGL(k_test_variates, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, kind, p, q, seed, sweep, dout, n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, dout, n * sizeof(double), hipMemcpyDeviceToHost));
    (void)hipFree(dout);
    return BMM_OK;
}

}  // extern "C"
