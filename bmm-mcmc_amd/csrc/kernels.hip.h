// kernels.hip.h -- gfx950 kernels of the cluster-allocation path.
//
// One observation per lane (wave64).  X is streamed either as bit planes packed once when the
// matrix is handed over (ceil(P/32) 32-bit words per observation; the default) or as the N x P
// int32 column-major matrix R hands over (lane i reads X[i + d*N]: a wave reads 256 contiguous
// bytes per feature, packed to bits in registers as they arrive).  The per-cluster log-predictive
// is K*ceil(P/W) LDS lookups into 2^W-entry group tables (one ds_read_b64 + one v_add_f64 per
// cluster per W features, W = 5, or 4 for big table images: bmm_spec.h; the category's constant term
// sits in group 0) instead of K*P multiply-adds; the observation's own cluster is scored from tables
// of its own with groups of 3.  All entries of a group sit in one 256- (128-) byte run, i.e. on different
// bank pairs, so the per-lane-indexed read is conflict-free.  Work is handed out per wave in chunks of 64
// observations from a counter in LDS.  Sufficient-statistic changes are accumulated as integers in
// LDS (a few movers: one at a time by the whole wave, one feature per lane; many: every mover lane
// walks its own set bits) and flushed with one global integer atomic per touched cell per
// workgroup -- order-independent, hence deterministic.
//
// Reference lines realised here (all /root/reference/src):
//   z | rest, finite K      collapsed_gibbs.cpp:86-182
//   z | rest, CRP           collapsed_gibbs_dp.cpp:108-242
//   z | pi, theta           stickbreaking.cpp:70-125, counts :164-186
//   theta-hat               collapsed_gibbs.cpp:205-219, collapsed_gibbs_dp.cpp:266-281
//   v, pi, theta draws      stickbreaking.cpp:187-229
//   alpha                   utils.cpp:6-14
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bmm_spec.h"

namespace bmm {

typedef __attribute__((address_space(3))) double lds_f64;  // LDS-qualified, keeps ds_read under volatile

constexpr int kMaxP = 128;      // fast path: 4 bit-words per observation
constexpr int lcm_(int a, int b) { int x = a; while (x % b) x += a; return x; }
// features per table-building chunk: whole groups of every width
constexpr int kChunkP = kMaxP / lcm_(lcm_(kGroupW, kGroupWAlt), kGroupWm) * lcm_(lcm_(kGroupW, kGroupWAlt), kGroupWm);
constexpr int kMaxCats = 64;    // fast path: clusters (+ the DP's new-cluster option)
constexpr int kMaxCatsAny = 1024;  // generic path

enum : int { MODE_COLLAPSED = 0, MODE_DP = 1, MODE_SB = 2, MODE_FULL = 3 };
// the two samplers that carry explicit (pi, theta) and resample z | pi, theta in one exact batch
__host__ __device__ inline bool explicit_params(int mode) { return mode == MODE_SB || mode == MODE_FULL; }

// Layout of the table image in global memory, in doubles:
//   Tp  [G][KT][M]    group tables against the full statistics (M = 2^W, W the shape's group width:
//                     ChainParams::W); group 0 carries the category's
//                     prior / weight term Cp, so a score is just the sum of G entries
//   Cp  [KT]          that term on its own (kept for inspection; the kernels do not read it)
//   Cm  [KT]          ... with the scored observation removed from its own cluster
//   Nk  [KT] int32    (two per double slot, padded to an even number of doubles)
//   E   [256]         2^(j/256), the table of the spec's expw_ (read per lane in the draw)
//   Tm  [Gm][KT][Mm]  group tables with the observation's own contribution removed, Cm in group 0
//                     (not SB); narrower groups, Mm = 2^kGroupWm entries each; Gm padded with all-zero
//                     groups to a multiple of kOwnSub
// A workgroup copies the head (Tp..Nk) into LDS, and Tm too when both fit in 160 KiB;
// otherwise Tm is gathered from global memory (L2-resident, 1/K of the lookups).
struct TableLayout {
    int G, KT, Gm, M;  // Gm = 0: no own-cluster tables; M = entries per group of Tp
    __host__ __device__ int tp() const { return 0; }
    __host__ __device__ int cp() const { return G * KT * M; }
    __host__ __device__ int cm() const { return cp() + KT; }
    __host__ __device__ int nk() const { return cm() + KT; }
    __host__ __device__ int et() const { return nk() + (KT + 1) / 2 + (((KT + 1) / 2) & 1); }
    __host__ __device__ int tm() const { return et() + 256; }
    __host__ __device__ int head() const { return tm(); }
    __host__ __device__ int gm_pad() const { return (Gm + kOwnSub - 1) / kOwnSub * kOwnSub; }
    __host__ __device__ int doubles() const { return tm() + gm_pad() * KT * kGroupMm; }
};

struct ChainParams {
    int mode;           // MODE_*
    int64_t N;          // observations held by this chain object (a shard, or all of them)
    int64_t Ntot;       // observations of the whole chain (= N unless the chain is sharded over ranks)
    int64_t obs0;       // global index of local observation 0 (keys the per-observation Philox counter)
    int P, W, G, Gm;    // features; features per lookup group of the tables (bmm_spec.h); groups of the
                        // tables, of the own-cluster tables
    int K;              // labels (K or maxK)
    int Kc;             // categories = K (+1 for DP)
    int KT;             // Kc rounded up to the kernel's accumulator count
    double beta, gamma, a, b;
    int sample_alpha;
    uint64_t seed;
};
__host__ __device__ inline TableLayout layout_of(const ChainParams& p, bool own_tables) {
    return TableLayout{p.G, p.KT, own_tables ? p.Gm : 0, 1 << p.W};
}

// The integer delta accumulators exist kDeltaReps times (replica r of dS at dS + r*K*P, of dNk at
// dNk + r*K).  Workgroup b of a launch adds into replica b % kDeltaReps: at the end of a short
// launch every workgroup flushes its histogram at the same moment, and device-scope atomics on one
// word serialise (about 12 ns each), so 250 workgroups on the same 1-2 k words cost microseconds;
// eight replicas cut the queue per word eightfold.  Consumers sum (and clear) the replicas.
#ifndef BMM_DELTA_REPS
#define BMM_DELTA_REPS 8
#endif
constexpr int kDeltaReps = BMM_DELTA_REPS;
__device__ __forceinline__ int32_t delta_take(int32_t* d, size_t idx, size_t stride) {
    int32_t v[kDeltaReps];
#pragma unroll
    for (int r = 0; r < kDeltaReps; ++r) v[r] = d[idx + r * stride];  // independent loads, one round trip
    int32_t t = 0;
#pragma unroll
    for (int r = 0; r < kDeltaReps; ++r) t += v[r];
    return t;
}
__device__ __forceinline__ void delta_clear(int32_t* d, size_t idx, size_t stride) {
#pragma unroll
    for (int r = 0; r < kDeltaReps; ++r) d[idx + r * stride] = 0;
}

// update_alpha_ (bmm_spec.h) with its four gamma variates -- independent Philox streams by construction --
// drawn by lanes 0..3 of one wave side by side instead of one after the other (each is a rejection loop
// of logs and square roots: about 1.5 us of latency apiece for a single lane); lane 0 combines them with
// the very operations of the serial form, so the result is bit-identical.  Call with the whole wave;
// the value is valid on lane 0.
__device__ __forceinline__ double update_alpha_wave(double alpha_old, double a, double b, double N, int K,
                                                    uint64_t seed, uint32_t sweep, int lane) {
    double g = 0.0;
    if (lane < 4) {
        const double shape = lane == 0 ? alpha_old + 1.0 : (lane == 1 ? N : (lane == 2 ? a + (double)K : a + (double)K - 1.0));
        Stream st = make_stream(seed, lane == 1 ? 1u : 0u, sweep,
                                lane < 2 ? kStreamAlphaEta : (lane == 2 ? kStreamAlphaG1 : kStreamAlphaG2));
        g = rgamma_(shape, st);
    }
    const double x = __shfl(g, 0), y = __shfl(g, 1), g1 = __shfl(g, 2), g2 = __shfl(g, 3);
    const double eta = div_(x, x + y);  // rbeta_
    const double b_eps = b - log_(eta);
    const double pi1 = a + (double)K - 1.0;
    const double pi2 = N * b_eps;
    const double pi = div_(pi1, pi1 + pi2);
    const double scale = div_(1.0, b_eps);
    const double ga = g1 * scale;
    const double gb = g2 * scale;
    return pi * ga + (1.0 - pi) * gb;
}

// ---------------------------------------------------------------------------------
// Table construction: one workgroup per category.  For the counting samplers it first
// folds the pending integer deltas of its cluster into the statistics.
// ---------------------------------------------------------------------------------
// e1/e0 hold the `pc` features of one chunk; its `gc` groups start at global group g0
// `c` is the category's constant term: it goes into the entries of global group 0
template <int W>
__device__ __forceinline__ void write_group_tables_w(const double* e1, const double* e0, int pc, int g0, int gc,
                                                   int KT, int k, double c, double* T) {
    constexpr int M = 1 << W;
    for (int idx = threadIdx.x; idx < gc * M; idx += blockDim.x) {
        const int g = idx / M;
        const unsigned m = idx % M;
        const double t = group_entry(e1, e0, g, pc, m, W);
        T[((size_t)(g0 + g) * KT + k) * M + m] = g0 + g == 0 ? c + t : t;
    }
}

// W = the shape's width (kGroupW or kGroupWAlt) or the own-cluster tables' kGroupWm; uniform
__device__ __forceinline__ void write_group_tables(int W, const double* e1, const double* e0, int pc, int c0,
                                                   int KT, int k, double c, double* T) {
    const int g0 = c0 / W, gc = (pc + W - 1) / W;
    if (W == kGroupW) write_group_tables_w<kGroupW>(e1, e0, pc, g0, gc, KT, k, c, T);
    else if (W == kGroupWAlt) write_group_tables_w<kGroupWAlt>(e1, e0, pc, g0, gc, KT, k, c, T);
    else write_group_tables_w<kGroupWm>(e1, e0, pc, g0, gc, KT, k, c, T);
}

constexpr int kCountTablesThreads = 576;
__global__ __launch_bounds__(kCountTablesThreads) void k_count_tables(ChainParams p, int32_t* __restrict__ Nk,
                                                                     int32_t* __restrict__ S,
                                                                     int32_t* __restrict__ dNk,
                                                                     int32_t* __restrict__ dS,
                                                                     const double* __restrict__ alpha_ptr,
                                                                     double* __restrict__ tab) {
    __shared__ double e1[kMaxP], e0[kMaxP], m1[kMaxP], m0[kMaxP], cst[4];  // cst: Cp, Cm, the two denominators
    const int k = blockIdx.x;
    const TableLayout L = layout_of(p, true);
    const int P = p.P;
    const bool is_label = k < p.K;
    // 576 threads, ONE log_ each on the critical path (a log_ is about 150 dependent instructions).  Waves
    // 0-7: role r = thread / 128 -- the term of feature d for x = 1 and for x = 0 against the full
    // statistics (r = 0, 1) and with the scored observation removed (r = 2, 3).  Wave 8: the cluster's
    // constants, one log per lane (the two denominators, the prior terms), combined by its lane 0 with the
    // operations of the serial form; the term threads subtract the denominators after a barrier.
    const int role = threadIdx.x >> 7;
    const int dl = role < 4 ? (threadIdx.x & 127) : kMaxP;
    // every load this workgroup depends on goes out first, in one round trip: the cluster size
    // (read by every thread: a broadcast, no LDS hand-over), the concentration, the first chunk of counts
    int32_t n_old = 0, n_dl = 0, s_old = 0, s_dl = 0;
    const size_t KP = (size_t)p.K * P;
    if (is_label) { n_old = Nk[k]; n_dl = delta_take(dNk, k, p.K); }
    if (is_label && dl < P) { s_old = S[(size_t)k * P + dl]; s_dl = delta_take(dS, (size_t)k * P + dl, KP); }
    const double alpha = *alpha_ptr;
    const int64_t n = (int64_t)n_old + n_dl;
    const double bg = p.beta + p.gamma;
    if (role == 4) {
        const int lane = threadIdx.x & 63;
        const bool dp_new = p.mode == MODE_DP && k == p.K;
        const double ak = p.mode == MODE_COLLAPSED ? div_(alpha, (double)p.K) : 0.0;
        double arg = 1.0;
        bool need = false;
        switch (lane) {
            case 0: arg = bg + (double)n; need = is_label && n > 0; break;                 // log(beta+gamma+n)
            case 1: arg = bg + (double)(n - 1); need = is_label && n > 1; break;           // ... with one removed
            case 2: arg = (double)n + ak; need = is_label && n > 0; break;                 // log(n + alpha/K), log n
            case 3: arg = (double)(n - 1) + ak; need = is_label && n > 1; break;
            case 4: arg = (double)(p.Ntot - 1) + alpha; need = true; break;                // log(N - 1 + alpha)
            case 5: arg = alpha; need = dp_new; break;
            case 6: arg = p.beta; need = dp_new; break;
            case 7: arg = bg; need = dp_new; break;
            default: break;
        }
        const double v = need ? log_(arg) : 0.0;
        const double den_p = __shfl(v, 0), den_m = __shfl(v, 1), ln = __shfl(v, 2), lm = __shfl(v, 3);
        const double ldN = __shfl(v, 4), la = __shfl(v, 5), lb = __shfl(v, 6), lbg = __shfl(v, 7);
        if (lane == 0) {
            double cp = neg_inf(), cm = neg_inf();
            if (is_label) {
                if (n > 0) cp = ln - ldN;
                if (n > 1) cm = lm - ldN;
            } else if (dp_new) {
                cp = (la - ldN) + (double)P * (lb - lbg);
            }
            tab[L.cp() + k] = cp;
            tab[L.cm() + k] = cm;
            cst[0] = cp; cst[1] = cm; cst[2] = den_p; cst[3] = den_m;  // read after the first barrier below
            reinterpret_cast<int32_t*>(tab + L.nk())[k] = (int32_t)n;
        }
    }
    for (int c0 = 0; c0 < P; c0 += kChunkP) {  // kChunkP features (whole groups) at a time
        const int pc = P - c0 < kChunkP ? P - c0 : kChunkP;
        int32_t s = 0;
        double raw = 0.0;
        bool have = false;
        if (dl < pc && is_label) {
            const int d = c0 + dl;
            s = c0 == 0 ? s_old + s_dl : S[(size_t)k * P + d] + delta_take(dS, (size_t)k * P + d, KP);
            // term_x1 / term_x0 of bmm_spec.h, the denominator subtracted below
            if (role == 0) { have = n > 0; raw = have ? log_(p.beta + (double)s) : 0.0; }
            else if (role == 1) { have = n > 0; raw = have ? log_((p.gamma + (double)n) - (double)s) : 0.0; }
            else if (role == 2) { have = n > 1 && s >= 1; raw = have ? log_(p.beta + (double)((int64_t)s - 1)) : 0.0; }
            else { have = n > 1 && s <= n - 1; raw = have ? log_((p.gamma + (double)(n - 1)) - (double)s) : 0.0; }
        }
        __syncthreads();  // the constants are in place; every role has read S + dS before either is rewritten
        if (dl < pc) {
            const double t = have ? raw - cst[role < 2 ? 2 : 3] : 0.0;
            (role == 0 ? e1 : role == 1 ? e0 : role == 2 ? m1 : m0)[dl] = t;
            if (role == 0 && is_label) {
                const int d = c0 + dl;
                S[(size_t)k * P + d] = s;
                delta_clear(dS, (size_t)k * P + d, KP);
            }
        }
        __syncthreads();
        write_group_tables(p.W, e1, e0, pc, c0, p.KT, k, cst[0], tab + L.tp());
        write_group_tables(kGroupWm, m1, m0, pc, c0, p.KT, k, cst[1], tab + L.tm());
        __syncthreads();
    }
    if (is_label && threadIdx.x == 0) {  // every thread read the old pair before the barriers above
        Nk[k] = (int32_t)n;
        delta_clear(dNk, k, p.K);
    }
    if (k == 0 && threadIdx.x < 256) tab[L.et() + threadIdx.x] = exp256_table()[threadIdx.x];
}

// Stick-breaking: theta_kd ~ Beta(beta + V_kd, gamma + c_k - V_kd) (stickbreaking.cpp:217-229),
// unless draw == 0 (initial theta is used as given), then the tables of cluster k.
__global__ __launch_bounds__(256) void k_sb_theta_tables(ChainParams p, const int32_t* __restrict__ Nk,
                                                         const int32_t* __restrict__ S,
                                                         const double* __restrict__ pi,
                                                         double* __restrict__ theta, int draw,
                                                         uint32_t sweep, double* __restrict__ theta_trace,
                                                         double* __restrict__ tab,
                                                         double* __restrict__ alpha_ptr,
                                                         double* __restrict__ alpha_trace,
                                                         const int* __restrict__ viable) {
    // 256 threads: a Beta draw is X / (X + Y) of two gammas on their own streams, so threads 0-127 draw
    // X and log theta for feature d while threads 128-255 draw Y and log(1 - theta): half the latency
    __shared__ double e1[kMaxP], e0[kMaxP], gam[2][kMaxP], cst;
    if (blockIdx.x == (unsigned)p.KT) {
        // one workgroup past the clusters' (launched with the draws of a sweep only): the concentration, four
        // gammas side by side (update_alpha_wave; stickbreaking.cpp:233-235, full_gibbs.cpp:228-230), from the count
        // k_sb_params left.  Nothing in this kernel reads alpha; the next sweep's k_sb_params does.
        if (threadIdx.x < 64) {
            const double alpha_prev = *alpha_ptr;
            double alpha_new = alpha_prev;
            if (p.sample_alpha)
                alpha_new = update_alpha_wave(alpha_prev, p.a, p.b, (double)p.Ntot, *viable, p.seed, sweep, threadIdx.x);
            if (threadIdx.x == 0) {
                if (p.sample_alpha) *alpha_ptr = alpha_new;
                if (alpha_trace) *alpha_trace = alpha_new;
            }
        }
        return;
    }
    const int k = blockIdx.x;
    const TableLayout L = layout_of(p, false);
    const int P = p.P, K = p.K;
    const bool is_label = k < K;
    const int half = threadIdx.x >> 7, dl = threadIdx.x & 127;
    if (threadIdx.x == 255) {  // the constants, ahead of its own feature (if P reaches 128)
        cst = is_label ? log_(pi[k]) : neg_inf();  // read after the barriers below
        tab[L.cp() + k] = cst;
        tab[L.cm() + k] = neg_inf();
        reinterpret_cast<int32_t*>(tab + L.nk())[k] = is_label ? Nk[k] : 0;
    }
    for (int c0 = 0; c0 < P; c0 += kChunkP) {
        const int pc = P - c0 < kChunkP ? P - c0 : kChunkP;
        const int d = c0 + dl;
        if (draw && is_label && dl < pc) {
            const int32_t ck = Nk[k], V = S[(size_t)k * P + d];
            const uint32_t c0s = (uint32_t)((size_t)k * P + d);
            Stream st = make_stream(p.seed, c0s, sweep, half ? kStreamThetaB : kStreamThetaA);
            gam[half][dl] = rgamma_(half ? (p.gamma + (double)ck) - (double)V : p.beta + (double)V, st);
        }
        __syncthreads();
        if (dl < pc) {
            double t = 0.0;
            if (is_label) {
                const double x = gam[0][dl];
                const double th = draw ? div_(x, x + gam[1][dl]) : theta[k + (size_t)d * K];  // rbeta_
                if (half == 0) {
                    if (draw) theta[k + (size_t)d * K] = th;
                    if (theta_trace) theta_trace[k + (size_t)d * K] = th;
                    t = log_(th);
                } else {
                    t = log_(1.0 - th);
                }
            }
            if (half == 0) e1[dl] = t; else e0[dl] = t;
        }
        __syncthreads();
        write_group_tables(p.W, e1, e0, pc, c0, p.KT, k, cst, tab + L.tp());
        __syncthreads();
    }
    if (k == 0) tab[L.et() + threadIdx.x] = exp256_table()[threadIdx.x];  // 256 threads
}

// Stick-breaking: fold deltas, v_k ~ Beta(1 + c_k, alpha + sum_{l>k} c_l), pi by stick
// breaking, K_viable, alpha (stickbreaking.cpp:164-214, 233-235).  One workgroup.
__global__ __launch_bounds__(1024) void k_sb_params(ChainParams p, int32_t* __restrict__ Nk,
                                                   int32_t* __restrict__ S, int32_t* __restrict__ dNk,
                                                   int32_t* __restrict__ dS, const double* __restrict__ alpha_ptr,
                                                   double* __restrict__ pi, uint32_t sweep,
                                                   double* __restrict__ pi_trace, int pi_stride,
                                                   int* __restrict__ viable_out,
                                                   int32_t* __restrict__ nk_trace) {
    __shared__ int32_t ck[kMaxCatsAny];
    __shared__ double v[kMaxCatsAny];
    const int K = p.K, P = p.P;
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        const int32_t n = Nk[k] + delta_take(dNk, k, K);
        Nk[k] = n; delta_clear(dNk, k, K); ck[k] = n;
        if (nk_trace) nk_trace[k] = n;
    }
    __syncthreads();
    const double alpha_prev = *alpha_ptr;
    // the first wave draws the sticks (they need the cluster sizes only) while the others fold the
    // feature counts, which only the next kernel reads
    const int nfold = blockDim.x > 64 ? blockDim.x - 64 : blockDim.x;
    const int t0 = blockDim.x > 64 ? (int)threadIdx.x - 64 : (int)threadIdx.x;
    for (int idx = t0 < 0 ? K * P : t0; idx < K * P; idx += nfold) {
        S[idx] += delta_take(dS, idx, (size_t)K * P); delta_clear(dS, idx, (size_t)K * P);
    }
    const int ndraw = blockDim.x > 64 ? 64 : blockDim.x;
    for (int k = threadIdx.x < (unsigned)ndraw ? threadIdx.x : K; k < K; k += ndraw) {
        Stream sa = make_stream(p.seed, (uint32_t)k, sweep, kStreamStickA);
        if (p.mode == MODE_FULL) {
            // pi ~ Dirichlet(alpha/K + c_k) through gammas (full_gibbs.cpp:10-27, 203-210)
            v[k] = rgamma_(div_(alpha_prev, (double)K) + (double)ck[k], sa);
        } else {
            int64_t prev = 0;
            for (int l = k + 1; l < K; ++l) prev += ck[l];
            Stream sb = make_stream(p.seed, (uint32_t)k, sweep, kStreamStickB);
            v[k] = rbeta_(1.0 + (double)ck[k], alpha_prev + (double)prev, sa, sb);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int viable = 0;
        if (p.mode == MODE_FULL) {
            double sum_term = 0.0;
            for (int k = 0; k < K; ++k) sum_term = sum_term + v[k];
            for (int k = 0; k < K; ++k) {
                const double pk = div_(v[k], sum_term);
                pi[k] = pk;
                if (pi_trace) pi_trace[(size_t)k * pi_stride] = pk;
            }
            viable = K;  // update_alpha(..., N, K) at full_gibbs.cpp:228-230
        } else {
            v[K - 1] = 1.0;
            double cumprod = 1.0;
            for (int k = 0; k < K; ++k) {
                const double pk = k == 0 ? v[0] : cumprod * v[k];
                if (pk > 0.01) ++viable;
                cumprod = k == 0 ? 1.0 - v[0] : cumprod * (1.0 - v[k]);
                pi[k] = pk;
                if (pi_trace) pi_trace[(size_t)k * pi_stride] = pk;
            }
        }
        // the concentration's update needs this count and nothing else of the sweep: it runs beside the theta
        // draws, in a workgroup of its own of the next kernel (k_sb_theta_tables), off this kernel's critical path
        *viable_out = viable;
    }
}

// Fold replicas 1.. of the delta accumulators into replica 0 and clear them: ahead of anything that
// hands the deltas to the host or to a collective (bmm_chain_get_counts, bmm_chain_shard_deltas).
__global__ __launch_bounds__(256) void k_reduce_deltas(ChainParams p, int32_t* __restrict__ dNk,
                                                       int32_t* __restrict__ dS) {
    const int KP = p.K * p.P;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= KP + p.K) return;
    int32_t* d = idx < KP ? dS : dNk;
    const size_t i = idx < KP ? idx : idx - KP, stride = idx < KP ? KP : p.K;
    const int32_t v = delta_take(d, i, stride);
    delta_clear(d, i, stride);
    d[i] = v;
}

// Counting samplers, end of sweep: fold what the last batch left, theta-hat = S/Nk
// (NaN for an empty cluster in the finite sampler, 0 for an unused DP label), alpha.
__global__ __launch_bounds__(1024) void k_count_sweep_end(ChainParams p, int32_t* __restrict__ Nk,
                                                         int32_t* __restrict__ S,
                                                         int32_t* __restrict__ dNk,
                                                         int32_t* __restrict__ dS,
                                                         double* __restrict__ alpha_ptr, uint32_t sweep,
                                                         double* __restrict__ theta_trace,
                                                         double* __restrict__ alpha_trace,
                                                         int32_t* __restrict__ nk_trace) {
    __shared__ int32_t nk[kMaxCatsAny];
    const int K = p.K, P = p.P;
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        const int32_t n = Nk[k] + delta_take(dNk, k, K);
        Nk[k] = n; delta_clear(dNk, k, K); nk[k] = n;
        if (nk_trace) nk_trace[k] = n;
    }
    __syncthreads();
    // the first wave draws alpha (below) while the others fold the counts
    const int nfold = blockDim.x > 64 ? blockDim.x - 64 : blockDim.x;
    const int t0 = blockDim.x > 64 ? (int)threadIdx.x - 64 : (int)threadIdx.x;
    for (int idx = t0 < 0 ? K * P : t0; idx < K * P; idx += nfold) {
        const int k = idx / P, d = idx % P;
        const int32_t s = S[idx] + delta_take(dS, idx, (size_t)K * P);
        S[idx] = s; delta_clear(dS, idx, (size_t)K * P);
        if (theta_trace) {
            double t;
            if (p.mode == MODE_DP && nk[k] == 0) t = 0.0;
            else t = div_((double)s, (double)nk[k]);
            theta_trace[k + d * K] = t;
        }
    }
    if (threadIdx.x < 64) {  // the first wave: the concentration, four gammas side by side
        double al = *alpha_ptr;
        if (p.sample_alpha) {
            int Kc = K;
            if (p.mode == MODE_DP) { Kc = 0; for (int k = 0; k < K; ++k) Kc += nk[k] > 0; }
            al = update_alpha_wave(al, p.a, p.b, (double)p.Ntot, Kc, p.seed, sweep, threadIdx.x);
        }
        if (threadIdx.x == 0) {
            if (p.sample_alpha) *alpha_ptr = al;
            if (alpha_trace) *alpha_trace = al;
        }
    }
}

// ---------------------------------------------------------------------------------
// The z-resample kernel.
// ---------------------------------------------------------------------------------
struct ResampleArgs {
    const int32_t* X;
    const uint32_t* Xb;    // bit planes of X: word w of observation i at Xb[w * N + i] (k_pack_bits)
    const int32_t* z_in;   // 0-based labels of the previous sweep, -1 = unassigned
    int32_t* z_out;
    const double* tab;     // table image (TableLayout)
    int32_t* dNk;          // global delta accumulators
    int32_t* dS;
    int64_t lo, hi;        // batch [lo, hi)
    uint32_t sweep;
    int minus_in_lds;      // 1: the Tm tables were sized into LDS too
    double* wts;           // or null: this sweep also emits the draw's weights, wts[k * N + i] for the Kc
    double* wtot;          //   categories in order, and their total wtot[i] (k_probs_finish normalises)
    unsigned long long* diag;  // BMM_DIAG builds: [5] cycle sums (score, pack, draw, movers, prologue)
    int* dbg_flag;         // -DBMM_DEBUG_HOOKS builds: set when a kernel meets a label outside its range
    int dbg_inject;        //   ... and a test's way to make one (BMM_DEBUG_BADLABEL)
    // SELF kernels (workgroups that build their own table image): the folded statistics, the deltas the previous
    // launch left (dNk / dS above are the set this launch flushes into), the concentration, and the count of
    // workgroups that have read all of that (self_fold_prev)
    int32_t* Nk;
    int32_t* S;
    int32_t* dNk_prev;
    int32_t* dS_prev;
    const double* alpha_ptr;
    int* self_done;
};

// The test variant of the library (-DBMM_DEBUG_HOOKS) checks every label a resample kernel is about to count
// with: 0 <= new < K, -1 <= old < K.  The LDS histogram is indexed by label without a bound, so a wrong
// experimental kernel would otherwise scribble over LDS (round 2 lost a GPU call to exactly that).  A bad
// label raises the chain's flag -- the host turns it into BMM_E_STATE at the next synchronisation -- and the
// observation is left unassigned and uncounted.  The product build carries none of this.
#ifdef BMM_DEBUG_HOOKS
__device__ __forceinline__ void dbg_check_labels(const ResampleArgs& a, bool valid, bool first, int K, int& zn, int& zo) {
    if ((a.dbg_inject & 1) && valid && first) zn = K + 3;
    if (valid && (zn < 0 || zn >= K || zo < -1 || zo >= K)) {
        atomicOr(a.dbg_flag, 1);
        zn = -1; zo = -1;
    }
}
#define BMM_DBG_LABELS(a, valid, first, K, zn, zo) dbg_check_labels(a, valid, first, K, zn, zo)
#else
#define BMM_DBG_LABELS(a, valid, first, K, zn, zo)
#endif

// Where a lane sits in a tile of NT consecutive observations.  Lanes past the end of the
// batch re-read its last observation (and never write back).
struct TilePos {
    const char* base;   // wave-uniform byte address of X[wave_base + 0*N]
    uint32_t voff;      // this lane's byte offset from it (>= 0)
    int64_t i, ic;      // own observation, clamped observation
    bool valid;
};
__device__ __forceinline__ TilePos tile_pos(const ResampleArgs& a, int64_t tile, int NT, int tid, int lane) {
    TilePos t;
    const int64_t tile_base = a.lo + tile * NT;
    t.i = tile_base + tid;
    t.valid = t.i < a.hi;
    int64_t wave_base = tile_base + (int64_t)__builtin_amdgcn_readfirstlane(tid & ~63);  // SGPR
    wave_base = wave_base < a.hi ? wave_base : a.hi - 1;
    const int64_t room = a.hi - 1 - wave_base;
    const int lane_off = lane < room ? lane : (int)room;
    t.ic = wave_base + lane_off;
    t.voff = (uint32_t)lane_off * 4u;
    t.base = reinterpret_cast<const char*>(a.X + wave_base);
    return t;
}
// Issue the loads of bit-word wd (features 32*wd ...): 32 coalesced dword loads per lane,
// wave-uniform base + zero-extended 32-bit lane offset (global_load saddr form).
// buffer form: the descriptor (SGPRs) carries the wave-uniform column base and is advanced
// by the column stride between loads; the lane offset is one shared VGPR.  No per-load
// address registers: the STG loads of one stage are in flight from STG + 1 VGPRs.
// STG = features per pipeline stage (16 or 32 = 4 or 8 lookup groups); template parameter below

template <int STG>
__device__ __forceinline__ void issue_stage(const TilePos& t, int64_t N, int P, int h, uint32_t (&st)[STG]) {
    const int d0 = h * STG;
    const int nb = P - d0 < STG ? P - d0 : STG;
    const int64_t stride = N * 4;
    const char* col = t.base + (int64_t)d0 * stride;
    // a partial last stage loads whole groups of four only (the tail re-reads feature P-1: same
    // cache lines; pack_stage masks it off)
#pragma unroll
    for (int u0 = 0; u0 < STG; u0 += 4) {
        if (u0 < nb) {
#pragma unroll
            for (int u = u0; u < u0 + 4; ++u) {
                const __amdgpu_buffer_rsrc_t rsrc =
                    __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(col), 0, 0x7fffffff, 0x00020000);
                st[u] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)t.voff, 0, 0);
                col += d0 + u + 1 < P ? stride : 0;
            }
        }
    }
}
template <int STG>
__device__ __forceinline__ uint32_t pack_stage(int P, int h, const uint32_t (&st)[STG]) {
    const int nb = P - h * STG < STG ? P - h * STG : STG;
    uint32_t v = 0;
#pragma unroll
    for (int u = 0; u < STG; ++u) v |= st[u] << u;  // X is validated to be 0/1 when it is set
    return nb >= 32 ? v : (v & ((1u << nb) - 1u));
}
// bits of stage h live in word h*STG/32 at bit (h*STG)%32
template <int STG>
__device__ __forceinline__ void put_stage(uint32_t v, int h, uint32_t& b0, uint32_t& b1, uint32_t& b2, uint32_t& b3) {
    const uint32_t sh = STG == 32 ? v : v << ((h & 1) * 16);
    const int w = STG == 32 ? h : h >> 1;
    b0 |= w == 0 ? sh : 0u;
    b1 |= w == 1 ? sh : 0u;
    b2 |= w == 2 ? sh : 0u;
    b3 |= w == 3 ? sh : 0u;
}
// Word w of the observation's 128-bit pattern (w uniform; past the last word: 0).  The field of lookup
// group g, bits [g*W, (g+1)*W), may straddle two words.
__device__ __forceinline__ uint32_t word_of(int w, uint32_t b0, uint32_t b1, uint32_t b2, uint32_t b3) {
    return w == 0 ? b0 : (w == 1 ? b1 : (w == 2 ? b2 : (w == 3 ? b3 : 0u)));
}

// Sufficient-statistic deltas of one wave's movers into the workgroup's LDS histogram (integer LDS
// atomics: order-independent, deterministic).  Cluster sizes: every mover lane adds its own +1 / -1.
// Feature counts, two forms chosen per wave by the number of movers:
//   few movers    one mover at a time by the whole wave: its bits broadcast by readlane, one feature
//                 per lane (conflict-free), about 40 instructions per mover;
//   many movers   every mover lane walks the set bits of its own words (LDS atomics resolve lanes that
//                 meet on a cell): about 10 instructions per set bit of the densest mover, whatever the
//                 number of movers -- the first sweeps from a random allocation, where all 64 lanes
//                 move, and samplers whose posteriors keep many observations undecided.
__device__ __forceinline__ void count_movers(bool moves, int zfrom, int zto, uint32_t b0, uint32_t b1,
                                             uint32_t b2, uint32_t b3, int32_t* hist, int K, int P, int lane) {
    unsigned long long movers = __ballot(moves);
    if (!movers) return;
    if (moves) {
        atomicAdd(&hist[K * P + zto], 1);
        if (zfrom >= 0) atomicAdd(&hist[K * P + zfrom], -1);
    }
    const int thr = P >= 16 ? P >> 3 : 2;
    if ((int)__popcll(movers) > thr) {  // uniform
        if (moves) {
            int32_t* const hn = hist + zto * P;
            int32_t* const ho = hist + (zfrom < 0 ? 0 : zfrom) * P;
            const int W = (P + 31) >> 5;
#pragma unroll 1
            for (int w = 0; w < W; ++w) {
                uint32_t bits = w == 0 ? b0 : (w == 1 ? b1 : (w == 2 ? b2 : b3));
                while (bits) {
                    const int d = (w << 5) + __ffs((int)bits) - 1;
                    bits &= bits - 1;
                    atomicAdd(&hn[d], 1);
                    if (zfrom >= 0) atomicAdd(&ho[d], -1);
                }
            }
        }
        return;
    }
    while (movers) {
        const int src = __ffsll((long long)movers) - 1;
        movers &= movers - 1;
        const int mzn = __builtin_amdgcn_readlane(zto, src);
        const int mzo = __builtin_amdgcn_readlane(zfrom, src);
        const uint32_t w0 = __builtin_amdgcn_readlane(b0, src), w1 = __builtin_amdgcn_readlane(b1, src);
        const uint32_t w2 = __builtin_amdgcn_readlane(b2, src), w3 = __builtin_amdgcn_readlane(b3, src);
        const uint32_t lo_w = lane < 32 ? w0 : w1, hi_w = lane < 32 ? w2 : w3;
        const int sh = lane & 31;
        if (lane < P && ((lo_w >> sh) & 1u)) {
            atomicAdd(&hist[mzn * P + lane], 1);
            if (mzo >= 0) atomicAdd(&hist[mzo * P + lane], -1);
        }
        if (lane + 64 < P && ((hi_w >> sh) & 1u)) {
            atomicAdd(&hist[mzn * P + lane + 64], 1);
            if (mzo >= 0) atomicAdd(&hist[mzo * P + lane + 64], -1);
        }
    }
}
__device__ __forceinline__ void flush_hist(const int32_t* hist, int K, int P, int32_t* dS, int32_t* dNk,
                                           int tid, int nt) {
    for (int i = tid; i < K * P + K; i += nt) {
        const int32_t v = hist[i];
        if (v != 0) {
            if (i < K * P) atomicAdd(&dS[i], v);
            else atomicAdd(&dNk[i - K * P], v);
        }
    }
}

// X as bit planes: word w of observation i (features 32w .. 32w+31, feature d at bit d % 32) at
// Xb[w * N + i]; bits past P are zero.  X is constant over the whole chain, so the resample kernel
// streams 4 * ceil(P / 32) bytes per observation and sweep instead of 4 * P.
__global__ __launch_bounds__(256) void k_pack_bits(const int32_t* __restrict__ X, int64_t rows, int64_t ldx,
                                                   int P, uint32_t* __restrict__ Xb, int64_t N) {
    // X: `rows` observations, feature d of observation i at X[i + d * ldx]; Xb already offset to the
    // first of them, planes N words apart
    const int W = (P + 31) / 32;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < rows; i += (int64_t)gridDim.x * 256) {
        for (int w = 0; w < W; ++w) {
            uint32_t v = 0;
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                const int d = w * 32 + j;
                if (d < P) v |= ((uint32_t)X[i + (int64_t)d * ldx] & 1u) << j;
            }
            Xb[(int64_t)w * N + i] = v;
        }
    }
}
// the (up to four) words of observation i; W = ceil(P / 32) is wave-uniform
__device__ __forceinline__ void load_words(const uint32_t* Xb, int64_t N, int W, int64_t i, uint32_t& w0,
                                           uint32_t& w1, uint32_t& w2, uint32_t& w3) {
    w0 = Xb[i];
    w1 = W > 1 ? Xb[N + i] : 0u;
    w2 = W > 2 ? Xb[2 * N + i] : 0u;
    w3 = W > 3 ? Xb[3 * N + i] : 0u;
}

// One pass over X when it is handed over: every cell must be 0 or 1 (the packing above
// shifts the loaded words without masking).  flag[0] is set when one is not.
__global__ __launch_bounds__(256) void k_validate_binary(const uint4* __restrict__ X4, int64_t n16,
                                                         const uint32_t* __restrict__ X, int64_t n,
                                                         int* __restrict__ flag) {
    uint32_t bad = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) {
        const uint4 v = X4[i];
        bad |= (v.x | v.y | v.z | v.w) & ~1u;
    }
    for (int64_t i = n16 * 4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        bad |= X[i] & ~1u;
    if (__any(bad != 0) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

// Statistics of a given allocation (the collapsed sampler's initial labels,
// collapsed_gibbs.cpp:60-63): every labelled observation counts as arriving.
__global__ __launch_bounds__(256) void k_count_labels(ChainParams p, const int32_t* __restrict__ X,
                                                      const uint32_t* __restrict__ Xb,
                                                      const int32_t* __restrict__ z, int32_t* dNk,
                                                      int32_t* dS) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int32_t* const hist = reinterpret_cast<int32_t*>(smem);
    const int P = p.P, K = p.K, tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < K * P + K; i += 256) hist[i] = 0;
    __syncthreads();
    ResampleArgs a{};
    a.X = X; a.lo = 0; a.hi = p.N;
    const int64_t ntiles = (p.N + 255) / 256;
    const int nstages = (P + 15) / 16;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const TilePos pos = tile_pos(a, tile, 256, tid, lane);
        uint32_t st[16], b0 = 0, b1 = 0, b2 = 0, b3 = 0;
#pragma unroll
        for (int u = 0; u < 16; ++u) st[u] = 0;
        if (Xb) {  // bit planes (uniform)
            load_words(Xb, p.N, (P + 31) / 32, pos.ic, b0, b1, b2, b3);
        } else {
#pragma unroll 1
            for (int h = 0; h < nstages; ++h) {
                issue_stage<16>(pos, p.N, P, h, st);
                put_stage<16>(pack_stage<16>(P, h, st), h, b0, b1, b2, b3);
            }
        }
        const int zl = z[pos.ic];
        count_movers(pos.valid && zl >= 0, -1, zl, b0, b1, b2, b3, hist, K, P, lane);
    }
    __syncthreads();
    flush_hist(hist, K, P, dS + (size_t)(blockIdx.x % kDeltaReps) * K * P, dNk + (blockIdx.x % kDeltaReps) * K, tid, 256);
}

// Diagnostic build only (-DBMM_DIAG, never shipped): per-wave cycle stamps of the phases.
#ifdef BMM_DIAG
__device__ __forceinline__ unsigned long long diag_stamp() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define DIAG(...) __VA_ARGS__
#else
#define DIAG(...)
#endif

// The z-resample kernel.  Software pipeline per wave, two levels:
//  * HBM: while tile t is scored, the STG loads of feature stage h of tile t+1 are in
//    flight; they are issued at the top of outer iteration h and packed at its bottom, after the
//    stage's own STG/4 lookup groups of tile t (nothing in flight is ever loop-carried);
//  * LDS: the K lookups of a group are issued together and added as they return (volatile:
//    one ds_read_b64 each -- the compiler would otherwise pair them into ds_read2_b64,
//    which moves half the bytes per clock).
// MINUS says where the own-cluster ("minus self") tables are: 0 none (stick-breaking),
// 1 in LDS, 2 in global memory.  It is a template parameter because a possible VMEM load in
// the lookup loop makes the compiler wait vmcnt(0) there, which would drain the HBM prefetch.
// SPLIT = 2 (bit planes only, for more than 32 accumulators): an observation is shared by lanes l
// and l + 32 of a wave, each scoring half of the categories -- half the accumulator registers per
// lane, so twice the waves per SIMD; the halves meet through three lane exchanges (maximum, running
// sum, count), all in the order the one-lane form uses, so the draw is bit-identical.
#ifndef BMM_LOOKUP_PRIO
#define BMM_LOOKUP_PRIO 2
#endif

// SELF: k_count_tables' work done by every resample workgroup for itself, straight into its LDS image -- for the
// finite sampler on shapes so small that the whole table build is at most two logs per thread (K (4P + 5) <= 2 NT:
// BASELINE config 2, K = 3, P = 20, is 255 logs: one per thread of a 256-thread workgroup; each further round of
// logs costs a launch about 2 us, and from the third on the table launch it replaces was cheaper).  Such shapes are bound by launches, not by work: a sweep of config 2
// is 8 table launches + 8 resample launches + 1, each a few microseconds, and this form drops the 8 table launches
// (every workgroup of a launch building the image of a bigger shape itself was measured in round 2: 8 000 logs per
// workgroup cost six times what the launch did).  The statistics are then folded by one of the launch's own
// workgroups (self_fold_prev, below).  Same functions, same operands, same order as k_count_tables and write_group_tables: the
// image is bit-identical.
// scratch (LDS, doubles): terms [K][4][P] (x=1 / x=0 against the full statistics, then with the scored
// observation removed), logs [K][8], consts [KT][2] (Cp, Cm).
__host__ __device__ inline size_t self_scratch_doubles(int K, int KT, int P) { return (size_t)K * 4 * P + (size_t)K * 8 + (size_t)KT * 2; }
__host__ __device__ inline bool self_tables_fit(int mode, int K, int P, int threads) {
    return mode == MODE_COLLAPSED && P <= kChunkP && (long)K * (4 * P + 5) <= 2L * threads;
}
// SELF kernels read the statistics themselves when a workgroup starts -- Nk, S and the deltas of the previous
// launch -- so all three must stay as the previous launch left them until every workgroup of THIS launch has
// read them: a workgroup may start late (another chain's kernels on the device, another process), after others
// of its launch have already finished and flushed.  Two sets of delta accumulators therefore take turns: a
// launch flushes into the one that is empty and reads the other; each workgroup takes a ticket (a counter in
// global memory) once its reads are done, and the one that draws the last ticket -- nobody will read the old
// values again -- folds the previous launch's deltas into Nk and S and clears them, while the others are already
// scoring.  The host swaps the two sets after every such launch, so that to every other kernel (k_count_tables
// in a sweep that emits probabilities, k_count_sweep_end, the accessors) "the deltas" are the pending set as
// before and the other set is zero.  Integer sums: the result does not depend on who folds or when.
template <int NT>
__device__ __forceinline__ void self_fold_prev(const ResampleArgs& a, int K, int P, int tid) {
    const int KP = K * P;
    for (int i = tid; i < KP + K; i += NT) {
        if (i < KP) {
            const int32_t v = delta_take(a.dS_prev, (size_t)i, (size_t)KP);
            if (v != 0) a.S[i] += v;
            delta_clear(a.dS_prev, (size_t)i, (size_t)KP);
        } else {
            const int32_t v = delta_take(a.dNk_prev, (size_t)(i - KP), (size_t)K);
            if (v != 0) a.Nk[i - KP] += v;
            delta_clear(a.dNk_prev, (size_t)(i - KP), (size_t)K);
        }
    }
    if (tid == 0) *a.self_done = 0;  // every ticket of this launch has been drawn
}

template <int KT, int NT, int GW>
__device__ __forceinline__ void build_tables_self(const ChainParams& p, const ResampleArgs& a, const TableLayout& L,
                                                  double* lds, double* scratch, int tid, int* is_last) {
    constexpr int GM = 1 << GW;
    const int K = p.K, P = p.P;
    const size_t KP = (size_t)K * P;
    double* const terms = scratch;
    double* const logs = scratch + (size_t)K * 4 * P;
    double* const consts = logs + (size_t)K * 8;
    const double alpha = *a.alpha_ptr;
    const double bg = p.beta + p.gamma;
    const double ak = div_(alpha, (double)K);
    const int nterm = K * 4 * P, nitem = nterm + K * 5;
    // phase A: every raw log, at most two per thread, held in registers across the barrier
    constexpr int R = 2;
    double raw[R];
    bool have[R];
#pragma unroll
    for (int q = 0; q < R; ++q) {
        const int t = tid + q * NT;
        raw[q] = 0.0; have[q] = false;
        if (t < nterm) {
            const int k = t / (4 * P), role = (t / P) & 3, d = t % P;
            const int64_t n = (int64_t)a.Nk[k] + delta_take(a.dNk_prev, k, K);
            const int32_t sd = a.S[(size_t)k * P + d] + delta_take(a.dS_prev, (size_t)k * P + d, KP);
            // term_x1 / term_x0 of bmm_spec.h as k_count_tables evaluates them, the denominator subtracted below
            if (role == 0) { have[q] = n > 0; raw[q] = have[q] ? log_(p.beta + (double)sd) : 0.0; }
            else if (role == 1) { have[q] = n > 0; raw[q] = have[q] ? log_((p.gamma + (double)n) - (double)sd) : 0.0; }
            else if (role == 2) { have[q] = n > 1 && sd >= 1; raw[q] = have[q] ? log_(p.beta + (double)((int64_t)sd - 1)) : 0.0; }
            else { have[q] = n > 1 && sd <= n - 1; raw[q] = have[q] ? log_((p.gamma + (double)(n - 1)) - (double)sd) : 0.0; }
        } else if (t < nitem) {
            const int k = (t - nterm) / 5, lane = (t - nterm) % 5;
            const int64_t n = (int64_t)a.Nk[k] + delta_take(a.dNk_prev, k, K);
            double arg = 1.0;
            bool need = false;
            switch (lane) {  // the logs of k_count_tables' ninth wave, lanes 0-4 (5-7 belong to the DP's new cluster)
                case 0: arg = bg + (double)n; need = n > 0; break;
                case 1: arg = bg + (double)(n - 1); need = n > 1; break;
                case 2: arg = (double)n + ak; need = n > 0; break;
                case 3: arg = (double)(n - 1) + ak; need = n > 1; break;
                case 4: arg = (double)(p.Ntot - 1) + alpha; need = true; break;
                default: break;
            }
            logs[(size_t)k * 8 + lane] = need ? log_(arg) : 0.0;
        }
    }
    __syncthreads();
    // phase B: the terms (denominator subtracted), the constants, the cluster sizes
#pragma unroll
    for (int q = 0; q < R; ++q) {
        const int t = tid + q * NT;
        if (t < nterm) {
            const int k = t / (4 * P), role = (t / P) & 3;
            terms[t] = have[q] ? raw[q] - logs[(size_t)k * 8 + (role < 2 ? 0 : 1)] : 0.0;
        }
    }
    for (int k = tid; k < KT; k += NT) {
        double cp = neg_inf(), cm = neg_inf();
        int32_t n32 = 0;
        if (k < K) {
            const int64_t n = (int64_t)a.Nk[k] + delta_take(a.dNk_prev, k, K);
            const double* v = logs + (size_t)k * 8;
            if (n > 0) cp = v[2] - v[4];
            if (n > 1) cm = v[3] - v[4];
            n32 = (int32_t)n;
        }
        consts[2 * k] = cp; consts[2 * k + 1] = cm;
        lds[L.cp() + k] = cp;
        lds[L.cm() + k] = cm;
        reinterpret_cast<int32_t*>(lds + L.nk())[k] = n32;
    }
    __syncthreads();
    // every read of the statistics is done: the ticket (self_fold_prev); its answer is needed only after phase C
    int ticket = 0;
    if (tid == 0) ticket = atomicAdd(a.self_done, 1);
    // phase C: the group tables, laid out as write_group_tables lays them out (constant folded into group 0;
    // accumulators past K score -inf; the own-cluster tables padded with zero groups)
    for (int idx = tid; idx < L.G * KT * GM; idx += NT) {
        const int g = idx / (KT * GM), k = (idx / GM) % KT;
        const unsigned m = idx % GM;
        const double t = k < K ? group_entry(terms + (size_t)(k * 4 + 0) * P, terms + (size_t)(k * 4 + 1) * P, g, P, m, GW) : 0.0;
        lds[L.tp() + ((size_t)g * KT + k) * GM + m] = g == 0 ? consts[2 * k] + t : t;
    }
    const int gm_used = (P + kGroupWm - 1) / kGroupWm;
    for (int idx = tid; idx < L.gm_pad() * KT * kGroupMm; idx += NT) {
        const int g = idx / (KT * kGroupMm), k = (idx / kGroupMm) % KT;
        const unsigned m = idx % kGroupMm;
        double v = 0.0;
        if (g < gm_used) {
            const double t = k < K ? group_entry(terms + (size_t)(k * 4 + 2) * P, terms + (size_t)(k * 4 + 3) * P, g, P, m, kGroupWm) : 0.0;
            v = g == 0 ? consts[2 * k + 1] + t : t;
        }
        lds[L.tm() + ((size_t)g * KT + k) * kGroupMm + m] = v;
    }
    if (tid == 0) *is_last = ticket == (int)gridDim.x - 1;
    for (int i = tid; i < 256; i += NT) lds[L.et() + i] = exp256_table()[i];
}

// EMIT: the launch also writes the draw's weights and their total (a.wts, a.wtot) for the probability
// hand-off to the host's relabelling; a twin instantiation, so that the plain kernel carries no branch.
// GW: features per lookup group of the tables (the shape's width, ChainParams::W).
template <int KT, int NT, int MINUS, int STG, bool BITS = false, int SPLIT = 1, bool EMIT = false, int GW = kGroupW, bool SELF = false>
__global__ __launch_bounds__(NT) void k_resample(ChainParams p, ResampleArgs a) {
    static_assert(!SELF || (MINUS == 1 && BITS && SPLIT == 1 && !EMIT), "self-built tables: the finite sampler's plain bit-plane kernel");
    constexpr int GM = 1 << GW;  // entries per group table
    // Bit planes, one lane per observation: a wave runs its scoring loop at raised priority.  Scoring is
    // bound by the CU's LDS pipe, the draw by the SIMD's VALU; with only four waves per SIMD the VALU
    // starves whenever all four sit in their scoring loops, so a scoring wave gets its few instructions in
    // ahead of the drawing waves (it stalls on LDS most of the time anyway) and leaves the loop sooner:
    // C5 +3.5 %, c3 +2.6 % over raising the priority for the issue of the reads only (which was +2 % over
    // none); the int32 pipeline and the two-lane form lose 1-3 % with it (profiles/r02/README.md).
    constexpr bool kPrioKernel = BITS && SPLIT == 1;
    static_assert(SPLIT == 1 || (SPLIT == 2 && BITS && MINUS != 2 && KT % 2 == 0), "split form");
    constexpr int SB = BITS ? 32 : STG;  // start bits of the lookup groups one stage scores
    constexpr int KH = KT / SPLIT;      // accumulators per lane
    constexpr int OT = NT / SPLIT;      // observations per tile
    constexpr int CH = SPLIT == 2 ? (KH <= 20 ? KH : KH / 2)
                                  : (KT <= 24 ? KT : (KT <= 48 ? KT / 2 : KT / 4));  // lookups issued together
    static_assert(KH % CH == 0, "chunking");
    DIAG(const unsigned long long d_entry = diag_stamp();)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr bool has_minus = MINUS != 0;
    const TableLayout L{p.G, KT, has_minus ? p.Gm : 0, GM};
    double* const lds = reinterpret_cast<double*>(smem);
    const int lds_doubles = MINUS == 1 ? L.doubles() : L.head();
    const volatile lds_f64* const Tp = (const volatile lds_f64*)(lds + L.tp());
    const lds_f64* const TmL = (const lds_f64*)(lds + L.tm());
    const double* const TmG = a.tab + L.tm();
    const int32_t* const NkT = reinterpret_cast<const int32_t*>(lds + L.nk());
    const lds_f64* const ET = (const lds_f64*)(lds + L.et());
    int32_t* const hist = reinterpret_cast<int32_t*>(lds + lds_doubles);  // [K*P] then [K]
    const int P = p.P, G = p.G, K = p.K;
    const int tid = threadIdx.x, lane = tid & 63;
    const int half = SPLIT == 2 ? lane >> 5 : 0;  // which half of the categories this lane scores
    const int kb = half * KH;
    // Work is handed out per wave, in chunks of OW consecutive observations: workgroup b owns the chunks
    // [b * cpw, (b + 1) * cpw) of the batch; a wave's first chunk is fixed (its loads go out before the
    // tables are staged), every further one comes from a counter in LDS -- the waves of a workgroup then
    // finish within one chunk of each other whatever order the SIMDs served them in.  The statistics are
    // integer sums and every label is stored by observation index, so results do not depend on who took what.
    constexpr int OW = 64 / SPLIT;
    constexpr int NW = NT / 64;  // waves per workgroup
    const int64_t nchunks = (a.hi - a.lo + OW - 1) / OW;
    const int64_t cpw = (nchunks + gridDim.x - 1) / gridDim.x;
    const int64_t wg_c0 = (int64_t)blockIdx.x * cpw;
    const int64_t wg_cn = nchunks - wg_c0 < cpw ? nchunks - wg_c0 : cpw;  // chunks of this workgroup (may be <= 0)
    (void)OT;
    auto tpos = [&](int64_t t) -> TilePos {
        if (SPLIT == 1) return tile_pos(a, t, 64, lane, lane);
        TilePos q;  // lanes l and l + 32 of a wave stand on the same observation
        q.i = a.lo + t * OW + (lane & 31);
        q.valid = q.i < a.hi;
        q.ic = q.valid ? q.i : a.hi - 1;
        q.voff = 0; q.base = nullptr;
        return q;
    };
    // BITS: X comes as bit planes; a "stage" is then one 32-bit word = eight lookup groups, all
    // (up to four) words of the next tile being loaded with the first stage
    const int W = (P + 31) / 32;
    const int nstages = BITS ? W : (P + STG - 1) / STG;

    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int64_t tile = wg_c0 + wave;  // chunk index within the batch
    const bool has_tile = wave < wg_cn;
    int* const next_chunk = hist + K * P + K;  // LDS counter behind the histogram
    uint32_t st[STG];
#pragma unroll
    for (int u = 0; u < STG; ++u) st[u] = 0;
    TilePos pos = tpos(has_tile ? tile : 0);
    // first loads of the first tile go out before the tables are staged
    uint32_t b0 = 0, b1 = 0, b2 = 0, b3 = 0;
    if (has_tile) {
        if (BITS) load_words(a.Xb, p.N, W, pos.ic, b0, b1, b2, b3);
        else issue_stage<STG>(pos, p.N, P, 0, st);
    }
    if (SELF) {
#ifdef BMM_DEBUG_HOOKS
        // test variant, BMM_DEBUG_STRAGGLER: workgroup 0 starts about 100 us late, as it may on a device that other
        // work shares -- the others have flushed by then (tests/test_gpu_layouts.py)
        if ((a.dbg_inject & 2) && blockIdx.x == 0)
            for (int i = 0; i < 30; ++i) __builtin_amdgcn_s_sleep(127);
#endif
        // behind the histogram (and its chunk counter), on an 8-byte boundary
        double* const scratch = lds + lds_doubles + ((size_t)(K * P + K + 4) * sizeof(int32_t) + 7) / 8;
        build_tables_self<KT, NT, GW>(p, a, L, lds, scratch, tid, next_chunk + 1);  // (a spare word behind the counter)
    } else {
        // stage the table image: eight 16-byte loads in flight per lane (one L2 round trip per
        // eight, not per one)
        const double2* src = reinterpret_cast<const double2*>(a.tab);
        double2* dst = reinterpret_cast<double2*>(smem);
        const int n2 = lds_doubles / 2;
        for (int i0 = tid; i0 < n2; i0 += NT * 8) {
            double2 t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * NT;
                t[u] = src[i < n2 ? i : n2 - 1];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * NT;
                if (i < n2) dst[i] = t[u];
            }
        }
    }
    for (int i = tid; i < K * P + K; i += NT) hist[i] = 0;
    if (tid == 0) *next_chunk = NW;  // chunks 0 .. NW-1 of the workgroup are the waves' first ones
    __syncthreads();
    if (SELF && next_chunk[1]) self_fold_prev<NT>(a, K, P, tid);  // this workgroup was the last to read the statistics

    // DP bookkeeping shared by the whole batch (collapsed_gibbs_dp.cpp:166-171,212-231)
    int Kused = 0, new_label = -1;
    if (p.mode == MODE_DP) {
        for (int k = 0; k < K; ++k) {
            if (NkT[k] > 0) ++Kused;
            else if (new_label < 0) new_label = k;
        }
    }

    DIAG(unsigned long long d_nmov = 0, d_ntile = 0; unsigned long long d_score = 0, d_pack = 0, d_draw = 0, d_mov = 0, d_pro = 0; unsigned long long d_t = diag_stamp(); const unsigned long long d_staged = d_t;)
    if (has_tile) {
        // prologue: the rest of the first tile's features, nothing to overlap with yet
        if (!BITS) put_stage<STG>(pack_stage<STG>(P, 0, st), 0, b0, b1, b2, b3);
        // no accumulators are live yet, so 64 loads share one round trip
#pragma unroll 1
        for (int h0 = 1; !BITS && h0 < nstages; h0 += 64 / STG) {
            uint32_t s0[STG], s1[STG], s2[STG], s3[STG];
#pragma unroll
            for (int u = 0; u < STG; ++u) { s0[u] = 0; s1[u] = 0; s2[u] = 0; s3[u] = 0; }
            issue_stage<STG>(pos, p.N, P, h0, s0);
            if (h0 + 1 < nstages) issue_stage<STG>(pos, p.N, P, h0 + 1, s1);
            if (STG == 16 && h0 + 2 < nstages) issue_stage<STG>(pos, p.N, P, h0 + 2, s2);
            if (STG == 16 && h0 + 3 < nstages) issue_stage<STG>(pos, p.N, P, h0 + 3, s3);
            put_stage<STG>(pack_stage<STG>(P, h0, s0), h0, b0, b1, b2, b3);
            if (h0 + 1 < nstages) put_stage<STG>(pack_stage<STG>(P, h0 + 1, s1), h0 + 1, b0, b1, b2, b3);
            if (STG == 16 && h0 + 2 < nstages) put_stage<STG>(pack_stage<STG>(P, h0 + 2, s2), h0 + 2, b0, b1, b2, b3);
            if (STG == 16 && h0 + 3 < nstages) put_stage<STG>(pack_stage<STG>(P, h0 + 3, s3), h0 + 3, b0, b1, b2, b3);
        }
        int zo = a.z_in ? a.z_in[pos.ic] : -1;
        asm volatile("" : "+v"(zo));  // land it before the pipeline starts: no load may be
                                      // pending at a loop header (the compiler would drain
                                      // vmcnt(0) at the first register reuse inside the loop)
        DIAG({ const unsigned long long n_ = diag_stamp(); d_pro += n_ - d_t; d_t = n_; })

        int zn_prev = 0;
        int64_t i_prev = -1;  // < 0: nothing to store yet
        for (;;) {
            const int zoc = zo < 0 ? 0 : zo;
            double acc_own = 0.0;
            if (MINUS != 0) {
                // the observation's own cluster is scored from the "minus self" tables: one entry per (narrow)
                // group, a per-lane gather -- from LDS (MINUS = 1), or from global memory when they are too big
                // to sit there beside Tp (MINUS = 2: L2-resident) -- before the accumulators are live, kOwnSub
                // groups per round: their fields are the low bits of a copy of the pattern that is shifted
                // down between rounds, their gathers are in flight together and summed in group order.  The
                // table image is padded with zero groups to whole rounds (bits past P are zero, so a padding
                // group adds its entry 0 = 0.0).
                uint32_t r0 = b0, r1 = b1, r2 = b2, r3 = b3;
                constexpr int RB = kOwnSub * kGroupWm;  // bits per round
                static_assert(RB < 32, "a round's fields come out of one word");
                const int rounds = (p.Gm + kOwnSub - 1) / kOwnSub;
                size_t at0 = (size_t)zoc * kGroupMm;
#pragma unroll 1
                for (int it = 0; it < rounds; ++it) {
                    double ow[kOwnSub];
#pragma unroll
                    for (int v = 0; v < kOwnSub; ++v) {
                        // v_bfe_u32 by hand: the compiler turns the bit-field extract into shift + and + add
                        // (three instructions where bfe + shift-add are two)
                        unsigned f;
                        asm("v_bfe_u32 %0, %1, %2, %3" : "=v"(f) : "v"(r0), "n"(v * kGroupWm), "n"(kGroupWm));
                        const size_t at = at0 + (size_t)v * KT * kGroupMm + f;
                        ow[v] = MINUS == 1 ? TmL[at] : TmG[at];
                    }
                    r0 = __builtin_amdgcn_alignbit(r1, r0, RB);
                    r1 = __builtin_amdgcn_alignbit(r2, r1, RB);
                    r2 = __builtin_amdgcn_alignbit(r3, r2, RB);
                    r3 >>= RB;
                    at0 += (size_t)kOwnSub * KT * kGroupMm;
#pragma unroll
                    for (int v = 0; v < kOwnSub; ++v) acc_own = acc_own + ow[v];
                }
            }
            // the previous tile's labels go out here, ahead of this iteration's stage loads in
            // the in-order VMEM stream (and behind the gathers above, whose wait would otherwise cover
            // a write round trip): a store still pending at the loop latch would make the compiler
            // wait vmcnt(0) there
            if (i_prev >= 0) a.z_out[i_prev] = zn_prev;
            int nc = 0;
            if (lane == 0) nc = atomicAdd(next_chunk, 1);
            nc = __builtin_amdgcn_readfirstlane(nc);
            const int64_t next = wg_c0 + nc;
            const bool has_next = nc < wg_cn;  // uniform
            const TilePos npos = tpos(has_next ? next : tile);
            uint32_t n0 = 0, n1 = 0, n2 = 0, n3 = 0;
            int zo_next = -1;

            // ---- scoring: K * G conflict-free LDS lookups.  One outer iteration = one feature
            // stage: its loads for the NEXT tile are issued first, fly during the four lookup
            // groups of THIS tile, and are packed last -- nothing in flight is loop-carried.
            double acc[KH];
#pragma unroll
            for (int k = 0; k < KH; ++k) acc[k] = 0.0;
            if (kPrioKernel) __builtin_amdgcn_s_setprio(BMM_LOOKUP_PRIO);
#pragma unroll 1
            for (int h = 0; h < nstages; ++h) {
                if (has_next) {
                    if (BITS) { if (h == 0) load_words(a.Xb, p.N, W, npos.ic, n0, n1, n2, n3); }
                    else issue_stage<STG>(npos, p.N, P, h, st);
                }
                // the lookup groups whose first bit lies in this stage's span [SB*h, SB*(h+1)): one word
                // `cur` holds it, a field may run on into `nxt`
                const int wd = (SB * h) >> 5;
                const uint32_t cur = word_of(wd, b0, b1, b2, b3), nxt = word_of(wd + 1, b0, b1, b2, b3);
                // the next tile's previous labels ride along with its first stage
                if (has_next && h == 0 && a.z_in) zo_next = a.z_in[npos.ic];
                const int g_lo = (SB * h + GW - 1) / GW;
                int g_hi = (SB * (h + 1) + GW - 1) / GW;
                g_hi = g_hi < G ? g_hi : G;
#pragma unroll 1
                for (int g = g_lo; g < g_hi; ++g) {
                    const unsigned nib = __builtin_amdgcn_alignbit(nxt, cur, (unsigned)(g * GW - 32 * wd)) &
                                         (unsigned)(GM - 1);
                    const volatile lds_f64* row = Tp + (((size_t)g * KT + kb) * GM + nib);
#pragma unroll
                    for (int c0 = 0; c0 < KH; c0 += CH) {
                        double tv[CH];
#pragma unroll
                        for (int j = 0; j < CH; ++j) tv[j] = row[(c0 + j) * GM];
#pragma unroll
                        for (int j = 0; j < CH; ++j) acc[c0 + j] = acc[c0 + j] + tv[j];
                        if (CH < KH) __builtin_amdgcn_sched_barrier(0);  // keep the chunks apart
                    }
                }
                DIAG({ const unsigned long long n_ = diag_stamp(); d_score += n_ - d_t; d_t = n_; })
                if (!BITS && has_next) put_stage<STG>(pack_stage<STG>(P, h, st), h, n0, n1, n2, n3);
                DIAG({ const unsigned long long n_ = diag_stamp(); d_pack += n_ - d_t; d_t = n_; })
            }
            if (kPrioKernel) __builtin_amdgcn_s_setprio(0);
            // scores (the constant terms sit in group 0 of the tables); the observation's own
            // cluster is scored without itself
            double m = neg_inf();
#pragma unroll
            for (int k = 0; k < KH; ++k) {
                double sc = acc[k];
                if (has_minus && kb + k == zo) sc = acc_own;
                acc[k] = sc;
                m = __builtin_fmax(m, sc);  // v_max_f64; scores are never NaN
            }
            if (SPLIT == 2) m = __builtin_fmax(m, __shfl_xor(m, 32));
            // weights exp(score - max) and their running sum in label order; acc[k] becomes the CDF
            double run = 0.0;
#pragma unroll
            for (int k = 0; k < KH; ++k) {
                const double w = expw_tab(acc[k] - m, ET);
                if (EMIT && kb + k < p.Kc && pos.valid) a.wts[(int64_t)(kb + k) * p.N + pos.i] = w;
                if (SPLIT == 1) { run = run + w; acc[k] = run; }
                else acc[k] = w;
                if ((k & 1) == 1) __builtin_amdgcn_sched_barrier(0);  // two at a time: bounds the temporaries
            }
            if (SPLIT == 2) {
                // the running sum goes through the categories in label order: the upper half continues
                // from the lower half's total
#pragma unroll
                for (int k = 0; k < KH; ++k) run = run + acc[k];
                const double lower = __shfl(run, lane & 31);
                run = half ? lower : 0.0;
#pragma unroll
                for (int k = 0; k < KH; ++k) { run = run + acc[k]; acc[k] = run; }
                run = __shfl(run, (lane & 31) + 32);
            }
            const double tot = run;
            if (EMIT && half == 0 && pos.valid) a.wtot[pos.i] = tot;
            // u <= 1 - 2^-52, so u * tot < tot: the walk always ends on a category with weight
            const double t = z_uniform(p.seed, (uint64_t)(p.obs0 + pos.ic), a.sweep) * tot;
            int cnt = 0;
#pragma unroll
            for (int k = 0; k < KH; ++k) cnt += t >= acc[k] ? 1 : 0;
            if (SPLIT == 2) cnt += __shfl_xor(cnt, 32);
            int zn = cnt;
            if (!(m > neg_inf())) zn = zoc;  // every category impossible: keep (or 0)
            if (p.mode == MODE_DP && zn == K) {
                const int own_single = (zo >= 0 && NkT[zoc] == 1) ? 1 : 0;
                if (Kused - own_single < K - 1) {
                    zn = new_label;
                    if (own_single && (zn < 0 || zo < zn)) zn = zo;
                } else {
                    int best = -1, bs = 0;
                    for (int k = 0; k < K; ++k) {
                        const int sz = NkT[k] - (k == zo ? 1 : 0);
                        if (sz > 0 && (best < 0 || sz < bs)) { best = k; bs = sz; }
                    }
                    zn = best >= 0 ? best : zoc;
                }
            }
            DIAG({ const unsigned long long n_ = diag_stamp(); d_draw += n_ - d_t; d_t = n_; })

            BMM_DBG_LABELS(a, pos.valid, pos.i == a.lo, K, zn, zo);
            DIAG(d_nmov += __popcll(__ballot(pos.valid && zn >= 0 && zn != zo));)
            count_movers(pos.valid && half == 0 && zn >= 0 && zn != zo, zo, zn, b0, b1, b2, b3, hist, K, P, lane);
            DIAG({ const unsigned long long n_ = diag_stamp(); d_mov += n_ - d_t; d_t = n_; })

            zn_prev = zn;
            i_prev = pos.valid && half == 0 ? pos.i : -1;
            if (!has_next) break;
            b0 = n0; b1 = n1; b2 = n2; b3 = n3;
            zo = zo_next;
            pos = npos;
            tile = next;
        }
        if (i_prev >= 0) a.z_out[i_prev] = zn_prev;
    }

    DIAG(const unsigned long long d_loop = diag_stamp();)
    __syncthreads();
    DIAG(const unsigned long long d_sync = diag_stamp();)
    flush_hist(hist, K, P, a.dS + (size_t)(blockIdx.x % kDeltaReps) * K * P, a.dNk + (blockIdx.x % kDeltaReps) * K, tid, NT);
    DIAG(asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); const unsigned long long d_flush = diag_stamp();)
    DIAG(if (a.diag && lane == 0) {
        atomicAdd(&a.diag[0], d_score); atomicAdd(&a.diag[1], d_pack); atomicAdd(&a.diag[2], d_draw);
        atomicAdd(&a.diag[3], d_mov); atomicAdd(&a.diag[4], d_pro); atomicAdd(&a.diag[5], 1ull);
        atomicAdd(&a.diag[6], d_nmov);
        // per-launch phases of a wave: table staging, tile loop, wait for the workgroup, flush
        atomicAdd(&a.diag[8], d_staged - d_entry); atomicAdd(&a.diag[9], d_loop - d_staged);
        atomicAdd(&a.diag[10], d_sync - d_loop); atomicAdd(&a.diag[11], d_flush - d_sync);
    })
}

// ---------------------------------------------------------------------------------
// Generic path: any P, up to kMaxCatsAny categories, no LDS residency.  Same arithmetic as
// k_resample (group lookups in g order, expw_, running sum and count in label order), with the
// tables gathered from global memory (L2) and the scores kept in a per-thread scratch column
// scr[k * stride + thread].  Clusters are accumulated sixteen at a time, X is re-read per chunk.
// Slow next to the resident kernel; it exists so that every shape the reference accepts runs.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ unsigned field_from_x(const int32_t* X, const uint32_t* Xb, int64_t N, int P,
                                                  int64_t i, int g, int GW) {
    if (Xb) {  // bit planes, any number of words: the field may straddle two of them
        const int o = g * GW, w = o >> 5, sh = o & 31, W = (P + 31) >> 5;
        uint32_t v = Xb[(int64_t)w * N + i] >> sh;
        if (sh + GW > 32 && w + 1 < W) v |= Xb[(int64_t)(w + 1) * N + i] << (32 - sh);
        return v & (unsigned)((1 << GW) - 1);
    }
    unsigned nib = 0;
    for (int j = 0; j < GW; ++j) {
        const int d = g * GW + j;
        if (d < P) nib |= ((unsigned)X[i + (int64_t)d * N] & 1u) << j;
    }
    return nib;
}

__global__ __launch_bounds__(256) void k_resample_generic(ChainParams p, ResampleArgs a, double* scr,
                                                          int64_t stride) {
    const bool has_minus = !explicit_params(p.mode);
    const TableLayout L = layout_of(p, has_minus);
    const int GM = L.M;
    const double* const Tp = a.tab + L.tp();
    const double* const Tm = a.tab + L.tm();
    const int32_t* const NkT = reinterpret_cast<const int32_t*>(a.tab + L.nk());
    const double* const ET = a.tab + L.et();
    const int P = p.P, G = p.G, K = p.K, Kc = p.Kc, KT = p.KT;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double* const my = scr + gid;

    int Kused = 0, new_label = -1;
    if (p.mode == MODE_DP) {
        for (int k = 0; k < K; ++k) {
            if (NkT[k] > 0) ++Kused;
            else if (new_label < 0) new_label = k;
        }
    }
    for (int64_t i = a.lo + gid; i < a.hi; i += (int64_t)gridDim.x * blockDim.x) {
        int zo = a.z_in ? a.z_in[i] : -1;
        const int zoc = zo < 0 ? 0 : zo;
        double acc_own = 0.0;
        if (has_minus)
            for (int g = 0; g < p.Gm; ++g)
                acc_own = acc_own + Tm[((size_t)g * KT + zoc) * kGroupMm + field_from_x(a.X, a.Xb, p.N, P, i, g, kGroupWm)];
        double m = neg_inf();
        for (int k0 = 0; k0 < Kc; k0 += 16) {
            double acc[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j] = 0.0;
            for (int g = 0; g < G; ++g) {
                const double* row = Tp + ((size_t)g * KT + k0) * GM + field_from_x(a.X, a.Xb, p.N, P, i, g, p.W);
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    if (k0 + j < Kc) acc[j] = acc[j] + row[j * GM];
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int k = k0 + j;
                if (k < Kc) {
                    double sc = acc[j];
                    if (has_minus && k == zo) sc = acc_own;
                    my[(int64_t)k * stride] = sc;
                    m = __builtin_fmax(m, sc);
                }
            }
        }
        double run = 0.0;
        for (int k = 0; k < Kc; ++k) {
            const double w = expw_tab(my[(int64_t)k * stride] - m, ET);
            if (a.wts) a.wts[(int64_t)k * p.N + i] = w;
            run = run + w;
            my[(int64_t)k * stride] = run;
        }
        const double tot = run;
        if (a.wts) a.wtot[i] = tot;
        const double t = z_uniform(p.seed, (uint64_t)(p.obs0 + i), a.sweep) * tot;
        int zn = 0;
        for (int k = 0; k < Kc; ++k) zn += t >= my[(int64_t)k * stride] ? 1 : 0;
        if (!(m > neg_inf())) zn = zoc;
        if (p.mode == MODE_DP && zn == K) {
            const int own_single = (zo >= 0 && NkT[zoc] == 1) ? 1 : 0;
            if (Kused - own_single < K - 1) {
                zn = new_label;
                if (own_single && (zn < 0 || zo < zn)) zn = zo;
            } else {
                int best = -1, bs = 0;
                for (int k = 0; k < K; ++k) {
                    const int sz = NkT[k] - (k == zo ? 1 : 0);
                    if (sz > 0 && (best < 0 || sz < bs)) { best = k; bs = sz; }
                }
                zn = best >= 0 ? best : zoc;
            }
        }
        BMM_DBG_LABELS(a, true, i == a.lo, K, zn, zo);
        a.z_out[i] = zn;
        if (zn >= 0 && zn != zo) {
            int32_t* const rNk = a.dNk + (blockIdx.x % kDeltaReps) * K;
            int32_t* const rS = a.dS + (size_t)(blockIdx.x % kDeltaReps) * K * P;
            atomicAdd(&rNk[zn], 1);
            if (zo >= 0) atomicAdd(&rNk[zo], -1);
            for (int d = 0; d < P; ++d)
                if (a.Xb ? (a.Xb[(int64_t)(d >> 5) * p.N + i] >> (d & 31)) & 1u : (uint32_t)a.X[i + (int64_t)d * p.N] & 1u) {
                    atomicAdd(&rS[(size_t)zn * P + d], 1);
                    if (zo >= 0) atomicAdd(&rS[(size_t)zo * P + d], -1);
                }
        }
    }
}

__global__ __launch_bounds__(256) void k_count_labels_generic(ChainParams p, const int32_t* __restrict__ X,
                                                              const uint32_t* __restrict__ Xb,
                                                              const int32_t* __restrict__ z, int32_t* dNk,
                                                              int32_t* dS) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < p.N; i += (int64_t)gridDim.x * blockDim.x) {
        const int zl = z[i];
        if (zl < 0) continue;
        atomicAdd(&dNk[zl], 1);
        for (int d = 0; d < p.P; ++d)
            if (Xb ? (Xb[(int64_t)(d >> 5) * p.N + i] >> (d & 31)) & 1u : (uint32_t)X[i + (int64_t)d * p.N] & 1u)
                atomicAdd(&dS[(size_t)zl * p.P + d], 1);
    }
}

// The allocation probabilities of one batch, normalised and filed by label: the matrix the reference
// stores for Stephens' relabelling (collapsed_gibbs.cpp:162-172, collapsed_gibbs_dp.cpp:190-200,
// stickbreaking.cpp:129-139).  wts/wtot are what the resample kernel of this batch emitted;
// probs is N x K column-major.  The DP's new-cluster mass goes under the label a new cluster would take
// for THIS observation (choices(K) = unused_clusters.top(), collapsed_gibbs_dp.cpp:169-170,193): the
// smallest free label of the batch (from the table image's cluster sizes) -- or the observation's own
// label when it sat alone in its cluster and that label is smaller, because removing it (:113-128) has
// just returned that label to the heap.  The draw itself (k_resample) opens the same label.
__global__ __launch_bounds__(256) void k_probs_finish(ChainParams p, const double* __restrict__ tab,
                                                      const double* __restrict__ wts,
                                                      const double* __restrict__ wtot,
                                                      const int32_t* __restrict__ z_in, int64_t lo, int64_t hi,
                                                      double* __restrict__ probs) {
    const TableLayout L = layout_of(p, !explicit_params(p.mode));
    const int32_t* const NkT = reinterpret_cast<const int32_t*>(tab + L.nk());
    int new_label = -1;
    if (p.mode == MODE_DP)
        for (int k = 0; k < p.K && new_label < 0; ++k)
            if (NkT[k] <= 0) new_label = k;
    for (int64_t i = lo + (int64_t)blockIdx.x * 256 + threadIdx.x; i < hi; i += (int64_t)gridDim.x * 256) {
        const double tot = wtot[i];
        int own_new = new_label;
        if (p.mode == MODE_DP && z_in) {
            const int zo = z_in[i];
            if (zo >= 0 && NkT[zo] == 1 && (new_label < 0 || zo < new_label)) own_new = zo;
        }
        for (int k = 0; k < p.Kc; ++k) {
            const int lbl = k < p.K ? k : own_new;  // in label order: the new-cluster mass lands last
            if (lbl >= 0) probs[i + (int64_t)lbl * p.N] = div_(wts[(int64_t)k * p.N + i], tot);
        }
    }
}

// The label trace on its way out: observations [i0, i0 + rows) of the [S][N] 0-based device trace as the
// rows x S block of R's S x N column-major matrix they occupy (out[(i - i0) * S + s]: that block is
// contiguous in the caller's buffer), labels 1-based.  T = int32_t: unassigned -> NA_integer_;
// T = uint8_t (at most 254 labels): unassigned -> 0, widened on the host, a quarter of the bytes over PCIe.
template <typename T>
__global__ __launch_bounds__(256) void k_trace_block(const int32_t* __restrict__ trace, int64_t N, int S,
                                                     int64_t i0, int64_t rows, T* __restrict__ out) {
    __shared__ int32_t tile[32][33];
    const int64_t b0 = (int64_t)blockIdx.x * 32;  // within the block of observations
    const int s0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int s = s0 + r;
        const int64_t li = b0 + tx;
        tile[r][tx] = (s < S && li < rows) ? trace[(size_t)s * N + i0 + li] : 0;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int64_t li = b0 + r;
        const int s = s0 + tx;
        if (s < S && li < rows) {
            const int32_t v = tile[tx][r];
            if (sizeof(T) == 1) out[(size_t)li * S + s] = (T)(v < 0 ? 0 : v + 1);
            else out[(size_t)li * S + s] = (T)(v < 0 ? (int32_t)0x80000000 : v + 1);
        }
    }
}

// initialK as R hands it over (1-based) -> the chain's 0-based label row; bad[0] receives the smallest
// index whose label lies outside 1..K (or stays at its initial all-ones value)
__global__ __launch_bounds__(256) void k_labels_from_r(const int32_t* __restrict__ z1, int64_t N, int K,
                                                       int32_t* __restrict__ z0, unsigned long long* bad) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (int64_t)gridDim.x * 256) {
        const int32_t v = z1[i];
        if (v < 1 || v > K) atomicMin(bad, (unsigned long long)i);
        z0[i] = v - 1;
    }
}

// ---- self-check kernels ----------------------------------------------------------
__global__ void k_test_math(int op, const double* in, const double* in2, double* out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = in[i];
    double y;
    if (op == 0) y = log_(x);
    else if (op == 1) y = exp_(x);
    else if (op == 2) y = div_(x, in2[i]);
    else if (op == 3) y = sqrt_(x);
    else y = expw_(x);
    out[i] = y;
}
__global__ void k_test_variates(int kind, double pp, double qq, uint64_t seed, uint32_t sweep, double* out,
                                int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double y;
    if (kind == 0) {
        Stream s = make_stream(seed, (uint32_t)i, sweep, kStreamThetaA);
        y = rgamma_(pp, s);
    } else if (kind == 1) {
        Stream sa = make_stream(seed, (uint32_t)i, sweep, kStreamThetaA);
        Stream sb = make_stream(seed, (uint32_t)i, sweep, kStreamThetaB);
        y = rbeta_(pp, qq, sa, sb);
    } else {
        y = update_alpha_(pp, 1.0, 1.0, 1000.0, (int)qq, seed + (uint64_t)i, sweep);
    }
    out[i] = y;
}

}  // namespace bmm
