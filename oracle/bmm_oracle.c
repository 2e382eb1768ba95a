/* bmm_oracle.c -- see bmm_oracle.h.  TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (header).
 * Build: oracle/Makefile (gcc -O2 -mfma -ffp-contract=off -fPIC -shared -pthread).
 * All file:line citations are into /root/reference.
 */
#include "bmm_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define GW 5  /* features per lookup group of the tables against the full statistics (= bmm::kGroupW) ... */
#define GW_ALT 4 /* ... unless the shape's table image is too big for that (oracle_group_width_for) */
#define GWM 3 /* ... of the own-cluster ("minus self") tables; = bmm::kGroupWm */
#define GMM (1 << GWM)

static __thread char g_err[256];
const char* oracle_last_error(void) { return g_err; }
int oracle_group_width(void) { return GW; }
int oracle_group_width_own(void) { return GWM; }
/* The spec's rule for the group width of a shape (DESIGN.md "Numerics"; sampler 0 collapsed, 1 DP, 2
 * stick-breaking, 3 full): 5 features per group when the whole table image of the shape -- tables of
 * 32-entry groups for the categories rounded up to a supported accumulator count, the constants, the
 * cluster sizes, the 256-entry exponential table, for the counting samplers the own-cluster tables (groups
 * of 3, padded to a multiple of 6 groups) -- and the integer histogram fit in 160 KiB; 4 otherwise. */
int oracle_group_width_for(int sampler, int K, int P) {
    static const int kts[] = {4, 8, 12, 16, 20, 24, 28, 32, 40, 48, 56, 64};
    const int cats = sampler == 1 ? K + 1 : K;
    int KT = -1;
    for (unsigned q = 0; q < sizeof kts / sizeof kts[0]; ++q) if (kts[q] >= cats) { KT = kts[q]; break; }
    if (KT < 0 || P > 128) return GW_ALT;
    const int counting = sampler == 0 || sampler == 1;
    const long G = (P + GW - 1) / GW, Gm = ((P + GWM - 1) / GWM + 5) / 6 * 6;
    const long nk = (KT + 1) / 2 + (((KT + 1) / 2) & 1);
    long doubles = G * KT * (1 << GW) + 2 * KT + nk + 256 + (counting ? Gm * KT * GMM : 0);
    long bytes = doubles * 8 + ((long)K * P + K + 4) * 4;
    return bytes <= 163840 ? GW : GW_ALT;
}

static int fail(const char* msg) {
    snprintf(g_err, sizeof g_err, "%s", msg);
    return 1;
}

/* ------------------------------------------------------------------ numerics */
static inline uint64_t d2u(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }
static inline double u2d(uint64_t u) { double x; memcpy(&x, &u, 8); return x; }
#define O_NEG_INF (u2d(0xfff0000000000000ull))
#define O_POS_INF (u2d(0x7ff0000000000000ull))

/* Philox4x32-10 (Salmon, Moraes, Dror, Shaw 2011): ten rounds of two 32x32->64
 * multiplies, key bumped by the Weyl constants between rounds. */
void oracle_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

double oracle_u01(uint32_t a, uint32_t b) {
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * 0x1p-53;
}

/* Philox2x32-10: one multiply per round; the two output words make the one uniform of a draw. */
void oracle_philox2x32_10(const uint32_t ctr[2], uint32_t key, uint32_t out[2]) {
    uint32_t c0 = ctr[0], c1 = ctr[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t pr = (uint64_t)0xD256D193u * c0;
        uint32_t n0 = (uint32_t)(pr >> 32) ^ key ^ c1;
        c1 = (uint32_t)pr;
        c0 = n0;
        key += 0x9E3779B9u;
    }
    out[0] = c0; out[1] = c1;
}

/* 52 random bits as the mantissa of a double in [1,2), minus 1: a uniform in [0, 1 - 2^-52] */
double oracle_u52(uint32_t a, uint32_t b) {
    uint64_t bits = ((uint64_t)(0x3ff00000u | (a >> 12)) << 32) | (uint32_t)((a << 20) | (b >> 12));
    return u2d(bits) - 1.0;
}

/* the uniform of z_i in sweep j: counter (i mod 2^32, sweep), key folded from the seed (and from the
 * high word of i) */
double oracle_z_uniform(uint64_t seed, uint64_t i, uint32_t sweep) {
    uint32_t h = (uint32_t)(i >> 32);
    uint32_t key = ((uint32_t)seed ^ ((uint32_t)(seed >> 32) * 0x85EBCA6Bu)) ^ ((h << 16) | (h >> 16));
    uint32_t c[2] = {(uint32_t)i, sweep}, o[2];
    oracle_philox2x32_10(c, key, o);
    return oracle_u52(o[0], o[1]);
}

/* log: m in [sqrt(.5), sqrt(2)), f = m-1, s = f/(2+f), even polynomial in s. */
double oracle_log(double x) {
    uint64_t ix = d2u(x);
    if (x == 0.0) return O_NEG_INF;
    if ((int64_t)ix < 0) return u2d(0x7ff8000000000000ull);
    if ((ix >> 52) == 0x7ff) return x;
    int e = 0;
    if ((ix >> 52) == 0) { x *= 0x1p54; ix = d2u(x); e = -54; }
    e += (int)(ix >> 52) - 1023;
    uint64_t m = (ix & 0x000fffffffffffffull) | 0x3ff0000000000000ull;
    if (m >= 0x3ff6a09e667f3bcdull) { m -= 0x0010000000000000ull; e += 1; }
    double f = u2d(m) - 1.0;
    double s = f / (2.0 + f);
    double z = s * s, w = z * z;
    double t1 = w * __builtin_fma(w, __builtin_fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01),
                                  3.999999999940941908e-01);
    double t2 = z * __builtin_fma(w, __builtin_fma(w, __builtin_fma(w, 1.479819860511658591e-01,
                                  1.818357216161805012e-01), 2.857142874366239149e-01),
                                  6.666666666666735130e-01);
    double R = t2 + t1;
    double hfsq = 0.5 * f * f;
    double dk = (double)e;
    return dk * 6.93147180369123816490e-01 -
           ((hfsq - (s * (hfsq + R) + dk * 1.90821492927058770002e-10)) - f);
}

/* exp: 64k + j = round(x * 64/ln2), r = x - (64k+j) ln2/64 in two parts, e^r - 1 by a degree-5
 * Taylor polynomial, times the tabulated 2^(j/64), times 2^k; x < -708 flushes to 0. */
static const double o_exp2_64[64] = {
    0x1.0000000000000p+0, 0x1.02c9a3e778061p+0, 0x1.059b0d3158574p+0, 0x1.0874518759bc8p+0,
    0x1.0b5586cf9890fp+0, 0x1.0e3ec32d3d1a2p+0, 0x1.11301d0125b51p+0, 0x1.1429aaea92de0p+0,
    0x1.172b83c7d517bp+0, 0x1.1a35beb6fcb75p+0, 0x1.1d4873168b9aap+0, 0x1.2063b88628cd6p+0,
    0x1.2387a6e756238p+0, 0x1.26b4565e27cddp+0, 0x1.29e9df51fdee1p+0, 0x1.2d285a6e4030bp+0,
    0x1.306fe0a31b715p+0, 0x1.33c08b26416ffp+0, 0x1.371a7373aa9cbp+0, 0x1.3a7db34e59ff7p+0,
    0x1.3dea64c123422p+0, 0x1.4160a21f72e2ap+0, 0x1.44e086061892dp+0, 0x1.486a2b5c13cd0p+0,
    0x1.4bfdad5362a27p+0, 0x1.4f9b2769d2ca7p+0, 0x1.5342b569d4f82p+0, 0x1.56f4736b527dap+0,
    0x1.5ab07dd485429p+0, 0x1.5e76f15ad2148p+0, 0x1.6247eb03a5585p+0, 0x1.6623882552225p+0,
    0x1.6a09e667f3bcdp+0, 0x1.6dfb23c651a2fp+0, 0x1.71f75e8ec5f74p+0, 0x1.75feb564267c9p+0,
    0x1.7a11473eb0187p+0, 0x1.7e2f336cf4e62p+0, 0x1.82589994cce13p+0, 0x1.868d99b4492edp+0,
    0x1.8ace5422aa0dbp+0, 0x1.8f1ae99157736p+0, 0x1.93737b0cdc5e5p+0, 0x1.97d829fde4e50p+0,
    0x1.9c49182a3f090p+0, 0x1.a0c667b5de565p+0, 0x1.a5503b23e255dp+0, 0x1.a9e6b5579fdbfp+0,
    0x1.ae89f995ad3adp+0, 0x1.b33a2b84f15fbp+0, 0x1.b7f76f2fb5e47p+0, 0x1.bcc1e904bc1d2p+0,
    0x1.c199bdd85529cp+0, 0x1.c67f12e57d14bp+0, 0x1.cb720dcef9069p+0, 0x1.d072d4a07897cp+0,
    0x1.d5818dcfba487p+0, 0x1.da9e603db3285p+0, 0x1.dfc97337b9b5fp+0, 0x1.e502ee78b3ff6p+0,
    0x1.ea4afa2a490dap+0, 0x1.efa1bee615a27p+0, 0x1.f50765b6e4540p+0, 0x1.fa7c1819e90d8p+0,
};
double oracle_exp(double x) {
    if (x != x) return x;
    if (x > 709.782712893384) return O_POS_INF;
    if (x < -708.0) return 0.0;
    double kd = __builtin_floor(__builtin_fma(x, 0x1.71547652b82fep+6, 0.5));
    double r = __builtin_fma(-kd, 0x1.62e42fefa39efp-7, x);
    r = __builtin_fma(-kd, 0x1.abc9e3b39803fp-62, r);
    int ki = (int)kd;
    int j = ki & 63, k = ki >> 6;
    double c = __builtin_fma(r, 8.3333333333333332177e-03, 4.1666666666666664354e-02);
    c = __builtin_fma(r, c, 1.6666666666666665741e-01);
    c = __builtin_fma(r, c, 0.5);
    double q = __builtin_fma(r * r, c, r);
    double t = o_exp2_64[j];
    double p = __builtin_fma(t, q, t);
    if (k > 1023) { p *= 2.0; k -= 1; }
    return p * u2d((uint64_t)(k + 1023) << 52);
}

/* the draw's weight exponential, x <= 0: 256 k + j = nearest integer to x * 256/ln2 (ties to even,
 * via the 1.5 * 2^52 sum), r = x - (256 k + j) ln2/256 in two parts, degree-4 polynomial for e^r - 1,
 * tabulated 2^(j/256), scaled by ldexp; -inf and NaN give 0 */
#include "exp256_table.h"
double oracle_expw(double x) {
    double xs = x > -1000.0 ? x : -1000.0; /* NaN -> -1000 */
    double km = __builtin_fma(xs, 0x1.71547652b82fep+8, 0x1.8p52);
    int32_t ki = (int32_t)(uint32_t)d2u(km);
    double kd = km - 0x1.8p52;
    double r = __builtin_fma(-kd, 0x1.62e42fefa39efp-9, xs);
    r = __builtin_fma(-kd, 0x1.abc9e3b39803fp-64, r);
    int j = ki & 255, k = ki >> 8;
    double c = __builtin_fma(r, 4.1666666666666664354e-02, 1.6666666666666665741e-01);
    c = __builtin_fma(r, c, 0.5);
    double q = __builtin_fma(r * r, c, r);
    double t = o_exp2_256[j];
    return ldexp(__builtin_fma(t, q, t), k);
}
void oracle_expw_array(const double* x, double* y, int64_t n) {
    for (int64_t i = 0; i < n; ++i) y[i] = oracle_expw(x[i]);
}

void oracle_log_array(const double* x, double* y, int64_t n) {
    for (int64_t i = 0; i < n; ++i) y[i] = oracle_log(x[i]);
}
void oracle_exp_array(const double* x, double* y, int64_t n) {
    for (int64_t i = 0; i < n; ++i) y[i] = oracle_exp(x[i]);
}

typedef struct { uint32_t c0, ctr, c2, sid, k0, k1; } ostream;
static ostream mk_stream(uint64_t seed, uint32_t c0, uint32_t sweep, uint32_t sid) {
    ostream s = {c0, 0u, sweep, sid, (uint32_t)seed, (uint32_t)(seed >> 32)};
    return s;
}
static void st_next(ostream* s, uint32_t o[4]) {
    uint32_t c[4] = {s->c0, s->ctr, s->c2, s->sid}, k[2] = {s->k0, s->k1};
    oracle_philox4x32_10(c, k, o);
    s->ctr++;
}
/* Marsaglia polar normal */
static double o_rnorm(ostream* s) {
    for (;;) {
        uint32_t o[4];
        st_next(s, o);
        double v1 = 2.0 * oracle_u01(o[0], o[1]) - 1.0, v2 = 2.0 * oracle_u01(o[2], o[3]) - 1.0;
        double q = v1 * v1 + v2 * v2;
        if (q < 1.0 && q > 0.0) return v1 * sqrt((-2.0 * oracle_log(q)) / q);
    }
}
/* Marsaglia-Tsang gamma, scale 1 */
static double o_rgamma(double shape, ostream* s) {
    if (!(shape > 0.0)) return 0.0;
    double boost = 1.0;
    if (shape < 1.0) {
        uint32_t o[4];
        st_next(s, o);
        boost = oracle_exp(oracle_log(1.0 - oracle_u01(o[0], o[1])) / shape);
        shape = shape + 1.0;
    }
    double d = shape - 0.33333333333333331483, c = 1.0 / sqrt(9.0 * d);
    for (;;) {
        double x = o_rnorm(s), v = 1.0 + c * x;
        if (v <= 0.0) continue;
        v = v * v * v;
        uint32_t o[4];
        st_next(s, o);
        double u = 1.0 - oracle_u01(o[0], o[1]);
        double x2 = x * x;
        if (oracle_log(u) < 0.5 * x2 + d - d * v + d * oracle_log(v)) return d * v * boost;
    }
}
double oracle_rgamma(double shape, uint64_t seed, uint32_t c0, uint32_t sweep, uint32_t stream) {
    ostream s = mk_stream(seed, c0, sweep, stream);
    return o_rgamma(shape, &s);
}
double oracle_rbeta(double p, double q, uint64_t seed, uint32_t c0a, uint32_t c0b, uint32_t sweep,
                    uint32_t stream_a, uint32_t stream_b) {
    ostream sa = mk_stream(seed, c0a, sweep, stream_a), sb = mk_stream(seed, c0b, sweep, stream_b);
    double x = o_rgamma(p, &sa), y = o_rgamma(q, &sb);
    return x / (x + y);
}
/* utils.cpp:6-14 (Escobar & West), with the build's variate generators */
double oracle_update_alpha(double alpha_old, double a, double b, double N, int K, uint64_t seed,
                           uint32_t sweep) {
    double eta = oracle_rbeta(alpha_old + 1.0, N, seed, 0, 1, sweep, 5, 5);
    double b_eps = b - oracle_log(eta);
    double pi1 = a + (double)K - 1.0, pi2 = N * b_eps;
    double pi = pi1 / (pi1 + pi2);
    double scale = 1.0 / b_eps;
    double ga = oracle_rgamma(a + (double)K, seed, 0, sweep, 6) * scale;
    double gb = oracle_rgamma(a + (double)K - 1.0, seed, 0, sweep, 7) * scale;
    return pi * ga + (1.0 - pi) * gb;
}

/* Inverse-CDF draw shared by every sampler: weights w[0..n), one uniform u <= 1 - 2^-52.
 * c_k = w[0]+..+w[k] in label order, t = u * c_{n-1}; picks the number of k with t >= c_k, i.e.
 * the first k with t < c_k.  t < c_{n-1} always holds for a finite positive total, so the pick
 * is a category that raised the running sum.  Returns -1 when the total is not positive. */
static int draw_index(const double* w, int n, double u) {
    double c = 0.0;
    for (int k = 0; k < n; ++k) c = c + w[k];
    if (!(c > 0.0)) return -1;
    double t = u * c;
    int cnt = 0;
    c = 0.0;
    for (int k = 0; k < n; ++k) {
        c = c + w[k];
        if (t >= c) ++cnt;
    }
    return cnt < n ? cnt : -1;
}
/* scores -> weights expw(score - max); returns 0 if max is -inf (degenerate) */
static int scores_to_weights(const double* score, int n, double* w) {
    double m = O_NEG_INF;
    for (int k = 0; k < n; ++k) if (score[k] > m) m = score[k];
    if (m == O_NEG_INF) return 0;
    for (int k = 0; k < n; ++k) w[k] = oracle_expw(score[k] - m);
    return 1;
}

/* ------------------------------------------------------------------ spec tables */
/* Group table of one cluster from counts (n observations, s[d] ones), optionally
 * with the scored observation's own contribution removed (minus = 1: n-1, and s-1
 * on the x=1 side).  Groups of W features: T has G * 2^W entries, G = ceil(P / W) (the callers use GW for
 * the full tables and GWM for the minus-self ones).  Combinations no member can exhibit get 0. */
static void counts_group_table(double beta, double gamma, int P, int W, int64_t n, const int32_t* s,
                               int minus, double* e1, double* e0, double* T) {
    const int G = (P + W - 1) / W;
    const unsigned M = 1u << W;
    int64_t ne = n - minus;
    if (ne <= 0) { memset(T, 0, sizeof(double) * (size_t)G * M); return; }
    double den = oracle_log((beta + gamma) + (double)ne);
    for (int d = 0; d < P; ++d) {
        int64_t s1 = (int64_t)s[d] - minus;
        e1[d] = s1 < 0 ? 0.0 : oracle_log(beta + (double)s1) - den;
        e0[d] = s[d] > ne ? 0.0 : oracle_log((gamma + (double)ne) - (double)s[d]) - den;
    }
    for (int g = 0; g < G; ++g)
        for (unsigned m = 0; m < M; ++m) {
            double t = 0.0;
            for (int j = 0; j < W; ++j) {
                int d = g * W + j;
                if (d < P) t = t + (((m >> j) & 1u) ? e1[d] : e0[d]);
            }
            T[g * M + m] = t;
        }
}
static void theta_group_table(int P, int W, int K, int k, const double* theta /*K x P colmajor*/,
                              double* T) {
    const int G = (P + W - 1) / W;
    const unsigned M = 1u << W;
    for (int g = 0; g < G; ++g)
        for (unsigned m = 0; m < M; ++m) {
            double t = 0.0;
            for (int j = 0; j < W; ++j) {
                int d = g * W + j;
                if (d < P) {
                    double th = theta[k + (size_t)d * K];
                    t = t + (((m >> j) & 1u) ? oracle_log(th) : oracle_log(1.0 - th));
                }
            }
            T[g * M + m] = t;
        }
}
/* score = ((C + T[0]) + T[1]) + ...: the category's constant term enters first (the HIP
 * path stores C + T[0][m] as the entries of group 0) */
static inline double table_sum(double C, const double* T, const uint8_t* nib, int G, int M) {
    double acc = C;
    for (int g = 0; g < G; ++g) acc = acc + T[g * M + nib[g]];
    return acc;
}
/* group fields of every observation, W features per group: nib[i*G + g] = sum_j x[i, g*W+j] << j */
static uint8_t* pack_nibbles(const int32_t* X, int64_t N, int P, int W) {
    const int G = (P + W - 1) / W;
    uint8_t* nib = (uint8_t*)calloc((size_t)N * G, 1);
    if (!nib) return NULL;
    for (int d = 0; d < P; ++d) {
        const int32_t* col = X + (size_t)d * N;
        int g = d / W, j = d % W;
        for (int64_t i = 0; i < N; ++i) nib[(size_t)i * G + g] |= (uint8_t)((col[i] & 1) << j);
    }
    return nib;
}

/* ------------------------------------------------------------------ conditionals */
/* collapsed_gibbs.cpp:89-150 for one observation, nothing else changed */
void oracle_collapsed_cond_literal(const int32_t* X, int64_t N, int P, const int32_t* z, int64_t i,
                                   int K, double alpha, double beta, double gamma, double* raw,
                                   double* norm) {
    double probs_sum = 0;
    for (int k = 0; k < K; ++k) {
        int Nk = 0;
        for (int64_t c = 0; c < N; ++c) if (c != i && z[c] - 1 == k) ++Nk;
        double dummy;
        if (Nk > 0) {
            double LHS = log(Nk + (alpha / K)) - log(N - 1 + alpha);
            double logLH = 0;
            for (int d = 0; d < P; ++d) {
                int sum_xd = 0;
                for (int64_t c = 0; c < N; ++c)
                    if (c != i && z[c] - 1 == k) sum_xd += X[c + (size_t)d * N];
                int xnd = X[i + (size_t)d * N];
                double left = xnd * log(beta + sum_xd);
                double right = (1 - xnd) * log(gamma + Nk - sum_xd);
                double denom = log(beta + gamma + Nk);
                logLH += left + right - denom;
            }
            dummy = exp(LHS + logLH);
        } else {
            dummy = 0;
        }
        probs_sum += dummy;
        raw[k] = dummy;
    }
    for (int k = 0; k < K; ++k) norm[k] = raw[k] / probs_sum;
}

/* counts with observation i left in, from 1-based labels */
static void count_stats(const int32_t* X, int64_t N, int P, const int32_t* z, int K, int32_t* Nk,
                        int32_t* S /*K*P row-major by cluster*/) {
    memset(Nk, 0, sizeof(int32_t) * K);
    memset(S, 0, sizeof(int32_t) * (size_t)K * P);
    for (int64_t c = 0; c < N; ++c) {
        int k = z[c] - 1;
        if (k < 0 || k >= K) continue;
        Nk[k]++;
        for (int d = 0; d < P; ++d) S[(size_t)k * P + d] += X[c + (size_t)d * N] & 1;
    }
}

void oracle_collapsed_cond_spec(const int32_t* X, int64_t N, int P, const int32_t* z, int64_t i,
                                int K, double alpha, double beta, double gamma, double* score,
                                double* norm) {
    const int gw = oracle_group_width_for(0, K, P), GM = 1 << gw;
    int G = (P + gw - 1) / gw, Gm = (P + GWM - 1) / GWM;
    int32_t* Nk = (int32_t*)malloc(sizeof(int32_t) * K);
    int32_t* S = (int32_t*)malloc(sizeof(int32_t) * (size_t)K * P);
    double* e1 = (double*)malloc(sizeof(double) * P * 2);
    double* T = (double*)malloc(sizeof(double) * ((size_t)G * GM + (size_t)Gm * GMM));
    double* w = (double*)malloc(sizeof(double) * K);
    uint8_t* nib = pack_nibbles(X, N, P, gw);
    uint8_t* nibm = pack_nibbles(X, N, P, GWM);
    count_stats(X, N, P, z, K, Nk, S);
    int zo = z[i] - 1;
    double ldN = oracle_log((double)(N - 1) + alpha);
    for (int k = 0; k < K; ++k) {
        int minus = (k == zo);
        int64_t ne = Nk[k] - minus;
        counts_group_table(beta, gamma, P, minus ? GWM : gw, Nk[k], S + (size_t)k * P, minus, e1, e1 + P, T);
        double C = ne > 0 ? oracle_log((double)ne + alpha / (double)K) - ldN : O_NEG_INF;
        score[k] = minus ? table_sum(C, T, nibm + (size_t)i * Gm, Gm, GMM) : table_sum(C, T, nib + (size_t)i * G, G, GM);
    }
    if (scores_to_weights(score, K, w)) {
        double tot = 0.0;
        for (int k = 0; k < K; ++k) tot = tot + w[k];
        for (int k = 0; k < K; ++k) norm[k] = w[k] / tot;
    }
    free(Nk); free(S); free(e1); free(T); free(w); free(nib); free(nibm);
}

/* collapsed_gibbs_dp.cpp:71,102-106,140-186 for one observation; clusters = labels 1..K */
void oracle_dp_cond_literal(const int32_t* X, int64_t N, int P, const int32_t* z, int64_t i, int K,
                            double alpha, double beta, double gamma, double* logw, double* norm) {
    double RHS_newk = P * (log(beta) - log(beta + gamma));
    double left_denom = log(N - 1 + alpha);
    double probs_newk = log(alpha) - left_denom + RHS_newk;
    for (int k = 0; k < K; ++k) {
        int Nk = 0;
        for (int64_t c = 0; c < N; ++c) if (c != i && z[c] - 1 == k) ++Nk;
        double LHS = log(Nk) - left_denom;
        double logLH = 0;
        double denom = log(beta + gamma + Nk);
        for (int d = 0; d < P; ++d) {
            int sum_xd = 0;
            for (int64_t c = 0; c < N; ++c)
                if (c != i && z[c] - 1 == k) sum_xd += X[c + (size_t)d * N];
            int xnd = X[i + (size_t)d * N];
            double left = xnd * log(beta + sum_xd);
            double right = (1 - xnd) * log(gamma + Nk - sum_xd);
            logLH += left + right - denom;
        }
        logw[k] = LHS + logLH;
    }
    logw[K] = probs_newk;
    double max_prob = logw[0];
    for (int k = 1; k <= K; ++k) if (logw[k] > max_prob) max_prob = logw[k];
    double sumprob = 0;
    for (int k = 0; k <= K; ++k) { norm[k] = exp(logw[k] - max_prob); sumprob += norm[k]; }
    for (int k = 0; k <= K; ++k) norm[k] /= sumprob;
}

static double dp_new_score(double alpha, double beta, double gamma, int P, double ldN) {
    return (oracle_log(alpha) - ldN) + (double)P * (oracle_log(beta) - oracle_log(beta + gamma));
}

void oracle_dp_cond_spec(const int32_t* X, int64_t N, int P, const int32_t* z, int64_t i, int K,
                         double alpha, double beta, double gamma, double* logw, double* norm) {
    const int gw = oracle_group_width_for(1, K, P), GM = 1 << gw;
    int G = (P + gw - 1) / gw, Gm = (P + GWM - 1) / GWM;
    int32_t* Nk = (int32_t*)malloc(sizeof(int32_t) * K);
    int32_t* S = (int32_t*)malloc(sizeof(int32_t) * (size_t)K * P);
    double* e1 = (double*)malloc(sizeof(double) * P * 2);
    double* T = (double*)malloc(sizeof(double) * ((size_t)G * GM + (size_t)Gm * GMM));
    double* w = (double*)malloc(sizeof(double) * (K + 1));
    uint8_t* nib = pack_nibbles(X, N, P, gw);
    uint8_t* nibm = pack_nibbles(X, N, P, GWM);
    count_stats(X, N, P, z, K, Nk, S);
    int zo = z[i] - 1;
    double ldN = oracle_log((double)(N - 1) + alpha);
    for (int k = 0; k < K; ++k) {
        int minus = (k == zo);
        int64_t ne = Nk[k] - minus;
        counts_group_table(beta, gamma, P, minus ? GWM : gw, Nk[k], S + (size_t)k * P, minus, e1, e1 + P, T);
        double C = ne > 0 ? oracle_log((double)ne) - ldN : O_NEG_INF;
        logw[k] = minus ? table_sum(C, T, nibm + (size_t)i * Gm, Gm, GMM) : table_sum(C, T, nib + (size_t)i * G, G, GM);
    }
    logw[K] = dp_new_score(alpha, beta, gamma, P, ldN) + 0.0;
    if (scores_to_weights(logw, K + 1, w)) {
        double tot = 0.0;
        for (int k = 0; k <= K; ++k) tot = tot + w[k];
        for (int k = 0; k <= K; ++k) norm[k] = w[k] / tot;
    }
    free(Nk); free(S); free(e1); free(T); free(w); free(nib); free(nibm);
}

/* stickbreaking.cpp:75-105 for one observation */
void oracle_sb_cond_literal(const int32_t* X, int64_t N, int P, int64_t i, int K, const double* pi,
                            const double* theta, double* raw, double* norm) {
    double cum_probs = 0;
    for (int k = 0; k < K; ++k) {
        double loglh = 0;
        for (int d = 0; d < P; ++d) {
            int x = X[i + (size_t)d * N];
            double th = theta[k + (size_t)d * K];
            loglh += x * log(th) + (1 - x) * log(1 - th);
        }
        double dummy = exp(log(pi[k]) + loglh);
        raw[k] = dummy;
        cum_probs += dummy;
    }
    for (int k = 0; k < K; ++k) norm[k] = raw[k] / cum_probs;
}
void oracle_sb_cond_spec(const int32_t* X, int64_t N, int P, int64_t i, int K, const double* pi,
                         const double* theta, double* score, double* norm) {
    const int gw = oracle_group_width_for(2, K, P), GM = 1 << gw;
    int G = (P + gw - 1) / gw;
    double* T = (double*)malloc(sizeof(double) * (size_t)G * GM);
    double* w = (double*)malloc(sizeof(double) * K);
    uint8_t* nib = pack_nibbles(X, N, P, gw);
    for (int k = 0; k < K; ++k) {
        theta_group_table(P, gw, K, k, theta, T);
        score[k] = table_sum(oracle_log(pi[k]), T, nib + (size_t)i * G, G, GM);
    }
    if (scores_to_weights(score, K, w)) {
        double tot = 0.0;
        for (int k = 0; k < K; ++k) tot = tot + w[k];
        for (int k = 0; k < K; ++k) norm[k] = w[k] / tot;
    }
    free(T); free(w); free(nib);
}

/* ------------------------------------------------------------------ literal samplers */
typedef struct { int* v; int n, cap; } ilist;
static void il_push(ilist* l, int x) {
    if (l->n == l->cap) { l->cap = l->cap ? 2 * l->cap : 16; l->v = (int*)realloc(l->v, sizeof(int) * l->cap); }
    l->v[l->n++] = x;
}
static void il_remove(ilist* l, int x) { /* erase(remove(...)) : drop every x, keep order */
    int w = 0;
    for (int r = 0; r < l->n; ++r) if (l->v[r] != x) l->v[w++] = l->v[r];
    l->n = w;
}

/* collapsed_gibbs.cpp:24-244 with relabel = FALSE, debug = FALSE; the rmultinom draw
 * (:154) is replaced by one Philox uniform and an inverse-CDF walk in label order. */
int oracle_collapsed_literal(const int32_t* X, int64_t N, int P, const int32_t* z0, int nsamples,
                             int K, double alpha, double beta, double gamma, double a, double b,
                             int burnin, uint64_t seed, int32_t* z_out, double* theta_out,
                             double* alpha_out) {
    if (nsamples < 1 || burnin < 0 || burnin > nsamples) return fail("bad nsamples/burnin");
    int S = nsamples - burnin;
    int32_t* zprev = (int32_t*)malloc(sizeof(int32_t) * N);
    int32_t* zcur = (int32_t*)malloc(sizeof(int32_t) * N);
    double* alpha_sampled = (double*)malloc(sizeof(double) * nsamples);
    double* probs = (double*)malloc(sizeof(double) * K);
    ilist* clusters = (ilist*)calloc(K, sizeof(ilist));
    for (int64_t i = 0; i < N; ++i) {
        if (z0[i] < 1 || z0[i] > K) return fail("initialK out of range");
        zprev[i] = z0[i];
        il_push(&clusters[z0[i] - 1], (int)i); /* :61-63 */
    }
    if (alpha == 0) alpha_sampled[0] = 1; else for (int j = 0; j < nsamples; ++j) alpha_sampled[j] = alpha; /* :50-54 */
    if (burnin == 0) { /* row 0 = initial state (:46); thetas slice 0 is never written */
        for (int64_t i = 0; i < N; ++i) z_out[0 + (size_t)i * S] = z0[i];
        for (int q = 0; q < K * P; ++q) theta_out[q] = NAN;
        alpha_out[0] = alpha_sampled[0];
    }
    for (int j = 1; j < nsamples; ++j) { /* :84 */
        for (int64_t i = 0; i < N; ++i) { /* :86 */
            int curr_cluster = zprev[i] - 1; /* :89 */
            il_remove(&clusters[curr_cluster], (int)i);
            double probs_sum = 0;
            for (int k = 0; k < K; ++k) { /* :99 */
                ilist* Ck = &clusters[k];
                int Nk = Ck->n;
                double dummy;
                if (Nk > 0) {
                    double LHS = log(Nk + (alpha_sampled[j - 1] / K)) - log(N - 1 + alpha_sampled[j - 1]);
                    double logLH = 0;
                    for (int d = 0; d < P; ++d) {
                        int sum_xd = 0;
                        for (int c = 0; c < Ck->n; ++c) sum_xd += X[Ck->v[c] + (size_t)d * N]; /* :112-114 */
                        int xnd = X[i + (size_t)d * N];
                        double left = xnd * log(beta + sum_xd);
                        double right = (1 - xnd) * log(gamma + Nk - sum_xd);
                        double denom = log(beta + gamma + Nk);
                        double full = left + right - denom;
                        logLH += full;
                    }
                    dummy = exp(LHS + logLH); /* :130 */
                } else {
                    dummy = 0; /* :132 */
                }
                probs_sum += dummy;
                probs[k] = dummy;
            }
            for (int k = 0; k < K; ++k) probs[k] /= probs_sum; /* :148-150 */
            int pick = draw_index(probs, K, oracle_z_uniform(seed, (uint64_t)i, (uint32_t)j));
            if (pick < 0) pick = curr_cluster;
            zcur[i] = pick + 1; /* :177 */
            il_push(&clusters[pick], (int)i);
            zprev[i] = zcur[i]; /* next sweep reads z_out(j-1, i); within sweep i is never re-read */
        }
        if (j >= burnin) {
            int s = j - burnin;
            for (int64_t i = 0; i < N; ++i) z_out[s + (size_t)i * S] = zcur[i];
            for (int k = 0; k < K; ++k) { /* :205-219 */
                ilist* Ck = &clusters[k];
                int Nk = Ck->n;
                for (int d = 0; d < P; ++d) {
                    int dsum = 0;
                    for (int c = 0; c < Ck->n; ++c) dsum += X[Ck->v[c] + (size_t)d * N];
                    theta_out[k + (size_t)d * K + (size_t)s * K * P] = dsum / (double)Nk;
                }
            }
        }
        if (alpha == 0) /* :222-224 */
            alpha_sampled[j] = oracle_update_alpha(alpha_sampled[j - 1], a, b, (double)N, K, seed, (uint32_t)j);
        if (j >= burnin) alpha_out[j - burnin] = alpha_sampled[j];
    }
    for (int k = 0; k < K; ++k) free(clusters[k].v);
    free(clusters); free(zprev); free(zcur); free(alpha_sampled); free(probs);
    return 0;
}

/* smallest cluster among those with size > 0 (sizes after own removal), ties -> lowest
 * label.  The reference (collapsed_gibbs_dp.cpp:220-229) returns the POSITION in
 * used_clusters instead of the label; this build returns the label. */
static int smallest_used_label(const int* size, int maxK) {
    int best = -1;
    for (int k = 0; k < maxK; ++k)
        if (size[k] > 0 && (best < 0 || size[k] < size[best])) best = k;
    return best;
}

/* collapsed_gibbs_dp.cpp:27-300, relabel = FALSE.  Deviations (all in DESIGN.md):
 * categories are walked in label order (the reference keeps creation order and draws
 * with RcppArmadillo::sample); truncation assigns to the smallest cluster's label;
 * alpha is redrawn once per sweep with the final K (the reference redraws per
 * observation and keeps the last, :236-240). */
int oracle_dp_literal(const int32_t* X, int64_t N, int P, int nsamples, double alpha, double beta,
                      double gamma, double a, double b, int burnin, int maxK, uint64_t seed,
                      int32_t* z_out, double* theta_out, double* alpha_out) {
    if (beta != gamma) return fail("Error: sampler currently not implemented for non-symmetric priors on beta and gamma"); /* :48-50 */
    if (nsamples < 1 || burnin < 0 || burnin > nsamples) return fail("bad nsamples/burnin");
    int S = nsamples - burnin;
    ilist* clusters = (ilist*)calloc(maxK, sizeof(ilist));
    int32_t* alloc = (int32_t*)malloc(sizeof(int32_t) * N); /* 0-based label, -1 unassigned */
    double* alpha_sampled = (double*)malloc(sizeof(double) * nsamples);
    double* probs = (double*)malloc(sizeof(double) * (maxK + 1));
    double* w = (double*)malloc(sizeof(double) * (maxK + 1));
    int* size = (int*)malloc(sizeof(int) * maxK);
    for (int64_t i = 0; i < N; ++i) alloc[i] = -1;
    if (alpha == 0) alpha_sampled[0] = 1; else for (int j = 0; j < nsamples; ++j) alpha_sampled[j] = alpha;
    double RHS_newk = P * (log(beta) - log(beta + gamma)); /* :71 */
    if (burnin == 0) { /* row 0 never written (:53); thetas zero-filled (:77) */
        for (int64_t i = 0; i < N; ++i) z_out[0 + (size_t)i * S] = ORACLE_NA_INT;
        for (int q = 0; q < maxK * P; ++q) theta_out[q] = 0.0;
        alpha_out[0] = alpha_sampled[0];
    }
    int K = 0;
    for (int j = 1; j < nsamples; ++j) {
        double left_denom = log(N - 1 + alpha_sampled[j - 1]);           /* :102 */
        double probs_newk = log(alpha_sampled[j - 1]) - left_denom + RHS_newk; /* :106 */
        for (int64_t i = 0; i < N; ++i) {
            if (j > 1) { /* :112-131 */
                int curr_cluster = alloc[i];
                il_remove(&clusters[curr_cluster], (int)i);
                if (clusters[curr_cluster].n == 0) K--;
            }
            for (int k = 0; k < maxK; ++k) { /* label order over used clusters */
                ilist* Ck = &clusters[k];
                int Nk = Ck->n;
                size[k] = Nk;
                if (Nk == 0) { probs[k] = -INFINITY; continue; }
                double LHS = log(Nk) - left_denom;
                double logLH = 0;
                double denom = log(beta + gamma + Nk);
                for (int d = 0; d < P; ++d) {
                    int sum_xd = 0;
                    for (int c = 0; c < Ck->n; ++c) sum_xd += X[Ck->v[c] + (size_t)d * N];
                    int xnd = X[i + (size_t)d * N];
                    double left = xnd * log(beta + sum_xd);
                    double right = (1 - xnd) * log(gamma + Nk - sum_xd);
                    logLH += left + right - denom;
                }
                probs[k] = LHS + logLH;
            }
            probs[maxK] = probs_newk;
            double max_prob = probs[maxK]; /* :174-186 */
            for (int k = 0; k < maxK; ++k) if (probs[k] > max_prob) max_prob = probs[k];
            double sumprob = 0;
            for (int k = 0; k <= maxK; ++k) { w[k] = exp(probs[k] - max_prob); sumprob += w[k]; }
            for (int k = 0; k <= maxK; ++k) w[k] /= sumprob;
            int ret = draw_index(w, maxK + 1, oracle_z_uniform(seed, (uint64_t)i, (uint32_t)j));
            if (ret == maxK) { /* :212-231 */
                if (K < maxK - 1) {
                    int new_cluster = 0; /* unused_clusters.top(): smallest free label */
                    while (clusters[new_cluster].n > 0) ++new_cluster;
                    ret = new_cluster;
                    K++;
                } else {
                    ret = smallest_used_label(size, maxK);
                }
            }
            il_push(&clusters[ret], (int)i);
            alloc[i] = ret;
        }
        if (j >= burnin) {
            int s = j - burnin;
            for (int64_t i = 0; i < N; ++i) z_out[s + (size_t)i * S] = alloc[i] + 1;
            for (int k = 0; k < maxK; ++k) { /* :266-281 */
                ilist* Ck = &clusters[k];
                for (int d = 0; d < P; ++d) {
                    double v = 0.0;
                    if (Ck->n > 0) {
                        int dsum = 0;
                        for (int c = 0; c < Ck->n; ++c) dsum += X[Ck->v[c] + (size_t)d * N];
                        v = dsum / (double)Ck->n;
                    }
                    theta_out[k + (size_t)d * maxK + (size_t)s * maxK * P] = v;
                }
            }
        }
        if (alpha == 0)
            alpha_sampled[j] = oracle_update_alpha(alpha_sampled[j - 1], a, b, (double)N, K, seed, (uint32_t)j);
        if (j >= burnin) alpha_out[j - burnin] = alpha_sampled[j];
    }
    for (int k = 0; k < maxK; ++k) free(clusters[k].v);
    free(clusters); free(alloc); free(alpha_sampled); free(probs); free(w); free(size);
    return 0;
}

/* parameter draws shared by the two stick-breaking restatements (stickbreaking.cpp:164-235) */
static void sb_draw_params(int full, int maxK, int P, const int32_t* ck, const int32_t* Vkd /*k*P+d*/,
                           double alpha_prev, double beta, double gamma, uint64_t seed, uint32_t j,
                           double* pi, double* theta /*maxK x P colmajor*/, int* K_viable) {
    double* v = (double*)malloc(sizeof(double) * maxK);
    int64_t num_previous_clusters = 0;
    if (full) { /* full_gibbs.cpp:203-210 with rdirichlet_cpp :10-27 */
        double sum_term = 0;
        for (int k = 0; k < maxK; ++k) {
            double dirich = (alpha_prev / maxK) + ck[k];
            v[k] = oracle_rgamma(dirich, seed, (uint32_t)k, j, 1);
            sum_term += v[k];
        }
        for (int k = 0; k < maxK; ++k) pi[k] = v[k] / sum_term;
        *K_viable = maxK; /* update_alpha(..., N, K), full_gibbs.cpp:228-230 */
        goto thetas;
    }
    for (int k = maxK - 1; k >= 0; --k) { /* :170-194 */
        double beta1 = 1 + ck[k];
        double beta2 = alpha_prev + (double)num_previous_clusters;
        v[k] = oracle_rbeta(beta1, beta2, seed, (uint32_t)k, (uint32_t)k, j, 1, 2);
        num_previous_clusters += ck[k];
    }
    v[maxK - 1] = 1; /* :195 */
    int kv = 0;
    pi[0] = v[0];
    if (pi[0] > 0.01) kv++;
    double cumprod = 1 - v[0];
    for (int k = 1; k < maxK; ++k) { /* :206-212 */
        pi[k] = cumprod * v[k];
        if (pi[k] > 0.01) kv++;
        cumprod *= (1 - v[k]);
    }
    *K_viable = kv;
thetas:
    for (int k = 0; k < maxK; ++k)
        for (int d = 0; d < P; ++d) { /* stickbreaking.cpp:217-229, full_gibbs.cpp:213-226 */
            int32_t V = Vkd[(size_t)k * P + d];
            uint32_t c0 = (uint32_t)((size_t)k * P + d);
            theta[k + (size_t)d * maxK] =
                oracle_rbeta(beta + (double)V, (gamma + (double)ck[k]) - (double)V, seed, c0, c0, j, 3, 4);
        }
    free(v);
}

static int sb_common(int literal, int full, const int32_t* X, int64_t N, int P, const double* pi0,
                     const double* theta0, int nsamples, int maxK, double alpha, double beta,
                     double gamma, double a, double b, int burnin, uint64_t seed, double* pi_out,
                     int32_t* z_out, double* theta_out, double* alpha_out) {
    if (nsamples < 1 || burnin < 0 || burnin > nsamples) return fail("bad nsamples/burnin");
    const int gw = oracle_group_width_for(full ? 3 : 2, maxK, P), GM = 1 << gw;
    int S = nsamples - burnin, G = (P + gw - 1) / gw;
    double* pi = (double*)malloc(sizeof(double) * maxK);
    double* theta = (double*)malloc(sizeof(double) * (size_t)maxK * P);
    double* alpha_sampled = (double*)malloc(sizeof(double) * nsamples);
    double* s = (double*)malloc(sizeof(double) * maxK);
    double* w = (double*)malloc(sizeof(double) * maxK);
    double* T = (double*)malloc(sizeof(double) * (size_t)maxK * G * GM);
    double* C = (double*)malloc(sizeof(double) * maxK);
    int32_t* zcur = (int32_t*)malloc(sizeof(int32_t) * N);
    int32_t* ck = (int32_t*)malloc(sizeof(int32_t) * maxK);
    int32_t* Vkd = (int32_t*)malloc(sizeof(int32_t) * (size_t)maxK * P);
    uint8_t* nib = literal ? NULL : pack_nibbles(X, N, P, gw);
    memcpy(pi, pi0, sizeof(double) * maxK);
    memcpy(theta, theta0, sizeof(double) * (size_t)maxK * P);
    if (alpha == 0) alpha_sampled[0] = 1; else for (int j = 0; j < nsamples; ++j) alpha_sampled[j] = alpha;
    if (burnin == 0) { /* slice 0 = initial values (:43,:50); z_out row 0 never written */
        for (int k = 0; k < maxK; ++k) pi_out[0 + (size_t)k * S] = pi0[k];
        memcpy(theta_out, theta0, sizeof(double) * (size_t)maxK * P);
        for (int64_t i = 0; i < N; ++i) z_out[0 + (size_t)i * S] = ORACLE_NA_INT;
        alpha_out[0] = alpha_sampled[0];
    }
    for (int j = 1; j < nsamples; ++j) {
        if (!literal)
            for (int k = 0; k < maxK; ++k) {
                theta_group_table(P, gw, maxK, k, theta, T + (size_t)k * G * GM);
                C[k] = oracle_log(pi[k]);
            }
        for (int64_t i = 0; i < N; ++i) { /* :70-125 */
            int pick;
            double u = oracle_z_uniform(seed, (uint64_t)i, (uint32_t)j);
            if (literal) {
                double cum_probs = 0;
                for (int k = 0; k < maxK; ++k) {
                    double loglh = 0;
                    for (int d = 0; d < P; ++d) {
                        int x = X[i + (size_t)d * N];
                        double th = theta[k + (size_t)d * maxK];
                        /* :80 is x*log(th) + (1-x)*log(1-th); 0*(-inf) there is NaN, here the
                         * unselected term is dropped so th = 0 or 1 stays finite-or-(-inf) */
                        loglh += x ? log(th) : log(1 - th);
                    }
                    double dummy = exp(log(pi[k]) + loglh); /* :89 */
                    s[k] = dummy;
                    cum_probs += dummy;
                }
                for (int p = 0; p < maxK; ++p) s[p] /= cum_probs; /* :103-105 */
                pick = draw_index(s, maxK, u);
            } else {
                for (int k = 0; k < maxK; ++k)
                    s[k] = table_sum(C[k], T + (size_t)k * G * GM, nib + (size_t)i * G, G, GM);
                pick = scores_to_weights(s, maxK, w) ? draw_index(w, maxK, u) : -1;
            }
            if (pick < 0) pick = (j > 1) ? zcur[i] : 0;
            zcur[i] = pick;
        }
        memset(ck, 0, sizeof(int32_t) * maxK); /* :164-186 */
        memset(Vkd, 0, sizeof(int32_t) * (size_t)maxK * P);
        for (int64_t i = 0; i < N; ++i) {
            int k = zcur[i];
            ck[k]++;
            for (int d = 0; d < P; ++d) Vkd[(size_t)k * P + d] += X[i + (size_t)d * N] & 1;
        }
        int K_viable;
        sb_draw_params(full, maxK, P, ck, Vkd, alpha_sampled[j - 1], beta, gamma, seed, (uint32_t)j, pi, theta, &K_viable);
        if (alpha == 0) /* :233-235 */
            alpha_sampled[j] = oracle_update_alpha(alpha_sampled[j - 1], a, b, (double)N, K_viable, seed, (uint32_t)j);
        if (j >= burnin) {
            int sidx = j - burnin;
            for (int64_t i = 0; i < N; ++i) z_out[sidx + (size_t)i * S] = zcur[i] + 1;
            for (int k = 0; k < maxK; ++k) pi_out[sidx + (size_t)k * S] = pi[k];
            memcpy(theta_out + (size_t)sidx * maxK * P, theta, sizeof(double) * (size_t)maxK * P);
            alpha_out[sidx] = alpha_sampled[j];
        }
    }
    free(pi); free(theta); free(alpha_sampled); free(s); free(w); free(T); free(C);
    free(zcur); free(ck); free(Vkd); free(nib);
    return 0;
}

int oracle_sb_literal(const int32_t* X, int64_t N, int P, const double* pi0, const double* theta0,
                      int nsamples, int maxK, double alpha, double beta, double gamma, double a,
                      double b, int burnin, uint64_t seed, double* pi_out, int32_t* z_out,
                      double* theta_out, double* alpha_out) {
    return sb_common(1, 0, X, N, P, pi0, theta0, nsamples, maxK, alpha, beta, gamma, a, b, burnin, seed,
                     pi_out, z_out, theta_out, alpha_out);
}
int oracle_sb_run(const int32_t* X, int64_t N, int P, const double* pi0, const double* theta0,
                  int nsamples, int maxK, double alpha, double beta, double gamma, double a, double b,
                  int burnin, uint64_t seed, double* pi_out, int32_t* z_out, double* theta_out,
                  double* alpha_out) {
    return sb_common(0, 0, X, N, P, pi0, theta0, nsamples, maxK, alpha, beta, gamma, a, b, burnin, seed,
                     pi_out, z_out, theta_out, alpha_out);
}

/* full_gibbs.cpp:32-249 (gibbs_cpp): the z-step is stickbreaking.cpp's with K for maxK */
int oracle_full_literal(const int32_t* X, int64_t N, int P, const double* pi0, const double* theta0,
                        int nsamples, int K, double alpha, double beta, double gamma, double a, double b,
                        int burnin, uint64_t seed, double* pi_out, int32_t* z_out, double* theta_out,
                        double* alpha_out) {
    return sb_common(1, 1, X, N, P, pi0, theta0, nsamples, K, alpha, beta, gamma, a, b, burnin, seed,
                     pi_out, z_out, theta_out, alpha_out);
}
int oracle_full_run(const int32_t* X, int64_t N, int P, const double* pi0, const double* theta0,
                    int nsamples, int K, double alpha, double beta, double gamma, double a, double b,
                    int burnin, uint64_t seed, double* pi_out, int32_t* z_out, double* theta_out,
                    double* alpha_out) {
    return sb_common(0, 1, X, N, P, pi0, theta0, nsamples, K, alpha, beta, gamma, a, b, burnin, seed,
                     pi_out, z_out, theta_out, alpha_out);
}

/* ------------------------------------------------------------------ sufficient-statistics chains */
typedef struct {
    int sampler; /* 0 collapsed, 1 dp */
    int64_t N; int P, K, gw, G, Gm;  /* K = number of labels (K or maxK); groups of gw / of GWM features */
    const int32_t* X;
    uint8_t* nib; uint8_t* nibm;
    int32_t* z;      /* current 0-based label, -1 unassigned */
    int32_t* znew;   /* batch scratch */
    int32_t* Nk; int32_t* S; /* K, K*P */
    double* Tp; double* Tm;  /* K*G*GM, K*Gm*GMM */
    double* Cp; double* Cm;  /* K (+1 for dp new) */
    unsigned char* dirty;
    double* score; double* w; double* e;
    double alpha_cur, beta, gamma, a, b;
    double alpha_consts; /* the alpha Cp/Cm were computed with (NaN: none yet) */
    int sample_alpha;
    uint64_t seed;
    int64_t batch;
} ochain;

static void chain_free(ochain* c) {
    free(c->nib); free(c->nibm); free(c->z); free(c->znew); free(c->Nk); free(c->S); free(c->Tp); free(c->Tm);
    free(c->Cp); free(c->Cm); free(c->dirty); free(c->score); free(c->w); free(c->e);
}
static int chain_init(ochain* c, int sampler, const int32_t* X, int64_t N, int P, int K,
                      const int32_t* z0_1based, double alpha, double beta, double gamma, double a,
                      double b, int64_t batch, uint64_t seed) {
    memset(c, 0, sizeof *c);
    c->sampler = sampler; c->N = N; c->P = P; c->K = K; c->gw = oracle_group_width_for(sampler, K, P);
    c->G = (P + c->gw - 1) / c->gw; c->Gm = (P + GWM - 1) / GWM; c->X = X;
    c->beta = beta; c->gamma = gamma; c->a = a; c->b = b; c->seed = seed;
    c->batch = batch < 1 ? 1 : (batch > N ? N : batch);
    c->sample_alpha = (alpha == 0);
    c->alpha_cur = c->sample_alpha ? 1.0 : alpha;
    size_t tg = (size_t)K * c->G * ((size_t)1 << c->gw), tgm = (size_t)K * c->Gm * GMM;
    c->nib = pack_nibbles(X, N, P, c->gw);
    c->nibm = pack_nibbles(X, N, P, GWM);
    c->z = (int32_t*)malloc(sizeof(int32_t) * N);
    c->znew = (int32_t*)malloc(sizeof(int32_t) * N);
    c->Nk = (int32_t*)calloc(K, sizeof(int32_t));
    c->S = (int32_t*)calloc((size_t)K * P, sizeof(int32_t));
    c->Tp = (double*)calloc(tg, sizeof(double));
    c->Tm = (double*)calloc(tgm, sizeof(double));
    c->Cp = (double*)calloc(K + 1, sizeof(double));
    c->Cm = (double*)calloc(K + 1, sizeof(double));
    c->dirty = (unsigned char*)malloc(K);
    c->score = (double*)malloc(sizeof(double) * (K + 1));
    c->w = (double*)malloc(sizeof(double) * (K + 1));
    c->e = (double*)malloc(sizeof(double) * 2 * P);
    if (!c->nib || !c->nibm || !c->z || !c->znew || !c->Tp || !c->Tm) return fail("out of memory");
    memset(c->dirty, 1, K);
    c->alpha_consts = NAN;
    for (int64_t i = 0; i < N; ++i) {
        int k = z0_1based ? z0_1based[i] - 1 : -1;
        if (z0_1based && (k < 0 || k >= K)) return fail("initialK out of range");
        c->z[i] = k;
        if (k >= 0) {
            c->Nk[k]++;
            for (int d = 0; d < P; ++d) c->S[(size_t)k * P + d] += X[i + (size_t)d * N] & 1;
        }
    }
    return 0;
}
static void chain_apply(ochain* c, int64_t i, int zn) {
    int zo = c->z[i];
    if (zo == zn) return;
    const int P = c->P;
    if (zo >= 0) {
        c->Nk[zo]--; c->dirty[zo] = 1;
        for (int d = 0; d < P; ++d) c->S[(size_t)zo * P + d] -= c->X[i + (size_t)d * c->N] & 1;
    }
    c->Nk[zn]++; c->dirty[zn] = 1;
    for (int d = 0; d < P; ++d) c->S[(size_t)zn * P + d] += c->X[i + (size_t)d * c->N] & 1;
    c->z[i] = zn;
}
/* one batch [lo, hi) of sweep j against statistics frozen at batch start */
static void chain_batch(ochain* c, int64_t lo, int64_t hi, uint32_t j) {
    const int K = c->K, G = c->G, Gm = c->Gm, P = c->P;
    const int GM = 1 << c->gw;
    const size_t tk = (size_t)G * GM, tkm = (size_t)Gm * GMM;
    const double alpha = c->alpha_cur;
    const double ldN = oracle_log((double)(c->N - 1) + alpha);
    int Kused = 0, new_label = -1;
    /* tables and constants are functions of (Nk, S, alpha) only: a cluster no observation entered or
     * left since they were last computed, under the same alpha, keeps them (same values, computed once
     * -- what makes batch = 1 affordable at N = 10^6) */
    const int all = !(c->alpha_consts == alpha);
    for (int k = 0; k < K; ++k) {
        const int64_t n = c->Nk[k];
        if (c->dirty[k] || all) {
            if (c->sampler == 0) {
                c->Cp[k] = n > 0 ? oracle_log((double)n + alpha / (double)K) - ldN : O_NEG_INF;
                c->Cm[k] = n > 1 ? oracle_log((double)(n - 1) + alpha / (double)K) - ldN : O_NEG_INF;
            } else {
                c->Cp[k] = n > 0 ? oracle_log((double)n) - ldN : O_NEG_INF;
                c->Cm[k] = n > 1 ? oracle_log((double)(n - 1)) - ldN : O_NEG_INF;
            }
        }
        if (c->dirty[k]) {
            counts_group_table(c->beta, c->gamma, P, c->gw, c->Nk[k], c->S + (size_t)k * P, 0, c->e, c->e + P, c->Tp + k * tk);
            counts_group_table(c->beta, c->gamma, P, GWM, c->Nk[k], c->S + (size_t)k * P, 1, c->e, c->e + P, c->Tm + k * tkm);
            c->dirty[k] = 0;
        }
        if (n > 0) Kused++; else if (new_label < 0) new_label = k;
    }
    const int ncat = c->sampler == 1 ? K + 1 : K;
    if (c->sampler == 1 && all) c->Cp[K] = dp_new_score(alpha, c->beta, c->gamma, P, ldN);
    c->alpha_consts = alpha;
    for (int64_t i = lo; i < hi; ++i) {
        const uint8_t* nb = c->nib + (size_t)i * G;
        const uint8_t* nbm = c->nibm + (size_t)i * Gm;
        const int zo = c->z[i];
        for (int k = 0; k < K; ++k) {
            if (k == zo) c->score[k] = table_sum(c->Cm[k], c->Tm + k * tkm, nbm, Gm, GMM);
            else c->score[k] = table_sum(c->Cp[k], c->Tp + k * tk, nb, G, GM);
        }
        if (c->sampler == 1) c->score[K] = c->Cp[K] + 0.0;
        int pick = scores_to_weights(c->score, ncat, c->w)
                       ? draw_index(c->w, ncat, oracle_z_uniform(c->seed, (uint64_t)i, j)) : -1;
        if (pick < 0) pick = zo >= 0 ? zo : 0;
        if (c->sampler == 1 && pick == K) {
            const int own_single = (zo >= 0 && c->Nk[zo] == 1);
            if (Kused - own_single < K - 1) { /* collapsed_gibbs_dp.cpp:213 */
                pick = new_label;
                if (own_single && (pick < 0 || zo < pick)) pick = zo;
            } else {
                int best = -1; int64_t bs = 0;
                for (int k = 0; k < K; ++k) {
                    int64_t sz = (int64_t)c->Nk[k] - (k == zo);
                    if (sz > 0 && (best < 0 || sz < bs)) { best = k; bs = sz; }
                }
                pick = best >= 0 ? best : (zo >= 0 ? zo : 0);
            }
        }
        c->znew[i] = pick;
    }
    for (int64_t i = lo; i < hi; ++i) chain_apply(c, i, c->znew[i]);
}
/* batch schedule of sweep j: uniform batches, except the DP's first sweep, which seats
 * observations with batch lengths min(batch, max(1, start)) = 1,1,2,4,... */
static void chain_sweep(ochain* c, uint32_t j) {
    int64_t lo = 0;
    while (lo < c->N) {
        int64_t len = c->batch;
        if (c->sampler == 1 && j == 1) { int64_t dbl = lo < 1 ? 1 : lo; if (dbl < len) len = dbl; }
        int64_t hi = lo + len > c->N ? c->N : lo + len;
        chain_batch(c, lo, hi, j);
        lo = hi;
    }
    if (c->sample_alpha) {
        int Kc = c->K;
        if (c->sampler == 1) { Kc = 0; for (int k = 0; k < c->K; ++k) if (c->Nk[k] > 0) Kc++; }
        c->alpha_cur = oracle_update_alpha(c->alpha_cur, c->a, c->b, (double)c->N, Kc, c->seed, j);
    }
}
static void chain_emit(const ochain* c, int s, int S, int32_t* z_out, double* theta_out, double* alpha_out) {
    const int K = c->K, P = c->P;
    if (z_out) for (int64_t i = 0; i < c->N; ++i) z_out[s + (size_t)i * S] = c->z[i] + 1;
    for (int k = 0; k < K; ++k)
        for (int d = 0; d < P; ++d) {
            double v;
            if (c->sampler == 1 && c->Nk[k] == 0) v = 0.0; /* dp: unused labels stay 0 (:77) */
            else v = (double)c->S[(size_t)k * P + d] / (double)c->Nk[k]; /* 0/0 = NaN (collapsed :214) */
            theta_out[k + (size_t)d * K + (size_t)s * K * P] = v;
        }
    alpha_out[s] = c->alpha_cur;
}

/* z_out may be null (no label trace); nk_out, if given, receives the cluster sizes after every kept
 * sweep (row s = sweep burnin + s, K entries) and z_last the labels (1-based) after the last sweep */
static int run_counts_chain(int sampler, const int32_t* X, int64_t N, int P, const int32_t* z0,
                            int nsamples, int K, double alpha, double beta, double gamma, double a,
                            double b, int burnin, int64_t batch, uint64_t seed, int32_t* z_out,
                            double* theta_out, double* alpha_out, int32_t* nk_out, int32_t* z_last) {
    if (nsamples < 1 || burnin < 0 || burnin > nsamples) return fail("bad nsamples/burnin");
    if (sampler == 1 && beta != gamma)
        return fail("Error: sampler currently not implemented for non-symmetric priors on beta and gamma");
    ochain c;
    if (chain_init(&c, sampler, X, N, P, K, z0, alpha, beta, gamma, a, b, batch, seed)) { chain_free(&c); return 1; }
    int S = nsamples - burnin;
    if (burnin == 0) {
        if (z_out) for (int64_t i = 0; i < N; ++i) z_out[0 + (size_t)i * S] = z0 ? z0[i] : ORACLE_NA_INT;
        for (int q = 0; q < K * P; ++q) theta_out[q] = sampler == 1 ? 0.0 : NAN;
        alpha_out[0] = c.alpha_cur;
        if (nk_out) memcpy(nk_out, c.Nk, sizeof(int32_t) * K);
    }
    /* alpha_sampled(j) keeps its previous value when alpha is fixed; when sampled the
     * sweep consumes alpha_sampled(j-1) and then draws alpha_sampled(j) */
    const char* progress = getenv("ORACLE_PROGRESS"); /* fixture generation only: a line per sweep on stderr */
    for (int j = 1; j < nsamples; ++j) {
        chain_sweep(&c, (uint32_t)j);
        if (j >= burnin) {
            chain_emit(&c, j - burnin, S, z_out, theta_out, alpha_out);
            if (nk_out) memcpy(nk_out + (size_t)(j - burnin) * K, c.Nk, sizeof(int32_t) * K);
        }
        if (progress) fprintf(stderr, "[oracle %s seed %llu] sweep %d / %d\n", progress, (unsigned long long)seed, j, nsamples - 1);
    }
    if (z_last) for (int64_t i = 0; i < N; ++i) z_last[i] = c.z[i] < 0 ? ORACLE_NA_INT : c.z[i] + 1;
    chain_free(&c);
    return 0;
}

int oracle_collapsed_run(const int32_t* X, int64_t N, int P, const int32_t* z0, int nsamples, int K,
                         double alpha, double beta, double gamma, double a, double b, int burnin,
                         int64_t batch, uint64_t seed, int32_t* z_out, double* theta_out,
                         double* alpha_out) {
    if (!z0) return fail("initialK required");
    return run_counts_chain(0, X, N, P, z0, nsamples, K, alpha, beta, gamma, a, b, burnin, batch, seed,
                            z_out, theta_out, alpha_out, NULL, NULL);
}
int oracle_dp_run(const int32_t* X, int64_t N, int P, int nsamples, double alpha, double beta,
                  double gamma, double a, double b, int burnin, int maxK, int64_t batch,
                  uint64_t seed, int32_t* z_out, double* theta_out, double* alpha_out) {
    return run_counts_chain(1, X, N, P, NULL, nsamples, maxK, alpha, beta, gamma, a, b, burnin, batch,
                            seed, z_out, theta_out, alpha_out, NULL, NULL);
}
/* The same chains without the S x N label trace (which does not fit at N = 10^6 and hundreds of kept
 * sweeps): cluster sizes per kept sweep, theta-hat, alpha, and the labels after the last sweep.
 * sampler 0: collapsed (z0 required), 1: dp (z0 ignored, K = maxK).  For tests/golden/make_tolerance_fixtures.py. */
int oracle_counts_summary(int sampler, const int32_t* X, int64_t N, int P, const int32_t* z0, int nsamples,
                          int K, double alpha, double beta, double gamma, double a, double b, int burnin,
                          int64_t batch, uint64_t seed, int32_t* nk_out, double* theta_out,
                          double* alpha_out, int32_t* z_last) {
    if (sampler != 0 && sampler != 1) return fail("sampler must be 0 (collapsed) or 1 (dp)");
    if (sampler == 0 && !z0) return fail("initialK required");
    return run_counts_chain(sampler, X, N, P, sampler == 0 ? z0 : NULL, nsamples, K, alpha, beta, gamma, a, b,
                            burnin, batch, seed, NULL, theta_out, alpha_out, nk_out, z_last);
}

/* ------------------------------------------------------------------ CPU baseline timing */
typedef struct {
    int sampler; const int32_t* X; int64_t N; int P, K, sweeps; int64_t batch; uint64_t seed;
    pthread_barrier_t* bar; double t0, t1; int rc;
} tjob;

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
static void* time_worker(void* arg) {
    tjob* t = (tjob*)arg;
    t->rc = 0;
    if (t->sampler == 2) {
        /* stick-breaking: time through the public entry (set-up is O(N P), one sweep's worth
         * of packing; reported as part of the time) */
        int maxK = t->K;
        double* pi0 = (double*)malloc(sizeof(double) * maxK);
        double* th0 = (double*)malloc(sizeof(double) * (size_t)maxK * t->P);
        for (int k = 0; k < maxK; ++k) pi0[k] = 1.0 / maxK;
        for (size_t q = 0; q < (size_t)maxK * t->P; ++q)
            th0[q] = 0.1 + 0.8 * oracle_u01((uint32_t)(q * 2654435761u), (uint32_t)(q * 40503u + 7));
        int S = 1, ns = t->sweeps + 1;
        double* pio = (double*)malloc(sizeof(double) * maxK * S);
        int32_t* zo = (int32_t*)malloc(sizeof(int32_t) * t->N * S);
        double* tho = (double*)malloc(sizeof(double) * (size_t)maxK * t->P * S);
        double ao[1];
        pthread_barrier_wait(t->bar);
        t->t0 = now_s();
        t->rc = oracle_sb_run(t->X, t->N, t->P, pi0, th0, ns, maxK, 0.0, 0.5, 0.5, 1, 1, ns - 1, t->seed,
                              pio, zo, tho, ao);
        t->t1 = now_s();
        free(pi0); free(th0); free(pio); free(zo); free(tho);
        return NULL;
    }
    ochain c;
    int32_t* z0 = NULL;
    if (t->sampler == 0) {
        z0 = (int32_t*)malloc(sizeof(int32_t) * t->N);
        for (int64_t i = 0; i < t->N; ++i)
            z0[i] = 1 + (int32_t)(oracle_z_uniform(t->seed ^ 0x5eedull, (uint64_t)i, 0) * t->K);
    }
    if (chain_init(&c, t->sampler, t->X, t->N, t->P, t->K, z0, 0.0, 0.5, 0.5, 1, 1, t->batch, t->seed)) {
        t->rc = 1; chain_free(&c); free(z0);
        pthread_barrier_wait(t->bar);
        return NULL;
    }
    pthread_barrier_wait(t->bar);
    t->t0 = now_s();
    for (int j = 1; j <= t->sweeps; ++j) chain_sweep(&c, (uint32_t)j);
    t->t1 = now_s();
    chain_free(&c); free(z0);
    return NULL;
}
double oracle_time_sweeps(int sampler, const int32_t* X, int64_t N, int P, int K, int sweeps,
                          int64_t batch, uint64_t seed, int nthreads) {
    if (nthreads < 1 || sweeps < 1) { fail("bad nthreads/sweeps"); return -1.0; }
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * nthreads);
    tjob* jobs = (tjob*)calloc(nthreads, sizeof(tjob));
    pthread_barrier_t bar;
    pthread_barrier_init(&bar, NULL, nthreads);
    for (int t = 0; t < nthreads; ++t) {
        tjob j = {sampler, X, N, P, K, sweeps, batch, seed + (uint64_t)t, &bar, 0, 0, 0};
        jobs[t] = j;
        pthread_create(&th[t], NULL, time_worker, &jobs[t]);
    }
    double t0 = 1e300, t1 = 0;
    int rc = 0;
    for (int t = 0; t < nthreads; ++t) {
        pthread_join(th[t], NULL);
        if (jobs[t].t0 < t0) t0 = jobs[t].t0;
        if (jobs[t].t1 > t1) t1 = jobs[t].t1;
        rc |= jobs[t].rc;
    }
    pthread_barrier_destroy(&bar);
    free(th); free(jobs);
    if (rc) { fail("chain failed"); return -1.0; }
    return t1 - t0;
}
