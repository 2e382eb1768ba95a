// bmm_spec.h -- the numerics of the allocation path, written once for host and device.
//
// Everything the categorical draw of z_n depends on is defined here with IEEE-754
// binary64 add / mul / fma / div / sqrt / floor and integer bit operations only, in a
// fixed order, so that a gfx950 wave and an x86 core produce the same bits (the build
// passes -ffp-contract=off to both compilers; every fused operation is an explicit
// bmm::fma_). No libm transcendental is called: log/exp are implemented below.
// tests/test_gpu_math.py checks device-vs-host bit equality of every function here;
// oracle/bmm_oracle.c restates the same arithmetic independently in plain C.
//
// Reference arithmetic being restated (all /root/reference):
//   collapsed conditional   src/collapsed_gibbs.cpp:99-137
//   DP conditional          src/collapsed_gibbs_dp.cpp:71,102-106,140-186
//   stick-breaking z-step   src/stickbreaking.cpp:75-105
//   alpha update            src/utils.cpp:6-14
// The reference draws through R's Mersenne-Twister (rmultinom / rbeta / rgamma); this
// build draws through Philox (counter-based, Salmon et al. 2011): Philox2x32-10 for the one
// uniform per observation and sweep that decides z_n, Philox4x32-10 for the parameter variates;
// see DESIGN.md.
#pragma once
#include <stdint.h>

#include "bmm_exp256.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define BMM_HD __host__ __device__ __forceinline__
#else
#define BMM_HD inline
#endif

namespace bmm {

// ---------------------------------------------------------------- constants
// Features per lookup group.  The tables against the full statistics use groups of kGroupW features
// (32-entry tables: 20 % fewer lookups and adds per observation than groups of four) whenever the table
// image of the shape fits in the 160 KiB of LDS that way, and groups of kGroupWAlt otherwise
// (group_width_for: a pure function of sampler, K and P, part of the arithmetic because it fixes the order
// of the sums).  The own-cluster ("minus self") tables are read once per observation, not once per
// category, so they use narrow groups (small tables) throughout.
constexpr int kGroupW = 5;
constexpr int kGroupWAlt = 4;
constexpr int kGroupWm = 3;
constexpr int kGroupMm = 1 << kGroupWm;
constexpr int kOwnSub = 6;  // the own-cluster tables are padded with zero groups to a multiple of this

// Philox stream ids (counter word 3)
enum : uint32_t {
    kStreamZ = 0,        // (reserved: the per-observation uniform is Philox2x32 under its own key, z_uniform)
    kStreamStickA = 1,   // v_k ~ Beta: first gamma;  c0 = k, c1 = block counter, c2 = sweep
    kStreamStickB = 2,   // v_k ~ Beta: second gamma
    kStreamThetaA = 3,   // theta_kd ~ Beta: first gamma; c0 = k*P+d
    kStreamThetaB = 4,
    kStreamAlphaEta = 5, // eta ~ Beta(alpha+1, N): first gamma (c0 = 0), second gamma (c0 = 1)
    kStreamAlphaG1 = 6,  // Gamma(a+K)
    kStreamAlphaG2 = 7,  // Gamma(a+K-1)
};

// ---------------------------------------------------------------- bit helpers
BMM_HD uint64_t dbits(double x) {
    union { double d; uint64_t u; } c; c.d = x; return c.u;
}
BMM_HD double dfrom(uint64_t u) {
    union { double d; uint64_t u; } c; c.u = u; return c.d;
}
BMM_HD double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
BMM_HD double floor_(double a) { return __builtin_floor(a); }
// IEEE correctly-rounded on both sides (checked bitwise on the GPU by test_gpu_math).
BMM_HD double div_(double a, double b) { return a / b; }
BMM_HD double sqrt_(double a) { return __builtin_sqrt(a); }

BMM_HD double neg_inf() { return dfrom(0xfff0000000000000ull); }
BMM_HD double pos_inf() { return dfrom(0x7ff0000000000000ull); }
BMM_HD double qnan() { return dfrom(0x7ff8000000000000ull); }

// ---------------------------------------------------------------- Philox4x32-10
struct U4 { uint32_t x, y, z, w; };

BMM_HD uint32_t mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32); }

BMM_HD U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = mulhi32(M0, c0), lo0 = M0 * c0;
        const uint32_t hi1 = mulhi32(M1, c2), lo1 = M1 * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += W0; k1 += W1;
    }
    U4 o; o.x = c0; o.y = c1; o.z = c2; o.w = c3; return o;
}

// Philox2x32-10: one 32x32->64 multiply per round; two output words = the one uniform a
// categorical draw needs (a 4x32 block would throw half of its four multiplies per round away).
BMM_HD void philox2x32_10(uint32_t c0, uint32_t c1, uint32_t k, uint32_t& o0, uint32_t& o1) {
    const uint32_t M = 0xD256D193u, W = 0x9E3779B9u;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int r = 0; r < 10; ++r) {
        const uint64_t pr = (uint64_t)M * (uint64_t)c0;
        c0 = (uint32_t)(pr >> 32) ^ k ^ c1;
        c1 = (uint32_t)pr;
        k += W;
    }
    o0 = c0; o1 = c1;
}

// 52-bit uniform in [0,1): the words fill the mantissa of a double in [1,2), minus 1 (no
// integer-to-double conversions).  Its largest value is 1 - 2^-52, and u * t < t then holds in
// binary64 for every finite t > 0, so an inverse-CDF walk against t = u * total always ends inside.
BMM_HD double u52(uint32_t a, uint32_t b) {
    const uint32_t hi = 0x3ff00000u | (a >> 12);
    const uint32_t lo = (a << 20) | (b >> 12);
    return dfrom(((uint64_t)hi << 32) | lo) - 1.0;
}

// 53-bit uniform in [0,1) from two 32-bit words (parameter variates).
BMM_HD double u01(uint32_t a, uint32_t b) {
    const double hi = (double)(a >> 5), lo = (double)(b >> 6);
    return (hi * 67108864.0 + lo) * 1.1102230246251565404e-16;  // 2^-53
}
// uniform in (0,1]: 1 - u01
BMM_HD double u01_open0(uint32_t a, uint32_t b) { return 1.0 - u01(a, b); }

// A counted stream of Philox blocks: (c0, block counter, c2, stream id) under one key.
struct Stream {
    uint32_t c0, ctr, c2, sid, k0, k1;
    BMM_HD U4 next() { U4 r = philox4x32_10(c0, ctr, c2, sid, k0, k1); ++ctr; return r; }
};
BMM_HD Stream make_stream(uint64_t seed, uint32_t c0, uint32_t c2, uint32_t sid) {
    Stream s; s.c0 = c0; s.ctr = 0; s.c2 = c2; s.sid = sid;
    s.k0 = (uint32_t)seed; s.k1 = (uint32_t)(seed >> 32); return s;
}

// The uniform that decides z_i in sweep j: Philox2x32-10 at counter (i mod 2^32, sweep) under a
// 32-bit key folded from the 64-bit seed (and from the high word of i, zero below 2^32 observations).
BMM_HD uint32_t z_key(uint64_t seed, uint64_t i) {
    const uint32_t h = (uint32_t)(i >> 32);
    return ((uint32_t)seed ^ ((uint32_t)(seed >> 32) * 0x85EBCA6Bu)) ^ ((h << 16) | (h >> 16));
}
BMM_HD double z_uniform(uint64_t seed, uint64_t i, uint32_t sweep) {
    uint32_t a, b;
    philox2x32_10((uint32_t)i, sweep, z_key(seed, i), a, b);
    return u52(a, b);
}

// ---------------------------------------------------------------- log / exp
// log(x): argument reduced to m in [sqrt(1/2), sqrt(2)), f = m-1, s = f/(2+f),
// log(1+f) = f - f^2/2 + s*(f^2/2 + R(s^2)), R an even minimax polynomial of degree 14
// (coefficients: the classic Remez set for this reduction). Error < 1 ulp.
BMM_HD double log_(double x) {
    uint64_t ix = dbits(x);
    if (x == 0.0) return neg_inf();
    if ((int64_t)ix < 0) return qnan();
    if ((ix >> 52) == 0x7ffull) return x;  // +inf, nan
    int e = 0;
    if ((ix >> 52) == 0) {  // subnormal: scale by 2^54
        x = x * 18014398509481984.0; ix = dbits(x); e = -54;
    }
    e += (int)(ix >> 52) - 1023;
    uint64_t m = (ix & 0x000fffffffffffffull) | 0x3ff0000000000000ull;
    if (m >= 0x3ff6a09e667f3bcdull) { m -= 0x0010000000000000ull; e += 1; }
    const double f = dfrom(m) - 1.0;
    const double s = div_(f, 2.0 + f);
    const double z = s * s, w = z * z;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
                 Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                 Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    const double t1 = w * fma_(w, fma_(w, Lg6, Lg4), Lg2);
    const double t2 = z * fma_(w, fma_(w, fma_(w, Lg7, Lg5), Lg3), Lg1);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)e;
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
}

// exp(x) = 2^k * 2^(j/64) * e^r with 64k + j = round(x * 64/ln2), r = x - (64k+j) ln2/64
// (two-part), |r| <= ln2/128: e^r - 1 by a degree-5 Taylor polynomial (truncation < 2^-54),
// 2^(j/64) from a 64-entry table of correctly rounded doubles (tools/gen_exp_table.py).
// Error < 1 ulp.  Results below 2^-1021 are flushed to 0 (x < -708); the sampler never needs them.
#define BMM_EXP2_64_TABLE \
    0x1.0000000000000p+0, 0x1.02c9a3e778061p+0, 0x1.059b0d3158574p+0, 0x1.0874518759bc8p+0, \
    0x1.0b5586cf9890fp+0, 0x1.0e3ec32d3d1a2p+0, 0x1.11301d0125b51p+0, 0x1.1429aaea92de0p+0, \
    0x1.172b83c7d517bp+0, 0x1.1a35beb6fcb75p+0, 0x1.1d4873168b9aap+0, 0x1.2063b88628cd6p+0, \
    0x1.2387a6e756238p+0, 0x1.26b4565e27cddp+0, 0x1.29e9df51fdee1p+0, 0x1.2d285a6e4030bp+0, \
    0x1.306fe0a31b715p+0, 0x1.33c08b26416ffp+0, 0x1.371a7373aa9cbp+0, 0x1.3a7db34e59ff7p+0, \
    0x1.3dea64c123422p+0, 0x1.4160a21f72e2ap+0, 0x1.44e086061892dp+0, 0x1.486a2b5c13cd0p+0, \
    0x1.4bfdad5362a27p+0, 0x1.4f9b2769d2ca7p+0, 0x1.5342b569d4f82p+0, 0x1.56f4736b527dap+0, \
    0x1.5ab07dd485429p+0, 0x1.5e76f15ad2148p+0, 0x1.6247eb03a5585p+0, 0x1.6623882552225p+0, \
    0x1.6a09e667f3bcdp+0, 0x1.6dfb23c651a2fp+0, 0x1.71f75e8ec5f74p+0, 0x1.75feb564267c9p+0, \
    0x1.7a11473eb0187p+0, 0x1.7e2f336cf4e62p+0, 0x1.82589994cce13p+0, 0x1.868d99b4492edp+0, \
    0x1.8ace5422aa0dbp+0, 0x1.8f1ae99157736p+0, 0x1.93737b0cdc5e5p+0, 0x1.97d829fde4e50p+0, \
    0x1.9c49182a3f090p+0, 0x1.a0c667b5de565p+0, 0x1.a5503b23e255dp+0, 0x1.a9e6b5579fdbfp+0, \
    0x1.ae89f995ad3adp+0, 0x1.b33a2b84f15fbp+0, 0x1.b7f76f2fb5e47p+0, 0x1.bcc1e904bc1d2p+0, \
    0x1.c199bdd85529cp+0, 0x1.c67f12e57d14bp+0, 0x1.cb720dcef9069p+0, 0x1.d072d4a07897cp+0, \
    0x1.d5818dcfba487p+0, 0x1.da9e603db3285p+0, 0x1.dfc97337b9b5fp+0, 0x1.e502ee78b3ff6p+0, \
    0x1.ea4afa2a490dap+0, 0x1.efa1bee615a27p+0, 0x1.f50765b6e4540p+0, 0x1.fa7c1819e90d8p+0,
static const double kExp2Host[64] = {BMM_EXP2_64_TABLE};
#if defined(__HIPCC__)
__constant__ double kExp2Dev[64] = {BMM_EXP2_64_TABLE};
#endif
BMM_HD const double* exp2_table() {
#if defined(__HIP_DEVICE_COMPILE__)
    return kExp2Dev;
#else
    return kExp2Host;
#endif
}

// core for an argument already known to be in [-708, 709.79]; Tab is any pointer-like to the table
template <class Tab>
BMM_HD double exp_core(double xs, Tab T, int& k_out) {
    const double kd = floor_(fma_(xs, 9.23324826168936567e+01, 0.5));   // 64/ln2
    double r = fma_(-kd, 1.08304246962491454e-02, xs);                   // ln2/64, high part
    r = fma_(-kd, 3.62351064663484299e-19, r);                           // low part
    const int ki = (int)kd;
    const int j = ki & 63;
    k_out = ki >> 6;
    double c = fma_(r, 8.3333333333333332177e-03, 4.1666666666666664354e-02);  // 1/120, 1/24
    c = fma_(r, c, 1.6666666666666665741e-01);                                  // 1/6
    c = fma_(r, c, 0.5);
    const double q = fma_(r * r, c, r);
    const double t = T[j];
    return fma_(t, q, t);
}

template <class Tab>
BMM_HD double exp_tab(double x, Tab T) {
    // written select-style (no early returns) so that the device code is branch-free
    const bool is_nan = x != x;
    const bool over = x > 709.782712893384;
    const bool under = x < -708.0;
    const double xs = (is_nan || over || under) ? 0.0 : x;
    int k;
    double p = exp_core(xs, T, k);
    const bool top = k > 1023;
    p = top ? p * 2.0 : p;
    k = top ? k - 1 : k;
    const double y = p * dfrom((uint64_t)(k + 1023) << 52);
    return is_nan ? x : (over ? pos_inf() : (under ? 0.0 : y));
}
BMM_HD double exp_(double x) { return exp_tab(x, exp2_table()); }

// The weight exponential of the draw: expw_(x) for x <= 0 (max-shifted scores; -inf and NaN
// arguments give exactly 0).  256 k + j = round-to-nearest-even(x * 256/ln2) by the 1.5 * 2^52
// trick (the integer is read from the low word of the sum: no floor, no conversion),
// r = x - (256 k + j) ln2/256 in two parts, |r| <= ln2/512: e^r - 1 by a degree-4 polynomial
// (truncation < 2^-54), 2^(j/256) from a 256-entry table of correctly rounded doubles
// (bmm_exp256.h), the scale 2^k by ldexp.  Arguments below -745.2 underflow to 0 through ldexp's
// own IEEE rounding; nothing is flushed by a comparison.  Error < 1 ulp in the normal range.
static const double kExp256Host[256] = {BMM_EXP2_256_TABLE};
#if defined(__HIPCC__)
__constant__ double kExp256Dev[256] = {BMM_EXP2_256_TABLE};
#endif
BMM_HD const double* exp256_table() {
#if defined(__HIP_DEVICE_COMPILE__)
    return kExp256Dev;
#else
    return kExp256Host;
#endif
}
BMM_HD double ldexp_(double p, int k) { return __builtin_ldexp(p, k); }
template <class Tab>
BMM_HD double expw_tab(double x, Tab T) {
    const double xs = __builtin_fmax(x, -1000.0);                        // NaN -> -1000
    const double km = fma_(xs, 0x1.71547652b82fep+8, 6755399441055744.0);  // 256/ln2, 1.5 * 2^52
    const int32_t ki = (int32_t)(uint32_t)dbits(km);
    const double kd = km - 6755399441055744.0;
    double r = fma_(-kd, 0x1.62e42fefa39efp-9, xs);                       // ln2/256, high part
    r = fma_(-kd, 0x1.abc9e3b39803fp-64, r);                             // low part
    const int j = ki & 255;
    const int k = ki >> 8;
    double c = fma_(r, 4.1666666666666664354e-02, 1.6666666666666665741e-01);  // 1/24, 1/6
    c = fma_(r, c, 0.5);
    const double q = fma_(r * r, c, r);
    const double t = T[j];
    return ldexp_(fma_(t, q, t), k);
}
BMM_HD double expw_(double x) { return expw_tab(x, exp256_table()); }

// ---------------------------------------------------------------- variates
// Standard normal by the Marsaglia polar method (log and sqrt only).
BMM_HD double rnorm_(Stream& st) {
    for (;;) {
        const U4 r = st.next();
        const double v1 = 2.0 * u01(r.x, r.y) - 1.0, v2 = 2.0 * u01(r.z, r.w) - 1.0;
        const double s = v1 * v1 + v2 * v2;
        if (s < 1.0 && s > 0.0) return v1 * sqrt_(div_(-2.0 * log_(s), s));
    }
}

// Gamma(shape, scale 1), Marsaglia & Tsang (2000); shape < 1 by the u^(1/shape) boost.
BMM_HD double rgamma_(double shape, Stream& st) {
    if (!(shape > 0.0)) return 0.0;  // R::rgamma(0, .) is 0
    double boost = 1.0;
    if (shape < 1.0) {
        const U4 r = st.next();
        boost = exp_(div_(log_(u01_open0(r.x, r.y)), shape));
        shape = shape + 1.0;
    }
    const double d = shape - 0.33333333333333331483;
    const double c = div_(1.0, sqrt_(9.0 * d));
    for (;;) {
        const double x = rnorm_(st);
        double v = 1.0 + c * x;
        if (v <= 0.0) continue;
        v = v * v * v;
        const U4 r = st.next();
        const double u = u01_open0(r.x, r.y);
        const double x2 = x * x;
        if (log_(u) < 0.5 * x2 + d - d * v + d * log_(v)) return d * v * boost;
    }
}

// Beta(p, q) = X / (X + Y), X ~ Gamma(p), Y ~ Gamma(q), two separate streams.
BMM_HD double rbeta_(double p, double q, Stream& sa, Stream& sb) {
    const double x = rgamma_(p, sa), y = rgamma_(q, sb);
    return div_(x, x + y);
}

// Escobar & West auxiliary-variable update of the concentration (utils.cpp:6-14).
BMM_HD double update_alpha_(double alpha_old, double a, double b, double N, int K,
                            uint64_t seed, uint32_t sweep) {
    Stream s0 = make_stream(seed, 0, sweep, kStreamAlphaEta);
    Stream s1 = make_stream(seed, 1, sweep, kStreamAlphaEta);
    const double eta = rbeta_(alpha_old + 1.0, N, s0, s1);
    const double b_eps = b - log_(eta);
    const double pi1 = a + (double)K - 1.0;
    const double pi2 = N * b_eps;
    const double pi = div_(pi1, pi1 + pi2);
    Stream g1 = make_stream(seed, 0, sweep, kStreamAlphaG1);
    Stream g2 = make_stream(seed, 0, sweep, kStreamAlphaG2);
    const double scale = div_(1.0, b_eps);
    const double ga = rgamma_(a + (double)K, g1) * scale;
    const double gb = rgamma_(a + (double)K - 1.0, g2) * scale;
    return pi * ga + (1.0 - pi) * gb;
}

// ---------------------------------------------------------------- conditional tables
// Per-feature log terms of the Beta-Bernoulli predictive of one cluster holding n
// observations with s1 of them 1 in this feature when the scored observation has x=1,
// s0 when it has x=0 (collapsed_gibbs.cpp:116-120; the "minus self" variant passes
// n-1, s-1 and s). `den` = log(beta+gamma+n) is hoisted by the caller.
BMM_HD double term_x1(double beta, int64_t s1, double den) { return log_(beta + (double)s1) - den; }
BMM_HD double term_x0(double gamma, int64_t n, int64_t s0, double den) {
    return log_((gamma + (double)n) - (double)s0) - den;
}

// One group-table entry: features [g*W, g*W+W) of one cluster under bit pattern m.
BMM_HD double group_entry(const double* e1, const double* e0, int g, int P, unsigned m, int W) {
    double t = 0.0;
    for (int j = 0; j < W; ++j) {
        const int d = g * W + j;
        if (d < P) t = t + (((m >> j) & 1u) ? e1[d] : e0[d]);
    }
    return t;
}

}  // namespace bmm
