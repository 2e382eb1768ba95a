/* bmmmcmc_shim.c -- .Call shim between R and the C ABI in include/bmm_mcmc.h.
 *
 * Plain C on R's own API (Rinternals.h); no Rcpp, no Armadillo.  It OWNS R_init_bmmmcmc and
 * registers every .Call symbol of the package:
 *
 *   the four samplers under exactly the names and arities of the reference's generated glue
 *   (/root/reference/src/RcppExports.cpp:137-146: collapsed 13, dp 12, gibbs_cpp 14,
 *   stickbreaking 14), so the package's unchanged R/RcppExports.R binds: the chain's seed is
 *   drawn inside, from R's RNG stream between GetRNGstate and PutRNGstate (set.seed() fixes the
 *   chain, as RNGScope does in the reference, RcppExports.cpp:14), batch = library default, device 0;
 *
 *   the same four with trailing (seed, batch, chains, devices) under new names, *_ex, which this
 *   build's R wrappers (../R/gibbs.R) call;
 *
 *   the three entry points this build does not touch (_bmmmcmc_rdirichlet_cpp 1, _bmmmcmc_my_lpsolve 1,
 *   _bmmmcmc_my_stephens_batch 2; RcppExports.cpp:140,142,143): stand-alone they raise a clear R
 *   error; with -DBMM_SHIM_FORWARD they are the package's own functions, linked in beside the shim
 *   (INTEGRATION.md section 1 shows the two Makevars lines that rename the generated file's
 *   R_init_bmmmcmc and its four sampler wrappers out of the way -- no source of the package is edited);
 *
 *   relabel = TRUE: with -DBMM_SHIM_FORWARD every sampler entry point hands such a call to relabel_glue.cpp
 *   (bmm_glue_*_relabel, C++ on Rcpp/RcppArmadillo beside the package's own stephens.cpp), which runs
 *   bmm_*_run_probs and feeds the package's unchanged Stephens code; stand-alone (no package objects to
 *   call) it is an R error that says so;
 *
 *   _bmmmcmc_set_progress(every): the reference prints "Sample j" at every sweep (collapsed_gibbs.cpp:85);
 *   here a run is silent unless this was called with every > 0 (or debug = TRUE: every sweep).
 *
 * `df` may be any numeric matrix: Rcpp's IntegerMatrix parameter coerces a REALSXP (what matrix(c(0, 1, ...))
 * is in R) or logical matrix (RcppExports.cpp:15,38,70,118), and so does this shim (coerceVector under
 * PROTECT; an INTSXP is used in place, no copy).
 *
 * R is not installed in the build image: tests/test_r_shim.py syntax-checks this file against a
 * test-only declaration header of the R API symbols it uses and compares the registration table
 * with the reference's.  Where R exists:
 *
 *   R CMD SHLIB -o bmmmcmc.so bmmmcmc_shim.c -I../../include -L../lib -lbmmmcmc_hip
 *
 * Ownership: inputs are R-owned and read-only; outputs are fresh R allocations (PROTECTed here)
 * whose raw pointers are handed to the C ABI, which fills them and retains nothing.  Errors: the
 * C ABI returns a status; this shim raises an R error with bmm_last_error() after everything on
 * the device has been released by the callee.  No R API call is made off the main thread
 * (bmm_multi_run's worker threads never see a SEXP).
 */
#include <R.h>
#include <Rinternals.h>
#include <R_ext/Rdynload.h>
#include <R_ext/Random.h>
#include <stdint.h>

#include "bmm_mcmc.h"

#define MAX_CHAINS 64

/* ---------------------------------------------------------------- small helpers */
/* df as an INTSXP matrix: itself, or a coerced copy the caller keeps PROTECTed (*nprot is bumped).  What
 * Rcpp::traits::input_parameter<IntegerMatrix> does for the reference (src/RcppExports.cpp:15). */
static SEXP as_int_matrix(SEXP df, int* nprot) {
    if (!isMatrix(df)) error("data must be a matrix with one observation per row");
    switch (TYPEOF(df)) {
        case INTSXP: return df;
        case REALSXP: case LGLSXP: case RAWSXP: {
            SEXP x = PROTECT(coerceVector(df, INTSXP));
            ++*nprot;
            return x;
        }
        default: error("data must be a numeric (0/1) matrix");
    }
    return R_NilValue; /* not reached */
}

static SEXP na_perm(int S, int K) { /* the reference returns uninitialised memory here */
    SEXP p = PROTECT(allocMatrix(INTSXP, S, K));
    for (R_xlen_t i = 0; i < (R_xlen_t)S * K; ++i) INTEGER(p)[i] = NA_INTEGER;
    UNPROTECT(1);
    return p;
}

static SEXP named_list(int n, const char** names, SEXP* vals) {
    SEXP out = PROTECT(allocVector(VECSXP, n)), nm = PROTECT(allocVector(STRSXP, n));
    for (int i = 0; i < n; ++i) {
        SET_VECTOR_ELT(out, i, vals[i]);
        SET_STRING_ELT(nm, i, mkChar(names[i]));
    }
    setAttrib(out, R_NamesSymbol, nm);
    UNPROTECT(2);
    return out;
}

static SEXP cube(int a, int b, int c) {
    SEXP dim = PROTECT(allocVector(INTSXP, 3));
    INTEGER(dim)[0] = a; INTEGER(dim)[1] = b; INTEGER(dim)[2] = c;
    SEXP x = PROTECT(allocArray(REALSXP, dim));
    UNPROTECT(2);
    return x;
}

/* relabel = TRUE runs the package's own Stephens code (src/stephens.cpp) on the probability matrices
 * bmm_*_run_probs produce: that needs the package's objects and relabel_glue.cpp beside this shim
 * (-DBMM_SHIM_FORWARD).  One chain per call. */
static int wants_relabel(SEXP relabel, int chains) {
    if (asLogical(relabel) != TRUE) return 0;
#ifndef BMM_SHIM_FORWARD
    error("relabel=TRUE feeds the package's own Stephens code (src/stephens.cpp) from bmm_*_run_probs: build "
          "this shim with -DBMM_SHIM_FORWARD and r-shim/relabel_glue.cpp into the package (INTEGRATION.md section 2)");
#endif
    if (chains != 1) error("relabel=TRUE is offered for one chain per call");
    return 1;
}
#ifdef BMM_SHIM_FORWARD
/* relabel_glue.cpp: the reference's argument lists (relabel and debug dropped) + seed, batch, device */
extern SEXP bmm_glue_collapsed_relabel(SEXP df, SEXP initialK, SEXP nsamples, SEXP K, SEXP alpha, SEXP beta, SEXP gamma,
                                       SEXP a, SEXP b, SEXP burnin, SEXP burnrelabel, SEXP seed, SEXP batch, SEXP device);
extern SEXP bmm_glue_dp_relabel(SEXP df, SEXP nsamples, SEXP alpha, SEXP beta, SEXP gamma, SEXP a, SEXP b, SEXP burnin,
                                SEXP burnrelabel, SEXP maxK, SEXP seed, SEXP batch, SEXP device);
extern SEXP bmm_glue_sb_relabel(SEXP df, SEXP initialPi, SEXP initialTheta, SEXP nsamples, SEXP maxK, SEXP alpha, SEXP beta,
                                SEXP gamma, SEXP a, SEXP b, SEXP burnin, SEXP burnrelabel, SEXP seed, SEXP device);
extern SEXP bmm_glue_full_relabel(SEXP df, SEXP initialPi, SEXP initialTheta, SEXP nsamples, SEXP K, SEXP alpha, SEXP beta,
                                  SEXP gamma, SEXP a, SEXP b, SEXP burnin, SEXP burnrelabel, SEXP seed, SEXP device);
static SEXP real_scalar(double v) { return ScalarReal(v); }
#endif

/* ---------------------------------------------------------------- progress */
static int g_progress_every = 0;
static int shim_progress(void* user, int sample, int nsamples, int k_used) {
    if (k_used >= 0) Rprintf("Sample %d\tK: %d\n", sample, k_used); /* collapsed_gibbs_dp.cpp:99 */
    else Rprintf("Sample %d\n", sample);                             /* collapsed_gibbs.cpp:85 */
    return 0;
}
/* the hook is on for the duration of one run: every `every` sweeps when _bmmmcmc_set_progress asked for it,
 * every sweep when debug = TRUE (the reference prints every sweep and, with debug, much more) */
static void progress_on(SEXP debug) {
    const int every = asLogical(debug) == TRUE ? 1 : g_progress_every;
    bmm_set_progress(every > 0 ? shim_progress : NULL, NULL, every);
}
static void progress_off(void) { bmm_set_progress(NULL, NULL, 0); }
SEXP _bmmmcmc_set_progress(SEXP every) {
    const int e = asInteger(every);
    if (e == NA_INTEGER || e < 0) error("every must be a non-negative whole number (0 = silent)");
    const int old = g_progress_every;
    g_progress_every = e;
    return ScalarInteger(old);
}

/* one 53-bit integer from R's stream: set.seed() determines the chain */
static uint64_t seed_from_r(void) {
    GetRNGstate();
    const double u = unif_rand();
    PutRNGstate();
    return (uint64_t)(u * 9007199254740992.0);
}

/* seed argument of the *_ex entry points: NULL -> R's stream; else a whole number in [0, 2^53] */
static uint64_t seed_arg(SEXP seed) {
    if (seed == R_NilValue) return seed_from_r();
    const double v = asReal(seed);
    if (ISNAN(v) || v < 0.0 || v > 9007199254740992.0 || v != (double)(uint64_t)v)
        error("seed must be a whole number in [0, 2^53] (or NULL to draw it from R's RNG)");
    return (uint64_t)v;
}

static int64_t batch_arg(SEXP batch) {
    if (batch == R_NilValue) return 0; /* library default */
    const double v = asReal(batch);
    if (ISNAN(v) || v < 0.0 || v > 9007199254740992.0) error("batch must be NULL or a non-negative number");
    return (int64_t)v;
}

static int chains_arg(SEXP chains) {
    const int n = chains == R_NilValue ? 1 : asInteger(chains);
    if (n == NA_INTEGER || n < 1 || n > MAX_CHAINS) error("chains must be between 1 and %d", MAX_CHAINS);
    return n;
}

/* devices: NULL (every chain on device 0) or one device per chain; returns dev (or NULL) */
static const int* devices_arg(SEXP devices, int chains, int* dev) {
    if (devices == R_NilValue) return NULL;
    SEXP d = PROTECT(coerceVector(devices, INTSXP));
    if (XLENGTH(d) != chains) { UNPROTECT(1); error("devices must name one device per chain"); }
    for (int c = 0; c < chains; ++c) dev[c] = INTEGER(d)[c];
    UNPROTECT(1);
    return dev;
}

typedef struct {
    int N, P, ns, K, bi, S;
} dims_t;

static dims_t dims_of(SEXP df, SEXP nsamples, SEXP K, SEXP burnin) {
    if (!isMatrix(df)) error("data must be a matrix with one observation per row");
    dims_t d;
    d.N = nrows(df); d.P = ncols(df); d.ns = asInteger(nsamples); d.K = asInteger(K); d.bi = asInteger(burnin);
    d.S = d.ns - d.bi;
    if (d.ns == NA_INTEGER || d.K == NA_INTEGER || d.bi == NA_INTEGER) error("nsamples, K and burnin must be numbers");
    if (d.S < 1) error("burnin must be smaller than nsamples");
    return d;
}

/* ---------------------------------------------------------------- the counting samplers */
/* sampler: BMM_SAMPLER_COLLAPSED (initialK: N x chains integer) or BMM_SAMPLER_DP (initialK unused) */
static SEXP counting_run(int sampler, SEXP df, SEXP initialK, dims_t d, SEXP alpha, SEXP beta, SEXP gamma,
                         SEXP a, SEXP b, uint64_t seed, int64_t batch, int chains, const int* dev, SEXP debug) {
    const int32_t* z0[MAX_CHAINS];
    int32_t* zs[MAX_CHAINS];
    double *ths[MAX_CHAINS], *als[MAX_CHAINS];
    int nprot = 0;
    SEXP x = as_int_matrix(df, &nprot);
    SEXP z0s = R_NilValue;
    if (sampler == BMM_SAMPLER_COLLAPSED) {
        z0s = PROTECT(coerceVector(initialK, INTSXP)); ++nprot;
        if (XLENGTH(z0s) != (R_xlen_t)d.N * chains) { UNPROTECT(nprot); error("initialK must hold one label per observation and chain"); }
        for (int c = 0; c < chains; ++c) z0[c] = INTEGER(z0s) + (R_xlen_t)c * d.N;
    }
    SEXP out = PROTECT(allocVector(VECSXP, chains)); ++nprot;
    const char* nm[] = {"alpha", "permutations", "z", "theta"};
    for (int c = 0; c < chains; ++c) {
        SEXP z = PROTECT(allocMatrix(INTSXP, d.S, d.N)), th = PROTECT(cube(d.K, d.P, d.S));
        SEXP al = PROTECT(allocMatrix(REALSXP, d.S, 1)), pm = PROTECT(na_perm(d.S, d.K));
        SEXP v[] = {al, pm, z, th};
        SET_VECTOR_ELT(out, c, named_list(4, nm, v));
        UNPROTECT(4);
        zs[c] = INTEGER(z); ths[c] = REAL(th); als[c] = REAL(al);
    }
    int rc;
    progress_on(debug);
    if (chains == 1 && sampler == BMM_SAMPLER_COLLAPSED)
        rc = bmm_collapsed_run(INTEGER(x), d.N, d.P, z0[0], d.ns, d.K, asReal(alpha), asReal(beta), asReal(gamma),
                               asReal(a), asReal(b), d.bi, batch, seed, dev ? dev[0] : 0, zs[0], ths[0], als[0]);
    else if (chains == 1)
        rc = bmm_dp_run(INTEGER(x), d.N, d.P, d.ns, asReal(alpha), asReal(beta), asReal(gamma), asReal(a), asReal(b),
                        d.bi, d.K, batch, seed, dev ? dev[0] : 0, zs[0], ths[0], als[0]);
    else
        rc = bmm_multi_run(sampler, chains, dev, INTEGER(x), d.N, d.P, sampler == BMM_SAMPLER_COLLAPSED ? z0 : NULL,
                           NULL, NULL, d.ns, d.K, asReal(alpha), asReal(beta), asReal(gamma), asReal(a), asReal(b),
                           d.bi, batch, seed, NULL, zs, ths, als);
    progress_off();
    if (rc) { UNPROTECT(nprot); error("%s", bmm_last_error()); }
    SEXP ret = chains == 1 ? VECTOR_ELT(out, 0) : out; /* one chain: the reference's list itself */
    UNPROTECT(nprot);
    return ret;
}

/* relabel = TRUE: the call goes to relabel_glue.cpp with the seed, batch and device this shim resolved */
#ifdef BMM_SHIM_FORWARD
static SEXP relabel_counting(int sampler, SEXP df, SEXP initialK, SEXP nsamples, SEXP K, SEXP alpha, SEXP beta, SEXP gamma,
                             SEXP a, SEXP b, SEXP burnin, SEXP burnrelabel, uint64_t seed, int64_t batch, int device) {
    SEXP s = PROTECT(real_scalar((double)seed)), bt = PROTECT(real_scalar((double)batch)), dv = PROTECT(ScalarInteger(device));
    SEXP r = sampler == BMM_SAMPLER_COLLAPSED
                 ? bmm_glue_collapsed_relabel(df, initialK, nsamples, K, alpha, beta, gamma, a, b, burnin, burnrelabel, s, bt, dv)
                 : bmm_glue_dp_relabel(df, nsamples, alpha, beta, gamma, a, b, burnin, burnrelabel, K, s, bt, dv);
    UNPROTECT(3);
    return r;
}
static SEXP relabel_explicit(int sampler, SEXP df, SEXP initialPi, SEXP initialTheta, SEXP nsamples, SEXP K, SEXP alpha,
                             SEXP beta, SEXP gamma, SEXP a, SEXP b, SEXP burnin, SEXP burnrelabel, uint64_t seed, int device) {
    SEXP s = PROTECT(real_scalar((double)seed)), dv = PROTECT(ScalarInteger(device));
    SEXP r = (sampler == BMM_SAMPLER_FULL ? bmm_glue_full_relabel : bmm_glue_sb_relabel)(
        df, initialPi, initialTheta, nsamples, K, alpha, beta, gamma, a, b, burnin, burnrelabel, s, dv);
    UNPROTECT(2);
    return r;
}
#else
#define relabel_counting(...) R_NilValue /* wants_relabel() has raised the error */
#define relabel_explicit(...) R_NilValue
#endif

/* collapsed_gibbs_cpp(df, initialK, nsamples, K, alpha, beta, gamma, a, b, burnin, relabel, burnrelabel, debug)
 * -- src/RcppExports.cpp:10, 13 arguments */
SEXP _bmmmcmc_collapsed_gibbs_cpp(SEXP df, SEXP initialK, SEXP nsamples, SEXP K, SEXP alpha, SEXP beta,
                                  SEXP gamma, SEXP a, SEXP b, SEXP burnin, SEXP relabel, SEXP burnrelabel,
                                  SEXP debug) {
    if (wants_relabel(relabel, 1))
        return relabel_counting(BMM_SAMPLER_COLLAPSED, df, initialK, nsamples, K, alpha, beta, gamma, a, b, burnin,
                                burnrelabel, seed_from_r(), 0, 0);
    return counting_run(BMM_SAMPLER_COLLAPSED, df, initialK, dims_of(df, nsamples, K, burnin), alpha, beta, gamma, a, b,
                        seed_from_r(), 0, 1, NULL, debug);
}
SEXP _bmmmcmc_collapsed_gibbs_ex(SEXP df, SEXP initialK, SEXP nsamples, SEXP K, SEXP alpha, SEXP beta, SEXP gamma,
                                 SEXP a, SEXP b, SEXP burnin, SEXP relabel, SEXP burnrelabel, SEXP debug,
                                 SEXP seed, SEXP batch, SEXP chains, SEXP devices) {
    int dev[MAX_CHAINS];
    const int n = chains_arg(chains);
    const int* dv = devices_arg(devices, n, dev);
    if (wants_relabel(relabel, n))
        return relabel_counting(BMM_SAMPLER_COLLAPSED, df, initialK, nsamples, K, alpha, beta, gamma, a, b, burnin,
                                burnrelabel, seed_arg(seed), batch_arg(batch), dv ? dv[0] : 0);
    return counting_run(BMM_SAMPLER_COLLAPSED, df, initialK, dims_of(df, nsamples, K, burnin), alpha, beta, gamma, a, b,
                        seed_arg(seed), batch_arg(batch), n, dv, debug);
}

/* collapsed_gibbs_dp_cpp(df, nsamples, alpha, beta, gamma, a, b, burnin, relabel, burnrelabel, maxK, debug)
 * -- src/RcppExports.cpp:33, 12 arguments */
SEXP _bmmmcmc_collapsed_gibbs_dp_cpp(SEXP df, SEXP nsamples, SEXP alpha, SEXP beta, SEXP gamma, SEXP a,
                                     SEXP b, SEXP burnin, SEXP relabel, SEXP burnrelabel, SEXP maxK,
                                     SEXP debug) {
    if (wants_relabel(relabel, 1))
        return relabel_counting(BMM_SAMPLER_DP, df, R_NilValue, nsamples, maxK, alpha, beta, gamma, a, b, burnin,
                                burnrelabel, seed_from_r(), 0, 0);
    return counting_run(BMM_SAMPLER_DP, df, R_NilValue, dims_of(df, nsamples, maxK, burnin), alpha, beta, gamma, a, b,
                        seed_from_r(), 0, 1, NULL, debug);
}
SEXP _bmmmcmc_collapsed_gibbs_dp_ex(SEXP df, SEXP nsamples, SEXP alpha, SEXP beta, SEXP gamma, SEXP a, SEXP b,
                                    SEXP burnin, SEXP relabel, SEXP burnrelabel, SEXP maxK, SEXP debug,
                                    SEXP seed, SEXP batch, SEXP chains, SEXP devices) {
    int dev[MAX_CHAINS];
    const int n = chains_arg(chains);
    const int* dv = devices_arg(devices, n, dev);
    if (wants_relabel(relabel, n))
        return relabel_counting(BMM_SAMPLER_DP, df, R_NilValue, nsamples, maxK, alpha, beta, gamma, a, b, burnin,
                                burnrelabel, seed_arg(seed), batch_arg(batch), dv ? dv[0] : 0);
    return counting_run(BMM_SAMPLER_DP, df, R_NilValue, dims_of(df, nsamples, maxK, burnin), alpha, beta, gamma, a, b,
                        seed_arg(seed), batch_arg(batch), n, dv, debug);
}

/* ---------------------------------------------------------------- the explicit-parameter samplers */
/* initialPi: K x chains, initialTheta: (K*P) x chains (each column a K x P matrix, column-major) */
static SEXP explicit_run(int sampler, SEXP df, SEXP initialPi, SEXP initialTheta, dims_t d, SEXP alpha, SEXP beta,
                         SEXP gamma, SEXP a, SEXP b, uint64_t seed, int chains, const int* dev, SEXP debug) {
    const double *pi0[MAX_CHAINS], *th0[MAX_CHAINS];
    int32_t* zs[MAX_CHAINS];
    double *ths[MAX_CHAINS], *als[MAX_CHAINS], *pis[MAX_CHAINS];
    int nprot = 0;
    SEXP x = as_int_matrix(df, &nprot);
    SEXP p0 = PROTECT(coerceVector(initialPi, REALSXP)), t0 = PROTECT(coerceVector(initialTheta, REALSXP));
    nprot += 2;
    const R_xlen_t kp = (R_xlen_t)d.K * d.P;
    if (XLENGTH(p0) != (R_xlen_t)d.K * chains || XLENGTH(t0) != kp * chains) {
        UNPROTECT(nprot);
        error("initialPi/initialTheta have the wrong size");
    }
    for (int c = 0; c < chains; ++c) { pi0[c] = REAL(p0) + (R_xlen_t)c * d.K; th0[c] = REAL(t0) + c * kp; }
    SEXP out = PROTECT(allocVector(VECSXP, chains)); ++nprot;
    const char* nm[] = {"pi", "alpha", "permutations", "z", "theta"};
    for (int c = 0; c < chains; ++c) {
        SEXP z = PROTECT(allocMatrix(INTSXP, d.S, d.N)), th = PROTECT(cube(d.K, d.P, d.S));
        SEXP al = PROTECT(allocMatrix(REALSXP, d.S, 1)), pi = PROTECT(allocMatrix(REALSXP, d.S, d.K));
        SEXP pm = PROTECT(na_perm(d.S, d.K));
        SEXP v[] = {pi, al, pm, z, th};
        SET_VECTOR_ELT(out, c, named_list(5, nm, v));
        UNPROTECT(5);
        zs[c] = INTEGER(z); ths[c] = REAL(th); als[c] = REAL(al); pis[c] = REAL(pi);
    }
    int rc;
    progress_on(debug);
    if (chains == 1)
        rc = (sampler == BMM_SAMPLER_FULL ? bmm_full_run : bmm_sb_run)(
            INTEGER(x), d.N, d.P, pi0[0], th0[0], d.ns, d.K, asReal(alpha), asReal(beta), asReal(gamma), asReal(a),
            asReal(b), d.bi, seed, dev ? dev[0] : 0, pis[0], zs[0], ths[0], als[0]);
    else
        rc = bmm_multi_run(sampler, chains, dev, INTEGER(x), d.N, d.P, NULL, pi0, th0, d.ns, d.K, asReal(alpha),
                           asReal(beta), asReal(gamma), asReal(a), asReal(b), d.bi, 0, seed, pis, zs, ths, als);
    progress_off();
    if (rc) { UNPROTECT(nprot); error("%s", bmm_last_error()); }
    SEXP ret = chains == 1 ? VECTOR_ELT(out, 0) : out;
    UNPROTECT(nprot);
    return ret;
}

/* gibbs_stickbreaking_cpp(df, initialPi, initialTheta, nsamples, maxK, alpha, beta, gamma, a, b, burnin,
 * relabel, burnrelabel, debug) -- src/RcppExports.cpp:113, 14 arguments */
SEXP _bmmmcmc_gibbs_stickbreaking_cpp(SEXP df, SEXP initialPi, SEXP initialTheta, SEXP nsamples, SEXP maxK,
                                      SEXP alpha, SEXP beta, SEXP gamma, SEXP a, SEXP b, SEXP burnin,
                                      SEXP relabel, SEXP burnrelabel, SEXP debug) {
    if (wants_relabel(relabel, 1))
        return relabel_explicit(BMM_SAMPLER_SB, df, initialPi, initialTheta, nsamples, maxK, alpha, beta, gamma, a, b,
                                burnin, burnrelabel, seed_from_r(), 0);
    return explicit_run(BMM_SAMPLER_SB, df, initialPi, initialTheta, dims_of(df, nsamples, maxK, burnin), alpha, beta,
                        gamma, a, b, seed_from_r(), 1, NULL, debug);
}
SEXP _bmmmcmc_gibbs_stickbreaking_ex(SEXP df, SEXP initialPi, SEXP initialTheta, SEXP nsamples, SEXP maxK,
                                     SEXP alpha, SEXP beta, SEXP gamma, SEXP a, SEXP b, SEXP burnin, SEXP relabel,
                                     SEXP burnrelabel, SEXP debug, SEXP seed, SEXP chains, SEXP devices) {
    int dev[MAX_CHAINS];
    const int n = chains_arg(chains);
    const int* dv = devices_arg(devices, n, dev);
    if (wants_relabel(relabel, n))
        return relabel_explicit(BMM_SAMPLER_SB, df, initialPi, initialTheta, nsamples, maxK, alpha, beta, gamma, a, b,
                                burnin, burnrelabel, seed_arg(seed), dv ? dv[0] : 0);
    return explicit_run(BMM_SAMPLER_SB, df, initialPi, initialTheta, dims_of(df, nsamples, maxK, burnin), alpha, beta,
                        gamma, a, b, seed_arg(seed), n, dv, debug);
}

/* gibbs_cpp(df, initialPi, initialTheta, nsamples, K, alpha, beta, gamma, a, b, burnin, relabel, burnrelabel,
 * debug) -- src/RcppExports.cpp:65, 14 arguments */
SEXP _bmmmcmc_gibbs_cpp(SEXP df, SEXP initialPi, SEXP initialTheta, SEXP nsamples, SEXP K, SEXP alpha,
                        SEXP beta, SEXP gamma, SEXP a, SEXP b, SEXP burnin, SEXP relabel, SEXP burnrelabel,
                        SEXP debug) {
    if (wants_relabel(relabel, 1))
        return relabel_explicit(BMM_SAMPLER_FULL, df, initialPi, initialTheta, nsamples, K, alpha, beta, gamma, a, b,
                                burnin, burnrelabel, seed_from_r(), 0);
    return explicit_run(BMM_SAMPLER_FULL, df, initialPi, initialTheta, dims_of(df, nsamples, K, burnin), alpha, beta,
                        gamma, a, b, seed_from_r(), 1, NULL, debug);
}
SEXP _bmmmcmc_gibbs_ex(SEXP df, SEXP initialPi, SEXP initialTheta, SEXP nsamples, SEXP K, SEXP alpha, SEXP beta,
                       SEXP gamma, SEXP a, SEXP b, SEXP burnin, SEXP relabel, SEXP burnrelabel, SEXP debug,
                       SEXP seed, SEXP chains, SEXP devices) {
    int dev[MAX_CHAINS];
    const int n = chains_arg(chains);
    const int* dv = devices_arg(devices, n, dev);
    if (wants_relabel(relabel, n))
        return relabel_explicit(BMM_SAMPLER_FULL, df, initialPi, initialTheta, nsamples, K, alpha, beta, gamma, a, b,
                                burnin, burnrelabel, seed_arg(seed), dv ? dv[0] : 0);
    return explicit_run(BMM_SAMPLER_FULL, df, initialPi, initialTheta, dims_of(df, nsamples, K, burnin), alpha, beta,
                        gamma, a, b, seed_arg(seed), n, dv, debug);
}

/* ---------------------------------------------------------------- the entry points this build does not touch */
#ifdef BMM_SHIM_FORWARD
/* the package's own functions (generated glue over src/full_gibbs.cpp:10, src/my_lpsolve.cpp:6,
 * src/stephens.cpp:6), linked in beside the shim: registered as they are */
extern SEXP _bmmmcmc_rdirichlet_cpp(SEXP alpha_m);
extern SEXP _bmmmcmc_my_lpsolve(SEXP x);
extern SEXP _bmmmcmc_my_stephens_batch(SEXP p, SEXP debug);
#else
static void stays_in_reference(const char* what) {
    error("%s is host code of the bmmmcmc package that this GPU build leaves untouched; build the shim "
          "with -DBMM_SHIM_FORWARD beside the package's own objects to keep it callable (INTEGRATION.md)", what);
}
SEXP _bmmmcmc_rdirichlet_cpp(SEXP alpha_m) { stays_in_reference("rdirichlet_cpp"); return R_NilValue; }
SEXP _bmmmcmc_my_lpsolve(SEXP x) { stays_in_reference("my_lpsolve"); return R_NilValue; }
SEXP _bmmmcmc_my_stephens_batch(SEXP p, SEXP debug) { stays_in_reference("my_stephens_batch"); return R_NilValue; }
#endif

/* The first seven rows are the reference's table, name for name and arity for arity
 * (src/RcppExports.cpp:137-146); tests/test_r_shim.py holds them to it. */
static const R_CallMethodDef CallEntries[] = {
    {"_bmmmcmc_collapsed_gibbs_cpp", (DL_FUNC)&_bmmmcmc_collapsed_gibbs_cpp, 13},
    {"_bmmmcmc_collapsed_gibbs_dp_cpp", (DL_FUNC)&_bmmmcmc_collapsed_gibbs_dp_cpp, 12},
    {"_bmmmcmc_rdirichlet_cpp", (DL_FUNC)&_bmmmcmc_rdirichlet_cpp, 1},
    {"_bmmmcmc_gibbs_cpp", (DL_FUNC)&_bmmmcmc_gibbs_cpp, 14},
    {"_bmmmcmc_my_lpsolve", (DL_FUNC)&_bmmmcmc_my_lpsolve, 1},
    {"_bmmmcmc_my_stephens_batch", (DL_FUNC)&_bmmmcmc_my_stephens_batch, 2},
    {"_bmmmcmc_gibbs_stickbreaking_cpp", (DL_FUNC)&_bmmmcmc_gibbs_stickbreaking_cpp, 14},
    {"_bmmmcmc_collapsed_gibbs_ex", (DL_FUNC)&_bmmmcmc_collapsed_gibbs_ex, 17},
    {"_bmmmcmc_collapsed_gibbs_dp_ex", (DL_FUNC)&_bmmmcmc_collapsed_gibbs_dp_ex, 16},
    {"_bmmmcmc_gibbs_ex", (DL_FUNC)&_bmmmcmc_gibbs_ex, 17},
    {"_bmmmcmc_gibbs_stickbreaking_ex", (DL_FUNC)&_bmmmcmc_gibbs_stickbreaking_ex, 17},
    {"_bmmmcmc_set_progress", (DL_FUNC)&_bmmmcmc_set_progress, 1},
    {NULL, NULL, 0}};

void R_init_bmmmcmc(DllInfo* dll) {
    R_registerRoutines(dll, NULL, CallEntries, NULL, NULL);
    R_useDynamicSymbols(dll, FALSE);
}
