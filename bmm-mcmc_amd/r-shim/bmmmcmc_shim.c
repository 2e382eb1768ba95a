/* bmmmcmc_shim.c -- .Call shim between R and the C ABI in include/bmm_mcmc.h.
 *
 * Plain C on R's own API (Rinternals.h); no Rcpp, no Armadillo.  It registers the
 * three hot entry points under exactly the names and arities the reference's
 * generated glue uses (/root/reference/src/RcppExports.cpp:137-146), with two extra
 * trailing arguments (seed, batch), so the package's R/RcppExports.R-style callers
 * keep working.  R is not installed in the build image, so this file is compiled
 * only where `R CMD SHLIB` exists:
 *
 *   R CMD SHLIB -o bmmmcmc.so bmmmcmc_shim.c -I../../include -L../lib -lbmmmcmc_hip
 *
 * Ownership: inputs are R-owned and read-only; outputs are fresh R allocations
 * (PROTECTed here) whose raw pointers are handed to the C ABI, which fills them and
 * retains nothing.  Errors: the C ABI returns a status; this shim raises an R error
 * with bmm_last_error() after everything on the device has been released by the
 * callee.  No R API call is made off the main thread.
 */
#include <R.h>
#include <Rinternals.h>
#include <R_ext/Rdynload.h>
#include <stdint.h>

#include "bmm_mcmc.h"

static void need_int_matrix(SEXP df) {
    if (TYPEOF(df) != INTSXP || !isMatrix(df)) error("data must be an integer matrix");
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

static void no_relabel(SEXP relabel) {
    if (asLogical(relabel) == TRUE)
        error("relabel=TRUE stays on the reference's host path (stephens.cpp / my_lpsolve.cpp)");
}

/* collapsed_gibbs_cpp(df, initialK, nsamples, K, alpha, beta, gamma, a, b, burnin, relabel,
 *                     burnrelabel, debug [, seed, batch]) */
SEXP _bmmmcmc_collapsed_gibbs_cpp(SEXP df, SEXP initialK, SEXP nsamples, SEXP K, SEXP alpha, SEXP beta,
                                  SEXP gamma, SEXP a, SEXP b, SEXP burnin, SEXP relabel, SEXP burnrelabel,
                                  SEXP debug, SEXP seed, SEXP batch) {
    need_int_matrix(df);
    no_relabel(relabel);
    const int N = nrows(df), P = ncols(df), ns = asInteger(nsamples), k = asInteger(K), bi = asInteger(burnin);
    const int S = ns - bi;
    if (S < 1) error("burnin must be smaller than nsamples");
    SEXP z0 = PROTECT(coerceVector(initialK, INTSXP));
    SEXP z = PROTECT(allocMatrix(INTSXP, S, N)), th = PROTECT(cube(k, P, S)), al = PROTECT(allocMatrix(REALSXP, S, 1));
    const int rc = bmm_collapsed_run(INTEGER(df), N, P, INTEGER(z0), ns, k, asReal(alpha), asReal(beta),
                                     asReal(gamma), asReal(a), asReal(b), bi, (int64_t)asReal(batch),
                                     (uint64_t)asReal(seed), 0, INTEGER(z), REAL(th), REAL(al));
    if (rc) { UNPROTECT(4); error("%s", bmm_last_error()); }
    SEXP pm = PROTECT(na_perm(S, k));
    const char* nm[] = {"alpha", "permutations", "z", "theta"};
    SEXP v[] = {al, pm, z, th};
    SEXP out = named_list(4, nm, v);
    UNPROTECT(5);
    return out;
}

/* collapsed_gibbs_dp_cpp(df, nsamples, alpha, beta, gamma, a, b, burnin, relabel, burnrelabel,
 *                        maxK, debug [, seed, batch]) */
SEXP _bmmmcmc_collapsed_gibbs_dp_cpp(SEXP df, SEXP nsamples, SEXP alpha, SEXP beta, SEXP gamma, SEXP a,
                                     SEXP b, SEXP burnin, SEXP relabel, SEXP burnrelabel, SEXP maxK,
                                     SEXP debug, SEXP seed, SEXP batch) {
    need_int_matrix(df);
    no_relabel(relabel);
    const int N = nrows(df), P = ncols(df), ns = asInteger(nsamples), k = asInteger(maxK), bi = asInteger(burnin);
    const int S = ns - bi;
    if (S < 1) error("burnin must be smaller than nsamples");
    SEXP z = PROTECT(allocMatrix(INTSXP, S, N)), th = PROTECT(cube(k, P, S)), al = PROTECT(allocMatrix(REALSXP, S, 1));
    const int rc = bmm_dp_run(INTEGER(df), N, P, ns, asReal(alpha), asReal(beta), asReal(gamma), asReal(a),
                              asReal(b), bi, k, (int64_t)asReal(batch), (uint64_t)asReal(seed), 0, INTEGER(z),
                              REAL(th), REAL(al));
    if (rc) { UNPROTECT(3); error("%s", bmm_last_error()); }
    SEXP pm = PROTECT(na_perm(S, k));
    const char* nm[] = {"alpha", "permutations", "z", "theta"};
    SEXP v[] = {al, pm, z, th};
    SEXP out = named_list(4, nm, v);
    UNPROTECT(4);
    return out;
}

/* gibbs_stickbreaking_cpp / gibbs_cpp (df, initialPi, initialTheta, nsamples, maxK or K, alpha, beta,
 *                         gamma, a, b, burnin, relabel, burnrelabel, debug [, seed]) */
static SEXP explicit_params_run(int full, SEXP df, SEXP initialPi, SEXP initialTheta, SEXP nsamples, SEXP maxK,
                                SEXP alpha, SEXP beta, SEXP gamma, SEXP a, SEXP b, SEXP burnin,
                                SEXP relabel, SEXP burnrelabel, SEXP debug, SEXP seed) {
    need_int_matrix(df);
    no_relabel(relabel);
    const int N = nrows(df), P = ncols(df), ns = asInteger(nsamples), k = asInteger(maxK), bi = asInteger(burnin);
    const int S = ns - bi;
    if (S < 1) error("burnin must be smaller than nsamples");
    SEXP pi0 = PROTECT(coerceVector(initialPi, REALSXP)), th0 = PROTECT(coerceVector(initialTheta, REALSXP));
    if (XLENGTH(pi0) != k || XLENGTH(th0) != (R_xlen_t)k * P) { UNPROTECT(2); error("initialPi/initialTheta have the wrong size"); }
    SEXP z = PROTECT(allocMatrix(INTSXP, S, N)), th = PROTECT(cube(k, P, S)), al = PROTECT(allocMatrix(REALSXP, S, 1));
    SEXP pi = PROTECT(allocMatrix(REALSXP, S, k));
    const int rc = (full ? bmm_full_run : bmm_sb_run)(INTEGER(df), N, P, REAL(pi0), REAL(th0), ns, k, asReal(alpha),
                                                      asReal(beta), asReal(gamma), asReal(a), asReal(b), bi,
                                                      (uint64_t)asReal(seed), 0, REAL(pi), INTEGER(z), REAL(th),
                                                      REAL(al));
    if (rc) { UNPROTECT(6); error("%s", bmm_last_error()); }
    SEXP pm = PROTECT(na_perm(S, k));
    const char* nm[] = {"pi", "alpha", "permutations", "z", "theta"};
    SEXP v[] = {pi, al, pm, z, th};
    SEXP out = named_list(5, nm, v);
    UNPROTECT(7);
    return out;
}

SEXP _bmmmcmc_gibbs_stickbreaking_cpp(SEXP df, SEXP initialPi, SEXP initialTheta, SEXP nsamples, SEXP maxK,
                                      SEXP alpha, SEXP beta, SEXP gamma, SEXP a, SEXP b, SEXP burnin,
                                      SEXP relabel, SEXP burnrelabel, SEXP debug, SEXP seed) {
    return explicit_params_run(0, df, initialPi, initialTheta, nsamples, maxK, alpha, beta, gamma, a, b, burnin,
                               relabel, burnrelabel, debug, seed);
}
/* gibbs_cpp, src/full_gibbs.cpp:32 */
SEXP _bmmmcmc_gibbs_cpp(SEXP df, SEXP initialPi, SEXP initialTheta, SEXP nsamples, SEXP K, SEXP alpha,
                        SEXP beta, SEXP gamma, SEXP a, SEXP b, SEXP burnin, SEXP relabel, SEXP burnrelabel,
                        SEXP debug, SEXP seed) {
    return explicit_params_run(1, df, initialPi, initialTheta, nsamples, K, alpha, beta, gamma, a, b, burnin,
                               relabel, burnrelabel, debug, seed);
}

static const R_CallMethodDef CallEntries[] = {
    {"_bmmmcmc_collapsed_gibbs_cpp", (DL_FUNC)&_bmmmcmc_collapsed_gibbs_cpp, 15},
    {"_bmmmcmc_collapsed_gibbs_dp_cpp", (DL_FUNC)&_bmmmcmc_collapsed_gibbs_dp_cpp, 14},
    {"_bmmmcmc_gibbs_stickbreaking_cpp", (DL_FUNC)&_bmmmcmc_gibbs_stickbreaking_cpp, 15},
    {"_bmmmcmc_gibbs_cpp", (DL_FUNC)&_bmmmcmc_gibbs_cpp, 15},
    {NULL, NULL, 0}};

void R_init_bmmmcmc(DllInfo* dll) {
    R_registerRoutines(dll, NULL, CallEntries, NULL, NULL);
    R_useDynamicSymbols(dll, FALSE);
}
