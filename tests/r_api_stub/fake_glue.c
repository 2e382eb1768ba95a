/* fake_glue.c -- TEST-ONLY stand-ins for what bmmmcmc_shim.c links against under -DBMM_SHIM_FORWARD: this
 * repo's own relabel_glue.cpp (C++ on RcppArmadillo, not compilable in an image without R) and the three
 * entry points of the package the shim forwards untouched.  Each records which function was reached and with
 * what seed / batch / device the shim resolved, as a named list, so that tests/test_r_shim_exec.py can run the
 * shim's relabel = TRUE dispatch.  Nothing of the reference is restated here. */
#include "Rinternals.h"

static SEXP record(const char* who, SEXP a, SEXP b, SEXP c) {
    SEXP out = PROTECT(Rf_allocVector(VECSXP, 4)), nm = PROTECT(Rf_allocVector(STRSXP, 4));
    const char* names[] = {"who", "seed", "batch", "device"};
    SEXP w = PROTECT(Rf_allocVector(STRSXP, 1));
    SET_STRING_ELT(w, 0, Rf_mkChar(who));
    SET_VECTOR_ELT(out, 0, w);
    SET_VECTOR_ELT(out, 1, a);
    SET_VECTOR_ELT(out, 2, b);
    SET_VECTOR_ELT(out, 3, c);
    for (int i = 0; i < 4; ++i) SET_STRING_ELT(nm, i, Rf_mkChar(names[i]));
    Rf_setAttrib(out, R_NamesSymbol, nm);
    UNPROTECT(3);
    return out;
}
SEXP bmm_glue_collapsed_relabel(SEXP df, SEXP initialK, SEXP nsamples, SEXP K, SEXP alpha, SEXP beta, SEXP gamma, SEXP a, SEXP b,
                                SEXP burnin, SEXP burnrelabel, SEXP seed, SEXP batch, SEXP device) {
    return record("collapsed", seed, batch, device);
}
SEXP bmm_glue_dp_relabel(SEXP df, SEXP nsamples, SEXP alpha, SEXP beta, SEXP gamma, SEXP a, SEXP b, SEXP burnin, SEXP burnrelabel,
                         SEXP maxK, SEXP seed, SEXP batch, SEXP device) {
    return record("dp", seed, batch, device);
}
SEXP bmm_glue_sb_relabel(SEXP df, SEXP initialPi, SEXP initialTheta, SEXP nsamples, SEXP maxK, SEXP alpha, SEXP beta, SEXP gamma,
                         SEXP a, SEXP b, SEXP burnin, SEXP burnrelabel, SEXP seed, SEXP device) {
    return record("sb", seed, burnrelabel, device);
}
SEXP bmm_glue_full_relabel(SEXP df, SEXP initialPi, SEXP initialTheta, SEXP nsamples, SEXP K, SEXP alpha, SEXP beta, SEXP gamma,
                           SEXP a, SEXP b, SEXP burnin, SEXP burnrelabel, SEXP seed, SEXP device) {
    return record("full", seed, burnrelabel, device);
}
SEXP _bmmmcmc_rdirichlet_cpp(SEXP alpha_m) { return record("rdirichlet_cpp", alpha_m, R_NilValue, R_NilValue); }
SEXP _bmmmcmc_my_lpsolve(SEXP x) { return record("my_lpsolve", x, R_NilValue, R_NilValue); }
SEXP _bmmmcmc_my_stephens_batch(SEXP p, SEXP debug) { return record("my_stephens_batch", p, debug, R_NilValue); }
