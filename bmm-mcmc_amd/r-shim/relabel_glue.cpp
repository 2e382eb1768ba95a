// relabel_glue.cpp -- relabel = TRUE end to end for all four samplers: the package's own, unchanged Stephens
// code (src/stephens.cpp: my_stephens_batch, my_stephens_online; src/my_lpsolve.cpp underneath) fed from the
// device by bmm_{collapsed,dp,sb,full}_run_probs (include/bmm_mcmc.h, "relabel = TRUE").
//
// C++ on Rcpp/RcppArmadillo like the files it stands in for, because the functions it calls take arma::cube /
// arma::mat.  It belongs in the package's src/ beside stephens.cpp and is reached from bmmmcmc_shim.c built with
// -DBMM_SHIM_FORWARD: the shim's sampler entry points hand a relabel = TRUE call to the four extern "C"
// functions below (plain SEXP in and out, so the C shim needs no Rcpp).  Neither R nor RcppArmadillo is in the
// build image, so this file is NOT compiled or tested there; the data path it relies on is
// (tests/test_gpu_relabel.py drives the same hooks from Python, with the same bookkeeping, and checks the
// matrices against the oracle's conditionals).
//
// It restates only the reference's bookkeeping around the two Stephens calls:
//   collapsed  src/collapsed_gibbs.cpp:187-201, 215-217, 229-243
//   dp         src/collapsed_gibbs_dp.cpp:247-261, 277-279, 285-299
//   sb         src/stickbreaking.cpp:145-159, 225-227, 238-254
//   full       src/full_gibbs.cpp:162-176, 221-223, 233-249
// i.e. Q from my_stephens_batch at j = burnin - 1; per kept sweep my_stephens_online -> permutation,
// z relabelled (perm(z - 1) + 1), theta rows moved to perm(k); the list gains z_original / theta_original.
//
// [[Rcpp::depends(RcppArmadillo)]]
#include <RcppArmadillo.h>

#include <functional>

#include "bmm_mcmc.h"
#include "stephens.h"  // the package's: my_stephens_batch, my_stephens_online

namespace {

struct RelabelState {
    int N, K, burnin;
    arma::cube probs_out;         // N x K x burnrelabel, filled by the library before batch_done
    arma::mat Q;
    arma::Mat<int> permutations;  // S x K
};

int batch_done(void* user, int /*j*/, const double* /*probs == probs_out.memptr()*/) {
    RelabelState* s = static_cast<RelabelState*>(user);
    try {
        Rcpp::Rcout << "Running Stephens Batch relabelling to identify initial Q values\n";  // collapsed_gibbs_dp.cpp:249
        s->Q = my_stephens_batch(s->probs_out, false);
        return 0;
    } catch (...) { return 1; }
}

int on_sample(void* user, int j, const double* probs) {
    RelabelState* s = static_cast<RelabelState*>(user);
    try {
        // a view of the library's N x K column-major matrix: no copy (copy_aux_mem = false)
        const arma::mat probs_sample(const_cast<double*>(probs), s->N, s->K, false, true);
        std::pair<arma::Row<int>, arma::mat> out = my_stephens_online(s->Q, probs_sample, j, false);
        s->Q = out.second;
        s->permutations.row(j - s->burnin) = out.first;
        return 0;
    } catch (...) { return 1; }
}

// run(hooks, z, theta, alpha, pi) calls one bmm_*_run_probs and returns its status; pi is used (and returned
// first in the list, stickbreaking.cpp:240) only when with_pi
typedef std::function<int(const bmm_relabel_hooks*, int32_t*, double*, double*, double*)> run_fn;

Rcpp::List relabelled(int N, int P, int K, int nsamples, int burnin, int burnrelabel, bool with_pi, const run_fn& run) {
    const int S = nsamples - burnin;
    if (S < 1) Rcpp::stop("burnin must be smaller than nsamples");
    if (burnrelabel < 1) Rcpp::stop("burnrelabel must be at least 1 with relabel = TRUE");
    RelabelState st{N, K, burnin, arma::cube(N, K, burnrelabel, arma::fill::zeros), arma::mat(), arma::Mat<int>(S, K)};
    bmm_relabel_hooks hooks{burnrelabel, st.probs_out.memptr(), batch_done, on_sample, &st};
    arma::Mat<int> z(S, N);
    arma::cube theta(K, P, S);
    arma::vec alpha_out(S);
    arma::mat pi(with_pi ? S : 0, with_pi ? K : 0);
    const int rc = run(&hooks, z.memptr(), theta.memptr(), alpha_out.memptr(), with_pi ? pi.memptr() : nullptr);
    if (rc) Rcpp::stop(bmm_last_error());
    arma::Mat<int> z_relabelled(S, N);
    arma::cube thetas_relabelled(K, P, S, arma::fill::zeros);
    for (int s = 0; s < S; ++s) {
        const arma::Row<int> perm = st.permutations.row(s);
        for (int i = 0; i < N; ++i) z_relabelled(s, i) = perm(z(s, i) - 1) + 1;
        for (int k = 0; k < K; ++k)
            for (int d = 0; d < P; ++d) thetas_relabelled(perm(k), d, s) = theta(k, d, s);
    }
    Rcpp::List ret;
    if (with_pi) ret["pi"] = pi;
    ret["alpha"] = alpha_out;
    ret["permutations"] = st.permutations;
    ret["z"] = z_relabelled;
    ret["theta"] = thetas_relabelled;
    ret["z_original"] = z;
    ret["theta_original"] = theta;
    return ret;
}

uint64_t seed_of(SEXP s) { return (uint64_t)Rcpp::as<double>(s); }

}  // namespace

// Entry points for bmmmcmc_shim.c: the reference's argument lists without relabel / debug, plus what the shim
// resolved (seed as a double below 2^53, batch, device).  BEGIN_RCPP / END_RCPP turn C++ exceptions into R errors.
extern "C" SEXP bmm_glue_collapsed_relabel(SEXP df_, SEXP initialK_, SEXP nsamples_, SEXP K_, SEXP alpha_, SEXP beta_,
                                           SEXP gamma_, SEXP a_, SEXP b_, SEXP burnin_, SEXP burnrelabel_, SEXP seed_,
                                           SEXP batch_, SEXP device_) {
    BEGIN_RCPP
    const Rcpp::IntegerMatrix df(df_);  // coerces a numeric matrix, as the generated glue does (RcppExports.cpp:15)
    const Rcpp::IntegerVector initialK(initialK_);
    const int nsamples = Rcpp::as<int>(nsamples_), K = Rcpp::as<int>(K_), burnin = Rcpp::as<int>(burnin_);
    const double alpha = Rcpp::as<double>(alpha_), beta = Rcpp::as<double>(beta_), gamma = Rcpp::as<double>(gamma_);
    const double a = Rcpp::as<double>(a_), b = Rcpp::as<double>(b_);
    const uint64_t seed = seed_of(seed_);
    const int64_t batch = (int64_t)Rcpp::as<double>(batch_);
    const int device = Rcpp::as<int>(device_);
    const int N = df.nrow(), P = df.ncol();
    if (initialK.size() != N) Rcpp::stop("initialK must hold one label per observation");
    return relabelled(N, P, K, nsamples, burnin, Rcpp::as<int>(burnrelabel_), false,
                      [&](const bmm_relabel_hooks* h, int32_t* z, double* th, double* al, double*) {
                          return bmm_collapsed_run_probs(df.begin(), N, P, initialK.begin(), nsamples, K, alpha, beta, gamma,
                                                         a, b, burnin, batch, seed, device, z, th, al, h);
                      });
    END_RCPP
}

extern "C" SEXP bmm_glue_dp_relabel(SEXP df_, SEXP nsamples_, SEXP alpha_, SEXP beta_, SEXP gamma_, SEXP a_, SEXP b_,
                                    SEXP burnin_, SEXP burnrelabel_, SEXP maxK_, SEXP seed_, SEXP batch_, SEXP device_) {
    BEGIN_RCPP
    const Rcpp::IntegerMatrix df(df_);
    const int nsamples = Rcpp::as<int>(nsamples_), maxK = Rcpp::as<int>(maxK_), burnin = Rcpp::as<int>(burnin_);
    const double alpha = Rcpp::as<double>(alpha_), beta = Rcpp::as<double>(beta_), gamma = Rcpp::as<double>(gamma_);
    const double a = Rcpp::as<double>(a_), b = Rcpp::as<double>(b_);
    const uint64_t seed = seed_of(seed_);
    const int64_t batch = (int64_t)Rcpp::as<double>(batch_);
    const int device = Rcpp::as<int>(device_);
    const int N = df.nrow(), P = df.ncol();
    // unused labels keep theta = 0 on both sides of the permutation (thetas_relab is zero-filled,
    // collapsed_gibbs_dp.cpp:78, and only used clusters are written, :266-279)
    return relabelled(N, P, maxK, nsamples, burnin, Rcpp::as<int>(burnrelabel_), false,
                      [&](const bmm_relabel_hooks* h, int32_t* z, double* th, double* al, double*) {
                          return bmm_dp_run_probs(df.begin(), N, P, nsamples, alpha, beta, gamma, a, b, burnin, maxK, batch,
                                                  seed, device, z, th, al, h);
                      });
    END_RCPP
}

static SEXP explicit_relabel(bool full, SEXP df_, SEXP initialPi_, SEXP initialTheta_, SEXP nsamples_, SEXP K_, SEXP alpha_,
                             SEXP beta_, SEXP gamma_, SEXP a_, SEXP b_, SEXP burnin_, SEXP burnrelabel_, SEXP seed_,
                             SEXP device_) {
    BEGIN_RCPP
    const Rcpp::IntegerMatrix df(df_);
    const Rcpp::NumericVector pi0(initialPi_);
    const Rcpp::NumericMatrix theta0(initialTheta_);
    const int nsamples = Rcpp::as<int>(nsamples_), K = Rcpp::as<int>(K_), burnin = Rcpp::as<int>(burnin_);
    const double alpha = Rcpp::as<double>(alpha_), beta = Rcpp::as<double>(beta_), gamma = Rcpp::as<double>(gamma_);
    const double a = Rcpp::as<double>(a_), b = Rcpp::as<double>(b_);
    const uint64_t seed = seed_of(seed_);
    const int device = Rcpp::as<int>(device_);
    const int N = df.nrow(), P = df.ncol();
    if (pi0.size() != K || theta0.nrow() != K || theta0.ncol() != P) Rcpp::stop("initialPi/initialTheta have the wrong size");
    return relabelled(N, P, K, nsamples, burnin, Rcpp::as<int>(burnrelabel_), true,
                      [&](const bmm_relabel_hooks* h, int32_t* z, double* th, double* al, double* pi) {
                          return (full ? bmm_full_run_probs : bmm_sb_run_probs)(df.begin(), N, P, pi0.begin(), theta0.begin(),
                                                                                nsamples, K, alpha, beta, gamma, a, b, burnin,
                                                                                seed, device, pi, z, th, al, h);
                      });
    END_RCPP
}

extern "C" SEXP bmm_glue_sb_relabel(SEXP df, SEXP initialPi, SEXP initialTheta, SEXP nsamples, SEXP maxK, SEXP alpha,
                                    SEXP beta, SEXP gamma, SEXP a, SEXP b, SEXP burnin, SEXP burnrelabel, SEXP seed,
                                    SEXP device) {
    return explicit_relabel(false, df, initialPi, initialTheta, nsamples, maxK, alpha, beta, gamma, a, b, burnin, burnrelabel,
                            seed, device);
}

extern "C" SEXP bmm_glue_full_relabel(SEXP df, SEXP initialPi, SEXP initialTheta, SEXP nsamples, SEXP K, SEXP alpha,
                                      SEXP beta, SEXP gamma, SEXP a, SEXP b, SEXP burnin, SEXP burnrelabel, SEXP seed,
                                      SEXP device) {
    return explicit_relabel(true, df, initialPi, initialTheta, nsamples, K, alpha, beta, gamma, a, b, burnin, burnrelabel, seed,
                            device);
}
