// relabel_glue.cpp -- relabel = TRUE end to end: the package's own, unchanged Stephens code
// (src/stephens.cpp: my_stephens_batch, my_stephens_online; src/my_lpsolve.cpp underneath) fed from
// the device by bmm_collapsed_run_probs (include/bmm_mcmc.h, "relabel = TRUE").
//
// C++ on Rcpp/RcppArmadillo like the file it replaces, because the functions it calls take
// arma::cube / arma::mat.  It belongs in the package's src/ beside stephens.cpp.  Neither R nor
// RcppArmadillo is in the build image, so this file is NOT compiled or tested there; the data path
// it relies on is (tests/test_gpu_relabel.py drives the same hooks from Python and checks the
// matrices against the oracle's conditionals).  It restates the reference's bookkeeping around the
// two calls, src/collapsed_gibbs.cpp:187-201, 215-217, 232-243, and nothing else.
//
// [[Rcpp::depends(RcppArmadillo)]]
#include <RcppArmadillo.h>

#include "bmm_mcmc.h"
#include "stephens.h"  // the package's: my_stephens_batch, my_stephens_online

namespace {

struct RelabelState {
    int N, K, burnin, burnrelabel;
    arma::cube probs_out;      // N x K x burnrelabel, filled by the library
    arma::mat Q;
    arma::Mat<int> permutations;  // S x K
};

int batch_done(void* user, int /*j*/, const double* /*probs == probs_out.memptr()*/) {
    RelabelState* s = static_cast<RelabelState*>(user);
    try {
        s->Q = my_stephens_batch(s->probs_out, false);  // collapsed_gibbs.cpp:190
        return 0;
    } catch (...) { return 1; }
}

int on_sample(void* user, int j, const double* probs) {
    RelabelState* s = static_cast<RelabelState*>(user);
    try {
        // a view of the library's N x K column-major matrix: no copy (copy_aux_mem = false)
        const arma::mat probs_sample(const_cast<double*>(probs), s->N, s->K, false, true);
        std::pair<arma::Row<int>, arma::mat> out = my_stephens_online(s->Q, probs_sample, j, false);  // :192
        s->Q = out.second;
        s->permutations.row(j - s->burnin) = out.first;  // :196
        return 0;
    } catch (...) { return 1; }
}

}  // namespace

// Same signature and returned list as collapsed_gibbs_cpp with relabel = TRUE (src/collapsed_gibbs.cpp:24-36,
// :232-243), plus the trailing seed / batch of the *_ex entry points.
// [[Rcpp::export]]
Rcpp::List collapsed_gibbs_relabel(Rcpp::IntegerMatrix df, Rcpp::IntegerVector initialK, int nsamples, int K,
                                   double alpha, double beta, double gamma, double a, double b, int burnin,
                                   int burnrelabel, double seed, double batch) {
    const int N = df.nrow(), P = df.ncol(), S = nsamples - burnin;
    RelabelState st{N, K, burnin, burnrelabel, arma::cube(N, K, burnrelabel, arma::fill::zeros), arma::mat(),
                    arma::Mat<int>(S, K)};
    bmm_relabel_hooks hooks{burnrelabel, st.probs_out.memptr(), batch_done, on_sample, &st};
    arma::Mat<int> z(S, N);
    arma::cube theta(K, P, S);
    arma::vec alpha_out(S);
    const int rc = bmm_collapsed_run_probs(df.begin(), N, P, initialK.begin(), nsamples, K, alpha, beta, gamma, a, b,
                                           burnin, (int64_t)batch, (uint64_t)seed, 0, z.memptr(), theta.memptr(),
                                           alpha_out.memptr(), &hooks);
    if (rc) Rcpp::stop(bmm_last_error());
    arma::Mat<int> z_relabelled(S, N);
    arma::cube thetas_relabelled(K, P, S);
    for (int s = 0; s < S; ++s) {
        const arma::Row<int> perm = st.permutations.row(s);
        for (int i = 0; i < N; ++i) z_relabelled(s, i) = perm(z(s, i) - 1) + 1;      // :198
        for (int k = 0; k < K; ++k)
            for (int d = 0; d < P; ++d) thetas_relabelled(perm(k), d, s) = theta(k, d, s);  // :216
    }
    Rcpp::List ret;  // names and order of :232-243
    ret["alpha"] = alpha_out;
    ret["permutations"] = st.permutations;
    ret["z"] = z_relabelled;
    ret["theta"] = thetas_relabelled;
    ret["z_original"] = z;
    ret["theta_original"] = theta;
    return ret;
}
