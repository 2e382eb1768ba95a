# R wrappers over the MI355X allocation path: the signatures of the reference's
# R/utils.R:23-47,95-107 with three optional trailing arguments (seed, batch, and for
# the stick-breaking sampler seed only).  Old calls stay valid; the returned lists have
# the reference's names, order, storage modes and dims.  When `seed` is NULL one integer
# is drawn from R's own RNG, so set.seed() still fixes the chain.

.bmm_seed <- function(seed) if (is.null(seed)) sample.int(.Machine$integer.max, 1) else seed

gibbs_dp <- function(data, nsamples, alpha=NULL, a=1, b=1, beta=0.5, gamma=0.5,
                     burnin=NULL, relabel=FALSE, burnrelabel=50, maxK=30, debug=FALSE,
                     seed=NULL, batch=0) {
    if (is.null(burnin)) burnin <- round(0.1 * nsamples)
    if (burnrelabel > burnin) burnrelabel <- round(0.1 * burnin)
    if (is.null(alpha)) alpha <- 0
    storage.mode(data) <- "integer"
    .Call('_bmmmcmc_collapsed_gibbs_dp_cpp', PACKAGE = 'bmmmcmc', data, nsamples, alpha, beta, gamma,
          a, b, burnin, relabel, burnrelabel, maxK, debug, as.numeric(.bmm_seed(seed)), as.numeric(batch))
}

gibbs_collapsed <- function(data, nsamples, K, alpha=NULL, beta=0.5, gamma=0.5,
                            a=1, b=1,
                            burnin=NULL, relabel=FALSE, burnrelabel=50, debug=FALSE,
                            seed=NULL, batch=0) {
    if (is.null(burnin)) burnin <- round(0.1 * nsamples)
    if (burnrelabel > burnin) burnrelabel <- round(0.1 * burnin)
    initial_K <- sample(1:K, nrow(data), replace=T)
    if (is.null(alpha)) alpha <- 0
    storage.mode(data) <- "integer"
    .Call('_bmmmcmc_collapsed_gibbs_cpp', PACKAGE = 'bmmmcmc', data, initial_K, nsamples, K, alpha,
          beta, gamma, a, b, burnin, relabel, burnrelabel, debug, as.numeric(.bmm_seed(seed)),
          as.numeric(batch))
}

gibbs_stickbreaking <- function(data, nsamples, maxK, alpha=NULL, beta=0.5, gamma=0.5, a=1, b=1,
                                burnin=NULL, relabel=FALSE, burnrelabel=50, debug=FALSE, seed=NULL) {
    if (is.null(burnin)) burnin <- round(0.1 * nsamples)
    initial_pi <- stats::runif(maxK)
    initial_pi <- exp(initial_pi)
    initial_pi <- initial_pi / sum(initial_pi)
    if (is.null(alpha)) alpha <- 0
    initial_theta <- matrix(stats::runif(maxK*ncol(data)), ncol=ncol(data), nrow=maxK)
    storage.mode(data) <- "integer"
    .Call('_bmmmcmc_gibbs_stickbreaking_cpp', PACKAGE = 'bmmmcmc', data, initial_pi, initial_theta,
          nsamples, maxK, alpha, beta, gamma, a, b, burnin, relabel, burnrelabel, debug,
          as.numeric(.bmm_seed(seed)))
}

gibbs_full <- function(data, nsamples, K, alpha=NULL, beta=0.5, gamma=0.5,
                       a=1, b=1,
                       burnin=NULL, relabel=FALSE, burnrelabel=50, debug=FALSE, seed=NULL) {
    if (is.null(burnin)) burnin <- round(0.1 * nsamples)
    initial_pi <- stats::runif(K)
    initial_pi <- exp(initial_pi)
    initial_pi <- initial_pi / sum(initial_pi)
    if (is.null(alpha)) alpha <- 0
    if (burnrelabel > burnin) burnrelabel <- round(0.1 * burnin)
    initial_theta <- matrix(stats::runif(K*ncol(data)), ncol=ncol(data), nrow=K)
    storage.mode(data) <- "integer"
    .Call('_bmmmcmc_gibbs_cpp', PACKAGE = 'bmmmcmc', data, initial_pi, initial_theta,
          nsamples, K, alpha, beta, gamma, a, b, burnin, relabel, burnrelabel, debug,
          as.numeric(.bmm_seed(seed)))
}
