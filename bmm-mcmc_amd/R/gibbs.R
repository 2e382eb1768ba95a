# R front end of the MI355X allocation path.
#
# Same call signatures as the reference's exported samplers (R/utils.R:23-24, 37-39, 64-66,
# 95-96 of stulacy/bmm-mcmc) with optional trailing `seed` / `batch`, so existing calls keep
# working; the lists that come back carry the reference's element names, order, storage
# modes and dims.  Everything below the `.Call` is bmm-mcmc_amd/r-shim/bmmmcmc_shim.c.

.bmm <- new.env()

# burn-in default (a tenth of the run) and the burnrelabel clamp the reference applies
.bmm$schedule <- function(nsamples, burnin, burnrelabel, clamp = TRUE) {
    burnin <- if (is.null(burnin)) round(nsamples / 10) else burnin
    if (clamp && burnrelabel > burnin) burnrelabel <- round(burnin / 10)
    list(burnin = burnin, burnrelabel = burnrelabel)
}

# NULL concentration means "sample it", which the native side encodes as 0
.bmm$conc <- function(alpha) if (is.null(alpha)) 0 else alpha

# one integer from R's own stream when no seed is given: set.seed() still fixes the chain
.bmm$key <- function(seed) as.numeric(if (is.null(seed)) sample.int(.Machine$integer.max, 1) else seed)

.bmm$ints <- function(data) {
    m <- as.matrix(data)
    storage.mode(m) <- "integer"
    m
}

# starting weights and success probabilities of the samplers that carry them explicitly
.bmm$start <- function(k, p) {
    w <- exp(stats::runif(k))
    list(pi = w / sum(w), theta = matrix(stats::runif(k * p), nrow = k, ncol = p))
}

gibbs_collapsed <- function(data, nsamples, K, alpha=NULL, beta=0.5, gamma=0.5, a=1, b=1,
                            burnin=NULL, relabel=FALSE, burnrelabel=50, debug=FALSE,
                            seed=NULL, batch=0) {
    s <- .bmm$schedule(nsamples, burnin, burnrelabel)
    x <- .bmm$ints(data)
    z0 <- sample.int(K, nrow(x), replace = TRUE)
    .Call("_bmmmcmc_collapsed_gibbs_cpp", PACKAGE = "bmmmcmc", x, z0, nsamples, K, .bmm$conc(alpha),
          beta, gamma, a, b, s$burnin, relabel, s$burnrelabel, debug, .bmm$key(seed), as.numeric(batch))
}

gibbs_dp <- function(data, nsamples, alpha=NULL, a=1, b=1, beta=0.5, gamma=0.5,
                     burnin=NULL, relabel=FALSE, burnrelabel=50, maxK=30, debug=FALSE,
                     seed=NULL, batch=0) {
    s <- .bmm$schedule(nsamples, burnin, burnrelabel)
    .Call("_bmmmcmc_collapsed_gibbs_dp_cpp", PACKAGE = "bmmmcmc", .bmm$ints(data), nsamples,
          .bmm$conc(alpha), beta, gamma, a, b, s$burnin, relabel, s$burnrelabel, maxK, debug,
          .bmm$key(seed), as.numeric(batch))
}

gibbs_full <- function(data, nsamples, K, alpha=NULL, beta=0.5, gamma=0.5, a=1, b=1,
                       burnin=NULL, relabel=FALSE, burnrelabel=50, debug=FALSE, seed=NULL) {
    s <- .bmm$schedule(nsamples, burnin, burnrelabel)
    x <- .bmm$ints(data)
    init <- .bmm$start(K, ncol(x))
    .Call("_bmmmcmc_gibbs_cpp", PACKAGE = "bmmmcmc", x, init$pi, init$theta, nsamples, K,
          .bmm$conc(alpha), beta, gamma, a, b, s$burnin, relabel, s$burnrelabel, debug, .bmm$key(seed))
}

gibbs_stickbreaking <- function(data, nsamples, maxK, alpha=NULL, beta=0.5, gamma=0.5, a=1, b=1,
                                burnin=NULL, relabel=FALSE, burnrelabel=50, debug=FALSE, seed=NULL) {
    # the reference does not clamp burnrelabel in this wrapper (R/utils.R:97-101)
    s <- .bmm$schedule(nsamples, burnin, burnrelabel, clamp = FALSE)
    x <- .bmm$ints(data)
    init <- .bmm$start(maxK, ncol(x))
    .Call("_bmmmcmc_gibbs_stickbreaking_cpp", PACKAGE = "bmmmcmc", x, init$pi, init$theta, nsamples,
          maxK, .bmm$conc(alpha), beta, gamma, a, b, s$burnin, relabel, s$burnrelabel, debug,
          .bmm$key(seed))
}
