# R front end of the MI355X allocation path.
#
# Same call signatures as the reference's exported samplers (R/utils.R:23-24, 37-39, 64-66,
# 95-96 of stulacy/bmm-mcmc) with optional trailing `seed`, `batch`, `chains`, `devices`, so
# existing calls keep working; the lists that come back carry the reference's element names,
# order, storage modes and dims.  Everything below the `.Call` is
# bmm-mcmc_amd/r-shim/bmmmcmc_shim.c, whose *_ex entry points these wrappers call; the package's
# own unchanged R/RcppExports.R binds too (same names, same arities: the shim then draws the seed
# from R's stream and uses the default batch).

.bmm <- new.env()

# burn-in default (a tenth of the run) and the burnrelabel clamp the reference applies
.bmm$schedule <- function(nsamples, burnin, burnrelabel, clamp = TRUE) {
    burnin <- if (is.null(burnin)) round(nsamples / 10) else burnin
    if (clamp && burnrelabel > burnin) burnrelabel <- round(burnin / 10)
    list(burnin = burnin, burnrelabel = burnrelabel)
}

# NULL concentration means "sample it", which the native side encodes as 0
.bmm$conc <- function(alpha) if (is.null(alpha)) 0 else alpha

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

# starting states of `chains` chains, one column per chain, drawn from R's stream in chain order
.bmm$starts <- function(chains, k, p) {
    st <- lapply(seq_len(chains), function(c) .bmm$start(k, p))
    list(pi = vapply(st, function(s) s$pi, numeric(k)),
         theta = vapply(st, function(s) as.numeric(s$theta), numeric(k * p)))
}

# Trailing arguments every wrapper gains (defaults keep the reference's calls valid):
#   seed     NULL: the shim draws the chain's key from R's RNG stream, so set.seed() fixes the chain
#   batch    NULL: library default (bmm_default_batch); 1: the reference's sequential scan
#   chains   independent chains in one call (keys seed, seed + 1, ...): returns a list of chain objects
#   devices  NULL: device 0; else one GPU index per chain (bmm_multi_run: one upload, one RCCL broadcast)

gibbs_collapsed <- function(data, nsamples, K, alpha=NULL, beta=0.5, gamma=0.5, a=1, b=1,
                            burnin=NULL, relabel=FALSE, burnrelabel=50, debug=FALSE,
                            seed=NULL, batch=NULL, chains=1, devices=NULL) {
    s <- .bmm$schedule(nsamples, burnin, burnrelabel)
    x <- .bmm$ints(data)
    z0 <- vapply(seq_len(chains), function(c) sample.int(K, nrow(x), replace = TRUE), integer(nrow(x)))
    .Call("_bmmmcmc_collapsed_gibbs_ex", PACKAGE = "bmmmcmc", x, z0, nsamples, K, .bmm$conc(alpha),
          beta, gamma, a, b, s$burnin, relabel, s$burnrelabel, debug, seed, batch, chains, devices)
}

gibbs_dp <- function(data, nsamples, alpha=NULL, a=1, b=1, beta=0.5, gamma=0.5,
                     burnin=NULL, relabel=FALSE, burnrelabel=50, maxK=30, debug=FALSE,
                     seed=NULL, batch=NULL, chains=1, devices=NULL) {
    s <- .bmm$schedule(nsamples, burnin, burnrelabel)
    .Call("_bmmmcmc_collapsed_gibbs_dp_ex", PACKAGE = "bmmmcmc", .bmm$ints(data), nsamples,
          .bmm$conc(alpha), beta, gamma, a, b, s$burnin, relabel, s$burnrelabel, maxK, debug,
          seed, batch, chains, devices)
}

gibbs_full <- function(data, nsamples, K, alpha=NULL, beta=0.5, gamma=0.5, a=1, b=1,
                       burnin=NULL, relabel=FALSE, burnrelabel=50, debug=FALSE,
                       seed=NULL, chains=1, devices=NULL) {
    s <- .bmm$schedule(nsamples, burnin, burnrelabel)
    x <- .bmm$ints(data)
    init <- .bmm$starts(chains, K, ncol(x))
    .Call("_bmmmcmc_gibbs_ex", PACKAGE = "bmmmcmc", x, init$pi, init$theta, nsamples, K,
          .bmm$conc(alpha), beta, gamma, a, b, s$burnin, relabel, s$burnrelabel, debug, seed, chains, devices)
}

gibbs_stickbreaking <- function(data, nsamples, maxK, alpha=NULL, beta=0.5, gamma=0.5, a=1, b=1,
                                burnin=NULL, relabel=FALSE, burnrelabel=50, debug=FALSE,
                                seed=NULL, chains=1, devices=NULL) {
    # the reference does not clamp burnrelabel in this wrapper (R/utils.R:97-101)
    s <- .bmm$schedule(nsamples, burnin, burnrelabel, clamp = FALSE)
    x <- .bmm$ints(data)
    init <- .bmm$starts(chains, maxK, ncol(x))
    .Call("_bmmmcmc_gibbs_stickbreaking_ex", PACKAGE = "bmmmcmc", x, init$pi, init$theta, nsamples,
          maxK, .bmm$conc(alpha), beta, gamma, a, b, s$burnin, relabel, s$burnrelabel, debug,
          seed, chains, devices)
}

# The reference prints "Sample j" at every sweep (src/collapsed_gibbs.cpp:85).  This build is silent unless
# asked: bmm_progress(100) prints that line every 100 sweeps (with the current number of clusters for
# gibbs_dp, src/collapsed_gibbs_dp.cpp:99), bmm_progress(0) turns it off again; debug = TRUE prints every sweep.
# Returns the previous setting, invisibly.
bmm_progress <- function(every = 0) {
    invisible(.Call("_bmmmcmc_set_progress", PACKAGE = "bmmmcmc", as.integer(every)))
}
