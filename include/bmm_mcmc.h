/* bmm_mcmc.h -- C ABI of the MI355X-native cluster-allocation path of bmm-mcmc.
 *
 * This is the drop-in boundary: the three *_run entry points take exactly what the
 * reference's .Call glue hands its C++ samplers and fill caller-owned buffers laid
 * out like the R objects it returns.  Plain pointers and sizes only; no R, Rcpp,
 * torch or HIP types.  Every function returns 0 on success or a BMM_E_* code, and
 * bmm_last_error() then describes the failure (thread-local).  Nothing here aborts.
 *
 * Layouts (all as R stores them):
 *   X          N x P int32, column-major: element (i,d) at X[i + d*N]   (IntegerMatrix,
 *              /root/reference/src/RcppExports.cpp:15)
 *   z_out      S x N int32, column-major, labels 1-based, S = nsamples - burnin
 *              (arma::Mat<int>::tail_rows, src/collapsed_gibbs.cpp:240)
 *   theta_out  K x P x S double, column-major                (src/collapsed_gibbs.cpp:241)
 *   alpha_out  S doubles (an S x 1 matrix in R)              (src/collapsed_gibbs.cpp:231)
 *   pi_out     S x maxK double, column-major                 (src/stickbreaking.cpp:240)
 * Rows the reference never writes (trace row 0 when burnin = 0) come back as
 * NA_integer_ (INT_MIN) / NaN / the initial values, see DESIGN.md "Quirks".
 *
 * `batch`: observations are resampled in consecutive batches of this many against
 * sufficient statistics frozen at batch start, each with its own contribution
 * removed exactly.  batch = 1 is the reference's sequential scan; batch <= 0 picks the
 * library default.  `seed` keys the Philox4x32-10 streams; same seed + same batch
 * => same chain, on any device.
 */
#ifndef BMM_MCMC_H
#define BMM_MCMC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BMM_OK 0
#define BMM_E_ARG 1         /* invalid argument (message says which) */
#define BMM_E_UNSUPPORTED 2 /* more than 1024 categories */
#define BMM_E_HIP 3         /* a HIP runtime call failed */
#define BMM_E_NODEVICE 4    /* no usable gfx950 device */
#define BMM_E_STATE 5       /* call sequence error on a resident chain */

#define BMM_SAMPLER_COLLAPSED 0
#define BMM_SAMPLER_DP 1
#define BMM_SAMPLER_SB 2
#define BMM_SAMPLER_FULL 3

#define BMM_NA_INTEGER (-2147483647 - 1)

const char* bmm_last_error(void);
/* features per lookup group of the spec arithmetic (DESIGN.md "Numerics") */
int bmm_spec_group_width(void);
/* library default batch size for N observations (used when batch <= 0): N/8 for the finite
 * sampler, N/16 for the DP sampler, N for stick-breaking; see DESIGN.md "Batches" */
int64_t bmm_default_batch(int sampler, int64_t N);

/* ---- drop-in entry points --------------------------------------------------------
 * Replaces collapsed_gibbs_cpp (src/collapsed_gibbs.cpp:24-36; .Call symbol
 * _bmmmcmc_collapsed_gibbs_cpp, src/RcppExports.cpp:11-31) with relabel = FALSE. */
int bmm_collapsed_run(const int32_t* X, int64_t N, int P, const int32_t* initialK, int nsamples,
                      int K, double alpha, double beta, double gamma, double a, double b, int burnin,
                      int64_t batch, uint64_t seed, int device, int32_t* z_out, double* theta_out,
                      double* alpha_out);
/* Replaces collapsed_gibbs_dp_cpp (src/collapsed_gibbs_dp.cpp:27-38; .Call symbol
 * _bmmmcmc_collapsed_gibbs_dp_cpp, src/RcppExports.cpp:34-53).  theta_out is maxK x P x S. */
int bmm_dp_run(const int32_t* X, int64_t N, int P, int nsamples, double alpha, double beta,
               double gamma, double a, double b, int burnin, int maxK, int64_t batch, uint64_t seed,
               int device, int32_t* z_out, double* theta_out, double* alpha_out);
/* Replaces gibbs_stickbreaking_cpp (src/stickbreaking.cpp:10-23; .Call symbol
 * _bmmmcmc_gibbs_stickbreaking_cpp, src/RcppExports.cpp:114-135).  initialTheta is
 * maxK x P column-major.  The z-step is exactly parallel, so there is no batch. */
int bmm_sb_run(const int32_t* X, int64_t N, int P, const double* initialPi,
               const double* initialTheta, int nsamples, int maxK, double alpha, double beta,
               double gamma, double a, double b, int burnin, uint64_t seed, int device,
               double* pi_out, int32_t* z_out, double* theta_out, double* alpha_out);

/* Replaces gibbs_cpp, the uncollapsed finite sampler (src/full_gibbs.cpp:32-45; .Call symbol
 * _bmmmcmc_gibbs_cpp, src/RcppExports.cpp:66-88): the z-step of the stick-breaking sampler with
 * pi ~ Dirichlet(alpha/K + counts) (full_gibbs.cpp:203-210).  SURVEY.md section 8 row f1. */
int bmm_full_run(const int32_t* X, int64_t N, int P, const double* initialPi,
                 const double* initialTheta, int nsamples, int K, double alpha, double beta,
                 double gamma, double a, double b, int burnin, uint64_t seed, int device,
                 double* pi_out, int32_t* z_out, double* theta_out, double* alpha_out);

/* ---- resident chains --------------------------------------------------------------
 * The same samplers with the data matrix and the chain state kept in HBM between
 * calls: what the benchmark and the one-chain-per-GPU driver use.  A chain is bound
 * to one device and one HIP stream; calls on one chain must not overlap. */
typedef struct bmm_chain bmm_chain;

int bmm_chain_create(bmm_chain** out, int sampler, int64_t N, int P, int K /* K or maxK */,
                     double alpha, double beta, double gamma, double a, double b, int64_t batch,
                     uint64_t seed, int device);
void bmm_chain_destroy(bmm_chain* c);
/* How the sweeps read X.  X is constant over the chain, so by default it is packed once, when
 * it is handed over, into bit planes (ceil(P/32) 32-bit words per observation: 24 instead of 408
 * bytes per observation and sweep at P = 100) and the int32 matrix is not kept: host data pass
 * through a staging buffer in slabs of rows, a matrix already on the device is read once and not
 * again after bmm_chain_set_data_device returns.  BMM_X_INT32 streams the IntegerMatrix layout R
 * hands over in place instead (no packing pass; a device matrix stays borrowed for the life of
 * the chain).  Same chain either way.  To be chosen before the data are set. */
#define BMM_X_BITPLANES 0
#define BMM_X_INT32 1
int bmm_chain_set_x_layout(bmm_chain* c, int layout);
int bmm_chain_get_x_layout(const bmm_chain* c, int* layout);
/* X from host memory or already on this device (see the layouts above for what is kept) */
int bmm_chain_set_data_host(bmm_chain* c, const int32_t* X);
int bmm_chain_set_data_device(bmm_chain* c, const void* dX);
/* starting state: collapsed needs 1-based labels; stick-breaking needs pi and theta;
 * dp starts from zero clusters and needs neither */
int bmm_chain_set_initial_labels(bmm_chain* c, const int32_t* z1);
int bmm_chain_set_initial_params(bmm_chain* c, const double* pi, const double* theta);
/* enqueue n more sweeps on the chain's stream (returns without waiting) */
int bmm_chain_sweeps(bmm_chain* c, int n);
int bmm_chain_sync(bmm_chain* c);
/* n more sweeps, returning the cluster sizes after each one (nk_out is n x K, row-major: sweep,
 * label) -- the per-sweep summary plot_gibbs derives from z (R/utils.R:146-173) without moving
 * the 4*N-byte label row off the device each sweep.  Waits. */
int bmm_chain_sweeps_counts(bmm_chain* c, int n, int32_t* nk_out);
int bmm_chain_sweep_index(const bmm_chain* c); /* sweeps done so far */
/* current state, copied to host: labels 1-based (NA where unassigned) */
int bmm_chain_get_labels(bmm_chain* c, int32_t* z1);
int bmm_chain_get_counts(bmm_chain* c, int32_t* Nk /*K*/, int32_t* S /*K*P, S[k*P+d]*/);
int bmm_chain_get_alpha(bmm_chain* c, double* alpha);
int bmm_chain_get_params(bmm_chain* c, double* pi /*K*/, double* theta /*K x P colmajor*/);
/* One more sweep, also returning that sweep's allocation probabilities: probs_out is N x K
 * column-major (host), row i = the normalised conditional observation i was drawn from, by label
 * (the DP's new-cluster mass under the label it would open).  This is the matrix the reference
 * stores for Stephens' relabelling (src/collapsed_gibbs.cpp:162-172, collapsed_gibbs_dp.cpp:190-200,
 * stickbreaking.cpp:129-139) -- the hand-off for relabel = TRUE, whose batch/online steps stay on the
 * host in the reference's own code.  Runs on the generic kernel; waits. */
int bmm_chain_sweep_probs(bmm_chain* c, double* probs_out);

/* ---- one chain sharded over several ranks --------------------
 * Exact for the stick-breaking and full samplers, whose z-step is independent across
 * observations given (pi, theta): each rank holds rows [first_row, first_row + N) of the N_total
 * and resamples them; the K*(P+1) integer statistic deltas are summed over ranks (one RCCL
 * all-reduce, done by the caller on the device pointers below); every rank then draws the same
 * pi, theta, alpha, because the Philox streams are keyed by (seed, global index) only.
 *   set_shard once before the first sweep; per sweep: shard_resample (returns with the deltas
 *   complete), all-reduce *dNk (K int32) and *dS (K*P int32) in place, shard_finish. */
int bmm_chain_set_shard(bmm_chain* c, int64_t N_total, int64_t first_row);
int bmm_chain_shard_resample(bmm_chain* c);
int bmm_chain_shard_deltas(bmm_chain* c, void** dNk, void** dS);
int bmm_chain_shard_finish(bmm_chain* c);

/* HIP-event timing of the z-resample kernel on the chain's own stream: turn on, run
 * sweeps, sync, read total milliseconds and launch count since it was turned on.
 * every = 0 off, 1 every sweep, n the launches of every n-th sweep only (an event pair costs
 * about 3 us of stream time per launch, which a sampled measurement keeps out of the total) */
int bmm_chain_profile(bmm_chain* c, int every);
int bmm_chain_profile_read(bmm_chain* c, double* resample_ms, int64_t* resample_launches);
/* batch size in effect (a defaulted one is rounded up to whole rounds of workgroups) */
int64_t bmm_chain_batch(const bmm_chain* c);
/* bytes of dynamic LDS and threads per workgroup the resample kernel uses for this shape */
int bmm_chain_kernel_shape(const bmm_chain* c, int* lds_bytes, int* threads, int* grid_max);

/* ---- device self-checks used by the parity tests (op: 0 log, 1 exp, 2 div by in2,
 * 3 sqrt; elementwise over n doubles, evaluated on the GPU with the spec arithmetic) */
int bmm_device_math(int device, int op, const double* in, const double* in2, double* out, int64_t n);
/* out[i] = the spec's variate number `kind` (0 gamma(shape p), 1 beta(p,q), 2 update_alpha
 * with alpha_old p, K = (int)q, N = 1000, a = b = 1) for stream index i, on the GPU */
int bmm_device_variates(int device, int kind, double p, double q, uint64_t seed, uint32_t sweep,
                        double* out, int64_t n);
int bmm_device_count(int* n);

#ifdef __cplusplus
}
#endif
#endif
