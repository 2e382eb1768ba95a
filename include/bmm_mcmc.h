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
 * removed exactly.  batch = 1 is the reference's sequential scan (src/collapsed_gibbs.cpp:86-182
 * updates the member lists after every draw); batch <= 0 picks bmm_default_batch(sampler, N), a
 * function of the sampler and N only.  `seed` keys the Philox streams; same seed + same batch
 * (given or defaulted) => same chain, bit for bit, on any device and with either X layout.
 *
 * TOLERANCE of a batch > 1 against the sequential scan (the north star's "stated floating-point
 * tolerance on posterior cluster proportions").  Quantity: the posterior-mean cluster
 * proportions, sorted descending (label-switching invariant), averaged over the kept sweeps.
 *   BMM_TOL_PROPORTIONS 0.015  |default-batch chain - batch-1 chain| per component, both averaged
 *                              over >= 3 seeds, on each of the reference's bundled data sets
 *                              (measured, 3 seeds x 400 kept sweeps: 0.0002 on K2_N100_P5, 0.0001 on
 *                              K2_N1000_P5, 0.005 on K3_N1000_P5, whose two 0.2 components overlap)
 *   BMM_TOL_THETA       0.05   the same comparison for theta-hat, per cell, clusters ordered by size
 *                              within each sweep (measured 0.0004, 0.0001, 0.037).  Against the
 *                              generating values the reference documents (R/bmm-mcmc.R:16-17, 34-35,
 *                              49-50) theta-hat of K2_N1000_P5 is within the same 0.05 (measured
 *                              0.035); on K3_N1000_P5 the posterior itself, at batch 1 as at the
 *                              default, sits up to 0.16 from them (finite sample, overlapping
 *                              components), so there the pin is the batch-1 chain, not the truth.
 * The DP sampler at its default batch (N/16): the two dominant components of K2_N1000_P5 within 0.05
 * of the batch-1 chain over 6 seeds (measured 0.038; noise-limited -- an unbounded-K chain wanders,
 * seed s.d. 0.08).
 * tests/test_gpu_tolerance.py holds the HIP path to all of it on the GPU, default batch against the
 * batch-1 oracle chain.  At the shapes the benchmark numbers are quoted on the batch-1 chain is a committed
 * fixture (tests/golden/tolerance_*.json: K = 20 with N = 1e6, P = 50 and N = 1e7, P = 100; K = 3, N = 1e5; five
 * DP shapes up to N = 1e6) and tests/test_gpu_tolerance_fixtures.py holds the default batch to the same two
 * numbers against it, per component including the 1/210 one: measured at most 2e-5 and 1e-3 from the generating
 * allocation (the bias of a batch shrinks with N; the bundled N <= 1000 sets above are the hard case).  For the
 * DP sampler there the quantity is the share of the observations per generating component (the sequential
 * scan itself seats a generating component as two clusters at N >= 1e5).  The tests run at exactly what
 * bmm_default_batch returns (N/8 and N/16 below 2^16 observations, N/4 for both samplers from there on), including a
 * K = 20 fixture at N = 2^16, the smallest N that gets the larger batch.
 * The stick-breaking and full samplers have no batch and no tolerance:
 * their z-step is exactly parallel (src/stickbreaking.cpp:69-92 reads only sweep j-1).
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
#define BMM_E_RCCL 6        /* RCCL could not be opened, or a collective failed */
#define BMM_E_CALLBACK 7    /* a relabel hook returned non-zero */

#define BMM_TOL_PROPORTIONS 0.015
#define BMM_TOL_THETA 0.05

#define BMM_SAMPLER_COLLAPSED 0
#define BMM_SAMPLER_DP 1
#define BMM_SAMPLER_SB 2
#define BMM_SAMPLER_FULL 3

#define BMM_NA_INTEGER (-2147483647 - 1)

const char* bmm_last_error(void);
/* features per lookup group of the spec arithmetic (DESIGN.md "Numerics"): of the tables against the full
 * statistics (preferred width; _for: the width the shape runs at -- narrower when its table image would not
 * fit in LDS otherwise; a pure function of its arguments, -1 for invalid ones), and of the own-cluster
 * ("minus self") tables */
int bmm_spec_group_width(void);
int bmm_spec_group_width_own(void);
int bmm_spec_group_width_for(int sampler, int K, int P);
/* library default batch size for N observations (used when batch <= 0): below 2^16 observations floor(N/8)
 * for the finite sampler and floor(N/16) for the DP sampler (at least 1); from 2^16 on floor(N/4) for
 * both -- the bias of a batch shrinks with N and is not measurable there (TOLERANCE above; DESIGN.md
 * section 2); N for stick-breaking and full.  Depends on nothing else. */
int64_t bmm_default_batch(int sampler, int64_t N);

/* ---- progress of the *_run entry points ---------------------------------------------------
 * The reference prints "Sample j" at every sweep (src/collapsed_gibbs.cpp:85, stickbreaking.cpp:67; the DP
 * sampler adds the current number of clusters, collapsed_gibbs_dp.cpp:99).  Here a run is silent unless the
 * caller installs a hook: after every `every`-th sweep the calling thread calls fn(user, sample, nsamples,
 * k_used) with sample = the reference's printed index of the sweep just finished (2 .. nsamples), k_used =
 * the DP sampler's clusters in use after it (-1 for the other samplers).  A non-zero return stops the run
 * (BMM_E_CALLBACK).  Per calling thread; applies to the single-chain *_run calls made afterwards; fn = NULL
 * or every <= 0 turns it off.  The sweeps stay enqueued ahead of the device: the hook follows events. */
typedef int (*bmm_progress_fn)(void* user, int sample, int nsamples, int k_used);
int bmm_set_progress(bmm_progress_fn fn, void* user, int every);
/* Wall-clock milliseconds the last single-chain *_run call of this thread spent in: [1] creating the chain,
 * its buffers and the starting state (the host's other cores validate and pack X meanwhile), [0] what was
 * left of the packing after that, plus the upload of the planes, [2] enqueueing the sweeps, [3] waiting for
 * the device to finish them, [4] the label trace on its way out (device transpose, PCIe, host copy into the
 * caller's matrix), [5] releasing the chain. */
#define BMM_RUN_PHASES 6
int bmm_last_run_phases(double* ms /* BMM_RUN_PHASES doubles */);
/* host threads the two ends of a run use (affinity mask, cgroup CPU quota, at most 16) */
int bmm_host_threads(void);
/* Between calls the library keeps up to eight 4 MiB pieces of pinned host staging and, per device, up to four
 * idle plain streams and up to six device blocks of at most 512 MiB in all (creating and destroying a stream
 * costs about 2 ms each on this runtime, pinning 4 MiB about 1 ms, and a device allocation now and then 10 ms:
 * together more than a 220-sweep drop-in call at N = 1e6 spends outside its sweeps).  This releases them; they
 * come back with the next call. */
int bmm_release_pools(void);

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

/* ---- relabel = TRUE: the per-sweep allocation probabilities for the host's Stephens code ----
 * The reference stores, per observation, the normalised conditional it drew z_i from: for the last
 * `burnrelabel` burn-in sweeps into the cube probs_out (N x K x burnrelabel,
 * src/collapsed_gibbs.cpp:76,162-167) that my_stephens_batch consumes once at j = burnin - 1
 * (:187-190), then every kept sweep's N x K matrix probs_sample for my_stephens_online (:168-172,
 * :191-192).  Stephens' algorithm and lp_solve stay host code of the reference, unchanged; the
 * *_run_probs entry points produce exactly those matrices on the device (from the resident
 * resample kernel's own weights) and hand them over through these hooks, called on the calling
 * thread.  Matrices are column-major, by label; the DP's new-cluster mass is filed under the label
 * it would open (src/collapsed_gibbs_dp.cpp:193).  A hook returning non-zero stops the run
 * (BMM_E_CALLBACK).  The chain itself does not depend on the relabelling, so sweep j + 1 already runs
 * while on_sample works on sweep j.  hooks == NULL: exactly the plain *_run. */
typedef int (*bmm_probs_fn)(void* user, int j /* sweep index, 1-based as the reference's loop */,
                            const double* probs /* host, valid during the call */);
typedef struct bmm_relabel_hooks {
    int burnrelabel;         /* sweeps of the batch window (clamped to burnin) */
    double* probs_batch;     /* N x K x burnrelabel doubles, filled before batch_done; caller-owned */
    bmm_probs_fn batch_done; /* once, after sweep burnin - 1; probs = probs_batch; may be NULL */
    bmm_probs_fn on_sample;  /* after every sweep j >= burnin; probs = that sweep's N x K; may be NULL */
    void* user;
} bmm_relabel_hooks;
int bmm_collapsed_run_probs(const int32_t* X, int64_t N, int P, const int32_t* initialK, int nsamples,
                            int K, double alpha, double beta, double gamma, double a, double b,
                            int burnin, int64_t batch, uint64_t seed, int device, int32_t* z_out,
                            double* theta_out, double* alpha_out, const bmm_relabel_hooks* hooks);
int bmm_dp_run_probs(const int32_t* X, int64_t N, int P, int nsamples, double alpha, double beta,
                     double gamma, double a, double b, int burnin, int maxK, int64_t batch,
                     uint64_t seed, int device, int32_t* z_out, double* theta_out, double* alpha_out,
                     const bmm_relabel_hooks* hooks);
int bmm_sb_run_probs(const int32_t* X, int64_t N, int P, const double* initialPi,
                     const double* initialTheta, int nsamples, int maxK, double alpha, double beta,
                     double gamma, double a, double b, int burnin, uint64_t seed, int device,
                     double* pi_out, int32_t* z_out, double* theta_out, double* alpha_out,
                     const bmm_relabel_hooks* hooks);
int bmm_full_run_probs(const int32_t* X, int64_t N, int P, const double* initialPi,
                       const double* initialTheta, int nsamples, int K, double alpha, double beta,
                       double gamma, double a, double b, int burnin, uint64_t seed, int device,
                       double* pi_out, int32_t* z_out, double* theta_out, double* alpha_out,
                       const bmm_relabel_hooks* hooks);

/* ---- several independent chains in one call (SURVEY.md section 8 rows b and e) ---------------
 * n_chains chains of one sampler over the same data, chain c keyed seed + c and resident on
 * devices[c] (all on device 0 when devices is NULL; a device may appear several times: its chains
 * share one copy of the data and overlap on streams with hardware queues of their own -- about four
 * chains per device is where that pays).  X is uploaded and packed into bit
 * planes once, on devices[0]; when the run spans several devices the planes -- 4 * ceil(P/32) bytes
 * per observation, not the int32 matrix -- are broadcast once with RCCL over xGMI inside this
 * process (ncclCommInitAll over the distinct devices, one ncclBroadcast; librccl is opened on
 * demand, BMM_E_RCCL if that fails).  That is the only collective: chains never communicate.
 * One host thread per chain drives it; nothing of R's API is touched off the calling thread.
 * Per-chain inputs and outputs are tables of n_chains pointers, each laid out as in the
 * single-chain entry point of the sampler (initialK for collapsed; initialPi/initialTheta and pi_out
 * for stick-breaking and full; unused tables may be NULL).  relabel is not offered here. */
int bmm_multi_run(int sampler, int n_chains, const int* devices, const int32_t* X, int64_t N, int P,
                  const int32_t* const* initialK, const double* const* initialPi,
                  const double* const* initialTheta, int nsamples, int K /* K or maxK */, double alpha,
                  double beta, double gamma, double a, double b, int burnin, int64_t batch, uint64_t seed,
                  double* const* pi_out, int32_t* const* z_out, double* const* theta_out,
                  double* const* alpha_out);
/* Where bmm_multi_run puts things for a device table (NULL: every chain on device 0), as pure bookkeeping --
 * no device is touched: devices_out[0 .. *n_devices_out) = the distinct devices in first-use order (the
 * broadcast list of the one ncclBroadcast, root first: devices_out[0] packs the data), holder_of_chain[c] =
 * the chain that holds the copy of the bit planes chain c reads (the first chain on c's device; a holder
 * names itself, every other chain shares with bmm_chain_share_data).  Both outputs have room for n_chains. */
int bmm_multi_plan(int n_chains, const int* devices, int* n_devices_out, int* devices_out, int* holder_of_chain);
/* The broadcast of bmm_multi_run on a test pattern of `words` 32-bit words over the listed distinct
 * devices, compared afterwards on every one of them.  With one device the collective still runs
 * (library opened, communicator of one rank, ncclBroadcast): what a one-GPU box can check. */
int bmm_multi_selfcheck(int n_devices, const int* devices, int64_t words);

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
/* X from host memory or already on this device (see the layouts above for what is kept).
 * set_data_device reads dX on the chain's own stream: it first waits for the whole device
 * (hipDeviceSynchronize), so a matrix a kernel or a collective of the caller's is still writing
 * is complete before it is validated and packed. */
int bmm_chain_set_data_host(bmm_chain* c, const int32_t* X);
int bmm_chain_set_data_device(bmm_chain* c, const void* dX);
/* Several chains on one device over the same data: `c` shares the bit planes `from` holds (same
 * device, N and P), before either has run a sweep.  The planes are reference-counted: chains may be
 * destroyed in any order, the last one frees them.  Both chains move to streams with hardware queues of
 * their own, so that their launches overlap (up to about four chains per device pay off).  What
 * bmm_multi_run does for chains that share a device. */
int bmm_chain_share_data(bmm_chain* c, bmm_chain* from);
/* The chain's bit planes on its device: ceil(P/32) planes of N 32-bit words (allocated on first
 * call).  A rank that received them from a broadcast (160 MB instead of the 4 GB int32 matrix at
 * K=20, N=1e7, P=100) declares them complete with bmm_chain_planes_filled, which waits for the device. */
int bmm_chain_planes(bmm_chain* c, void** dXb, int64_t* n_words);
int bmm_chain_planes_filled(bmm_chain* c);
/* Resident chains on different devices, one per device, in one process: chains[0] holds the data; the
 * others receive its bit planes with one RCCL broadcast (ncclCommInitAll + ncclBroadcast, as
 * bmm_multi_run does).  Then bmm_chains_sweeps drives them, one host thread each. */
int bmm_chains_broadcast_planes(bmm_chain* const* chains, int n_chains);
/* starting state: collapsed needs 1-based labels; stick-breaking needs pi and theta;
 * dp starts from zero clusters and needs neither */
int bmm_chain_set_initial_labels(bmm_chain* c, const int32_t* z1);
int bmm_chain_set_initial_params(bmm_chain* c, const double* pi, const double* theta);
/* enqueue n more sweeps on the chain's stream (returns without waiting) */
int bmm_chain_sweeps(bmm_chain* c, int n);
/* the same for several chains at once, one host thread per chain (chains on one device overlap) */
int bmm_chains_sweeps(bmm_chain* const* chains, int n_chains, int n);
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
 * stickbreaking.cpp:129-139) -- one sweep of what the *_run_probs entry points stream (SURVEY.md section 8
 * row f2): the weights come out of the resident resample kernel itself.  Waits. */
int bmm_chain_sweep_probs(bmm_chain* c, double* probs_out);

/* ---- one chain sharded over several ranks --------------------
 * Exact for the stick-breaking and full samplers, whose z-step is independent across
 * observations given (pi, theta): each rank holds rows [first_row, first_row + N) of the N_total
 * and resamples them; the K*(P+1) integer statistic deltas are summed over ranks (one RCCL
 * all-reduce, done by the caller on the device pointers below); every rank then draws the same
 * pi, theta, alpha, because the Philox streams are keyed by (seed, global index) only.
 *   set_shard once before the first sweep; per sweep: shard_resample (returns with the deltas
 *   complete), all-reduce *dNk (K int32) and *dS (K*P int32) in place, shard_finish.
 *   The _async form of shard_resample returns without waiting: the caller then orders its collective
 *   behind the chain's own HIP stream (bmm_chain_stream) and the finish behind the collective -- no
 *   host round trip inside a sweep (multi.ShardedChain does it with a torch ExternalStream). */
int bmm_chain_set_shard(bmm_chain* c, int64_t N_total, int64_t first_row);
int bmm_chain_shard_resample(bmm_chain* c);
int bmm_chain_shard_resample_async(bmm_chain* c);
int bmm_chain_stream(bmm_chain* c, void** hip_stream);
int bmm_chain_shard_deltas(bmm_chain* c, void** dNk, void** dS);
int bmm_chain_shard_finish(bmm_chain* c);

/* HIP-event timing of the z-resample kernel on the chain's own stream (start / stop events attached to
 * the launches): turn on, run sweeps, sync, read total milliseconds and launch count since it was turned on.
 * every = 0 off, 1 every sweep, n the launches of every n-th sweep only (an event pair costs
 * a few us of stream time per launch, which a sampled measurement keeps out of the total) */
int bmm_chain_profile(bmm_chain* c, int every);
int bmm_chain_profile_read(bmm_chain* c, double* resample_ms, int64_t* resample_launches);
/* batch size in effect */
int64_t bmm_chain_batch(const bmm_chain* c);
/* bytes of dynamic LDS and threads per workgroup the resample kernel uses for this shape */
int bmm_chain_kernel_shape(const bmm_chain* c, int* lds_bytes, int* threads, int* grid_max);
/* ... and its form: lanes of a wave per observation (1, or 2: two lanes share an observation, each scoring half
 * of the categories -- above 32 accumulators, and for launches too short to fill the chip otherwise), and whether
 * the resample workgroups build the table image themselves (small finite-sampler shapes: no table kernel between
 * batches).  Which form runs never changes a chain's values. */
int bmm_chain_kernel_form(const bmm_chain* c, int* lanes_per_observation, int* builds_own_tables);

/* ---- device self-checks used by the parity tests (op: 0 log, 1 exp, 2 div by in2, 3 sqrt,
 * 4 the draw's weight exponential expw; elementwise over n doubles, evaluated on the GPU with the
 * spec arithmetic) */
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
