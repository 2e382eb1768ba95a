"""bmm_mcmc_amd -- MI355X-native cluster-allocation path of the bmm-mcmc Gibbs samplers.

Host-side mirror of the reference's R wrappers (R/utils.R:23-47, 95-107): the same
function names, argument meaning, defaults and returned objects, over the C ABI in
include/bmm_mcmc.h.  (R is not installed in the build image; the R wrappers and the
.Call shim a maintainer would use are in bmm-mcmc_amd/R and bmm-mcmc_amd/r-shim, see
INTEGRATION.md.)  Returned arrays are laid out like the R objects: `z` is S x N with
1-based labels, `theta` is K x P x S, `alpha` is S x 1, `pi` is S x maxK.

There is no CPU fallback: without the built HIP library and a gfx950 device every
sampler call raises.
"""
import ctypes as _C

import numpy as _np

from . import _capi
from ._capi import BmmError, NA_INTEGER
from .rdata import read_rdata_matrix  # the package's bundled data sets (data/*.RData) without R

__all__ = ["gibbs_collapsed", "gibbs_dp", "gibbs_stickbreaking", "gibbs_full", "Chain", "BmmError", "NA_INTEGER", "set_progress",
           "default_batch", "sweep_chains", "broadcast_planes", "read_rdata_matrix", "chain_summary", "TOL_PROPORTIONS", "TOL_THETA"]

# include/bmm_mcmc.h: the stated tolerance of a batch > 1 against the reference's sequential scan
TOL_PROPORTIONS = 0.015
TOL_THETA = 0.05


def _seed(seed):
    # R draws through the session RNG, so set.seed() fixes the chain; here the global
    # NumPy RNG plays that part when no seed is given.
    if seed is None:
        return int(_np.random.randint(0, 2 ** 31 - 1))
    return int(seed) & 0xFFFFFFFFFFFFFFFF


def _burnin(burnin, nsamples):
    if burnin is None:
        burnin = int(round(0.1 * nsamples))  # R/utils.R:25,40,97 (round half even, as R)
    burnin = int(burnin)
    if not 0 <= burnin < nsamples:
        raise ValueError("burnin must be in [0, nsamples)")
    return burnin


def _na_perm(S, K):
    return _np.full((S, K), NA_INTEGER, dtype=_np.int32, order="F")  # uninitialised in the reference


def default_batch(sampler, N):
    return int(_capi.lib().bmm_default_batch(_capi.SAMPLER_CODE[sampler], N))


# ---------------------------------------------------------------- relabel = TRUE
_PROBS_FN = _C.CFUNCTYPE(_C.c_int, _C.c_void_p, _C.c_int, _C.POINTER(_C.c_double))


class _Hooks(_C.Structure):  # bmm_relabel_hooks
    _fields_ = [("burnrelabel", _C.c_int), ("probs_batch", _C.c_void_p), ("batch_done", _PROBS_FN),
                ("on_sample", _PROBS_FN), ("user", _C.c_void_p)]


class _Relabel:
    """The reference's relabel = TRUE bookkeeping (src/collapsed_gibbs.cpp:187-201, 215-217, 232-243) around
    host code that stays what it is: `stephens` supplies the two functions of src/stephens.h,
        stephens.batch(p)          p: (N, K, burnrelabel) cube      -> Q (N, K)       my_stephens_batch
        stephens.online(Q, p, j)   p: (N, K) of sweep j             -> (perm, Q_new)  my_stephens_online
    (perm 0-based, as arma::Row<int>).  The per-sweep probability matrices come from the device."""

    def __init__(self, stephens, N, K, nsamples, burnin, burnrelabel):
        if stephens is None:
            raise NotImplementedError(
                "relabel=TRUE runs Stephens' relabelling (src/stephens.cpp, src/my_lpsolve.cpp) on the host, and "
                "that code stays in the reference package: pass stephens=<object with batch(p) and "
                "online(Q, p, j)> bound to it; this build supplies the probability matrices it consumes")
        self.st, self.N, self.K, self.burnin = stephens, N, K, burnin
        self.S = nsamples - burnin
        self.W = max(0, int(burnrelabel))
        self.cube = _np.zeros((N, K, max(self.W, 1)), order="F")
        self.Q = None
        self.perms = _np.full((self.S, K), NA_INTEGER, dtype=_np.int32, order="F")
        self.error = None
        self._batch_cb = _PROBS_FN(self._batch_done)
        self._sample_cb = _PROBS_FN(self._on_sample)
        self.hooks = _Hooks(self.W, self.cube.ctypes.data_as(_C.c_void_p), self._batch_cb, self._sample_cb, None)

    def _batch_done(self, user, j, probs):
        try:
            self.Q = _np.asarray(self.st.batch(self.cube[:, :, :self.W]), dtype=_np.float64)
            return 0
        except Exception as e:  # an exception must not unwind through the C frames
            self.error = e
            return 1

    def _on_sample(self, user, j, probs):
        try:
            p = _np.ctypeslib.as_array(probs, shape=(self.K, self.N)).T  # N x K column-major, no copy
            perm, self.Q = self.st.online(self.Q, p, j)
            self.perms[j - self.burnin] = _np.asarray(perm, dtype=_np.int32)
            return 0
        except Exception as e:
            self.error = e
            return 1

    def ref(self):
        return _C.byref(self.hooks)

    def finish(self, rc, out):
        if self.error is not None:
            raise self.error
        _capi.check(rc)
        z, theta = out["z"], out["theta"]
        zr = _np.empty_like(z)
        thr = _np.empty_like(theta)
        for s in range(self.S):
            pm = self.perms[s]
            zr[s] = pm[z[s] - 1] + 1          # z_out_relabelled(j,i) = perm(z_out(j,i)-1) + 1   (:198)
            thr[pm, :, s] = theta[:, :, s]    # thetas_relab(perm(k), d, j) = theta(k, d, j)     (:216)
        out.update(permutations=self.perms, z=zr, theta=thr, z_original=z, theta_original=theta)
        return out


# ---------------------------------------------------------------- progress ("Sample j", as the reference prints it)
_PROGRESS_FN = _C.CFUNCTYPE(_C.c_int, _C.c_void_p, _C.c_int, _C.c_int, _C.c_int)
_progress_every = 0


def set_progress(every=0):
    """The reference prints "Sample j" at every sweep (src/collapsed_gibbs.cpp:85; gibbs_dp adds the number of
    clusters, src/collapsed_gibbs_dp.cpp:99).  Here a run is silent unless asked: set_progress(100) prints that
    line every 100 sweeps of the single-chain calls that follow, set_progress(0) turns it off; debug=True prints
    every sweep.  Returns the previous setting (the R wrapper is bmm_progress(), R/gibbs.R)."""
    global _progress_every
    old, _progress_every = _progress_every, max(0, int(every))
    return old


def _print_progress(user, sample, nsamples, k_used):
    print("Sample %d" % sample if k_used < 0 else "Sample %d\tK: %d" % (sample, k_used), flush=True)
    return 0


_progress_cb = _PROGRESS_FN(_print_progress)


class _progress:
    """the hook around one run: every sweep with debug=True, else what set_progress asked for"""

    def __init__(self, debug):
        self.every = 1 if debug else _progress_every

    def __enter__(self):
        if self.every:
            _capi.lib().bmm_set_progress(_progress_cb, None, _C.c_int(self.every))

    def __exit__(self, *exc):
        if self.every:
            _capi.lib().bmm_set_progress(_PROGRESS_FN(0), None, _C.c_int(0))


def _clamp_burnrelabel(burnrelabel, burnin):
    return int(round(0.1 * burnin)) if burnrelabel > burnin else int(burnrelabel)  # R/utils.R:26,41,72


def _ptr_table(arrays, ctype=_C.c_void_p):
    return (ctype * len(arrays))(*[a.ctypes.data for a in arrays])


def _devices(chains, devices):
    if devices is None:
        return None
    dv = [int(d) for d in devices]
    if len(dv) != chains:
        raise ValueError("devices must name one device per chain")
    return (_C.c_int * chains)(*dv)


def _multi(sampler, X, chains, devices, z0s, pi0s, th0s, nsamples, K, alpha, beta, gamma, a, b, burnin, batch,
           seed, with_pi):
    """bmm_multi_run: `chains` independent chains (seed + c) over one upload of the data."""
    N, P = X.shape
    S = nsamples - burnin
    zs = [_np.empty((S, N), dtype=_np.int32, order="F") for _ in range(chains)]  # every cell is written by the library
    ths = [_np.zeros((K, P, S), order="F") for _ in range(chains)]
    als = [_np.zeros((S, 1), order="F") for _ in range(chains)]
    pis = [_np.zeros((S, K), order="F") for _ in range(chains)] if with_pi else None
    rc = _capi.lib().bmm_multi_run(
        _C.c_int(_capi.SAMPLER_CODE[sampler]), _C.c_int(chains), _devices(chains, devices), _capi.vp(X),
        _C.c_int64(N), _C.c_int(P), _ptr_table(z0s) if z0s else None, _ptr_table(pi0s) if pi0s else None,
        _ptr_table(th0s) if th0s else None, _C.c_int(nsamples), _C.c_int(K),
        _C.c_double(0.0 if alpha is None else alpha), _C.c_double(beta), _C.c_double(gamma), _C.c_double(a),
        _C.c_double(b), _C.c_int(burnin), _C.c_int64(0 if batch is None else batch), _C.c_uint64(seed),
        _ptr_table(pis) if with_pi else None, _ptr_table(zs), _ptr_table(ths), _ptr_table(als))
    _capi.check(rc)
    out = []
    for c in range(chains):
        d = {"alpha": als[c], "permutations": _na_perm(S, K), "z": zs[c], "theta": ths[c]}
        if with_pi:
            d = {"pi": pis[c], **d}
        out.append(d)
    return out


def gibbs_collapsed(data, nsamples, K, alpha=None, beta=0.5, gamma=0.5, a=1, b=1, burnin=None,
                    relabel=False, burnrelabel=50, debug=False, *, seed=None, batch=None, device=0,
                    initial_K=None, chains=1, devices=None, stephens=None):
    """Collapsed Gibbs sampler, finite K (R/utils.R:37-47 -> src/collapsed_gibbs.cpp:24).

    Extra keyword-only arguments: `seed` (Philox key; default drawn from the global NumPy
    RNG), `batch` (observations resampled per frozen-statistics batch; 1 = the reference's
    sequential scan; None = library default), `device`, `initial_K` (1-based labels; default
    sampled uniformly as R/utils.R:42 does), `chains` / `devices` (several independent chains,
    seed + c, in one call: returns a list of chain objects), `stephens` (the host relabelling
    code for relabel=True, see _Relabel).
    """
    X = _capi.as_x(data)
    N, P = X.shape
    nsamples, K = int(nsamples), int(K)
    burnin = _burnin(burnin, nsamples)
    seed = _seed(seed)
    chains = int(chains)
    if chains > 1:
        if relabel:
            raise NotImplementedError("relabel=TRUE is offered per chain (chains=1)")
        z0s = [_np.ascontiguousarray(_np.random.default_rng(seed + c).integers(1, K + 1, N), dtype=_np.int32)
               for c in range(chains)] if initial_K is None else [_np.ascontiguousarray(z, dtype=_np.int32) for z in initial_K]
        return _multi("collapsed", X, chains, devices, z0s, None, None, nsamples, K, alpha, beta, gamma, a, b,
                      burnin, batch, seed, False)
    if initial_K is None:
        initial_K = _np.random.default_rng(seed).integers(1, K + 1, N)
    z0 = _np.ascontiguousarray(initial_K, dtype=_np.int32)
    if z0.shape != (N,):
        raise ValueError("initial_K must have one label per observation")
    S = nsamples - burnin
    rl = _Relabel(stephens, N, K, nsamples, burnin, _clamp_burnrelabel(burnrelabel, burnin)) if relabel else None
    z = _np.empty((S, N), dtype=_np.int32, order="F")  # every cell is written by the library
    theta = _np.zeros((K, P, S), order="F")
    al = _np.zeros((S, 1), order="F")
    with _progress(debug):
        rc = _capi.lib().bmm_collapsed_run_probs(
            _capi.vp(X), _C.c_int64(N), _C.c_int(P), _capi.vp(z0), _C.c_int(nsamples), _C.c_int(K),
            _C.c_double(0.0 if alpha is None else alpha), _C.c_double(beta), _C.c_double(gamma),
            _C.c_double(a), _C.c_double(b), _C.c_int(burnin), _C.c_int64(0 if batch is None else batch),
            _C.c_uint64(seed), _C.c_int(device), _capi.vp(z), _capi.vp(theta), _capi.vp(al),
            rl.ref() if rl else None)
    out = {"alpha": al, "permutations": _na_perm(S, K), "z": z, "theta": theta}
    if rl:
        return rl.finish(rc, out)
    _capi.check(rc)
    return out


def gibbs_dp(data, nsamples, alpha=None, a=1, b=1, beta=0.5, gamma=0.5, burnin=None, relabel=False,
             burnrelabel=50, maxK=30, debug=False, *, seed=None, batch=None, device=0, chains=1, devices=None,
             stephens=None):
    """Collapsed Gibbs sampler with a Dirichlet-process prior, truncated at maxK
    (R/utils.R:23-30 -> src/collapsed_gibbs_dp.cpp:27)."""
    X = _capi.as_x(data)
    N, P = X.shape
    nsamples, maxK = int(nsamples), int(maxK)
    burnin = _burnin(burnin, nsamples)
    seed = _seed(seed)
    if int(chains) > 1:
        if relabel:
            raise NotImplementedError("relabel=TRUE is offered per chain (chains=1)")
        return _multi("dp", X, int(chains), devices, None, None, None, nsamples, maxK, alpha, beta, gamma, a, b,
                      burnin, batch, seed, False)
    S = nsamples - burnin
    rl = _Relabel(stephens, N, maxK, nsamples, burnin, _clamp_burnrelabel(burnrelabel, burnin)) if relabel else None
    z = _np.empty((S, N), dtype=_np.int32, order="F")  # every cell is written by the library
    theta = _np.zeros((maxK, P, S), order="F")
    al = _np.zeros((S, 1), order="F")
    with _progress(debug):
        rc = _capi.lib().bmm_dp_run_probs(
            _capi.vp(X), _C.c_int64(N), _C.c_int(P), _C.c_int(nsamples),
            _C.c_double(0.0 if alpha is None else alpha), _C.c_double(beta), _C.c_double(gamma),
            _C.c_double(a), _C.c_double(b), _C.c_int(burnin), _C.c_int(maxK),
            _C.c_int64(0 if batch is None else batch), _C.c_uint64(seed), _C.c_int(device), _capi.vp(z),
            _capi.vp(theta), _capi.vp(al), rl.ref() if rl else None)
    out = {"alpha": al, "permutations": _na_perm(S, maxK), "z": z, "theta": theta}
    if rl:
        return rl.finish(rc, out)
    _capi.check(rc)
    return out


def _explicit(sampler, fn, clamp, data, nsamples, K, alpha, beta, gamma, a, b, burnin, relabel, burnrelabel, seed,
              device, initial_pi, initial_theta, chains, devices, stephens, debug=False):
    X = _capi.as_x(data)
    N, P = X.shape
    nsamples, K = int(nsamples), int(K)
    burnin = _burnin(burnin, nsamples)
    seed = _seed(seed)
    chains = int(chains)

    def start(sd, pi, th):
        rng = _np.random.default_rng(sd)
        if pi is None:  # R/utils.R:68-70, 98-100
            pi = _np.exp(rng.random(K))
            pi = pi / pi.sum()
        if th is None:  # R/utils.R:74, 103
            th = rng.random(K * P).reshape((K, P), order="F")
        pi = _np.ascontiguousarray(pi, dtype=_np.float64)
        th = _np.asfortranarray(th, dtype=_np.float64)
        if pi.shape != (K,) or th.shape != (K, P):
            raise ValueError("initial_pi must have K entries and initial_theta be K x P")
        return pi, th

    if chains > 1:
        if relabel:
            raise NotImplementedError("relabel=TRUE is offered per chain (chains=1)")
        st = [start(seed + c, initial_pi[c] if initial_pi is not None else None,
                    initial_theta[c] if initial_theta is not None else None) for c in range(chains)]
        return _multi(sampler, X, chains, devices, None, [p for p, _ in st], [t for _, t in st], nsamples, K,
                      alpha, beta, gamma, a, b, burnin, None, seed, True)
    pi0, th0 = start(seed, initial_pi, initial_theta)
    S = nsamples - burnin
    W = _clamp_burnrelabel(burnrelabel, burnin) if clamp else int(burnrelabel)  # R/utils.R:97-101 has no clamp
    rl = _Relabel(stephens, N, K, nsamples, burnin, W) if relabel else None
    z = _np.empty((S, N), dtype=_np.int32, order="F")  # every cell is written by the library
    theta = _np.zeros((K, P, S), order="F")
    al = _np.zeros((S, 1), order="F")
    pi = _np.zeros((S, K), order="F")
    with _progress(debug):
        rc = getattr(_capi.lib(), fn)(
            _capi.vp(X), _C.c_int64(N), _C.c_int(P), _capi.vp(pi0), _capi.vp(th0), _C.c_int(nsamples),
            _C.c_int(K), _C.c_double(0.0 if alpha is None else alpha), _C.c_double(beta),
            _C.c_double(gamma), _C.c_double(a), _C.c_double(b), _C.c_int(burnin), _C.c_uint64(seed),
            _C.c_int(device), _capi.vp(pi), _capi.vp(z), _capi.vp(theta), _capi.vp(al), rl.ref() if rl else None)
    out = {"pi": pi, "alpha": al, "permutations": _na_perm(S, K), "z": z, "theta": theta}
    if rl:
        return rl.finish(rc, out)
    _capi.check(rc)
    return out


def gibbs_stickbreaking(data, nsamples, maxK, alpha=None, beta=0.5, gamma=0.5, a=1, b=1, burnin=None,
                        relabel=False, burnrelabel=50, debug=False, *, seed=None, device=0, initial_pi=None,
                        initial_theta=None, chains=1, devices=None, stephens=None):
    """Blocked Gibbs sampler, truncated stick-breaking prior (R/utils.R:95-107 ->
    src/stickbreaking.cpp:10).  The z-step is exactly parallel, so there is no batch."""
    return _explicit("stickbreaking", "bmm_sb_run_probs", False, data, nsamples, maxK, alpha, beta, gamma, a, b,
                     burnin, relabel, burnrelabel, seed, device, initial_pi, initial_theta, chains, devices, stephens, debug)


def gibbs_full(data, nsamples, K, alpha=None, beta=0.5, gamma=0.5, a=1, b=1, burnin=None, relabel=False,
               burnrelabel=50, debug=False, *, seed=None, device=0, initial_pi=None, initial_theta=None, chains=1,
               devices=None, stephens=None):
    """Full (uncollapsed) Gibbs sampler, finite K (R/utils.R:64-78 -> src/full_gibbs.cpp:32)."""
    return _explicit("full", "bmm_full_run_probs", True, data, nsamples, K, alpha, beta, gamma, a, b, burnin,
                     relabel, burnrelabel, seed, device, initial_pi, initial_theta, chains, devices, stephens, debug)


class Chain:
    """One chain resident on one GPU (bmm_chain_* in include/bmm_mcmc.h): the data matrix
    and the chain state stay in HBM between `sweeps()` calls."""

    _CODE = _capi.SAMPLER_CODE

    X_LAYOUT = {"bits": 0, "int32": 1}

    def __init__(self, sampler, N, P, K, alpha=None, beta=0.5, gamma=0.5, a=1, b=1, batch=None, seed=0,
                 device=0, x_layout=None):
        self._h = _C.c_void_p()
        self.sampler, self.N, self.P, self.K = sampler, int(N), int(P), int(K)
        self._keep = None
        rc = _capi.lib().bmm_chain_create(
            _C.byref(self._h), _C.c_int(self._CODE[sampler]), _C.c_int64(N), _C.c_int(P), _C.c_int(K),
            _C.c_double(0.0 if alpha is None else alpha), _C.c_double(beta), _C.c_double(gamma),
            _C.c_double(a), _C.c_double(b), _C.c_int64(0 if batch is None else batch),
            _C.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF), _C.c_int(device))
        _capi.check(rc)
        if x_layout is not None:  # "bits" (default: X packed once into bit planes) or "int32" (as handed over)
            _capi.check(_capi.lib().bmm_chain_set_x_layout(self._h, _C.c_int(self.X_LAYOUT[x_layout])))

    def x_layout(self):
        v = _C.c_int(0)
        _capi.check(_capi.lib().bmm_chain_get_x_layout(self._h, _C.byref(v)))
        return "int32" if v.value else "bits"

    def close(self):
        if self._h:
            _capi.lib().bmm_chain_destroy(self._h)
            self._h = _C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_data(self, X):
        X = _capi.as_x(X)
        if X.shape != (self.N, self.P):
            raise ValueError("data shape does not match the chain")
        _capi.check(_capi.lib().bmm_chain_set_data_host(self._h, _capi.vp(X)))

    def set_data_device(self, ptr, keepalive=None):
        """Borrow an int32 column-major N x P matrix already on this device (e.g. a torch
        tensor's data_ptr()); `keepalive` is held so the owner outlives the chain."""
        self._keep = keepalive
        _capi.check(_capi.lib().bmm_chain_set_data_device(self._h, _C.c_void_p(int(ptr))))

    def share_data(self, other):
        """Share the bit planes `other` (same device, N, P) already holds: several chains over one copy
        (reference-counted in the library: chains may be closed in any order)."""
        _capi.check(_capi.lib().bmm_chain_share_data(self._h, other._h))

    def planes(self):
        """(device address, words) of the chain's bit planes, allocated on first call."""
        a, n = _C.c_void_p(), _C.c_int64(0)
        _capi.check(_capi.lib().bmm_chain_planes(self._h, _C.byref(a), _C.byref(n)))
        return a.value, n.value

    def planes_filled(self):
        _capi.check(_capi.lib().bmm_chain_planes_filled(self._h))

    def set_initial_labels(self, z1):
        z1 = _np.ascontiguousarray(z1, dtype=_np.int32)
        if z1.shape != (self.N,):
            raise ValueError("one label per observation")
        _capi.check(_capi.lib().bmm_chain_set_initial_labels(self._h, _capi.vp(z1)))

    def set_initial_params(self, pi, theta):
        pi = _np.ascontiguousarray(pi, dtype=_np.float64)
        theta = _np.asfortranarray(theta, dtype=_np.float64)
        if pi.shape != (self.K,) or theta.shape != (self.K, self.P):
            raise ValueError("pi must have K entries and theta be K x P")
        _capi.check(_capi.lib().bmm_chain_set_initial_params(self._h, _capi.vp(pi), _capi.vp(theta)))

    def sweeps(self, n):
        _capi.check(_capi.lib().bmm_chain_sweeps(self._h, _C.c_int(n)))

    # -- one chain over several ranks (stick-breaking / full): see multi.ShardedChain
    def set_shard(self, N_total, first_row):
        _capi.check(_capi.lib().bmm_chain_set_shard(self._h, _C.c_int64(N_total), _C.c_int64(first_row)))

    def shard_resample(self, wait=True):
        fn = _capi.lib().bmm_chain_shard_resample if wait else _capi.lib().bmm_chain_shard_resample_async
        _capi.check(fn(self._h))

    def stream(self):
        """The chain's HIP stream handle (for ordering a caller's collective behind it)."""
        st = _C.c_void_p()
        _capi.check(_capi.lib().bmm_chain_stream(self._h, _C.byref(st)))
        return st.value

    def shard_deltas(self):
        """Device addresses of the int32 statistic deltas (dNk: K, dS: K*P) to be summed over ranks."""
        a, b = _C.c_void_p(), _C.c_void_p()
        _capi.check(_capi.lib().bmm_chain_shard_deltas(self._h, _C.byref(a), _C.byref(b)))
        return a.value, b.value

    def shard_finish(self):
        _capi.check(_capi.lib().bmm_chain_shard_finish(self._h))

    def sweep_probs(self):
        """One more sweep; returns the (N, K) matrix of allocation probabilities it drew from (the
        input of the reference's host-side relabelling)."""
        out = _np.zeros((self.N, self.K), order="F")
        _capi.check(_capi.lib().bmm_chain_sweep_probs(self._h, _capi.vp(out)))
        return out

    def sweeps_counts(self, n):
        """n more sweeps; returns the (n, K) cluster sizes after each, computed on the device."""
        out = _np.zeros((n, self.K), dtype=_np.int32)
        _capi.check(_capi.lib().bmm_chain_sweeps_counts(self._h, _C.c_int(n), _capi.vp(out)))
        return out

    def sync(self):
        _capi.check(_capi.lib().bmm_chain_sync(self._h))

    @property
    def batch(self):
        return int(_capi.lib().bmm_chain_batch(self._h))

    @property
    def sweep_index(self):
        return int(_capi.lib().bmm_chain_sweep_index(self._h))

    def labels(self):
        z = _np.zeros(self.N, dtype=_np.int32)
        _capi.check(_capi.lib().bmm_chain_get_labels(self._h, _capi.vp(z)))
        return z

    def counts(self):
        Nk = _np.zeros(self.K, dtype=_np.int32)
        S = _np.zeros((self.K, self.P), dtype=_np.int32)
        _capi.check(_capi.lib().bmm_chain_get_counts(self._h, _capi.vp(Nk), _capi.vp(S)))
        return Nk, S

    def alpha(self):
        v = _C.c_double(0.0)
        _capi.check(_capi.lib().bmm_chain_get_alpha(self._h, _C.byref(v)))
        return v.value

    def params(self):
        pi = _np.zeros(self.K)
        theta = _np.zeros((self.K, self.P), order="F")
        _capi.check(_capi.lib().bmm_chain_get_params(self._h, _capi.vp(pi), _capi.vp(theta)))
        return pi, theta

    def profile(self, every=1):
        """Time the resample launches of every `every`-th sweep with HIP events (0/False: off)."""
        _capi.check(_capi.lib().bmm_chain_profile(self._h, _C.c_int(int(every))))

    def profile_read(self):
        ms, n = _C.c_double(0.0), _C.c_int64(0)
        _capi.check(_capi.lib().bmm_chain_profile_read(self._h, _C.byref(ms), _C.byref(n)))
        return ms.value, n.value

    def kernel_shape(self):
        lds, th, g = _C.c_int(0), _C.c_int(0), _C.c_int(0)
        _capi.check(_capi.lib().bmm_chain_kernel_shape(self._h, _C.byref(lds), _C.byref(th), _C.byref(g)))
        lanes, own = _C.c_int(0), _C.c_int(0)
        _capi.check(_capi.lib().bmm_chain_kernel_form(self._h, _C.byref(lanes), _C.byref(own)))
        return {"lds_bytes": lds.value, "threads": th.value, "grid_max": g.value,
                "lanes_per_observation": lanes.value, "builds_own_tables": bool(own.value)}


def sweep_chains(chains, n):
    """n more sweeps of every chain in `chains`, one host thread per chain (bmm_chains_sweeps): chains on
    one device overlap on their streams.  Returns without waiting; sync each chain afterwards."""
    tab = (_C.c_void_p * len(chains))(*[c._h.value for c in chains])
    _capi.check(_capi.lib().bmm_chains_sweeps(tab, _C.c_int(len(chains)), _C.c_int(int(n))))


def broadcast_planes(chains):
    """Resident chains on different devices of this process, one per device: chains[0] holds the data, the
    others receive its bit planes with one RCCL broadcast (bmm_chains_broadcast_planes)."""
    tab = (_C.c_void_p * len(chains))(*[c._h.value for c in chains])
    _capi.check(_capi.lib().bmm_chains_broadcast_planes(tab, _C.c_int(len(chains))))


def chain_summary(obj, cluster_threshold=0.1):
    """The numbers `plot_gibbs` draws from a returned chain object (R/utils.R:147-190), without the plots:
    `proportions` S x K, the share of the observations holding each label in each kept sample (labels that
    are NA -- the DP and explicit samplers' row 0 at burnin = 0 -- do not count); `clusters`, the 1-based
    labels that exceed `cluster_threshold` in some sample after the first (plot_gibbs drops sample 1), in
    order of first appearance; `theta` K x P x S restricted to those labels (others NaN), as the theta
    panel shows it."""
    z = _np.asarray(obj["z"])
    th = _np.asarray(obj["theta"], dtype=_np.float64)
    K = th.shape[0]
    S, N = z.shape
    props = _np.zeros((S, K))
    for s in range(S):
        lab = z[s]
        lab = lab[(lab >= 1) & (lab <= K)]
        if lab.size:
            props[s] = _np.bincount(lab - 1, minlength=K) / lab.size
    clusters = []
    for s in range(1, S):
        for k in _np.nonzero(props[s] > cluster_threshold)[0]:
            if k + 1 not in clusters:
                clusters.append(int(k) + 1)
    shown = _np.full_like(th, _np.nan)
    for k in clusters:
        shown[k - 1] = th[k - 1]
    return {"proportions": props, "clusters": clusters, "theta": shown}
