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

__all__ = ["gibbs_collapsed", "gibbs_dp", "gibbs_stickbreaking", "gibbs_full", "Chain", "BmmError", "NA_INTEGER",
           "default_batch"]


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


def _no_relabel(relabel):
    if relabel:
        raise NotImplementedError(
            "relabel=TRUE (Stephens 2000b via lp_solve) is host post-processing that stays in the "
            "reference package (src/stephens.cpp, src/my_lpsolve.cpp); this build does not produce "
            "the per-sweep probability matrices it consumes yet (SURVEY.md section 8 f2)")


def _na_perm(S, K):
    return _np.full((S, K), NA_INTEGER, dtype=_np.int32, order="F")  # uninitialised in the reference


def default_batch(sampler, N):
    return int(_capi.lib().bmm_default_batch(_capi.SAMPLER_CODE[sampler], N))


def gibbs_collapsed(data, nsamples, K, alpha=None, beta=0.5, gamma=0.5, a=1, b=1, burnin=None,
                    relabel=False, burnrelabel=50, debug=False, *, seed=None, batch=None, device=0,
                    initial_K=None):
    """Collapsed Gibbs sampler, finite K (R/utils.R:37-47 -> src/collapsed_gibbs.cpp:24).

    Extra keyword-only arguments: `seed` (Philox key; default drawn from the global NumPy
    RNG), `batch` (observations resampled per frozen-statistics batch; 1 = the reference's
    sequential scan; None = library default), `device`, `initial_K` (1-based labels; default
    sampled uniformly as R/utils.R:42 does).
    """
    _no_relabel(relabel)
    X = _capi.as_x(data)
    N, P = X.shape
    nsamples, K = int(nsamples), int(K)
    burnin = _burnin(burnin, nsamples)
    seed = _seed(seed)
    if initial_K is None:
        initial_K = _np.random.default_rng(seed).integers(1, K + 1, N)
    z0 = _np.ascontiguousarray(initial_K, dtype=_np.int32)
    if z0.shape != (N,):
        raise ValueError("initial_K must have one label per observation")
    S = nsamples - burnin
    z = _np.zeros((S, N), dtype=_np.int32, order="F")
    theta = _np.zeros((K, P, S), order="F")
    al = _np.zeros((S, 1), order="F")
    rc = _capi.lib().bmm_collapsed_run(
        _capi.vp(X), _C.c_int64(N), _C.c_int(P), _capi.vp(z0), _C.c_int(nsamples), _C.c_int(K),
        _C.c_double(0.0 if alpha is None else alpha), _C.c_double(beta), _C.c_double(gamma),
        _C.c_double(a), _C.c_double(b), _C.c_int(burnin), _C.c_int64(0 if batch is None else batch),
        _C.c_uint64(seed), _C.c_int(device), _capi.vp(z), _capi.vp(theta), _capi.vp(al))
    _capi.check(rc)
    return {"alpha": al, "permutations": _na_perm(S, K), "z": z, "theta": theta}


def gibbs_dp(data, nsamples, alpha=None, a=1, b=1, beta=0.5, gamma=0.5, burnin=None, relabel=False,
             burnrelabel=50, maxK=30, debug=False, *, seed=None, batch=None, device=0):
    """Collapsed Gibbs sampler with a Dirichlet-process prior, truncated at maxK
    (R/utils.R:23-30 -> src/collapsed_gibbs_dp.cpp:27)."""
    _no_relabel(relabel)
    X = _capi.as_x(data)
    N, P = X.shape
    nsamples, maxK = int(nsamples), int(maxK)
    burnin = _burnin(burnin, nsamples)
    seed = _seed(seed)
    S = nsamples - burnin
    z = _np.zeros((S, N), dtype=_np.int32, order="F")
    theta = _np.zeros((maxK, P, S), order="F")
    al = _np.zeros((S, 1), order="F")
    rc = _capi.lib().bmm_dp_run(
        _capi.vp(X), _C.c_int64(N), _C.c_int(P), _C.c_int(nsamples),
        _C.c_double(0.0 if alpha is None else alpha), _C.c_double(beta), _C.c_double(gamma),
        _C.c_double(a), _C.c_double(b), _C.c_int(burnin), _C.c_int(maxK),
        _C.c_int64(0 if batch is None else batch), _C.c_uint64(seed), _C.c_int(device), _capi.vp(z),
        _capi.vp(theta), _capi.vp(al))
    _capi.check(rc)
    return {"alpha": al, "permutations": _na_perm(S, maxK), "z": z, "theta": theta}


def gibbs_stickbreaking(data, nsamples, maxK, alpha=None, beta=0.5, gamma=0.5, a=1, b=1, burnin=None,
                        relabel=False, burnrelabel=50, debug=False, *, seed=None, device=0,
                        initial_pi=None, initial_theta=None):
    """Blocked Gibbs sampler, truncated stick-breaking prior
    (R/utils.R:95-107 -> src/stickbreaking.cpp:10)."""
    _no_relabel(relabel)
    X = _capi.as_x(data)
    N, P = X.shape
    nsamples, maxK = int(nsamples), int(maxK)
    burnin = _burnin(burnin, nsamples)
    seed = _seed(seed)
    rng = _np.random.default_rng(seed)
    if initial_pi is None:  # R/utils.R:98-100
        initial_pi = _np.exp(rng.random(maxK))
        initial_pi = initial_pi / initial_pi.sum()
    if initial_theta is None:  # R/utils.R:103
        initial_theta = rng.random(maxK * P).reshape((maxK, P), order="F")
    pi0 = _np.ascontiguousarray(initial_pi, dtype=_np.float64)
    th0 = _np.asfortranarray(initial_theta, dtype=_np.float64)
    if pi0.shape != (maxK,) or th0.shape != (maxK, P):
        raise ValueError("initial_pi must have maxK entries and initial_theta be maxK x P")
    S = nsamples - burnin
    z = _np.zeros((S, N), dtype=_np.int32, order="F")
    theta = _np.zeros((maxK, P, S), order="F")
    al = _np.zeros((S, 1), order="F")
    pi = _np.zeros((S, maxK), order="F")
    rc = _capi.lib().bmm_sb_run(
        _capi.vp(X), _C.c_int64(N), _C.c_int(P), _capi.vp(pi0), _capi.vp(th0), _C.c_int(nsamples),
        _C.c_int(maxK), _C.c_double(0.0 if alpha is None else alpha), _C.c_double(beta),
        _C.c_double(gamma), _C.c_double(a), _C.c_double(b), _C.c_int(burnin), _C.c_uint64(seed),
        _C.c_int(device), _capi.vp(pi), _capi.vp(z), _capi.vp(theta), _capi.vp(al))
    _capi.check(rc)
    return {"pi": pi, "alpha": al, "permutations": _na_perm(S, maxK), "z": z, "theta": theta}


def gibbs_full(data, nsamples, K, alpha=None, beta=0.5, gamma=0.5, a=1, b=1, burnin=None, relabel=False,
               burnrelabel=50, debug=False, *, seed=None, device=0, initial_pi=None, initial_theta=None):
    """Full (uncollapsed) Gibbs sampler, finite K (R/utils.R:64-78 -> src/full_gibbs.cpp:32)."""
    _no_relabel(relabel)
    X = _capi.as_x(data)
    N, P = X.shape
    nsamples, K = int(nsamples), int(K)
    burnin = _burnin(burnin, nsamples)
    seed = _seed(seed)
    rng = _np.random.default_rng(seed)
    if initial_pi is None:  # R/utils.R:68-70
        initial_pi = _np.exp(rng.random(K))
        initial_pi = initial_pi / initial_pi.sum()
    if initial_theta is None:  # R/utils.R:74
        initial_theta = rng.random(K * P).reshape((K, P), order="F")
    pi0 = _np.ascontiguousarray(initial_pi, dtype=_np.float64)
    th0 = _np.asfortranarray(initial_theta, dtype=_np.float64)
    if pi0.shape != (K,) or th0.shape != (K, P):
        raise ValueError("initial_pi must have K entries and initial_theta be K x P")
    S = nsamples - burnin
    z = _np.zeros((S, N), dtype=_np.int32, order="F")
    theta = _np.zeros((K, P, S), order="F")
    al = _np.zeros((S, 1), order="F")
    pi = _np.zeros((S, K), order="F")
    rc = _capi.lib().bmm_full_run(
        _capi.vp(X), _C.c_int64(N), _C.c_int(P), _capi.vp(pi0), _capi.vp(th0), _C.c_int(nsamples),
        _C.c_int(K), _C.c_double(0.0 if alpha is None else alpha), _C.c_double(beta), _C.c_double(gamma),
        _C.c_double(a), _C.c_double(b), _C.c_int(burnin), _C.c_uint64(seed), _C.c_int(device),
        _capi.vp(pi), _capi.vp(z), _capi.vp(theta), _capi.vp(al))
    _capi.check(rc)
    return {"pi": pi, "alpha": al, "permutations": _na_perm(S, K), "z": z, "theta": theta}


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

    def shard_resample(self):
        _capi.check(_capi.lib().bmm_chain_shard_resample(self._h))

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
        return {"lds_bytes": lds.value, "threads": th.value, "grid_max": g.value}
