"""ctypes front end of the CPU oracle (oracle/bmm_oracle.c).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never by the bmm-mcmc_amd package.  PARITY UNPINNED for RNG-dependent
outputs (see bmm_oracle.h).

Array conventions are R's: X is N x P (any integer array; passed Fortran-ordered int32),
z is S x N with 1-based labels, theta is K x P x S, pi is S x K, alpha is S x 1.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
NA_INT = -2147483648

_i32p = np.ctypeslib.ndpointer(np.int32, flags="F_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="F_CONTIGUOUS")


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = [os.path.join(_HERE, f) for f in ("bmm_oracle.c", "bmm_oracle.h", "exp256_table.h", "Makefile")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        L.oracle_last_error.restype = C.c_char_p
        L.oracle_u01.restype = C.c_double
        L.oracle_u01.argtypes = [C.c_uint32, C.c_uint32]
        L.oracle_z_uniform.restype = C.c_double
        L.oracle_z_uniform.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32]
        L.oracle_log.restype = C.c_double
        L.oracle_log.argtypes = [C.c_double]
        L.oracle_exp.restype = C.c_double
        L.oracle_exp.argtypes = [C.c_double]
        L.oracle_expw.restype = C.c_double
        L.oracle_expw.argtypes = [C.c_double]
        L.oracle_u52.restype = C.c_double
        L.oracle_u52.argtypes = [C.c_uint32, C.c_uint32]
        L.oracle_rgamma.restype = C.c_double
        L.oracle_rgamma.argtypes = [C.c_double, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
        L.oracle_rbeta.restype = C.c_double
        L.oracle_rbeta.argtypes = [C.c_double, C.c_double, C.c_uint64, C.c_uint32, C.c_uint32,
                                   C.c_uint32, C.c_uint32, C.c_uint32]
        L.oracle_update_alpha.restype = C.c_double
        L.oracle_update_alpha.argtypes = [C.c_double, C.c_double, C.c_double, C.c_double, C.c_int,
                                          C.c_uint64, C.c_uint32]
        L.oracle_time_sweeps.restype = C.c_double
        L.oracle_time_sweeps.argtypes = [C.c_int, _i32p, C.c_int64, C.c_int, C.c_int, C.c_int,
                                         C.c_int64, C.c_uint64, C.c_int]
        L.oracle_group_width.restype = C.c_int
        L.oracle_group_width_own.restype = C.c_int
        L.oracle_group_width_for.restype = C.c_int
        L.oracle_group_width_for.argtypes = [C.c_int, C.c_int, C.c_int]
        _LIB = L
    return _LIB


def _check(rc):
    if rc != 0:
        raise RuntimeError(lib().oracle_last_error().decode())


def _x(data):
    X = np.asfortranarray(np.asarray(data), dtype=np.int32)
    if X.ndim != 2:
        raise ValueError("data must be N x P")
    return X


# ---------------------------------------------------------------- numerics
def philox4x32_10(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().oracle_philox4x32_10(c, k, o)
    return tuple(int(v) for v in o)


def philox2x32_10(ctr, key):
    c = (C.c_uint32 * 2)(*ctr)
    o = (C.c_uint32 * 2)()
    lib().oracle_philox2x32_10(c, C.c_uint32(key), o)
    return tuple(int(v) for v in o)


def expw_array(x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.empty_like(x)
    lib().oracle_expw_array(x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), C.c_int64(x.size))
    return y


def z_uniform(seed, i, sweep):
    return lib().oracle_z_uniform(seed, i, sweep)


def log_array(x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.empty_like(x)
    lib().oracle_log_array(x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), C.c_int64(x.size))
    return y


def exp_array(x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.empty_like(x)
    lib().oracle_exp_array(x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), C.c_int64(x.size))
    return y


def rgamma(shape, seed, c0, sweep, stream):
    return lib().oracle_rgamma(shape, seed, c0, sweep, stream)


def rbeta(p, q, seed, c0a, c0b, sweep, stream_a, stream_b):
    return lib().oracle_rbeta(p, q, seed, c0a, c0b, sweep, stream_a, stream_b)


def update_alpha(alpha_old, a, b, N, K, seed, sweep):
    return lib().oracle_update_alpha(alpha_old, a, b, float(N), K, seed, sweep)


# ---------------------------------------------------------------- conditionals
def _cond(fn, X, z, i, K, alpha, beta, gamma, ncat):
    X = _x(X)
    N, P = X.shape
    z = np.ascontiguousarray(z, dtype=np.int32)
    a = np.zeros(ncat)
    b = np.zeros(ncat)
    getattr(lib(), fn)(X.ctypes.data_as(C.c_void_p), C.c_int64(N), C.c_int(P), z.ctypes.data_as(C.c_void_p),
                       C.c_int64(i), C.c_int(K), C.c_double(alpha), C.c_double(beta), C.c_double(gamma),
                       a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p))
    return a, b


def collapsed_cond(X, z, i, K, alpha, beta, gamma, spec=False):
    """(raw or score, normalised) conditional of observation i (0-based); z 1-based."""
    return _cond("oracle_collapsed_cond_spec" if spec else "oracle_collapsed_cond_literal",
                 X, z, i, K, alpha, beta, gamma, K)


def dp_cond(X, z, i, K, alpha, beta, gamma, spec=False):
    """(log-weights, normalised) over K existing clusters then the new-cluster option."""
    return _cond("oracle_dp_cond_spec" if spec else "oracle_dp_cond_literal",
                 X, z, i, K, alpha, beta, gamma, K + 1)


def sb_cond(X, i, pi, theta, spec=False):
    X = _x(X)
    N, P = X.shape
    pi = np.ascontiguousarray(pi, dtype=np.float64)
    K = pi.size
    theta = np.asfortranarray(theta, dtype=np.float64)
    assert theta.shape == (K, P)
    a = np.zeros(K)
    b = np.zeros(K)
    fn = "oracle_sb_cond_spec" if spec else "oracle_sb_cond_literal"
    getattr(lib(), fn)(X.ctypes.data_as(C.c_void_p), C.c_int64(N), C.c_int(P), C.c_int64(i), C.c_int(K),
                       pi.ctypes.data_as(C.c_void_p), theta.ctypes.data_as(C.c_void_p),
                       a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p))
    return a, b


# ---------------------------------------------------------------- samplers
def _outs(S, N, K, P):
    return (np.zeros((S, N), dtype=np.int32, order="F"), np.zeros((K, P, S), order="F"),
            np.zeros((S, 1), order="F"))


def _vp(a):
    return a.ctypes.data_as(C.c_void_p)


def collapsed(X, z0, nsamples, K, alpha, beta, gamma, a, b, burnin, seed, batch=1, literal=False):
    X = _x(X)
    N, P = X.shape
    z0 = np.ascontiguousarray(z0, dtype=np.int32)
    S = nsamples - burnin
    z, th, al = _outs(S, N, K, P)
    L = lib()
    if literal:
        rc = L.oracle_collapsed_literal(_vp(X), C.c_int64(N), C.c_int(P), _vp(z0), C.c_int(nsamples), C.c_int(K),
                                        C.c_double(alpha), C.c_double(beta), C.c_double(gamma), C.c_double(a),
                                        C.c_double(b), C.c_int(burnin), C.c_uint64(seed), _vp(z), _vp(th), _vp(al))
    else:
        rc = L.oracle_collapsed_run(_vp(X), C.c_int64(N), C.c_int(P), _vp(z0), C.c_int(nsamples), C.c_int(K),
                                    C.c_double(alpha), C.c_double(beta), C.c_double(gamma), C.c_double(a),
                                    C.c_double(b), C.c_int(burnin), C.c_int64(batch), C.c_uint64(seed),
                                    _vp(z), _vp(th), _vp(al))
    _check(rc)
    return {"alpha": al, "z": z, "theta": th}


def dp(X, nsamples, alpha, beta, gamma, a, b, burnin, maxK, seed, batch=1, literal=False):
    X = _x(X)
    N, P = X.shape
    S = nsamples - burnin
    z, th, al = _outs(S, N, maxK, P)
    L = lib()
    if literal:
        rc = L.oracle_dp_literal(_vp(X), C.c_int64(N), C.c_int(P), C.c_int(nsamples), C.c_double(alpha),
                                 C.c_double(beta), C.c_double(gamma), C.c_double(a), C.c_double(b),
                                 C.c_int(burnin), C.c_int(maxK), C.c_uint64(seed), _vp(z), _vp(th), _vp(al))
    else:
        rc = L.oracle_dp_run(_vp(X), C.c_int64(N), C.c_int(P), C.c_int(nsamples), C.c_double(alpha),
                             C.c_double(beta), C.c_double(gamma), C.c_double(a), C.c_double(b), C.c_int(burnin),
                             C.c_int(maxK), C.c_int64(batch), C.c_uint64(seed), _vp(z), _vp(th), _vp(al))
    _check(rc)
    return {"alpha": al, "z": z, "theta": th}


def counts_summary(sampler, X, z0, nsamples, K, alpha, beta, gamma, a, b, burnin, seed, batch=1):
    """The collapsed / dp `*_run` chain without the S x N label trace: {"nk": (S, K) cluster sizes after every
    kept sweep, "theta": K x P x S, "alpha": S x 1, "z_last": labels after the last sweep (1-based)}."""
    X = _x(X)
    N, P = X.shape
    code = {"collapsed": 0, "dp": 1}[sampler]
    S = nsamples - burnin
    nk = np.zeros((S, K), dtype=np.int32)
    th = np.zeros((K, P, S), order="F")
    al = np.zeros((S, 1), order="F")
    zl = np.zeros(N, dtype=np.int32)
    z0p = _vp(np.ascontiguousarray(z0, dtype=np.int32)) if z0 is not None else None
    rc = lib().oracle_counts_summary(C.c_int(code), _vp(X), C.c_int64(N), C.c_int(P), z0p, C.c_int(nsamples),
                                     C.c_int(K), C.c_double(alpha), C.c_double(beta), C.c_double(gamma),
                                     C.c_double(a), C.c_double(b), C.c_int(burnin), C.c_int64(batch),
                                     C.c_uint64(seed), _vp(nk), _vp(th), _vp(al), _vp(zl))
    _check(rc)
    return {"nk": nk, "theta": th, "alpha": al, "z_last": zl}


def full(X, pi0, theta0, nsamples, K, alpha, beta, gamma, a, b, burnin, seed, literal=False):
    """gibbs_cpp (full_gibbs.cpp): the stick-breaking z-step with a Dirichlet pi draw."""
    return stickbreaking(X, pi0, theta0, nsamples, K, alpha, beta, gamma, a, b, burnin, seed, literal=literal,
                         _fn=("oracle_full_literal" if literal else "oracle_full_run"))


def stickbreaking(X, pi0, theta0, nsamples, maxK, alpha, beta, gamma, a, b, burnin, seed, literal=False, _fn=None):
    X = _x(X)
    N, P = X.shape
    S = nsamples - burnin
    pi0 = np.ascontiguousarray(pi0, dtype=np.float64)
    theta0 = np.asfortranarray(theta0, dtype=np.float64)
    assert pi0.shape == (maxK,) and theta0.shape == (maxK, P)
    z, th, al = _outs(S, N, maxK, P)
    pi = np.zeros((S, maxK), order="F")
    fn = getattr(lib(), _fn) if _fn else (lib().oracle_sb_literal if literal else lib().oracle_sb_run)
    rc = fn(_vp(X), C.c_int64(N), C.c_int(P), _vp(pi0), _vp(theta0), C.c_int(nsamples), C.c_int(maxK),
            C.c_double(alpha), C.c_double(beta), C.c_double(gamma), C.c_double(a), C.c_double(b),
            C.c_int(burnin), C.c_uint64(seed), _vp(pi), _vp(z), _vp(th), _vp(al))
    _check(rc)
    return {"pi": pi, "alpha": al, "z": z, "theta": th}


def time_sweeps(sampler, X, K, sweeps, batch, seed, nthreads):
    """Wall seconds for `nthreads` independent chains of `sweeps` sweeps each (one per thread)."""
    X = _x(X)
    N, P = X.shape
    code = {"collapsed": 0, "dp": 1, "stickbreaking": 2}[sampler]
    t = lib().oracle_time_sweeps(code, X, N, P, K, sweeps, batch, seed, nthreads)
    if t < 0:
        raise RuntimeError(lib().oracle_last_error().decode())
    return t
