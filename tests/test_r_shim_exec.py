"""The .Call shim EXECUTED (R is not in the image): bmm-mcmc_amd/r-shim/bmmmcmc_shim.c is compiled together with
tests/r_api_stub/mock_r.c -- a test-only implementation of the few dozen R API functions the shim uses -- and
linked against the real libbmmmcmc_hip.so.  The test plays .Call: it runs R_init_bmmmcmc, looks routines up in
the table the shim registered, passes SEXPs and inspects the lists that come back, the PROTECT depth, R's RNG
stream and what was printed.

Without a GPU: registration as the reference's (src/RcppExports.cpp:137-146), argument checks, the REALSXP /
logical matrix that Rcpp's IntegerMatrix parameter accepts (RcppExports.cpp:15) reaching the library (which then
refuses for want of a device: no CPU fallback), relabel = TRUE stand-alone, PROTECT balance on every path.
With a GPU (-m gpu): the returned chain objects -- names, order, storage modes, dims of R/utils.R's lists
(collapsed_gibbs.cpp:229-243, stickbreaking.cpp:238-254) -- bit-equal to the Python mirror's for the same seed;
set.seed()-style determinism through R's stream; chains = 2; debug = TRUE / set_progress printing the
reference's per-sweep line."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest

import bmm_mcmc_amd as bm
from bmm_mcmc_amd import _capi, build as _build
from test_r_shim import EX_TABLE, REFERENCE_TABLE

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(ROOT, "tests", "r_api_stub")
SHIM = os.path.join(ROOT, "bmm-mcmc_amd", "r-shim", "bmmmcmc_shim.c")
LGLSXP, INTSXP, REALSXP, VECSXP = 10, 13, 14, 19
NA_INT = -2147483648


class MockR:
    def __init__(self, so):
        _capi.lib()  # the product library first, from its own path
        L = self.L = C.CDLL(so)
        for f in ("mock_nil", "mock_int", "mock_real", "mock_lgl", "mock_str", "mock_elt", "mock_call", "mock_data"):
            getattr(L, f).restype = C.c_void_p
        L.mock_int.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int]
        L.mock_real.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int]
        L.mock_call.argtypes = [C.c_char_p, C.c_int, C.c_void_p]
        for f in ("mock_type", "mock_ndim", "mock_protect_depth", "mock_rng_reads", "mock_dynamic_symbols"):
            getattr(L, f).restype = C.c_int
        for f in ("mock_type", "mock_ndim", "mock_length", "mock_data"):
            getattr(L, f).argtypes = [C.c_void_p]
        L.mock_dim.argtypes = [C.c_void_p, C.c_int]
        L.mock_elt.argtypes = [C.c_void_p, C.c_int]
        L.mock_name.argtypes = [C.c_void_p, C.c_int]
        L.mock_name.restype = C.c_char_p
        L.mock_length.restype = C.c_longlong
        L.mock_last_error.restype = C.c_char_p
        L.mock_printed.restype = C.c_char_p
        L.mock_set_seed.argtypes = [C.c_ulonglong]
        L.mock_init()

    # ---- R values
    def sexp(self, v):
        L = self.L
        if v is None:
            return L.mock_nil()
        if isinstance(v, bool):
            return L.mock_lgl(int(v))
        if isinstance(v, (int, np.integer)):
            return self.sexp(np.array([v], dtype=np.int32))
        if isinstance(v, float):
            return self.sexp(np.array([v], dtype=np.float64))
        a = np.asfortranarray(v)
        nr, nc = (a.shape if a.ndim == 2 else (0, 0))
        if a.dtype.kind == "b":   # a logical vector / matrix
            x = self.sexp(a.astype(np.int32))
            L.mock_set_type(C.c_void_p(x), LGLSXP)
            return x
        if a.dtype.kind in "iu":
            a = np.asfortranarray(a, dtype=np.int32)
            return L.mock_int(a.ctypes.data_as(C.c_void_p), a.size, nr, nc)
        a = np.asfortranarray(a, dtype=np.float64)
        return L.mock_real(a.ctypes.data_as(C.c_void_p), a.size, nr, nc)

    def call(self, name, *args):
        tab = (C.c_void_p * len(args))(*[self.sexp(a) for a in args])
        r = self.L.mock_call(name.encode(), len(args), tab)
        depth = self.L.mock_protect_depth()
        assert depth == 0, "PROTECT stack not balanced after %s: depth %d" % (name, depth)
        if not r:
            raise RuntimeError(self.L.mock_last_error().decode())
        return r

    def value(self, x):
        """an R value as Python: atomic vectors as NumPy arrays shaped by their dim, lists as ordered dicts"""
        L = self.L
        t, n = L.mock_type(x), L.mock_length(x)
        dims = tuple(L.mock_dim(x, k) for k in range(L.mock_ndim(x)))
        if t == VECSXP:
            names = [L.mock_name(x, i) for i in range(n)]
            vals = [self.value(L.mock_elt(x, i)) for i in range(n)]
            return dict(zip([s.decode() for s in names], vals)) if names and names[0] is not None else vals
        ct = {INTSXP: C.c_int32, LGLSXP: C.c_int32, REALSXP: C.c_double}[t]
        a = np.ctypeslib.as_array(C.cast(L.mock_data(x), C.POINTER(ct)), shape=(n,)).copy()
        return a.reshape(dims, order="F") if dims else a

    def printed(self):
        s = self.L.mock_printed().decode()
        self.L.mock_clear_printed()
        return s


@pytest.fixture(scope="module")
def R(tmp_path_factory):
    gcc = shutil.which("gcc")
    assert gcc
    if _build.stale(_build.LIB):
        _build.build()
    so = str(tmp_path_factory.mktemp("shim") / "libshim_under_mock_r.so")
    libdir = os.path.dirname(_build.LIB)
    subprocess.run([gcc, "-shared", "-fPIC", "-std=gnu11", "-O1", "-Wall", "-Wno-unused-parameter", "-Wno-cast-function-type",
                    "-I" + STUB, "-I" + os.path.join(ROOT, "include"), os.path.join(STUB, "mock_r.c"), SHIM,
                    "-L" + libdir, "-lbmmmcmc_hip", "-Wl,-rpath," + libdir, "-o", so], check=True, capture_output=True)
    return MockR(so)


@pytest.fixture(scope="module")
def Rfwd(tmp_path_factory):
    """the shim built with -DBMM_SHIM_FORWARD, beside test doubles of what it then links against
    (tests/r_api_stub/fake_glue.c): its relabel = TRUE dispatch and the three forwarded entry points, executed"""
    gcc = shutil.which("gcc")
    if _build.stale(_build.LIB):
        _build.build()
    so = str(tmp_path_factory.mktemp("shimfwd") / "libshim_forward_under_mock_r.so")
    libdir = os.path.dirname(_build.LIB)
    subprocess.run([gcc, "-shared", "-fPIC", "-std=gnu11", "-O1", "-Wall", "-Wno-unused-parameter", "-Wno-cast-function-type",
                    "-DBMM_SHIM_FORWARD", "-I" + STUB, "-I" + os.path.join(ROOT, "include"), os.path.join(STUB, "mock_r.c"),
                    os.path.join(STUB, "fake_glue.c"), SHIM, "-L" + libdir, "-lbmmmcmc_hip", "-Wl,-rpath," + libdir, "-o", so],
                   check=True, capture_output=True)
    return MockR(so)


def test_forward_build_hands_relabel_calls_to_the_glue(Rfwd):
    """-DBMM_SHIM_FORWARD: relabel = TRUE on any of the eight sampler entry points reaches relabel_glue.cpp's
    function for that sampler with the seed, batch and device the shim resolved (here: test doubles that report
    what they were given); the three untouched entry points are the package's own functions."""
    R = Rfwd
    X, z0 = _x(), np.ones(400, dtype=np.int32)

    def fields(r):
        L = R.L
        name_sexp = L.mock_elt(L.mock_elt(r, 0), 0)                  # STRSXP -> CHARSXP
        name = C.string_at(L.mock_data(name_sexp)).decode()
        vals = [R.value(L.mock_elt(r, i)) if L.mock_type(L.mock_elt(r, i)) in (INTSXP, REALSXP, LGLSXP) else None for i in (1, 2, 3)]
        return name, [None if v is None else float(v[0]) for v in vals]

    reads = R.L.mock_rng_reads()
    name, (seed, batch, dev) = fields(R.call("_bmmmcmc_collapsed_gibbs_cpp", X, z0, 10, 2, 0.0, 0.5, 0.5, 1.0, 1.0, 5, True, 3, False))
    assert name == "collapsed" and batch == 0 and dev == 0 and 0 <= seed < 2 ** 53 and seed == int(seed)
    assert R.L.mock_rng_reads() == reads + 1                          # the seed came from R's stream
    name, (seed, batch, dev) = fields(R.call("_bmmmcmc_collapsed_gibbs_dp_ex", X, 10, 0.0, 0.5, 0.5, 1.0, 1.0, 5, True, 3, 30, False,
                                             7.0, 64.0, 1, np.array([2], dtype=np.int32)))
    assert (name, seed, batch, dev) == ("dp", 7.0, 64.0, 2.0)
    pi0, th0 = np.ones(4) / 4, np.full((4, 6), 0.5)
    name, (seed, burnrelabel, dev) = fields(R.call("_bmmmcmc_gibbs_stickbreaking_cpp", X, pi0, th0, 10, 4, 0.0, 0.5, 0.5, 1.0, 1.0, 5,
                                                   True, 3, False))
    assert name == "sb" and burnrelabel == 3 and dev == 0
    name, (seed, _, dev) = fields(R.call("_bmmmcmc_gibbs_ex", X, pi0, th0, 10, 4, 0.0, 0.5, 0.5, 1.0, 1.0, 5, True, 3, False, 11.0, 1, None))
    assert (name, seed, dev) == ("full", 11.0, 0.0)
    with pytest.raises(RuntimeError, match="one chain per call"):
        R.call("_bmmmcmc_collapsed_gibbs_ex", X, np.ones((400, 2), dtype=np.int32), 10, 2, 0.0, 0.5, 0.5, 1.0, 1.0, 5, True, 3, False,
               None, None, 2, None)
    assert fields(R.call("_bmmmcmc_my_lpsolve", 1.0))[0] == "my_lpsolve"          # forwarded, not the error stub
    assert fields(R.call("_bmmmcmc_my_stephens_batch", 1.0, False))[0] == "my_stephens_batch"


def _x(N=400, P=6, seed=3):
    rng = np.random.default_rng(seed)
    return np.asfortranarray((rng.random((N, P)) < 0.4).astype(np.int32))


def test_registration_as_r_sees_it(R):
    rows, i = [], 0
    name, arity = C.c_char_p(), C.c_int()
    while R.L.mock_registered(i, C.byref(name), C.byref(arity)):
        rows.append((name.value.decode(), arity.value))
        i += 1
    assert rows[:7] == list(REFERENCE_TABLE.items())            # src/RcppExports.cpp:137-146, in its order
    assert dict(rows[7:]) == {**EX_TABLE, "_bmmmcmc_set_progress": 1}
    assert R.L.mock_dynamic_symbols() == 0                      # R_useDynamicSymbols(dll, FALSE), :150
    with pytest.raises(RuntimeError, match="not available for .Call"):
        R.call("collapsed_gibbs_relabel", 1)
    with pytest.raises(RuntimeError, match="Incorrect number of arguments"):
        R.call("_bmmmcmc_collapsed_gibbs_cpp", 1, 2)


def test_argument_errors_are_r_errors_and_leave_the_protect_stack_balanced(R):
    X, z0 = _x(), np.ones(400, dtype=np.int32)
    good = [X, z0, 10, 2, 0.0, 0.5, 0.5, 1.0, 1.0, 1, False, 50, False]
    with pytest.raises(RuntimeError, match="data must be a matrix"):
        R.call("_bmmmcmc_collapsed_gibbs_cpp", *([X.ravel()] + good[1:]))
    with pytest.raises(RuntimeError, match="burnin must be smaller"):
        R.call("_bmmmcmc_collapsed_gibbs_cpp", *(good[:9] + [10] + good[10:]))
    with pytest.raises(RuntimeError, match="one label per observation"):
        R.call("_bmmmcmc_collapsed_gibbs_cpp", *([X, z0[:7]] + good[2:]))
    with pytest.raises(RuntimeError, match="seed must be a whole number"):
        R.call("_bmmmcmc_collapsed_gibbs_ex", *(good + [1.5, None, 1, None]))
    with pytest.raises(RuntimeError, match="chains must be between"):
        R.call("_bmmmcmc_collapsed_gibbs_ex", *(good + [None, None, 0, None]))
    with pytest.raises(RuntimeError, match="one device per chain"):
        R.call("_bmmmcmc_collapsed_gibbs_ex", *(good + [None, None, 2, np.array([0], dtype=np.int32)]))
    with pytest.raises(RuntimeError, match="wrong size"):
        R.call("_bmmmcmc_gibbs_stickbreaking_cpp", X, np.ones(3) / 3, np.full((4, 6), 0.5), 10, 4, 0.0, 0.5, 0.5, 1.0, 1.0, 1,
               False, 50, False)
    # relabel = TRUE needs the package's Stephens code beside the shim (-DBMM_SHIM_FORWARD + relabel_glue.cpp)
    for name, args in (("_bmmmcmc_collapsed_gibbs_cpp", good[:10] + [True] + good[11:]),
                       ("_bmmmcmc_collapsed_gibbs_dp_cpp", [X, 10, 0.0, 0.5, 0.5, 1.0, 1.0, 1, True, 50, 30, False])):
        with pytest.raises(RuntimeError, match="BMM_SHIM_FORWARD"):
            R.call(name, *args)
    for name in ("_bmmmcmc_rdirichlet_cpp", "_bmmmcmc_my_lpsolve"):
        with pytest.raises(RuntimeError, match="leaves untouched"):
            R.call(name, 1.0)
    # the progress setter returns the previous setting
    assert R.value(R.call("_bmmmcmc_set_progress", 5))[0] == 0
    assert R.value(R.call("_bmmmcmc_set_progress", 0))[0] == 5
    with pytest.raises(RuntimeError, match="non-negative"):
        R.call("_bmmmcmc_set_progress", -1)


@pytest.mark.skipif(_capi.device_count() > 0, reason="a GPU is present")
def test_numeric_matrix_reaches_the_library_which_has_no_cpu_fallback(R):
    """matrix(c(0, 1, ...)) is REALSXP in R; Rcpp's IntegerMatrix parameter coerces it (RcppExports.cpp:15).
    Here the coerced call gets as far as the library, whose refusal (no device) comes back as an R error."""
    X, z0 = _x(), np.ones(400, dtype=np.int32)
    reads = R.L.mock_rng_reads()
    for data in (X.astype(np.float64), X):
        with pytest.raises(RuntimeError, match="no HIP device"):
            R.call("_bmmmcmc_collapsed_gibbs_cpp", data, z0, 10, 2, 0.0, 0.5, 0.5, 1.0, 1.0, 1, False, 50, False)
    assert R.L.mock_rng_reads() == reads + 2     # one read of R's stream per call: set.seed() fixes the chain


# ---------------------------------------------------------------- on the GPU
@pytest.mark.gpu
def test_collapsed_chain_object_equals_the_python_mirror(R):
    X = _x(3000, 12, 5)
    z0 = np.random.default_rng(1).integers(1, 4, 3000).astype(np.int32)
    want = bm.gibbs_collapsed(X, 12, 3, burnin=2, seed=77, initial_K=z0)
    for data in (X, X.astype(np.float64), X.astype(bool)):     # INTSXP in place; REALSXP / logical coerced
        got = R.value(R.call("_bmmmcmc_collapsed_gibbs_ex", data, z0, 12, 3, 0.0, 0.5, 0.5, 1.0, 1.0, 2, False, 50, False, 77.0, None, 1, None))
        assert list(got) == ["alpha", "permutations", "z", "theta"]            # collapsed_gibbs.cpp:229-243
        assert got["alpha"].shape == (10, 1) and got["alpha"].dtype == np.float64
        assert got["permutations"].shape == (10, 3) and (got["permutations"] == NA_INT).all()
        assert got["z"].shape == (10, 3000) and got["z"].dtype == np.int32
        assert got["theta"].shape == (3, 12, 10)
        assert np.array_equal(got["z"], want["z"]) and np.array_equal(got["theta"], want["theta"], equal_nan=True)
        assert np.array_equal(got["alpha"], want["alpha"])
    # batch = 1 (the reference's scan) through the trailing argument
    b1 = R.value(R.call("_bmmmcmc_collapsed_gibbs_ex", X, z0, 6, 3, 0.0, 0.5, 0.5, 1.0, 1.0, 1, False, 50, False, 77.0, 1.0, 1, None))
    assert np.array_equal(b1["z"], bm.gibbs_collapsed(X, 6, 3, burnin=1, seed=77, batch=1, initial_K=z0)["z"])


@pytest.mark.gpu
def test_reference_arity_entry_points_draw_the_seed_from_rs_stream(R):
    X = _x(2000, 10, 8)
    args = [X, 9, 0.0, 0.5, 0.5, 1.0, 1.0, 2, False, 50, 12, False]       # collapsed_gibbs_dp_cpp, 12 arguments
    R.L.mock_set_seed(2024)
    a = R.value(R.call("_bmmmcmc_collapsed_gibbs_dp_cpp", *args))
    b = R.value(R.call("_bmmmcmc_collapsed_gibbs_dp_cpp", *args))
    R.L.mock_set_seed(2024)                                                # set.seed(2024) again
    c = R.value(R.call("_bmmmcmc_collapsed_gibbs_dp_cpp", *args))
    assert list(a) == ["alpha", "permutations", "z", "theta"] and a["theta"].shape == (12, 10, 7)
    assert np.array_equal(a["z"], c["z"]) and not np.array_equal(a["z"], b["z"])
    assert a["z"].min() >= 1 and a["z"].max() <= 12


@pytest.mark.gpu
def test_explicit_samplers_chains_and_devices(R):
    X = _x(1500, 8, 2)
    rng = np.random.default_rng(3)
    K = 5
    pi0 = np.stack([rng.dirichlet(np.ones(K)) for _ in range(2)], axis=1)               # K x chains
    th0 = np.stack([rng.random(K * 8) for _ in range(2)], axis=1)                       # (K*P) x chains
    one = R.value(R.call("_bmmmcmc_gibbs_stickbreaking_ex", X, pi0[:, 0], th0[:, 0].reshape((K, 8), order="F"), 8, K, 0.0, 0.5,
                         0.5, 1.0, 1.0, 3, False, 50, False, 11.0, 1, None))
    assert list(one) == ["pi", "alpha", "permutations", "z", "theta"]                   # stickbreaking.cpp:238-254
    assert one["pi"].shape == (5, K) and one["theta"].shape == (K, 8, 5) and one["z"].shape == (5, 1500)
    want = bm.gibbs_stickbreaking(X, 8, K, burnin=3, seed=11, initial_pi=pi0[:, 0], initial_theta=th0[:, 0].reshape((K, 8), order="F"))
    assert np.array_equal(one["z"], want["z"]) and np.array_equal(one["pi"], want["pi"])
    two = R.value(R.call("_bmmmcmc_gibbs_ex", X, pi0, th0, 8, K, 0.0, 0.5, 0.5, 1.0, 1.0, 3, False, 50, False, 11.0, 2,
                         np.array([0, 0], dtype=np.int32)))
    assert isinstance(two, list) and len(two) == 2 and list(two[0]) == ["pi", "alpha", "permutations", "z", "theta"]
    w0 = bm.gibbs_full(X, 8, K, burnin=3, seed=11, initial_pi=pi0[:, 0], initial_theta=th0[:, 0].reshape((K, 8), order="F"))
    w1 = bm.gibbs_full(X, 8, K, burnin=3, seed=12, initial_pi=pi0[:, 1], initial_theta=th0[:, 1].reshape((K, 8), order="F"))
    assert np.array_equal(two[0]["z"], w0["z"]) and np.array_equal(two[1]["z"], w1["z"])      # chain c is keyed seed + c


@pytest.mark.gpu
def test_debug_and_set_progress_print_the_references_sample_line(R):
    X = _x(1000, 6, 4)
    z0 = np.ones(1000, dtype=np.int32)
    R.printed()
    R.call("_bmmmcmc_collapsed_gibbs_cpp", X, z0, 6, 2, 0.0, 0.5, 0.5, 1.0, 1.0, 1, False, 50, True)        # debug = TRUE
    assert R.printed() == "".join("Sample %d\n" % j for j in range(2, 7))                                    # collapsed_gibbs.cpp:85
    R.call("_bmmmcmc_collapsed_gibbs_cpp", X, z0, 6, 2, 0.0, 0.5, 0.5, 1.0, 1.0, 1, False, 50, False)
    assert R.printed() == ""
    R.call("_bmmmcmc_set_progress", 3)
    out = R.value(R.call("_bmmmcmc_collapsed_gibbs_dp_cpp", X, 8, 0.0, 0.5, 0.5, 1.0, 1.0, 0, False, 50, 9, False))
    lines = R.printed().splitlines()
    assert [l.split("\t")[0] for l in lines] == ["Sample 4", "Sample 7", "Sample 8"]                         # after sweeps 3, 6, 7
    assert lines[0] == "Sample 4\tK: %d" % len(np.unique(out["z"][3]))                                       # collapsed_gibbs_dp.cpp:99
    R.call("_bmmmcmc_set_progress", 0)
    # a failing run is an R error carrying the library's message, with nothing left protected
    bad = X.copy()
    bad[5, 2] = 7
    with pytest.raises(RuntimeError, match="binary"):
        R.call("_bmmmcmc_collapsed_gibbs_cpp", bad, z0, 6, 2, 0.0, 0.5, 0.5, 1.0, 1.0, 1, False, 50, False)
