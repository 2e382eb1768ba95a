"""No-GPU checks of the boundary: the library loads, exports every symbol the header
declares, and refuses to run without a device (no CPU fallback)."""
import os
import re

import numpy as np
import pytest

import bmm_mcmc_amd as bm
from bmm_mcmc_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "bmm_mcmc.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(bmm_[a-z_0-9]+)\s*\(", src)))


def test_header_and_loader_agree():
    assert _declared() == sorted(_capi.SYMBOLS)


def test_library_exports_every_declared_symbol():
    L = _capi.lib()
    for s in _declared():
        assert getattr(L, s) is not None


def test_spec_constants_match_oracle(oracle):
    assert _capi.lib().bmm_spec_group_width() == oracle.lib().oracle_group_width()
    assert _capi.lib().bmm_spec_group_width_own() == oracle.lib().oracle_group_width_own()


def test_default_batch_policy():
    assert bm.default_batch("stickbreaking", 1000) == 1000
    assert bm.default_batch("collapsed", 100) == 12
    # a pure function of (sampler, N): floor(N/8) (N/16 dp) below 2^16 observations, floor(N/4) from there
    # on, at least 1, never rounded -- the ratios the tolerance fixtures are held at
    assert bm.default_batch("collapsed", 10 ** 7) == 2_500_000 and bm.default_batch("dp", 10 ** 7) == 2_500_000
    for N in (1, 7, 100, 12345, 2 ** 16 - 1):
        assert bm.default_batch("collapsed", N) == max(1, N // 8)
        assert bm.default_batch("dp", N) == max(1, N // 16)
    for N in (2 ** 16, 10 ** 5, 4 * 10 ** 5, 10 ** 6, 8 * 3 * 2 ** 18 + 8, 2 ** 31 + 5):
        assert bm.default_batch("collapsed", N) == N // 4
        assert bm.default_batch("dp", N) == N // 4
    assert bm.default_batch("collapsed", 10 ** 6) == 250000
    assert bm.default_batch("dp", 10 ** 6) == 250000
    assert bm.default_batch("dp", 1600) == 100
    assert bm.default_batch("dp", 3) == 1
    assert bm.default_batch("full", 77) == 77


def test_argument_validation_needs_no_gpu():
    X = np.zeros((10, 3), dtype=np.int32)
    with pytest.raises(NotImplementedError, match="relabel"):
        bm.gibbs_collapsed(X, 10, 2, relabel=True)
    with pytest.raises(ValueError, match="binary"):      # caught before a narrowing cast could hide it;
        bm.gibbs_collapsed(X.astype(np.int64) + 2 ** 32, 10, 2)  # int32 input is checked by the library
    with pytest.raises(ValueError, match="burnin"):
        bm.gibbs_dp(X, 10, burnin=10)


def test_new_entry_points_validate_before_touching_a_device():
    import ctypes as C
    L = _capi.lib()
    tab = (C.c_void_p * 1)(None)
    rc = L.bmm_multi_run(C.c_int(0), C.c_int(0), None, None, C.c_int64(10), C.c_int(3), None, None, None, C.c_int(5),
                         C.c_int(2), C.c_double(1.0), C.c_double(0.5), C.c_double(0.5), C.c_double(1), C.c_double(1),
                         C.c_int(1), C.c_int64(0), C.c_uint64(1), None, tab, tab, tab)
    assert rc == 1 and b"n_chains" in L.bmm_last_error()
    rc = L.bmm_multi_run(C.c_int(9), C.c_int(1), None, None, C.c_int64(10), C.c_int(3), None, None, None, C.c_int(5),
                         C.c_int(2), C.c_double(1.0), C.c_double(0.5), C.c_double(0.5), C.c_double(1), C.c_double(1),
                         C.c_int(1), C.c_int64(0), C.c_uint64(1), None, tab, tab, tab)
    assert rc == 1 and b"sampler" in L.bmm_last_error()
    assert L.bmm_multi_selfcheck(C.c_int(0), None, C.c_int64(16)) == 1
    assert L.bmm_chains_sweeps(None, C.c_int(0), C.c_int(1)) == 1
    assert L.bmm_chain_share_data(None, None) == 1 and L.bmm_chain_planes(None, None, None) == 1
    # the stated tolerance is one number in the header and in the Python mirror
    hdr = open(os.path.join(ROOT, "include", "bmm_mcmc.h")).read()
    assert "#define BMM_TOL_PROPORTIONS %g" % bm.TOL_PROPORTIONS in hdr
    assert "#define BMM_TOL_THETA %g" % bm.TOL_THETA in hdr


@pytest.mark.skipif(_capi.device_count() > 0, reason="a GPU is present")
def test_no_cpu_fallback_without_a_device():
    X = np.zeros((10, 3), dtype=np.int32)
    with pytest.raises(bm.BmmError, match="no HIP device"):
        bm.gibbs_collapsed(X, 10, 2, seed=1)
    with pytest.raises(bm.BmmError, match="no HIP device"):
        bm.Chain("dp", 10, 3, 5)


def test_chain_summary_derives_what_plot_gibbs_plots():
    """R/utils.R:147-190: per-sample label shares, labels above the threshold after sample 1"""
    import numpy as np
    import bmm_mcmc_amd as bm
    z = np.array([[1, 1, 1, 1], [1, 1, 2, 2], [3, 1, 1, 1], [-2147483648, 1, 2, 2]], dtype=np.int32)
    theta = np.arange(3 * 2 * 4, dtype=float).reshape(3, 2, 4)
    s = bm.chain_summary({"z": z, "theta": theta}, cluster_threshold=0.3)
    assert np.allclose(s["proportions"][1], [0.5, 0.5, 0.0])
    assert np.allclose(s["proportions"][3], [1 / 3, 2 / 3, 0.0])
    assert s["clusters"] == [1, 2]  # label 3 never exceeds 0.3; sample 1 (all ones) is not consulted
    assert np.isnan(s["theta"][2]).all() and np.array_equal(s["theta"][:2], theta[:2])


def test_group_width_rule_is_the_oracles(oracle):
    """the per-shape group width is part of the arithmetic: library and oracle state the rule independently"""
    L, O = _capi.lib(), oracle.lib()
    L.bmm_spec_group_width_for.restype = O.oracle_group_width_for.restype = __import__("ctypes").c_int
    seen = set()
    for sampler in range(4):
        for K in (1, 2, 3, 4, 5, 12, 19, 20, 21, 24, 29, 30, 31, 32, 33, 40, 47, 48, 50, 56, 63, 64, 65, 100):
            for P in (1, 5, 20, 31, 32, 33, 50, 64, 65, 96, 100, 101, 110, 127, 128, 129, 513):
                w = L.bmm_spec_group_width_for(sampler, K, P)
                assert w == O.oracle_group_width_for(sampler, K, P), (sampler, K, P)
                seen.add(w)
    assert seen == {4, 5}
    assert L.bmm_spec_group_width_for(0, 20, 100) == 5 and L.bmm_spec_group_width_for(1, 30, 50) == 5  # C5, C3
    assert L.bmm_spec_group_width_for(2, 50, 50) == 5 and L.bmm_spec_group_width_for(0, 3, 20) == 5    # C4, C2
    assert L.bmm_spec_group_width_for(0, 20, 128) == 4 and L.bmm_spec_group_width_for(2, 64, 64) == 4
    assert L.bmm_spec_group_width_for(7, 3, 3) == -1


def test_multi_run_placement_is_pure_bookkeeping():
    """bmm_multi_plan: what bmm_multi_run does with a device table, without touching a device -- the distinct
    devices in first-use order (the RCCL broadcast list, root first) and, per chain, the chain holding its
    device's copy of the bit planes.  No multi-GPU box has run this path yet; its bookkeeping is held here."""
    import ctypes as C
    L = _capi.lib()

    def plan(devs):
        n = len(devs) if devs is not None else 5
        tab = (C.c_int * n)(*devs) if devs is not None else None
        nd, out, hold = C.c_int(0), (C.c_int * n)(), (C.c_int * n)()
        rc = L.bmm_multi_plan(C.c_int(n), tab, C.byref(nd), out, hold)
        return rc, list(out)[:nd.value], list(hold)

    assert plan([0, 1, 0, 1]) == (0, [0, 1], [0, 1, 0, 1])
    assert plan([3, 3, 3]) == (0, [3], [0, 0, 0])
    assert plan([2, 0, 1, 0, 2, 7]) == (0, [2, 0, 1, 7], [0, 1, 2, 1, 0, 5])
    assert plan(list(range(8))) == (0, list(range(8)), list(range(8)))           # the 8-GPU node: one chain per GPU
    assert plan(None) == (0, [0], [0, 0, 0, 0, 0])                               # devices = NULL: all on device 0
    rc, _, _ = plan([0, -1])
    assert rc == 1 and b"devices[1]" in L.bmm_last_error()
    assert L.bmm_multi_plan(C.c_int(0), None, None, None, None) == 1


# ---- the host-side ends of a *_run call, without a device (test-variant exports of the debug-hooks build)
def _dbg_cdll():
    import ctypes
    from bmm_mcmc_amd import build
    if build.stale(build.LIB_DBG):
        build.build(debug_variant=True)
    return ctypes.CDLL(build.LIB_DBG)


@pytest.mark.parametrize("N,P", [(1, 1), (63, 7), (4097, 32), (70001, 50), (33000, 100), (5000, 129)])
def test_host_pack_equals_numpy_bit_planes(N, P):
    """what the host's threads make of R's column-major int32 matrix: feature d at bit d % 32 of word d / 32, planes
    [w][N], bits past P zero -- the layout k_pack_bits writes on the device (tests/test_gpu_fullsize.py holds
    the two to each other through a chain)"""
    import ctypes
    lib = _dbg_cdll()
    rng = np.random.default_rng(N + P)
    X = np.asfortranarray((rng.random((N, P)) < 0.4).astype(np.int32))
    W = (P + 31) // 32
    out = np.full((W, N), 0xFFFFFFFF, dtype=np.uint32)
    seen = ctypes.c_uint32(7)
    rc = lib.bmm_dbg_host_pack(X.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(N), ctypes.c_int(P),
                               out.ctypes.data_as(ctypes.c_void_p), ctypes.byref(seen))
    assert rc == 0 and seen.value == int(X.max())
    want = np.zeros((W, N), dtype=np.uint32)
    for d in range(P):
        want[d // 32] |= X[:, d].astype(np.uint32) << np.uint32(d % 32)
    np.testing.assert_array_equal(out, want)
    # a cell that is not 0/1 shows in the OR the caller checks
    X[N // 2, P - 1] = 6
    lib.bmm_dbg_host_pack(X.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(N), ctypes.c_int(P),
                          out.ctypes.data_as(ctypes.c_void_p), ctypes.byref(seen))
    assert seen.value & ~1


@pytest.mark.parametrize("n,offset", [(0, 0), (1, 0), (15, 1), (16, 0), (1000, 3), (262147, 1), (1 << 20, 0)])
def test_host_label_copies_widen_and_mark_unassigned(n, offset):
    """trace_out's host half: one-byte labels (0 = unassigned) widened to R's int32 with NA_integer_, int32 labels
    copied as they are; destinations that do not start on a 16-byte boundary, lengths that are no multiple of
    the vector width, and the same threads reused job after job"""
    import ctypes
    lib = _dbg_cdll()
    rng = np.random.default_rng(n + offset)
    NA = -2147483648
    src8 = rng.integers(0, 255, size=n, dtype=np.uint8)
    if n > 4:
        src8[:3] = (0, 254, 1)
    buf = np.full(n + 8, 12345, dtype=np.int32)
    dst = buf[offset:offset + n]
    assert lib.bmm_dbg_host_labels(src8.ctypes.data_as(ctypes.c_void_p), 1, dst.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(n), 3) == 0
    want = src8.astype(np.int32)
    want[src8 == 0] = NA
    np.testing.assert_array_equal(dst, want)
    assert (buf[:offset] == 12345).all() and (buf[offset + n:] == 12345).all()
    src32 = rng.integers(1, 400, size=n, dtype=np.int32)
    assert lib.bmm_dbg_host_labels(src32.ctypes.data_as(ctypes.c_void_p), 0, dst.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(n), 2) == 0
    np.testing.assert_array_equal(dst, src32)
    assert (buf[:offset] == 12345).all() and (buf[offset + n:] == 12345).all()
