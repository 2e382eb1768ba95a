"""Round-3 additions at the drop-in boundary, through the C ABI on the GPU: the progress hook of the *_run
entry points (the reference's per-sweep "Sample j" line, collapsed_gibbs.cpp:85, collapsed_gibbs_dp.cpp:99),
the two ends of a run (host-side validate + pack on the way in, blocked label trace on the way out, narrow and
wide labels), the DP hand-off's new-cluster column for an observation that sits alone in its cluster
(collapsed_gibbs_dp.cpp:113-128, 169-170, 193), and the debug-hooks build's label-range check."""
import ctypes as C

import numpy as np
import pytest

import bmm_mcmc_amd as bm
from bmm_mcmc_amd import _capi
from util import load_dataset, synth

pytestmark = pytest.mark.gpu

PROGRESS_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int)


def _z0(N, K, seed):
    return np.random.default_rng(seed).integers(1, K + 1, N).astype(np.int32)


class _Progress:
    def __init__(self, every, stop_at=None):
        self.calls, self.stop_at = [], stop_at
        self._cb = PROGRESS_FN(self._fn)
        self.every = every

    def _fn(self, user, sample, nsamples, k_used):
        self.calls.append((sample, nsamples, k_used))
        return 1 if self.stop_at is not None and sample >= self.stop_at else 0

    def __enter__(self):
        _capi.check(_capi.lib().bmm_set_progress(self._cb, None, C.c_int(self.every)))
        return self

    def __exit__(self, *exc):
        _capi.lib().bmm_set_progress(PROGRESS_FN(0), None, C.c_int(0))


def test_progress_hook_reports_every_mth_sweep_and_the_chain_is_unchanged(oracle):
    X, _, _, _ = synth(6000, 20, 3, 4)
    z0 = _z0(6000, 3, 1)
    want = oracle.collapsed(X, z0, 12, 3, 0.0, 0.5, 0.5, 1, 1, 2, seed=5, batch=750)
    with _Progress(4) as p:
        got = bm.gibbs_collapsed(X, 12, 3, burnin=2, seed=5, batch=750, initial_K=z0)
    assert np.array_equal(got["z"], want["z"]) and np.array_equal(got["theta"], want["theta"], equal_nan=True)
    # 11 sweeps (j = 1..11), a mark after sweeps 4, 8 and 11: the reference's "Sample j + 1"
    assert p.calls == [(5, 12, -1), (9, 12, -1), (12, 12, -1)]
    silent = bm.gibbs_collapsed(X, 12, 3, burnin=2, seed=5, batch=750, initial_K=z0)   # hook removed again
    assert np.array_equal(silent["z"], want["z"])
    # DP: the hook gets the number of clusters in use after the sweep (collapsed_gibbs_dp.cpp:99 prints K)
    with _Progress(1) as p:
        got = bm.gibbs_dp(X, 8, burnin=0, maxK=12, seed=3, batch=500)
    assert [c[0] for c in p.calls] == list(range(2, 9))
    for (sample, _, k_used) in p.calls:
        assert k_used == len(np.unique(got["z"][sample - 1]))


def test_python_mirror_prints_the_sample_line(capfd):
    """debug=True prints "Sample j" every sweep as the reference does (collapsed_gibbs.cpp:85; with K for gibbs_dp,
    collapsed_gibbs_dp.cpp:99); bm.set_progress(M) every M sweeps; silent otherwise"""
    X, _, _, _ = synth(1500, 8, 2, 4)
    bm.gibbs_collapsed(X, 5, 2, burnin=1, seed=5)
    assert "Sample" not in capfd.readouterr().out
    bm.gibbs_collapsed(X, 5, 2, burnin=1, seed=5, debug=True)
    assert capfd.readouterr().out == "".join("Sample %d\n" % j for j in range(2, 6))
    assert bm.set_progress(2) == 0
    out = bm.gibbs_dp(X, 6, burnin=0, maxK=7, seed=5)
    lines = capfd.readouterr().out.splitlines()
    assert [l.split("\t")[0] for l in lines] == ["Sample 3", "Sample 5", "Sample 6"]
    assert lines[0] == "Sample 3\tK: %d" % len(np.unique(out["z"][2]))
    bm.gibbs_stickbreaking(X, 4, 3, burnin=1, seed=5, debug=True)
    assert capfd.readouterr().out == "Sample 2\nSample 3\nSample 4\n"
    assert bm.set_progress(0) == 2


def test_progress_hook_can_stop_the_run():
    X, _, _, _ = synth(2000, 10, 2, 4)
    with _Progress(2, stop_at=5) as p:
        with pytest.raises(bm.BmmError, match="progress hook stopped the run after sample 5") as e:
            bm.gibbs_collapsed(X, 40, 2, burnin=1, seed=5)
    assert e.value.code == 7 and [c[0] for c in p.calls] == [3, 5]
    assert bm.gibbs_collapsed(X, 4, 2, burnin=1, seed=1)["z"].shape == (3, 2000)   # and the library lives on


def test_pools_can_be_released_and_come_back(oracle):
    X, _, _, _ = synth(4000, 10, 3, 2)
    z0 = _z0(4000, 3, 5)
    want = oracle.collapsed(X, z0, 6, 3, 0.0, 0.5, 0.5, 1, 1, 1, seed=2, batch=500)
    for _ in range(3):
        got = bm.gibbs_collapsed(X, 6, 3, burnin=1, seed=2, batch=500, initial_K=z0)
        assert np.array_equal(got["z"], want["z"])
        assert _capi.lib().bmm_release_pools() == 0      # pinned staging and idle streams go; the next call re-makes them


def test_run_phases_and_host_threads_are_reported():
    X, _, _, _ = synth(50000, 40, 4, 2)
    bm.gibbs_collapsed(X, 30, 4, burnin=10, seed=2)
    ms = (C.c_double * 6)()
    _capi.check(_capi.lib().bmm_last_run_phases(ms))
    assert all(v >= 0 for v in ms) and sum(ms) > 0 and ms[0] > 0 and ms[4] > 0
    assert 1 <= _capi.lib().bmm_host_threads() <= 16


@pytest.mark.parametrize("K", [254, 255, 300])
def test_label_trace_narrow_and_wide(oracle, K):
    """up to 254 labels leave the device as one byte each and are widened by the host copy; more travel as
    int32 -- the same S x N matrix either way, NA where the reference never writes (row 0 at burnin = 0)"""
    X, _, _, _ = synth(1500, 12, 3, 6)
    z0 = _z0(1500, K, 2)
    z0[:3] = [K, 1, K]                      # the largest label is in the trace
    got = bm.gibbs_collapsed(X, 5, K, alpha=2.0, burnin=0, seed=3, batch=400, initial_K=z0)
    want = oracle.collapsed(X, z0, 5, K, 2.0, 0.5, 0.5, 1, 1, 0, seed=3, batch=400)
    assert np.array_equal(got["z"], want["z"]) and got["z"].max() == K
    got = bm.gibbs_dp(X, 4, alpha=1.0, burnin=0, maxK=K, seed=3, batch=400)
    want = oracle.dp(X, 4, 1.0, 0.5, 0.5, 1, 1, 0, K, seed=3, batch=400)
    assert np.array_equal(got["z"], want["z"]) and (got["z"][0] == bm.NA_INTEGER).all()


def test_many_blocks_of_the_label_trace(oracle):
    """S x N large enough for several 8 MiB blocks of the outgoing trace, with a ragged last one"""
    N, P, K = 300_007, 8, 3
    X, _, _, _ = synth(N, P, K, 12)
    z0 = _z0(N, K, 2)
    got = bm.gibbs_collapsed(X, 41, K, burnin=1, seed=9, initial_K=z0)          # S = 40: 209 715 rows per block
    want = oracle.collapsed(X, z0, 41, K, 0.0, 0.5, 0.5, 1, 1, 1, seed=9, batch=bm.default_batch("collapsed", N))
    assert np.array_equal(got["z"], want["z"])
    assert np.array_equal(got["theta"], want["theta"]) and np.array_equal(got["alpha"], want["alpha"])


def test_host_side_validation_names_the_problem():
    X, _, _, _ = synth(70000, 9, 2, 1)
    bad = X.copy()
    bad[69999, 8] = 3
    z0 = _z0(70000, 2, 1)
    z, th, al = np.zeros((1, 70000), np.int32, order="F"), np.zeros((2, 9, 1), order="F"), np.zeros((1, 1), order="F")
    L = _capi.lib()
    rc = L.bmm_collapsed_run(_capi.vp(bad), C.c_int64(70000), C.c_int(9), _capi.vp(z0), C.c_int(2), C.c_int(2),
                             C.c_double(0), C.c_double(.5), C.c_double(.5), C.c_double(1), C.c_double(1), C.c_int(1),
                             C.c_int64(0), C.c_uint64(1), C.c_int(0), _capi.vp(z), _capi.vp(th), _capi.vp(al))
    assert rc == 1 and b"binary" in L.bmm_last_error()
    z0[12345] = 3
    rc = L.bmm_collapsed_run(_capi.vp(X), C.c_int64(70000), C.c_int(9), _capi.vp(z0), C.c_int(2), C.c_int(2),
                             C.c_double(0), C.c_double(.5), C.c_double(.5), C.c_double(1), C.c_double(1), C.c_int(1),
                             C.c_int64(0), C.c_uint64(1), C.c_int(0), _capi.vp(z), _capi.vp(th), _capi.vp(al))
    assert rc == 1 and b"initialK[12345] = 3 outside 1..2" in L.bmm_last_error()


def test_dp_new_cluster_mass_of_a_singleton_goes_under_its_own_label(oracle):
    """An observation alone in its cluster frees its label when it is taken out (collapsed_gibbs_dp.cpp:113-128),
    so unused_clusters.top() -- the label a new cluster takes and the column its probability is filed under
    (:169-170, :193) -- is that label when it is smaller than every other free one.  The draw has always
    opened that label (k_resample); the emitted matrix files the mass there too."""
    X = load_dataset("K2_N100_P5")
    N, maxK, alpha = 100, 40, 6.0
    found = 0
    for seed in range(1, 30):
        with bm.Chain("dp", N, 5, maxK, alpha=alpha, batch=N, seed=seed) as ch:
            ch.set_data(X)
            ch.sweeps(3)
            zb = ch.labels()
            probs = ch.sweep_probs()
            za = ch.labels()
        size = np.bincount(zb - 1, minlength=maxK)
        free = int(np.flatnonzero(size == 0)[0])
        for i in range(N):
            own = zb[i] - 1
            _, norm = oracle.dp_cond(X, zb, i, maxK, alpha, 0.5, 0.5, spec=True)
            want = norm[:maxK].copy()
            lbl = own if size[own] == 1 and own < free else free
            want[lbl] = norm[maxK]           # an empty label's own weight is exactly 0
            assert np.array_equal(probs[i], want), (seed, i)
            if size[own] == 1 and own < free:
                found += 1
                if za[i] - 1 == own:         # it re-opened its own label: the row it was drawn from says so
                    assert probs[i, own] > 0
        if found >= 5:
            break
    assert found >= 5, "no singleton below the smallest free label in 30 seeds: the test lost its case"


def test_debug_variant_reports_a_label_out_of_range(dbg_lib):
    """-DBMM_DEBUG_HOOKS build: a kernel that produces a label outside [0, K) raises BMM_E_STATE instead of
    indexing the LDS histogram with it (BMM_DEBUG_BADLABEL makes the first observation of every batch one)"""
    X, _, _, _ = synth(5000, 16, 3, 4)
    z0 = _z0(5000, 3, 1)
    ok = bm.gibbs_collapsed(X, 4, 3, burnin=1, seed=5, initial_K=z0)
    assert ok["z"].min() >= 1
    dbg_lib.setenv("BMM_DEBUG_BADLABEL", "1")
    with pytest.raises(bm.BmmError, match=r"label outside \[0, K\)") as e:
        bm.gibbs_collapsed(X, 4, 3, burnin=1, seed=5, initial_K=z0)
    assert e.value.code == 5
    with bm.Chain("dp", 5000, 16, 6, seed=1, batch=512) as ch:      # resident API: reported at the next sync
        ch.set_data(X)
        ch.sweeps(1)
        with pytest.raises(bm.BmmError, match="label outside"):
            ch.sync()
    dbg_lib.delenv("BMM_DEBUG_BADLABEL")
    again = bm.gibbs_collapsed(X, 4, 3, burnin=1, seed=5, initial_K=z0)
    assert np.array_equal(again["z"], ok["z"])
