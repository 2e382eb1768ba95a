"""The two X layouts of the resident kernel -- bit planes packed once when the matrix is handed
over (default) and the int32 matrix streamed as R hands it over -- run the same chain: both
must equal the oracle bit for bit, for every word-boundary shape of P."""
import numpy as np
import pytest

import bmm_mcmc_amd as bm
from util import synth

pytestmark = pytest.mark.gpu


def _z0(N, K, seed):
    return np.random.default_rng(seed).integers(1, K + 1, N).astype(np.int32)


def _chain_labels(layout, X, K, z0, batch, sweeps=3):
    N, P = X.shape
    with bm.Chain("collapsed", N, P, K, batch=batch, seed=21, x_layout=layout) as c:
        assert c.x_layout() == layout
        c.set_data(X)
        c.set_initial_labels(z0)
        c.sweeps(sweeps)
        c.sync()
        return c.labels(), c.counts()


@pytest.mark.parametrize("P", [1, 4, 31, 32, 33, 63, 64, 65, 96, 97, 100, 127, 128])
def test_bit_planes_and_int32_run_the_same_chain(oracle, P):
    N, K, batch = 3000, 5, 700
    X, _, _, _ = synth(N, P, 3, 100 + P)
    z0 = _z0(N, K, 2)
    zb, (nb, sb) = _chain_labels("bits", X, K, z0, batch)
    zi, (ni, si) = _chain_labels("int32", X, K, z0, batch)
    assert np.array_equal(zb, zi) and np.array_equal(nb, ni) and np.array_equal(sb, si)
    want = oracle.collapsed(X, z0, 4, K, 0.0, 0.5, 0.5, 1, 1, 0, seed=21, batch=batch)
    assert np.array_equal(zb, want["z"][3])
    assert np.array_equal(sb, np.stack([X[zb == k + 1].sum(axis=0) for k in range(K)]))


@pytest.mark.parametrize("env", [None, "1"])
def test_run_entry_points_under_both_layouts(oracle, dbg_lib, env):
    # the *_run entry points have no layout argument: the test variant's BMM_X_LAYOUT_INT32 switches them
    if env is None:
        dbg_lib.delenv("BMM_X_LAYOUT_INT32", raising=False)
    else:
        dbg_lib.setenv("BMM_X_LAYOUT_INT32", env)
    X, _, _, _ = synth(4000, 50, 4, 3)
    z0 = _z0(4000, 20, 9)
    got = bm.gibbs_collapsed(X, 6, 20, burnin=0, seed=5, batch=512, initial_K=z0)
    want = oracle.collapsed(X, z0, 6, 20, 0.0, 0.5, 0.5, 1, 1, 0, seed=5, batch=512)
    for k in ("z", "theta", "alpha"):
        assert np.array_equal(got[k], want[k], equal_nan=True), k
    got = bm.gibbs_dp(X, 6, burnin=0, maxK=12, seed=5, batch=256)
    want = oracle.dp(X, 6, 0.0, 0.5, 0.5, 1, 1, 0, 12, seed=5, batch=256)
    for k in ("z", "theta", "alpha"):
        assert np.array_equal(got[k], want[k], equal_nan=True), k
    rng = np.random.default_rng(1)
    pi0 = rng.dirichlet(np.ones(10))
    th0 = rng.random((10, 50))
    got = bm.gibbs_stickbreaking(X, 5, 10, burnin=0, seed=8, initial_pi=pi0, initial_theta=th0)
    want = oracle.stickbreaking(X, pi0, th0, 5, 10, 0.0, 0.5, 0.5, 1, 1, 0, seed=8)
    for k in ("z", "theta", "alpha", "pi"):
        assert np.array_equal(got[k], want[k], equal_nan=True), k


def test_the_product_library_ignores_the_steering_environment(oracle, monkeypatch):
    """The product build reads no environment: with every switch set, a *_run call still runs the
    bit-plane resident kernel (same chain either way, so this is seen in the layout it reports)."""
    for k in ("BMM_X_LAYOUT_INT32", "BMM_DEBUG_GENERIC", "BMM_DEBUG_NOSPLIT"):
        monkeypatch.setenv(k, "1")
    monkeypatch.setenv("BMM_DEBUG_THREADS", "512")
    with bm.Chain("collapsed", 500, 8, 2, seed=1) as c:
        assert c.x_layout() == "bits"
        assert c.kernel_shape()["lds_bytes"] > 0   # resident kernel, not the generic path


def test_layout_is_fixed_once_the_data_are_set():
    X, _, _, _ = synth(500, 8, 2, 1)
    with bm.Chain("collapsed", 500, 8, 2, seed=1) as c:
        assert c.x_layout() == "bits"
        c.set_data(X)
        with pytest.raises(bm.BmmError):
            _capi_set(c, 1)


def _capi_set(c, layout):
    import ctypes
    from bmm_mcmc_amd import _capi
    _capi.check(_capi.lib().bmm_chain_set_x_layout(c._h, ctypes.c_int(layout)))


# more than 32 accumulators: two lanes share an observation (SPLIT = 2), with and without the
# own-cluster tables; the one-lane kernels (BMM_DEBUG_NOSPLIT) must give the same chain
@pytest.mark.parametrize("nosplit", [False, True])
def test_wide_category_counts_two_lanes_per_observation(oracle, dbg_lib, nosplit):
    if nosplit:
        dbg_lib.setenv("BMM_DEBUG_NOSPLIT", "1")
    else:
        dbg_lib.delenv("BMM_DEBUG_NOSPLIT", raising=False)
    for N, P, K, batch in [(3000, 40, 40, 700), (2011, 20, 64, 2011), (1500, 33, 56, 97), (2600, 50, 50, 650), (1800, 25, 52, 1800)]:   # (50, 52: the 52-accumulator kernels)
        X, _, _, _ = synth(N, P, 4, K)
        z0 = _z0(N, K, 3)
        got = bm.gibbs_collapsed(X, 6, K, burnin=0, seed=77, batch=batch, initial_K=z0)
        want = oracle.collapsed(X, z0, 6, K, 0.0, 0.5, 0.5, 1, 1, 0, seed=77, batch=batch)
        for k in ("z", "theta", "alpha"):
            assert np.array_equal(got[k], want[k], equal_nan=True), (k, N, P, K)
    X, _, _, _ = synth(4000, 24, 5, 9)
    for maxK in (47, 51):
        got = bm.gibbs_dp(X, 8, burnin=0, maxK=maxK, seed=31, batch=333)
        want = oracle.dp(X, 8, 0.0, 0.5, 0.5, 1, 1, 0, maxK, seed=31, batch=333)
        for k in ("z", "theta", "alpha"):
            assert np.array_equal(got[k], want[k], equal_nan=True), (k, maxK)
    rng = np.random.default_rng(4)
    pi0 = rng.dirichlet(np.ones(40))
    th0 = rng.random((40, 30))
    X, _, _, _ = synth(2500, 30, 3, 2)
    got = bm.gibbs_full(X, 5, 40, burnin=0, seed=6, initial_pi=pi0, initial_theta=th0)
    want = oracle.full(X, pi0, th0, 5, 40, 0.0, 0.5, 0.5, 1, 1, 0, seed=6)
    for k in ("z", "theta", "alpha", "pi"):
        assert np.array_equal(got[k], want[k], equal_nan=True), k


@pytest.mark.parametrize("nosplit", [False, True])
def test_both_forms_of_16_to_32_accumulators_on_short_launches(oracle, dbg_lib, nosplit):
    """Launches of at most 24 chunks per CU run 16-32 accumulators two lanes per observation (pick_kernel); the
    one-lane form of the same kernels is what long launches (C5) run.  Test-sized inputs are all short launches,
    so the one-lane form is reached here through BMM_DEBUG_NOSPLIT: both must be the oracle's chain, bit for bit."""
    if nosplit:
        dbg_lib.setenv("BMM_DEBUG_NOSPLIT", "1")
    else:
        dbg_lib.delenv("BMM_DEBUG_NOSPLIT", raising=False)
    for N, P, K, batch in [(3000, 100, 20, 400), (2500, 50, 16, 2500), (1999, 33, 24, 97), (1200, 64, 28, 300), (2100, 20, 32, 777)]:
        X, _, _, _ = synth(N, P, 4, K + P)
        z0 = _z0(N, K, 3)
        got = bm.gibbs_collapsed(X, 5, K, burnin=0, seed=5, batch=batch, initial_K=z0)
        want = oracle.collapsed(X, z0, 5, K, 0.0, 0.5, 0.5, 1, 1, 0, seed=5, batch=batch)
        for k in ("z", "theta", "alpha"):
            assert np.array_equal(got[k], want[k], equal_nan=True), (k, N, P, K)
    X, _, _, _ = synth(3000, 50, 6, 19)
    got = bm.gibbs_dp(X, 8, burnin=0, maxK=30, seed=31, batch=187)          # the c3 shape's 32 accumulators
    want = oracle.dp(X, 8, 0.0, 0.5, 0.5, 1, 1, 0, 30, seed=31, batch=187)
    for k in ("z", "theta", "alpha"):
        assert np.array_equal(got[k], want[k], equal_nan=True), k
    rng = np.random.default_rng(4)
    pi0 = rng.dirichlet(np.ones(24))
    th0 = 0.05 + 0.9 * rng.random((24, 40))
    X, _, _, _ = synth(2500, 40, 3, 2)
    got = bm.gibbs_stickbreaking(X, 5, 24, burnin=0, seed=6, initial_pi=pi0, initial_theta=th0)
    want = oracle.stickbreaking(X, pi0, th0, 5, 24, 0.0, 0.5, 0.5, 1, 1, 0, seed=6)
    for k in ("z", "theta", "alpha", "pi"):
        assert np.array_equal(got[k], want[k], equal_nan=True), k
    # and the kernel really differs: 1024 threads with two lanes per observation, the default size otherwise
    with bm.Chain("collapsed", 3000, 100, 20, seed=1, batch=400) as ch:
        shape = ch.kernel_shape()
    assert shape["threads"] == (1024 if not nosplit else shape["threads"])


@pytest.mark.parametrize("noself", [False, True])
def test_self_built_tables_equal_the_table_kernels(oracle, dbg_lib, noself):
    """Small finite-sampler shapes: the resample workgroups build the table image themselves (no k_count_tables
    between batches); BMM_DEBUG_NOSELF forces the ordinary path.  Both are the oracle's chain bit for bit -- over
    many batches per sweep (the statistics are folded only at the end of a sweep in the first form), with sampled
    and fixed alpha, with an emptied cluster, and when a sweep in between hands its probabilities to the host."""
    if noself:
        dbg_lib.setenv("BMM_DEBUG_NOSELF", "1")
    else:
        dbg_lib.delenv("BMM_DEBUG_NOSELF", raising=False)
    for N, P, K, batch, alpha in [(100000, 20, 3, 12500, 0.0), (5000, 5, 2, 64, 0.0), (3000, 30, 4, 1, 1.5), (777, 120, 1, 100, 0.0),
                                  (4000, 12, 9, 500, 0.3), (2500, 40, 3, 2500, 0.0)]:
        X, _, _, _ = synth(N, P, min(K, 3), N + P)
        z0 = _z0(N, K, 3)
        if K == 4:
            z0[z0 == 4] = 1                      # an empty cluster: stays empty (probability exactly 0)
        sweeps = 3 if batch == 1 else 6
        got = bm.gibbs_collapsed(X, sweeps, K, alpha=alpha if alpha else None, burnin=0, seed=5, batch=batch, initial_K=z0)
        want = oracle.collapsed(X, z0, sweeps, K, alpha, 0.5, 0.5, 1, 1, 0, seed=5, batch=batch)
        for k in ("z", "theta", "alpha"):
            assert np.array_equal(got[k], want[k], equal_nan=True), (k, N, P, K, batch)
    X, _, _, _ = synth(6000, 20, 3, 4)
    z0 = _z0(6000, 3, 2)
    with bm.Chain("collapsed", 6000, 20, 3, batch=750, seed=9) as ch:
        ch.set_data(X)
        ch.set_initial_labels(z0)
        ch.sweeps(2)
        probs = ch.sweep_probs()                 # sweep 3 runs the emitting twin on k_count_tables' image
        ch.sweeps(2)
        z = ch.labels()
        shape = ch.kernel_shape()
    want = oracle.collapsed(X, z0, 6, 3, 0.0, 0.5, 0.5, 1, 1, 5, seed=9, batch=750)
    assert np.array_equal(z, want["z"][0])
    np.testing.assert_allclose(probs.sum(axis=1), 1.0, rtol=0, atol=1e-14)
    assert shape["threads"] == 256


def test_self_built_tables_with_a_workgroup_that_starts_late(oracle, dbg_lib):
    """A workgroup of a table-building launch may start after others of the same launch have finished -- the device
    is shared with another chain's kernels, or another process -- and must still score against the statistics as the
    PREVIOUS launch left them.  BMM_DEBUG_STRAGGLER (test variant) holds workgroup 0 back by about 100 us, long after
    the other workgroups of these short launches have flushed their counts.  (Round 3 first read the statistics plus
    the pending deltas in every workgroup: a late one then saw part of its own launch -- found as a one-in-twenty
    mismatch of a chain that ran beside two others, tests/test_gpu_multi.py.)"""
    dbg_lib.setenv("BMM_DEBUG_STRAGGLER", "1")
    dbg_lib.delenv("BMM_DEBUG_NOSELF", raising=False)
    for N, P, K, batch in [(100000, 20, 3, 12500), (20000, 24, 3, 2048), (5000, 24, 3, 512)]:
        X, _, _, _ = synth(N, P, K, N + P)
        z0 = _z0(N, K, 3)
        with bm.Chain("collapsed", N, P, K, batch=batch, seed=5) as ch:
            ch.set_data(X)
            ch.set_initial_labels(z0)
            ch.sweeps(4)
            z, (nk, S), shape = ch.labels(), ch.counts(), ch.kernel_shape()
        assert shape["threads"] == 256                  # the table-building kernels
        want = oracle.collapsed(X, z0, 5, K, 0.0, 0.5, 0.5, 1, 1, 4, seed=5, batch=batch)
        assert np.array_equal(z, want["z"][0]), (N, P, K, batch, int((z != want["z"][0]).sum()))
        assert np.array_equal(nk, np.bincount(z - 1, minlength=K))
        assert np.array_equal(S, np.stack([X[z == k + 1].sum(axis=0) for k in range(K)]))


def test_kernel_forms_at_the_benchmark_shapes():
    """Which form of the resample kernel a shape and batch get (pick_kernel) -- never visible in a chain's values, so
    held here: the measured rules of profiles/r03/ab_smallsplit.log and ab_self_tables.log at the shapes they were
    measured on (an MI355X: 256 CUs)."""
    def form(sampler, N, P, K, batch=0):
        with bm.Chain(sampler, N, P, K, seed=1, batch=batch) as ch:
            s = ch.kernel_shape()
            return ch.batch, s["threads"], s["lanes_per_observation"], s["builds_own_tables"]
    # north-star shape: N/4 = 250 000 observations per launch fill the chip with one-lane workgroups of 1024 threads;
    # at 125 000 (the N/8 of the first half of round 3) two lanes per observation
    assert form("collapsed", 1_000_000, 50, 20) == (250_000, 1024, 1, False)
    assert form("collapsed", 1_000_000, 50, 20, batch=125_000) == (125_000, 1024, 2, False)
    assert form("collapsed", 1_000_000, 50, 20, batch=200_000)[2] == 1          # from 196 608 on: one lane
    # c3 (32 accumulators): 250 000 -> one-lane workgroups of 768 threads; 125 000 -> two lanes
    assert form("dp", 1_000_000, 50, 30) == (250_000, 768, 1, False)
    assert form("dp", 1_000_000, 50, 30, batch=125_000)[2] == 2
    # c2: workgroups of 256 threads that build their own tables; C5 and c4: the big kernels
    assert form("collapsed", 100_000, 20, 3) == (25_000, 256, 1, True)
    assert form("collapsed", 10_000_000, 100, 20)[:3] == (2_500_000, 1024, 1)
    assert form("stickbreaking", 1_000_000, 50, 50)[1:3] == (1024, 2)           # 52 accumulators: always two lanes
