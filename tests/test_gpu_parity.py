"""HIP path (through the C ABI and the Python mirror of the R wrappers) against the
oracle on the same inputs and seed: labels, theta, pi and alpha must be bit-identical
(integer z bit-exact under matched RNG; the floating-point outputs are produced by the
same binary64 operation sequence, so they are compared exactly too)."""
import numpy as np
import pytest

import bmm_mcmc_amd as bm
from util import load_dataset, proportions, synth

pytestmark = pytest.mark.gpu


def _z0(N, K, seed):
    return np.random.default_rng(seed).integers(1, K + 1, N).astype(np.int32)


def _same(got, want, keys):
    for k in keys:
        assert got[k].shape == want[k].shape, k
        assert np.array_equal(got[k], want[k], equal_nan=True), k


# ---------------------------------------------------------------- finite collapsed sampler
@pytest.mark.parametrize("batch", [1, 7, 100])
def test_collapsed_bundled_c1_small(oracle, batch):
    X = load_dataset("K2_N100_P5")
    z0 = _z0(100, 2, 3)
    got = bm.gibbs_collapsed(X, 60, 2, seed=11, batch=batch, initial_K=z0)
    want = oracle.collapsed(X, z0, 60, 2, 0.0, 0.5, 0.5, 1, 1, 6, seed=11, batch=batch)
    assert got["z"].shape == (54, 100) and got["theta"].shape == (2, 5, 54) and got["alpha"].shape == (54, 1)
    assert got["permutations"].shape == (54, 2) and (got["permutations"] == bm.NA_INTEGER).all()
    _same(got, want, ["z", "theta", "alpha"])


@pytest.mark.parametrize("batch", [1, 25])
def test_collapsed_config_c1_full(oracle, batch):
    # BASELINE config 1: bundled K2_N100_P5, 1 chain, 1000 iterations, burn-in 100 -- at batch = 1, the
    # reference's sequential scan (100 launches per sweep: the chain the reference itself runs), and batched
    X = load_dataset("K2_N100_P5")
    z0 = _z0(100, 2, 17)
    got = bm.gibbs_collapsed(X, 1000, 2, seed=2019, batch=batch, initial_K=z0)
    want = oracle.collapsed(X, z0, 1000, 2, 0.0, 0.5, 0.5, 1, 1, 100, seed=2019, batch=batch)
    _same(got, want, ["z", "theta", "alpha"])
    np.testing.assert_allclose(proportions(got["z"], 2), [0.7, 0.3], atol=0.05)


@pytest.mark.parametrize("N,P,K,batch", [(4000, 20, 3, 512), (3001, 33, 7, 1000), (2500, 100, 20, 2500),
                                         (1500, 128, 24, 300), (777, 1, 2, 50), (5000, 50, 20, 4096)])
def test_collapsed_synthetic_shapes(oracle, N, P, K, batch):
    X, _, _, _ = synth(N, P, min(K, 5), 18)
    z0 = _z0(N, K, 5)
    got = bm.gibbs_collapsed(X, 8, K, burnin=0, seed=99, batch=batch, initial_K=z0)
    want = oracle.collapsed(X, z0, 8, K, 0.0, 0.5, 0.5, 1, 1, 0, seed=99, batch=batch)
    _same(got, want, ["z", "theta", "alpha"])
    assert np.array_equal(got["z"][0], z0)  # burnin = 0: row 0 is the initial allocation


def test_collapsed_fixed_alpha_asymmetric_prior(oracle):
    X, _, _, _ = synth(3000, 17, 4, 21)
    z0 = _z0(3000, 4, 1)
    got = bm.gibbs_collapsed(X, 10, 4, alpha=2.5, beta=0.3, gamma=1.7, burnin=2, seed=5, batch=256, initial_K=z0)
    want = oracle.collapsed(X, z0, 10, 4, 2.5, 0.3, 1.7, 1, 1, 2, seed=5, batch=256)
    _same(got, want, ["z", "theta", "alpha"])
    assert (got["alpha"] == 2.5).all()


def test_collapsed_empty_cluster_stays_empty(oracle):
    # collapsed_gibbs.cpp:104,131-133: an emptied cluster has probability 0 for ever; theta NaN (:214)
    X = load_dataset("K3_N1000_P5")
    z0 = np.ones(1000, dtype=np.int32)
    z0[::2] = 2  # cluster 3 of 4 never used, cluster 4 holds one observation
    z0[7] = 4
    got = bm.gibbs_collapsed(X, 12, 4, burnin=1, seed=3, batch=64, initial_K=z0)
    want = oracle.collapsed(X, z0, 12, 4, 0.0, 0.5, 0.5, 1, 1, 1, seed=3, batch=64)
    _same(got, want, ["z", "theta", "alpha"])
    assert not (got["z"] == 3).any()
    assert np.isnan(got["theta"][2]).all()


def test_collapsed_same_seed_same_chain_other_seed_other_chain():
    X, _, _, _ = synth(6000, 12, 3, 18)
    z0 = _z0(6000, 3, 2)
    a = bm.gibbs_collapsed(X, 6, 3, seed=1, batch=1024, initial_K=z0)
    b = bm.gibbs_collapsed(X, 6, 3, seed=1, batch=1024, initial_K=z0)
    c = bm.gibbs_collapsed(X, 6, 3, seed=2, batch=1024, initial_K=z0)
    assert np.array_equal(a["z"], b["z"]) and not np.array_equal(a["z"], c["z"])


# ---------------------------------------------------------------- DP sampler
@pytest.mark.parametrize("batch", [1, 13, 100])
def test_dp_bundled(oracle, batch):
    X = load_dataset("K2_N100_P5")
    got = bm.gibbs_dp(X, 40, seed=77, batch=batch)
    want = oracle.dp(X, 40, 0.0, 0.5, 0.5, 1, 1, 4, 30, seed=77, batch=batch)
    assert got["theta"].shape == (30, 5, 36)
    _same(got, want, ["z", "theta", "alpha"])


@pytest.mark.parametrize("N,P,maxK,batch,alpha", [(3000, 20, 30, 256, 0.0), (2000, 50, 30, 2000, 0.0),
                                                  (1000, 5, 3, 50, 5.0), (2500, 64, 12, 500, 1.0)])
def test_dp_synthetic_and_truncation(oracle, N, P, maxK, batch, alpha):
    X, _, _, _ = synth(N, P, 5, 19)
    got = bm.gibbs_dp(X, 9, alpha=alpha, burnin=0, maxK=maxK, seed=4, batch=batch)
    want = oracle.dp(X, 9, alpha, 0.5, 0.5, 1, 1, 0, maxK, seed=4, batch=batch)
    _same(got, want, ["z", "theta", "alpha"])
    assert (got["z"][0] == bm.NA_INTEGER).all()      # row 0 is never written by the reference
    assert got["z"][1:].min() >= 1 and got["z"][1:].max() <= maxK


def test_dp_rejects_asymmetric_prior():
    X = load_dataset("K2_N100_P5")
    with pytest.raises(bm.BmmError, match="non-symmetric"):
        bm.gibbs_dp(X, 5, beta=0.5, gamma=0.7, seed=1)


# ---------------------------------------------------------------- stick-breaking sampler
def _sb_init(maxK, P, seed):
    rng = np.random.default_rng(seed)
    pi0 = np.exp(rng.random(maxK)); pi0 /= pi0.sum()
    return pi0, rng.random((maxK, P))


@pytest.mark.parametrize("name,maxK", [("K2_N100_P5", 6), ("K3_N1000_P5", 10)])
def test_sb_bundled(oracle, name, maxK):
    X = load_dataset(name)
    pi0, th0 = _sb_init(maxK, 5, 3)
    got = bm.gibbs_stickbreaking(X, 50, maxK, seed=31, initial_pi=pi0, initial_theta=th0)
    want = oracle.stickbreaking(X, pi0, th0, 50, maxK, 0.0, 0.5, 0.5, 1, 1, 5, seed=31)
    assert got["pi"].shape == (45, maxK)
    _same(got, want, ["z", "theta", "alpha", "pi"])


@pytest.mark.parametrize("N,P,maxK", [(4000, 50, 50), (3000, 20, 10), (1234, 97, 33), (2000, 128, 32), (1500, 60, 64)])
def test_sb_synthetic_shapes(oracle, N, P, maxK):
    X, _, _, _ = synth(N, P, 6, 20)
    pi0, th0 = _sb_init(maxK, P, 8)
    got = bm.gibbs_stickbreaking(X, 7, maxK, burnin=0, seed=12, initial_pi=pi0, initial_theta=th0)
    want = oracle.stickbreaking(X, pi0, th0, 7, maxK, 0.0, 0.5, 0.5, 1, 1, 0, seed=12)
    _same(got, want, ["z", "theta", "alpha", "pi"])
    assert np.array_equal(got["pi"][0], pi0) and np.array_equal(got["theta"][:, :, 0], th0)


# ---------------------------------------------------------------- full (uncollapsed) sampler, row f1
@pytest.mark.parametrize("name,K", [("K2_N100_P5", 2), ("K3_N1000_P5", 3)])
def test_full_bundled(oracle, name, K):
    X = load_dataset(name)
    pi0, th0 = _sb_init(K, 5, 5)
    got = bm.gibbs_full(X, 80, K, seed=23, initial_pi=pi0, initial_theta=th0)
    want = oracle.full(X, pi0, th0, 80, K, 0.0, 0.5, 0.5, 1, 1, 8, seed=23)
    assert got["pi"].shape == (72, K)
    _same(got, want, ["z", "theta", "alpha", "pi"])


@pytest.mark.parametrize("N,P,K,alpha", [(5000, 50, 20, 0.0), (2222, 77, 9, 3.0)])
def test_full_synthetic_shapes(oracle, N, P, K, alpha):
    X, _, _, _ = synth(N, P, 5, 23)
    pi0, th0 = _sb_init(K, P, 6)
    got = bm.gibbs_full(X, 7, K, alpha=alpha, burnin=0, seed=2, initial_pi=pi0, initial_theta=th0)
    want = oracle.full(X, pi0, th0, 7, K, alpha, 0.5, 0.5, 1, 1, 0, seed=2)
    _same(got, want, ["z", "theta", "alpha", "pi"])


# ---------------------------------------------------------------- resident chains and limits
def test_resident_chain_matches_run_and_counts_are_consistent(oracle):
    X, _, _, _ = synth(20000, 24, 4, 18)
    z0 = _z0(20000, 4, 9)
    with bm.Chain("collapsed", 20000, 24, 4, batch=4096, seed=5) as ch:
        ch.set_data(X)
        ch.set_initial_labels(z0)
        ch.sweeps(3)
        ch.sweeps(2)
        z = ch.labels()
        Nk, S = ch.counts()
        al = ch.alpha()
    want = oracle.collapsed(X, z0, 6, 4, 0.0, 0.5, 0.5, 1, 1, 5, seed=5, batch=4096)
    assert np.array_equal(z, want["z"][0])
    assert al == want["alpha"][0, 0]
    # size-independent property: the integer statistics equal a recount from the labels
    assert np.array_equal(Nk, np.bincount(z - 1, minlength=4))
    for k in range(4):
        assert np.array_equal(S[k], X[z == k + 1].sum(axis=0))


@pytest.mark.parametrize("N,P,K,batch", [(1, 3, 2, 1), (63, 1, 2, 63), (65, 16, 3, 7), (129, 17, 4, 129),
                                         (1000, 128, 2, 999)])
def test_collapsed_ragged_and_tiny(oracle, N, P, K, batch):
    # fewer observations than a wave, stages that end mid-word, batches that end mid-wave
    rng = np.random.default_rng(N)
    X = np.asfortranarray(rng.integers(0, 2, (N, P)).astype(np.int32))
    z0 = _z0(N, K, 1)
    got = bm.gibbs_collapsed(X, 6, K, burnin=0, seed=7, batch=batch, initial_K=z0)
    want = oracle.collapsed(X, z0, 6, K, 0.0, 0.5, 0.5, 1, 1, 0, seed=7, batch=batch)
    _same(got, want, ["z", "theta", "alpha"])


def test_all_zero_and_all_one_columns(oracle):
    # log(beta + 0) and log(gamma + 0) sides of the tables
    X = np.zeros((500, 9), dtype=np.int32)
    X[:, 3] = 1
    X[::3, 5] = 1
    X = np.asfortranarray(X)
    z0 = _z0(500, 3, 4)
    got = bm.gibbs_collapsed(X, 8, 3, burnin=1, seed=1, batch=50, initial_K=z0)
    want = oracle.collapsed(X, z0, 8, 3, 0.0, 0.5, 0.5, 1, 1, 1, seed=1, batch=50)
    _same(got, want, ["z", "theta", "alpha"])
    got = bm.gibbs_dp(X, 8, burnin=1, seed=1, batch=50, maxK=6)
    want = oracle.dp(X, 8, 0.0, 0.5, 0.5, 1, 1, 1, 6, seed=1, batch=50)
    _same(got, want, ["z", "theta", "alpha"])


def test_non_binary_data_is_rejected_by_the_c_abi():
    import ctypes as C
    from bmm_mcmc_amd import _capi
    X = np.asfortranarray(np.random.default_rng(0).integers(0, 2, (300, 7)).astype(np.int32))
    X[123, 4] = 2
    z0 = np.ones(300, dtype=np.int32)
    z = np.zeros((1, 300), dtype=np.int32, order="F"); th = np.zeros((2, 7, 1), order="F"); al = np.zeros((1, 1))
    rc = _capi.lib().bmm_collapsed_run(_capi.vp(X), C.c_int64(300), C.c_int(7), _capi.vp(z0), C.c_int(3), C.c_int(2),
                                       C.c_double(1.0), C.c_double(0.5), C.c_double(0.5), C.c_double(1), C.c_double(1),
                                       C.c_int(2), C.c_int64(0), C.c_uint64(1), C.c_int(0), _capi.vp(z), _capi.vp(th),
                                       _capi.vp(al))
    assert rc == 1 and b"binary" in _capi.lib().bmm_last_error()


def test_device_side_count_summaries_match_the_label_trace(oracle):
    # row f3: per-sweep cluster sizes without the S x N label matrix
    X, _, _, _ = synth(9000, 30, 5, 18)
    z0 = _z0(9000, 6, 3)
    with bm.Chain("collapsed", 9000, 30, 6, batch=1000, seed=8) as ch:
        ch.set_data(X)
        ch.set_initial_labels(z0)
        ch.sweeps(2)
        nk = ch.sweeps_counts(5)
    want = oracle.collapsed(X, z0, 8, 6, 0.0, 0.5, 0.5, 1, 1, 3, seed=8, batch=1000)
    for s in range(5):
        assert np.array_equal(nk[s], np.bincount(want["z"][s] - 1, minlength=6))
    pi0, th0 = _sb_init(7, 30, 2)
    with bm.Chain("stickbreaking", 9000, 30, 7, seed=8) as ch:
        ch.set_data(X)
        ch.set_initial_params(pi0, th0)
        nk = ch.sweeps_counts(4)
    want = oracle.stickbreaking(X, pi0, th0, 5, 7, 0.0, 0.5, 0.5, 1, 1, 1, seed=8)
    for s in range(4):
        assert np.array_equal(nk[s], np.bincount(want["z"][s] - 1, minlength=7))


# ---------------------------------------------------------------- one chain over several ranks (row f4)
@pytest.mark.parametrize("sampler", ["stickbreaking", "full"])
def test_sharded_chain_equals_the_single_chain(oracle, sampler):
    # two "virtual ranks" on one GPU: each holds half of the rows; the all-reduce of the statistic
    # deltas is done by hand on the same device buffers RCCL would reduce
    import torch
    from bmm_mcmc_amd import multi
    N, P, K, sweeps = 5001, 40, 9, 6
    X, _, _, _ = synth(N, P, 4, 41)
    pi0, th0 = _sb_init(K, P, 3)
    cut = 2048
    Xt = torch.from_numpy(np.ascontiguousarray(X.T)).cuda()          # (P, N) = N x P column-major
    shards = [Xt[:, :cut].contiguous(), Xt[:, cut:].contiguous()]
    chains = [multi.ShardedChain(sampler, shards[0], N, 0, K, pi0, th0, seed=77),
              multi.ShardedChain(sampler, shards[1], N, cut, K, pi0, th0, seed=77)]
    for _ in range(sweeps):
        for c in chains:
            c.chain.shard_resample()
        for name in ("d_nk", "d_s"):
            tot = getattr(chains[0], name) + getattr(chains[1], name)
            for c in chains:
                getattr(c, name).copy_(tot)
        torch.cuda.synchronize()
        for c in chains:
            c.chain.shard_finish()
    z = np.concatenate([c.chain.labels() for c in chains])
    pis = [c.chain.params() for c in chains]
    alphas = [c.chain.alpha() for c in chains]
    for c in chains:
        c.close()
    fn = oracle.stickbreaking if sampler == "stickbreaking" else oracle.full
    want = fn(X, pi0, th0, sweeps + 1, K, 0.0, 0.5, 0.5, 1, 1, sweeps, seed=77)
    assert np.array_equal(z, want["z"][0])
    for pi, theta in pis:                                             # identical on both ranks
        assert np.array_equal(pi, want["pi"][0]) and np.array_equal(theta, want["theta"][:, :, 0])
    assert alphas[0] == alphas[1] == want["alpha"][0, 0]


def test_sharding_is_refused_where_it_is_not_exact():
    with bm.Chain("collapsed", 100, 5, 2, seed=1) as ch:
        with pytest.raises(bm.BmmError, match="shard exactly"):
            ch.set_shard(200, 0)
    with bm.Chain("stickbreaking", 100, 5, 4, seed=1) as ch:
        with pytest.raises(bm.BmmError, match="outside N_total"):
            ch.set_shard(150, 100)


# ---------------------------------------------------------------- relabel hand-off (row f2)
def test_sweep_probabilities_are_the_conditionals_the_draw_used(oracle):
    X = load_dataset("K3_N1000_P5")[::4]                      # 250 observations
    N, K = X.shape[0], 3
    z0 = _z0(N, K, 5)
    with bm.Chain("collapsed", N, 5, K, alpha=1.5, batch=N, seed=4) as ch:
        ch.set_data(X)
        ch.set_initial_labels(z0)
        ch.sweeps(2)
        z_before = ch.labels()
        probs = ch.sweep_probs()
        z_after = ch.labels()
    want = oracle.collapsed(X, z0, 4, K, 1.5, 0.5, 0.5, 1, 1, 2, seed=4, batch=N)
    assert np.array_equal(z_before, want["z"][0]) and np.array_equal(z_after, want["z"][1])
    for i in range(N):                                        # batch = N: every row against the same state
        _, norm = oracle.collapsed_cond(X, z_before, i, K, 1.5, 0.5, 0.5, spec=True)
        assert np.array_equal(probs[i], norm)
    np.testing.assert_allclose(probs.sum(axis=1), 1.0, rtol=0, atol=1e-15)
    pi0, th0 = _sb_init(4, 5, 1)
    with bm.Chain("stickbreaking", N, 5, 4, alpha=1.0, seed=4) as ch:
        ch.set_data(X)
        ch.set_initial_params(pi0, th0)
        probs = ch.sweep_probs()
    for i in (0, 17, N - 1):
        _, norm = oracle.sb_cond(X, i, pi0, th0, spec=True)
        assert np.array_equal(probs[i], norm)
    # DP: the new-cluster mass is filed under the label it would open; batched (3 launches per sweep)
    with bm.Chain("dp", N, 5, 6, alpha=1.0, batch=100, seed=9) as ch:
        ch.set_data(X)
        ch.sweeps(3)
        zb = ch.labels()
        probs = ch.sweep_probs()
    assert probs.shape == (N, 6)
    np.testing.assert_allclose(probs.sum(axis=1), 1.0, rtol=0, atol=1e-14)
    used = np.bincount(zb - 1, minlength=6) > 0
    free = int(np.argmin(used)) if not used.all() else -1
    _, norm = oracle.dp_cond(X, zb, 0, 6, 1.0, 0.5, 0.5, spec=True)   # observation 0: first batch, state zb
    want0 = norm[:6].copy()
    if free >= 0:
        want0[free] = norm[6]
    assert np.array_equal(probs[0], want0)
    # the int32 layout has no emitting twin: its hand-off sweeps run on the generic kernel, same numbers
    with bm.Chain("collapsed", N, 5, K, alpha=1.5, batch=N, seed=4, x_layout="int32") as ch:
        ch.set_data(X)
        ch.set_initial_labels(z0)
        ch.sweeps(2)
        p32 = ch.sweep_probs()
    with bm.Chain("collapsed", N, 5, K, alpha=1.5, batch=N, seed=4) as ch:
        ch.set_data(X)
        ch.set_initial_labels(z0)
        ch.sweeps(2)
        pbits = ch.sweep_probs()
    assert np.array_equal(p32, pbits)


# ---------------------------------------------------------------- generic path (any shape)
@pytest.mark.parametrize("N,P,K,batch", [(1200, 200, 5, 300), (900, 40, 100, 900), (700, 513, 3, 64),
                                         (1500, 128, 60, 500)])
def test_collapsed_beyond_the_resident_kernel(oracle, N, P, K, batch):
    # P > 128, more than 64 clusters, or tables that do not fit in LDS: tables from global memory
    X, _, _, _ = synth(N, P, 4, 31)
    z0 = _z0(N, K, 2)
    got = bm.gibbs_collapsed(X, 5, K, burnin=0, seed=13, batch=batch, initial_K=z0)
    want = oracle.collapsed(X, z0, 5, K, 0.0, 0.5, 0.5, 1, 1, 0, seed=13, batch=batch)
    _same(got, want, ["z", "theta", "alpha"])


def test_dp_and_explicit_samplers_beyond_the_resident_kernel(oracle):
    X, _, _, _ = synth(800, 150, 4, 33)
    got = bm.gibbs_dp(X, 6, burnin=1, maxK=80, seed=3, batch=100)
    want = oracle.dp(X, 6, 0.0, 0.5, 0.5, 1, 1, 1, 80, seed=3, batch=100)
    _same(got, want, ["z", "theta", "alpha"])
    pi0, th0 = _sb_init(70, 150, 9)
    got = bm.gibbs_stickbreaking(X, 5, 70, burnin=0, seed=4, initial_pi=pi0, initial_theta=th0)
    want = oracle.stickbreaking(X, pi0, th0, 5, 70, 0.0, 0.5, 0.5, 1, 1, 0, seed=4)
    _same(got, want, ["z", "theta", "alpha", "pi"])
    got = bm.gibbs_full(X, 5, 70, burnin=0, seed=4, initial_pi=pi0, initial_theta=th0)
    want = oracle.full(X, pi0, th0, 5, 70, 0.0, 0.5, 0.5, 1, 1, 0, seed=4)
    _same(got, want, ["z", "theta", "alpha", "pi"])


def test_generic_path_equals_resident_path_on_an_ordinary_shape(oracle, dbg_lib):
    X, _, _, _ = synth(3000, 33, 4, 35)
    z0 = _z0(3000, 7, 6)
    dbg_lib.delenv("BMM_DEBUG_GENERIC", raising=False)
    fast = bm.gibbs_collapsed(X, 6, 7, burnin=0, seed=21, batch=512, initial_K=z0)
    dbg_lib.setenv("BMM_DEBUG_GENERIC", "1")   # read by the test variant of the library only
    slow = bm.gibbs_collapsed(X, 6, 7, burnin=0, seed=21, batch=512, initial_K=z0)
    _same(fast, slow, ["z", "theta", "alpha"])
    want = oracle.collapsed(X, z0, 6, 7, 0.0, 0.5, 0.5, 1, 1, 0, seed=21, batch=512)
    _same(fast, want, ["z", "theta", "alpha"])


def test_unsupported_shapes_fail_loudly():
    X = np.zeros((50, 10), dtype=np.int32)
    with pytest.raises(bm.BmmError, match="categories"):
        bm.gibbs_collapsed(X, 5, 2000, seed=1)
    with pytest.raises(bm.BmmError, match="initialK"):
        bm.gibbs_collapsed(X, 5, 2, seed=1, initial_K=np.full(50, 3))


# ---------------------------------------------------------------- the narrower group width (big table images)
def test_shapes_whose_tables_need_the_narrower_groups(oracle):
    """group width 4 instead of 5 where the 32-entry tables would not fit in LDS (bmm_spec_group_width_for):
    own-cluster tables still resident (K=20, P=112), two lanes per observation (stick-breaking K=64, P=64),
    and the emitting twin (sweep_probs) on such a shape -- all bit-equal to the oracle, which applies the
    same rule on its own"""
    from bmm_mcmc_amd import _capi
    W = _capi.lib().bmm_spec_group_width_for
    assert W(0, 20, 112) == 4 and W(2, 64, 64) == 4 and W(0, 20, 100) == 5
    N, P, K = 5000, 112, 20
    X, _, _, _ = synth(N, P, 5, 51)
    z0 = _z0(N, K, 52)
    for batch in (N, 700):
        got = bm.gibbs_collapsed(X, 5, K, burnin=0, seed=3, batch=batch, initial_K=z0)
        want = oracle.collapsed(X, z0, 5, K, 0.0, 0.5, 0.5, 1, 1, 0, seed=3, batch=batch)
        for k in ("z", "theta", "alpha"):
            assert np.array_equal(got[k], want[k], equal_nan=True), (k, batch)
    with bm.Chain("collapsed", N, P, K, alpha=1.5, batch=N, seed=4) as ch:
        assert ch.kernel_shape()["lds_bytes"] > 0            # resident, not the generic path
        ch.set_data(X)
        ch.set_initial_labels(z0)
        ch.sweeps(2)
        zb = ch.labels()
        probs = ch.sweep_probs()
    for i in (0, 1, 777, N - 1):
        _, norm = oracle.collapsed_cond(X, zb, i, K, 1.5, 0.5, 0.5, spec=True)
        assert np.array_equal(probs[i], norm)
    labels = {}
    for layout in ("bits", "int32"):                         # the int32 kernels at the narrower width too
        with bm.Chain("collapsed", N, P, K, batch=900, seed=6, x_layout=layout) as ch:
            ch.set_data(X)
            ch.set_initial_labels(z0)
            ch.sweeps(3)
            labels[layout] = ch.labels()
    assert np.array_equal(labels["bits"], labels["int32"])
    assert np.array_equal(labels["bits"], oracle.collapsed(X, z0, 4, K, 0.0, 0.5, 0.5, 1, 1, 3, seed=6, batch=900)["z"][0])
    N, P, K = 3000, 64, 64
    X, _, _, _ = synth(N, P, 6, 53)
    pi0, th0 = _sb_init(K, P, 2)
    got = bm.gibbs_stickbreaking(X, 4, K, burnin=0, seed=5, initial_pi=pi0, initial_theta=th0)
    want = oracle.stickbreaking(X, pi0, th0, 4, K, 0.0, 0.5, 0.5, 1, 1, 0, seed=5)
    for k in ("z", "theta", "alpha", "pi"):
        assert np.array_equal(got[k], want[k], equal_nan=True), k
