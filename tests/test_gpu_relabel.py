"""relabel = TRUE data path end to end through the C ABI: the *_run_probs entry
points hand the host's Stephens code the matrices the reference stores (src/collapsed_gibbs.cpp:162-172),
at the sweeps it stores them, produced by the resident resample kernel.  Stephens' algorithm itself is host
code of the reference and is not restated in the product; the `stephens` object below is a test double with
the two signatures of src/stephens.h that records what it is given."""
import numpy as np
import pytest

import bmm_mcmc_amd as bm
from util import load_dataset, synth

pytestmark = pytest.mark.gpu


def _z0(N, K, seed):
    return np.random.default_rng(seed).integers(1, K + 1, N).astype(np.int32)


class RecordingStephens:
    """my_stephens_batch / my_stephens_online stand-in: keeps its inputs; the permutation it returns is a
    fixed rotation, so the relabelled outputs are predictable."""

    def __init__(self, K):
        self.K = K
        self.cube = None
        self.samples = {}

    def batch(self, p):
        self.cube = np.array(p)
        return self.cube.mean(axis=2)

    def online(self, Q, p, j):
        self.samples[j] = np.array(p)
        perm = (np.arange(self.K) + j) % self.K
        return perm, (j * (Q + p[:, perm])) / (j + 1)        # the update of src/stephens.cpp:91


def test_collapsed_relabel_feeds_the_matrices_the_reference_stores(oracle):
    X = load_dataset("K3_N1000_P5")[::2]
    N, K, ns, burnin, W = X.shape[0], 3, 14, 6, 4
    z0 = _z0(N, K, 1)
    st = RecordingStephens(K)
    got = bm.gibbs_collapsed(X, ns, K, alpha=1.2, burnin=burnin, relabel=True, burnrelabel=W, seed=77, batch=N,
                             initial_K=z0, stephens=st)
    want = oracle.collapsed(X, z0, ns, K, 1.2, 0.5, 0.5, 1, 1, 0, seed=77, batch=N)   # every sweep kept
    zall = want["z"]                                        # row j = allocation after sweep j
    # the batch cube: sweeps burnin - W .. burnin - 1, each row the conditional of the draw (batch = N:
    # statistics frozen at the previous sweep's allocation, own contribution removed)
    assert st.cube.shape == (N, K, W)
    for q, j in enumerate(range(burnin - W, burnin)):
        for i in (0, 1, N // 2, N - 1):
            _, norm = oracle.collapsed_cond(X, zall[j - 1], i, K, 1.2, 0.5, 0.5, spec=True)
            assert np.array_equal(st.cube[i, :, q], norm), (j, i)
    # every kept sweep's matrix, in order
    assert sorted(st.samples) == list(range(burnin, ns))
    for j in (burnin, ns - 1):
        for i in (0, 7, N - 1):
            _, norm = oracle.collapsed_cond(X, zall[j - 1], i, K, 1.2, 0.5, 0.5, spec=True)
            assert np.array_equal(st.samples[j][i], norm), (j, i)
    # the chain does not depend on the relabelling; the outputs carry the reference's bookkeeping (:196-217)
    assert np.array_equal(got["z_original"], zall[burnin:])
    assert np.array_equal(got["theta_original"], want["theta"][:, :, burnin:], equal_nan=True)
    S = ns - burnin
    assert got["permutations"].shape == (S, K)
    for s in range(S):
        perm = (np.arange(K) + burnin + s) % K
        assert np.array_equal(got["permutations"][s], perm)
        assert np.array_equal(got["z"][s], perm[got["z_original"][s] - 1] + 1)
        assert np.array_equal(got["theta"][perm, :, s], got["theta_original"][:, :, s], equal_nan=True)


def test_relabel_path_batched_dp_and_stickbreaking(oracle):
    X, _, _, _ = synth(3000, 20, 3, 9)
    N = 3000
    st = RecordingStephens(8)
    got = bm.gibbs_dp(X, 9, alpha=1.0, burnin=4, relabel=True, burnrelabel=2, maxK=8, seed=3, batch=500, stephens=st)
    want = oracle.dp(X, 9, 1.0, 0.5, 0.5, 1, 1, 4, 8, seed=3, batch=500)
    assert np.array_equal(got["z_original"], want["z"])            # same chain as without relabel
    assert st.cube.shape == (N, 8, 2) and sorted(st.samples) == [4, 5, 6, 7, 8]
    for m in list(st.samples.values()) + [st.cube[:, :, 0], st.cube[:, :, 1]]:
        np.testing.assert_allclose(m.sum(axis=1), 1.0, rtol=0, atol=1e-14)
        assert (m >= 0).all()
    rng = np.random.default_rng(5)
    pi0, th0 = rng.dirichlet(np.ones(6)), rng.random((6, 20))
    st = RecordingStephens(6)
    # the stick-breaking wrapper does not clamp burnrelabel (R/utils.R:97-101): a window longer than the
    # burn-in leaves the slices of sweeps that do not exist at zero, as arma::fill::zeros does
    got = bm.gibbs_stickbreaking(X, 7, 6, alpha=1.0, burnin=3, relabel=True, burnrelabel=5, seed=8, initial_pi=pi0,
                                 initial_theta=th0, stephens=st)
    want = oracle.stickbreaking(X, pi0, th0, 7, 6, 1.0, 0.5, 0.5, 1, 1, 3, seed=8)
    assert np.array_equal(got["z_original"], want["z"]) and np.array_equal(got["pi"], want["pi"])
    assert st.cube.shape == (N, 6, 5)
    assert not st.cube[:, :, :3].any()                             # sweeps -2, -1, 0
    for i in (0, 99, N - 1):                                       # slice 3 = sweep 1: drawn from (pi0, theta0)
        _, norm = oracle.sb_cond(X, i, pi0, th0, spec=True)
        assert np.array_equal(st.cube[i, :, 3], norm)
    assert sorted(st.samples) == [3, 4, 5, 6]


@pytest.mark.parametrize("N,P,K,batch", [(1500, 30, 40, 1500),     # more than 32 categories: the 512-thread emitting twin
                                         (1200, 128, 24, 1200)])   # own-cluster tables beyond LDS (second tier)
def test_probability_hand_off_on_the_wide_and_second_tier_kernels(oracle, N, P, K, batch):
    X, _, _, _ = synth(N, P, 4, K)
    z0 = _z0(N, K, 2)
    st = RecordingStephens(K)
    ns, burnin = 5, 2
    got = bm.gibbs_collapsed(X, ns, K, alpha=0.7, burnin=burnin, relabel=True, burnrelabel=1, seed=19, batch=batch,
                             initial_K=z0, stephens=st)
    want = oracle.collapsed(X, z0, ns, K, 0.7, 0.5, 0.5, 1, 1, 0, seed=19, batch=batch)
    assert np.array_equal(got["z_original"], want["z"][burnin:])
    for j, m in [(1, st.cube[:, :, 0])] + [(j, st.samples[j]) for j in (2, 4)]:
        for i in (0, N // 3, N - 1):
            _, norm = oracle.collapsed_cond(X, want["z"][j - 1], i, K, 0.7, 0.5, 0.5, spec=True)
            assert np.array_equal(m[i], norm), (j, i)


def test_a_failing_hook_stops_the_run_cleanly():
    X, _, _, _ = synth(500, 8, 2, 1)

    class Boom(RecordingStephens):
        def online(self, Q, p, j):
            raise RuntimeError("stephens failed at %d" % j)

    with pytest.raises(RuntimeError, match="stephens failed at 2"):
        bm.gibbs_collapsed(X, 6, 2, burnin=2, relabel=True, burnrelabel=1, seed=1, stephens=Boom(2))
    # and the library is still usable afterwards
    assert bm.gibbs_collapsed(X, 4, 2, burnin=1, seed=1)["z"].shape == (3, 500)
