"""The batch contract's stated tolerance (include/bmm_mcmc.h, BMM_TOL_PROPORTIONS / BMM_TOL_THETA), held on
the GPU: the HIP path at the library's DEFAULT batch against the oracle's batch = 1 chain -- the reference's
sequential scan -- on the three data sets the reference bundles, over three seeds; and theta-hat against the
generating values the reference documents for them (R/bmm-mcmc.R:13-17, 31-35, 46-50), the only numbers the
reference holds.  Different batches are different chains, so the comparison is of posterior summaries:
sorted posterior-mean cluster proportions (label-switching invariant)."""
import itertools

import numpy as np
import pytest

import bmm_mcmc_amd as bm
from util import load_dataset, proportions

pytestmark = pytest.mark.gpu

# R/bmm-mcmc.R:13-17 (K2_N100_P5), 31-35 (K2_N1000_P5), 46-50 (K3_N1000_P5): mixing ratios and theta
TRUTH = {
    "K2_N100_P5": ([0.7, 0.3], [[0.7, 0.8, 0.2, 0.1, 0.1], [0.2, 0.2, 0.9, 0.8, 0.6]]),
    "K2_N1000_P5": ([0.7, 0.3], [[0.7, 0.8, 0.2, 0.1, 0.1], [0.2, 0.2, 0.9, 0.8, 0.6]]),
    "K3_N1000_P5": ([0.6, 0.2, 0.2], [[0.7, 0.8, 0.2, 0.1, 0.1], [0.3, 0.5, 0.9, 0.8, 0.6],
                                     [0.1, 0.2, 0.5, 0.4, 0.9]]),
}
SEEDS = (101, 202, 303)
NS, BURN = 600, 200


def _z0(N, K, seed):
    return np.random.default_rng(seed).integers(1, K + 1, N).astype(np.int32)


@pytest.mark.parametrize("name,K", [("K2_N100_P5", 2), ("K2_N1000_P5", 2), ("K3_N1000_P5", 3)])
def test_default_batch_proportions_within_the_stated_tolerance_of_the_sequential_scan(oracle, name, K):
    X = load_dataset(name)
    N = X.shape[0]
    assert bm.default_batch("collapsed", N) == max(1, N // 8) > 1
    hip, seq = [], []
    for s in SEEDS:
        z0 = _z0(N, K, s)
        got = bm.gibbs_collapsed(X, NS, K, burnin=BURN, seed=s, initial_K=z0)              # default batch
        want = oracle.collapsed(X, z0, NS, K, 0.0, 0.5, 0.5, 1, 1, BURN, seed=s, batch=1)  # the reference's scan
        hip.append(proportions(got["z"], K))
        seq.append(proportions(want["z"], K))
    hip, seq = np.mean(hip, axis=0), np.mean(seq, axis=0)
    assert np.abs(hip - seq).max() <= bm.TOL_PROPORTIONS, (hip, seq)
    # and both sit where the reference says the data were generated (finite-sample posterior: SURVEY 8c)
    truth = np.array(TRUTH[name][0])
    assert abs(hip[0] - truth[0]) < (0.05 if N <= 100 else 0.03)


def _theta_by_size(r, K):
    """posterior-mean theta-hat with the clusters of every sweep ordered by size (label-switching invariant)"""
    th = []
    for s in range(r["z"].shape[0]):
        order = np.argsort(-np.bincount(r["z"][s] - 1, minlength=K), kind="stable")
        th.append(r["theta"][order, :, s])
    return np.nanmean(th, axis=0)


@pytest.mark.parametrize("name,K,truth_tol", [("K2_N100_P5", 2, 0.15), ("K2_N1000_P5", 2, bm.TOL_THETA),
                                               ("K3_N1000_P5", 3, 0.2)])
def test_theta_hat_within_the_stated_tolerance(oracle, name, K, truth_tol):
    X = load_dataset(name)
    N = X.shape[0]
    hip, seq = [], []
    for s in SEEDS:
        z0 = _z0(N, K, s)
        hip.append(_theta_by_size(bm.gibbs_collapsed(X, NS, K, burnin=BURN, seed=s, initial_K=z0), K))
        seq.append(_theta_by_size(oracle.collapsed(X, z0, NS, K, 0.0, 0.5, 0.5, 1, 1, BURN, seed=s, batch=1), K))
    hip, seq = np.mean(hip, axis=0), np.mean(seq, axis=0)
    assert np.abs(hip - seq).max() <= bm.TOL_THETA, np.abs(hip - seq).max()
    # against the generating values the reference documents -- the only reference-held numbers there are.
    # K2_N1000_P5 meets the stated tolerance against them too; the N = 100 set and the overlapping K3 set
    # are finite-sample posteriors up to 0.12 / 0.16 away at batch 1 as at the default batch.
    truth = np.array(TRUTH[name][1])
    err = min(np.abs(hip[list(perm)] - truth).max() for perm in itertools.permutations(range(K)))
    assert err <= truth_tol, err


def test_dp_default_batch_against_the_sequential_scan(oracle):
    """The DP sampler at its default batch (N/16): the two dominant components of K2_N1000_P5 within the
    stated tolerance x 2 of the batch-1 chain (an unbounded-K posterior keeps some mass in small extra
    clusters, and a DP chain wanders more between seeds: six seeds here)."""
    X = load_dataset("K2_N1000_P5")
    hip, seq = [], []
    for s in (1, 2, 3, 4, 5, 6):
        got = bm.gibbs_dp(X, 400, burnin=150, maxK=30, seed=s)
        want = oracle.dp(X, 400, 0.0, 0.5, 0.5, 1, 1, 150, 30, seed=s, batch=1)
        hip.append(proportions(got["z"], 30)[:2])
        seq.append(proportions(want["z"], 30)[:2])
    hip, seq = np.mean(hip, axis=0), np.mean(seq, axis=0)
    assert np.abs(hip - seq).max() <= 2 * bm.TOL_PROPORTIONS + 0.02, (hip, seq)


def test_default_batch_on_a_larger_shuffled_mixture(oracle):
    """the same contract away from the bundled sets: N = 20 000, K = 4, P = 12 (SURVEY 8d's generator, rows
    shuffled), HIP at the default batch (N/8 = 2500) against the oracle's sequential scan, two seeds --
    proportions and theta-hat within the stated tolerances of each other and of the generating values"""
    from util import synth
    N, P, K = 20000, 12, 4
    X, _, theta_true, w_true = synth(N, P, K, 77)
    assert bm.default_batch("collapsed", N) == N // 8
    hip_p, seq_p, hip_t, seq_t = [], [], [], []
    for s in (11, 12):
        z0 = _z0(N, K, s)
        got = bm.gibbs_collapsed(X, 160, K, burnin=80, seed=s, initial_K=z0)
        want = oracle.collapsed(X, z0, 160, K, 0.0, 0.5, 0.5, 1, 1, 80, seed=s, batch=1)
        hip_p.append(proportions(got["z"], K)); seq_p.append(proportions(want["z"], K))
        hip_t.append(_theta_by_size(got, K)); seq_t.append(_theta_by_size(want, K))
    hip_p, seq_p = np.mean(hip_p, axis=0), np.mean(seq_p, axis=0)
    hip_t, seq_t = np.mean(hip_t, axis=0), np.mean(seq_t, axis=0)
    assert np.abs(hip_p - seq_p).max() <= bm.TOL_PROPORTIONS, (hip_p, seq_p)
    assert np.abs(hip_t - seq_t).max() <= bm.TOL_THETA, np.abs(hip_t - seq_t).max()
    assert np.abs(hip_p - np.sort(w_true)[::-1]).max() <= 0.02, hip_p
    order = np.argsort(-w_true, kind="stable")
    assert np.abs(hip_t - theta_true[order]).max() <= bm.TOL_THETA, np.abs(hip_t - theta_true[order]).max()
