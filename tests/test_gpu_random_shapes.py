"""A fixed pseudo-random sample of shapes, batch sizes and samplers against the oracle: every draw
lands somewhere in the dispatch table (accumulator counts 4..64, one or two lanes per observation,
workgroup sizes, own-cluster tables in LDS or in global memory, ragged last tiles and batches)."""
import os

import numpy as np
import pytest

import bmm_mcmc_amd as bm
from util import synth

pytestmark = pytest.mark.gpu


def _case(seed):
    r = np.random.default_rng(seed)
    sampler = ["collapsed", "collapsed", "dp", "stickbreaking", "full"][int(r.integers(0, 5))]
    K = int(r.choice([2, 3, 4, 5, 8, 9, 12, 13, 16, 20, 21, 24, 28, 31, 32, 33, 40, 47, 48, 56, 63, 64]))
    if sampler == "dp":
        K = max(K, 3)
    P = int(r.choice([1, 2, 3, 4, 5, 7, 8, 16, 20, 31, 32, 33, 50, 64, 65, 96, 100, 127, 128]))
    N = int(r.integers(1, 6000))
    batch = int(r.choice([1 if N < 300 else 37, 64, 100, 255, 256, 257, 1000, 1024, N, max(1, N // 8)]))
    return sampler, N, P, K, min(batch, N)


# BMM_RANDOM_CASES=N widens the sample for a one-off soak
@pytest.mark.parametrize("seed", range(100, 100 + int(os.environ.get("BMM_RANDOM_CASES", "48"))))
def test_random_shape_equals_oracle(oracle, seed):
    sampler, N, P, K, batch = _case(seed)
    X, _, _, _ = synth(N, P, min(K, 4), seed)
    rng = np.random.default_rng(seed + 1)
    sweeps = 4
    if sampler == "collapsed":
        z0 = rng.integers(1, K + 1, N).astype(np.int32)
        got = bm.gibbs_collapsed(X, sweeps, K, burnin=0, seed=seed, batch=batch, initial_K=z0)
        want = oracle.collapsed(X, z0, sweeps, K, 0.0, 0.5, 0.5, 1, 1, 0, seed=seed, batch=batch)
        keys = ("z", "theta", "alpha")
    elif sampler == "dp":
        got = bm.gibbs_dp(X, sweeps, burnin=0, maxK=K, seed=seed, batch=batch)
        want = oracle.dp(X, sweeps, 0.0, 0.5, 0.5, 1, 1, 0, K, seed=seed, batch=batch)
        keys = ("z", "theta", "alpha")
    else:
        pi0 = rng.dirichlet(np.ones(K))
        th0 = 0.05 + 0.9 * rng.random((K, P))
        if sampler == "stickbreaking":
            got = bm.gibbs_stickbreaking(X, sweeps, K, burnin=0, seed=seed, initial_pi=pi0, initial_theta=th0)
            want = oracle.stickbreaking(X, pi0, th0, sweeps, K, 0.0, 0.5, 0.5, 1, 1, 0, seed=seed)
        else:
            got = bm.gibbs_full(X, sweeps, K, burnin=0, seed=seed, initial_pi=pi0, initial_theta=th0)
            want = oracle.full(X, pi0, th0, sweeps, K, 0.0, 0.5, 0.5, 1, 1, 0, seed=seed)
        keys = ("z", "theta", "alpha", "pi")
    for k in keys:
        assert np.array_equal(got[k], want[k], equal_nan=True), (k, sampler, N, P, K, batch)
