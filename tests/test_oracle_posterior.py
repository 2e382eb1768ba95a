"""An independent pin of WHAT the samplers sample from: for data small enough to enumerate every allocation, the
exact posterior of the model the reference's formulas define -- computed here by brute force with scipy's log-beta /
log-gamma, nothing of the samplers' own arithmetic -- against the long-run frequencies of the oracle's chains at
batch 1 (which the HIP path equals bit for bit, tests/test_gpu_parity.py).  Not a restatement of the conditional
formulas (tests/test_oracle_kats.py does that): a sampler with a wrong conditional, a wrong variate generator or a
wrong bookkeeping rule has a different stationary law and fails here.

  gibbs_dp             CRP(alpha) prior over partitions x Beta-Bernoulli marginal likelihood
                       (src/collapsed_gibbs_dp.cpp:102-106, 140-186): all 877 partitions of 7 observations
  gibbs_stickbreaking  truncated stick-breaking prior, v_k ~ Beta(1, alpha), v_maxK = 1 (src/stickbreaking.cpp:187-214):
                       p(z) = prod_{k < maxK} B(1 + n_k, alpha + n_{>k}) / B(1, alpha); all 3^7 labelled allocations
  gibbs_full           pi ~ Dirichlet(alpha / K) (src/full_gibbs.cpp:203-210): p(z) = Dirichlet-multinomial; all 2^7
  update_alpha         Escobar & West's two-step draw (src/utils.cpp:6-14) leaves p(alpha | K, N) prop. to
                       Gamma(alpha; a, rate b) alpha^K Gamma(alpha) / Gamma(alpha + N) invariant: numerical quadrature

(The finite collapsed sampler is not in this list: the reference gives an emptied cluster probability exactly 0 for
ever, src/collapsed_gibbs.cpp:104,131-133, so on data this small its chain is absorbed in the one-cluster state; its
conditional is pinned by the known answers instead.)  Seeds are fixed, so the outcomes are deterministic; the
tolerances are about four standard errors of 2e5 correlated sweeps."""
import itertools

import numpy as np
import pytest
from scipy.special import betaln, gammaln

N, P = 7, 3
BETA = GAMMA = 0.5
SWEEPS, BURN = 200_000, 2_000


@pytest.fixture(scope="module")
def data():
    rng = np.random.default_rng(11)
    X = (rng.random((N, P)) < [0.8, 0.3, 0.6]).astype(np.int32)
    X[:3, 0] = 1
    X[4:, 0] = 0
    return np.asfortranarray(X)


def _block_loglik(X, rows):
    n, s = len(rows), X[rows].sum(axis=0)
    return float(np.sum(betaln(BETA + s, GAMMA + n - s) - betaln(BETA, GAMMA)))


def _coclustering(weights, states):
    """(distribution of the number of non-empty clusters, matrix of P(z_i = z_j))"""
    nb, co = np.zeros(N + 1), np.zeros((N, N))
    for w, z in zip(weights, states):
        z = np.asarray(z)
        nb[len(set(z.tolist()))] += w
        co += w * (z[:, None] == z[None, :])
    return nb, co


def _normalised(logw):
    w = np.exp(np.asarray(logw) - np.max(logw))
    return w / w.sum()


def _partitions(n):
    def rec(prefix, m):
        if len(prefix) == n:
            yield tuple(prefix)
            return
        for v in range(m + 1):
            yield from rec(prefix + [v], max(m, v + 1))
    return list(rec([0], 1))


def _canonical(z):
    seen, out = {}, []
    for v in z:
        out.append(seen.setdefault(int(v), len(seen)))
    return tuple(out)


def test_dp_chain_samples_the_crp_posterior_over_all_877_partitions(oracle, data):
    alpha = 1.3
    parts = _partitions(N)
    assert len(parts) == 877
    logw = []
    for p in parts:
        blocks = {}
        for i, b in enumerate(p):
            blocks.setdefault(b, []).append(i)
        logw.append(len(blocks) * np.log(alpha) + sum(gammaln(len(r)) + _block_loglik(data, r) for r in blocks.values()))
    exact = _normalised(logw)
    r = oracle.dp(data, SWEEPS + BURN, alpha, BETA, GAMMA, 1, 1, BURN, 12, seed=5, batch=1)   # maxK = 12 never binds
    index = {p: i for i, p in enumerate(parts)}
    emp = np.bincount([index[_canonical(row)] for row in r["z"]], minlength=len(parts)) / SWEEPS
    nb_e, co_e = _coclustering(exact, parts)
    nb_m, co_m = _coclustering(emp, parts)
    assert np.abs(nb_e - nb_m).max() < 0.004, (nb_e, nb_m)          # measured 6e-4
    assert np.abs(co_e - co_m).max() < 0.008                        # measured 2.2e-3; the entries span 0.12 .. 0.60
    top = np.argsort(-exact)[:20]
    assert np.abs(exact[top] - emp[top]).max() < 0.002              # single partitions of 1.6 % .. 0.6 %
    assert 0.5 * np.abs(exact - emp).sum() < 0.04                   # total variation over 877 cells: sampling noise 0.02


def test_stickbreaking_chain_samples_the_truncated_stick_posterior(oracle, data):
    K, alpha = 3, 1.7
    states = list(itertools.product(range(K), repeat=N))
    logw = []
    for z in states:
        z = np.array(z)
        n = np.bincount(z, minlength=K)
        w = sum(betaln(1 + n[k], alpha + n[k + 1:].sum()) - betaln(1, alpha) for k in range(K - 1))
        w += sum(_block_loglik(data, np.flatnonzero(z == k)) for k in range(K) if n[k])
        logw.append(w)
    exact = _normalised(logw)
    r = oracle.stickbreaking(data, np.ones(K) / K, np.full((K, P), 0.5), SWEEPS + BURN, K, alpha, BETA, GAMMA, 1, 1, BURN, seed=9)
    code = ((r["z"] - 1) * (K ** np.arange(N - 1, -1, -1))).sum(axis=1)
    emp = np.bincount(code, minlength=K ** N) / SWEEPS
    share = lambda w: sum(wi * np.bincount(np.array(z), minlength=K) for wi, z in zip(w, states)) / N
    assert np.abs(share(exact) - share(emp)).max() < 0.006          # label shares are NOT symmetric here: measured 2e-3
    _, co_e = _coclustering(exact, states)
    _, co_m = _coclustering(emp, states)
    assert np.abs(co_e - co_m).max() < 0.008                        # measured 1.5e-3
    # and the sticks themselves: E[pi_k | data] = sum_z p(z) E[pi_k | z], pi | z being the reference's Beta sticks
    def e_pi(z):
        n = np.bincount(np.array(z), minlength=K)
        v = [(1 + n[k]) / (1 + n[k] + alpha + n[k + 1:].sum()) for k in range(K - 1)] + [1.0]
        return np.array([v[k] * np.prod([1 - v[l] for l in range(k)]) for k in range(K)])
    want_pi = sum(w * e_pi(z) for w, z in zip(exact, states))
    assert np.abs(r["pi"].mean(axis=0) - want_pi).max() < 0.006


def test_full_chain_samples_the_dirichlet_multinomial_posterior(oracle, data):
    K, alpha = 2, 2.0
    states = list(itertools.product(range(K), repeat=N))
    logw = []
    for z in states:
        z = np.array(z)
        n = np.bincount(z, minlength=K)
        w = float(np.sum(gammaln(alpha / K + n) - gammaln(alpha / K)))
        w += sum(_block_loglik(data, np.flatnonzero(z == k)) for k in range(K) if n[k])
        logw.append(w)
    exact = _normalised(logw)
    r = oracle.full(data, np.ones(K) / K, np.full((K, P), 0.5), SWEEPS + BURN, K, alpha, BETA, GAMMA, 1, 1, BURN, seed=4)
    code = ((r["z"] - 1) * (K ** np.arange(N - 1, -1, -1))).sum(axis=1)
    emp = np.bincount(code, minlength=K ** N) / SWEEPS
    assert 0.5 * np.abs(exact - emp).sum() < 0.03                   # all 128 labelled allocations: measured 0.010
    _, co_e = _coclustering(exact, states)
    _, co_m = _coclustering(emp, states)
    assert np.abs(co_e - co_m).max() < 0.01                         # measured 3.4e-3
    # theta | z ~ Beta(beta + s, gamma + n - s): posterior mean of theta for the label holding observation 0
    want = sum(w * (BETA + data[np.array(z) == z[0]].sum(axis=0)) / (BETA + GAMMA + (np.array(z) == z[0]).sum())
               for w, z in zip(exact, states))
    k0 = r["z"][:, 0] - 1
    got = r["theta"][k0, :, np.arange(SWEEPS)].mean(axis=0)
    assert np.abs(got - want).max() < 0.01, (got, want)


def test_update_alpha_leaves_the_escobar_west_posterior_invariant(oracle):
    """utils.cpp:6-14 with this build's variate generators: 1e5 successive draws of alpha for fixed K and N against
    p(alpha | K, N) by quadrature"""
    a, b, Nn, K = 1.0, 1.0, 500.0, 6
    grid = np.linspace(1e-4, 30.0, 300_001)
    logp = (a - 1) * np.log(grid) - b * grid + K * np.log(grid) + gammaln(grid) - gammaln(grid + Nn)
    p = np.exp(logp - logp.max())
    p /= p.sum()
    mean, var = float((p * grid).sum()), float((p * grid ** 2).sum() - (p * grid).sum() ** 2)
    alpha, draws = 1.0, []
    for j in range(1, 100_001):
        alpha = oracle.update_alpha(alpha, a, b, Nn, K, 77, j)
        draws.append(alpha)
    draws = np.array(draws[1000:])
    assert abs(draws.mean() - mean) < 0.02 * mean and abs(draws.var() - var) < 0.06 * var, (draws.mean(), mean, draws.var(), var)
    q_exact = np.interp([0.1, 0.5, 0.9], np.cumsum(p), grid)
    assert np.abs(np.quantile(draws, [0.1, 0.5, 0.9]) - q_exact).max() < 0.03 * q_exact[2]
