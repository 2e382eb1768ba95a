"""Known-answer vectors for the RNG-free conditionals, computed with NumPy fp64
directly from the reference's formulas (not from the oracle, not from the HIP path).

    collapsed   src/collapsed_gibbs.cpp:105-130,148-150
    dp          src/collapsed_gibbs_dp.cpp:71,102-106,145-160,174-186
    stick-br.   src/stickbreaking.cpp:80,89,103-105
    theta-hat   src/collapsed_gibbs.cpp:205-219

The reference holds no golden vectors of its own and cannot be run here (no R), so
these are restatements of its arithmetic, not outputs of it: PARITY UNPINNED.
State: labels z_i = 1 + (i mod K) on each bundled dataset; alpha = 1, beta = gamma = 0.5.
Writes kats.json next to this file.  Needs only the committed *.txt fixtures.
"""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def load(name):
    with open(os.path.join(HERE, name + ".txt")) as f:
        return np.array([[int(c) for c in line.strip()] for line in f if line.strip()], dtype=np.int64)


def collapsed(X, z, i, K, alpha, beta, gamma):
    N, P = X.shape
    keep = np.arange(N) != i
    raw = np.zeros(K)
    for k in range(K):
        mem = keep & (z == k + 1)
        Nk = int(mem.sum())
        if Nk == 0:
            continue
        LHS = np.log(Nk + alpha / K) - np.log(N - 1 + alpha)
        logLH = 0.0
        for d in range(P):
            s = int(X[mem, d].sum())
            x = int(X[i, d])
            logLH += x * np.log(beta + s) + (1 - x) * np.log(gamma + Nk - s) - np.log(beta + gamma + Nk)
        raw[k] = np.exp(LHS + logLH)
    return raw, raw / raw.sum()


def dp(X, z, i, K, alpha, beta, gamma):
    N, P = X.shape
    keep = np.arange(N) != i
    left_denom = np.log(N - 1 + alpha)
    logw = np.zeros(K + 1)
    for k in range(K):
        mem = keep & (z == k + 1)
        Nk = int(mem.sum())
        LHS = np.log(Nk) - left_denom
        denom = np.log(beta + gamma + Nk)
        logLH = 0.0
        for d in range(P):
            s = int(X[mem, d].sum())
            x = int(X[i, d])
            logLH += x * np.log(beta + s) + (1 - x) * np.log(gamma + Nk - s) - denom
        logw[k] = LHS + logLH
    logw[K] = np.log(alpha) - left_denom + P * (np.log(beta) - np.log(beta + gamma))
    w = np.exp(logw - logw.max())
    return logw, w / w.sum()


def sb(X, i, pi, theta):
    K, P = theta.shape
    raw = np.zeros(K)
    for k in range(K):
        loglh = 0.0
        for d in range(P):
            x = int(X[i, d])
            loglh += x * np.log(theta[k, d]) + (1 - x) * np.log(1 - theta[k, d])
        raw[k] = np.exp(np.log(pi[k]) + loglh)
    return raw, raw / raw.sum()


TRUTH = {  # R/bmm-mcmc.R:13-17, 31-35, 46-50
    "K2_N100_P5": ([0.7, 0.3], [[0.7, 0.8, 0.2, 0.1, 0.1], [0.2, 0.2, 0.9, 0.8, 0.6]]),
    "K2_N1000_P5": ([0.7, 0.3], [[0.7, 0.8, 0.2, 0.1, 0.1], [0.2, 0.2, 0.9, 0.8, 0.6]]),
    "K3_N1000_P5": ([0.6, 0.2, 0.2], [[0.7, 0.8, 0.2, 0.1, 0.1], [0.3, 0.5, 0.9, 0.8, 0.6],
                                      [0.1, 0.2, 0.5, 0.4, 0.9]]),
}

out = {"alpha": 1.0, "beta": 0.5, "gamma": 0.5, "datasets": {}}
for name, (pi, theta) in TRUTH.items():
    X = load(name)
    N, P = X.shape
    K = len(pi)
    z = 1 + (np.arange(N) % K)
    Nk = [int((z == k + 1).sum()) for k in range(K)]
    S = [[int(X[z == k + 1, d].sum()) for d in range(P)] for k in range(K)]
    rec = {"K": K, "N": N, "P": P, "Nk": Nk, "S": S,
           "theta_hat": [[S[k][d] / Nk[k] for d in range(P)] for k in range(K)],
           "pi_true": pi, "theta_true": theta, "cases": []}
    for i in (0, 1, N // 2, N - 1):
        craw, cnorm = collapsed(X, z, i, K, 1.0, 0.5, 0.5)
        dlog, dnorm = dp(X, z, i, K, 1.0, 0.5, 0.5)
        sraw, snorm = sb(X, i, np.array(pi), np.array(theta))
        rec["cases"].append({"i": i, "x": [int(v) for v in X[i]],
                             "collapsed_raw": craw.tolist(), "collapsed_norm": cnorm.tolist(),
                             "dp_logw": dlog.tolist(), "dp_norm": dnorm.tolist(),
                             "sb_raw": sraw.tolist(), "sb_norm": snorm.tolist()})
    out["datasets"][name] = rec
with open(os.path.join(HERE, "kats.json"), "w") as f:
    json.dump(out, f, indent=1)
c = out["datasets"]["K2_N100_P5"]["cases"]
print(c[0]["collapsed_raw"], c[0]["dp_logw"], c[0]["sb_norm"])
print(c[3]["collapsed_norm"], c[3]["dp_norm"], c[3]["sb_norm"])
