"""Helpers shared by the tests: committed fixtures, synthetic data, summaries."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_dataset(name):
    with open(os.path.join(GOLDEN, name + ".txt")) as f:
        rows = [[int(c) for c in line.strip()] for line in f if line.strip()]
    return np.asfortranarray(np.array(rows, dtype=np.int32))


def load_kats():
    with open(os.path.join(GOLDEN, "kats.json")) as f:
        return json.load(f)


def proportions(z, K):
    """Posterior-mean cluster proportions, sorted descending (label-switching invariant)."""
    z = np.asarray(z)
    S = z.shape[0]
    props = np.stack([np.sort(np.bincount(z[s] - 1, minlength=K)[:K] / z.shape[1])[::-1] for s in range(S)])
    return props.mean(axis=0)


def synth(N, P, K_true, seed, shuffle=True):
    """SURVEY.md section 8(d) generator: weights ∝ (K,K-1,..,1), theta = 0.1 + 0.8 U."""
    rng = np.random.Generator(np.random.PCG64(seed))
    w = np.arange(K_true, 0, -1, dtype=np.float64)
    w /= w.sum()
    n = np.round(N * w).astype(np.int64)
    n[-1] = N - n[:-1].sum()
    theta = 0.1 + 0.8 * rng.random((K_true, P))
    labels = np.repeat(np.arange(K_true), n)
    X = (rng.random((N, P)) < theta[labels]).astype(np.int32)
    if shuffle:
        perm = rng.permutation(N)
        X, labels = X[perm], labels[perm]
    return np.asfortranarray(X), labels, theta, w
