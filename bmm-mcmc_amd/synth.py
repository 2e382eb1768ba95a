"""Synthetic Bernoulli-mixture data of the shape the benchmarks name (SURVEY.md section 8d,
mirroring the recipe in the reference's R/simulate_data.R:7-18: per-cluster Bernoulli
columns, cluster weights proportional to K, K-1, ..., 1, theta = 0.1 + 0.8 U, rows shuffled).

`device_matrix` builds the N x P matrix directly in HBM in R's layout (int32, column-major:
stored as a contiguous (P, N) tensor) so that benchmark inputs never cross PCIe.
torch is used for device memory and its RNG only.
"""
import numpy as np

WORKLOADS = {
    # name: (sampler, K or maxK, K_true, N, P, data seed)   -- BASELINE.md section 3
    "c2": ("collapsed", 3, 3, 100_000, 20, 18),
    "c3": ("dp", 30, 10, 1_000_000, 50, 19),
    "c4": ("stickbreaking", 50, 10, 1_000_000, 50, 20),
    "c5": ("collapsed", 20, 20, 10_000_000, 100, 21),
    "ns": ("collapsed", 20, 20, 1_000_000, 50, 22),  # the north-star point K=20, N=1e6, P=50
}


def truth(K_true, P, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    w = np.arange(K_true, 0, -1, dtype=np.float64)
    w /= w.sum()
    theta = 0.1 + 0.8 * rng.random((K_true, P))
    return w, theta


def counts(N, w):
    n = np.round(N * w).astype(np.int64)
    n[-1] = N - n[:-1].sum()
    return n


def device_matrix(N, P, K_true, seed, device):
    """(X, labels): X int32 (P, N) contiguous on `device` = N x P column-major; labels int64 (N,)."""
    import torch
    w, theta = truth(K_true, P, seed)
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    n = torch.as_tensor(counts(N, w), device=device)
    labels = torch.repeat_interleave(torch.arange(K_true, device=device), n)
    labels = labels[torch.randperm(N, generator=g, device=device)]
    th = torch.as_tensor(theta, dtype=torch.float32, device=device)
    X = torch.empty((P, N), dtype=torch.int32, device=device)
    for d in range(P):
        X[d] = (torch.rand(N, generator=g, device=device) < th[labels, d]).to(torch.int32)
    return X, labels


def host_matrix(N, P, K_true, seed, shuffle=True):
    """NumPy twin for test-sized inputs: (X Fortran-ordered int32 N x P, labels, theta, w)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    w = np.arange(K_true, 0, -1, dtype=np.float64)
    w /= w.sum()
    n = counts(N, w)
    theta = 0.1 + 0.8 * rng.random((K_true, P))
    labels = np.repeat(np.arange(K_true), n)
    X = (rng.random((N, P)) < theta[labels]).astype(np.int32)
    if shuffle:
        perm = rng.permutation(N)
        X, labels = X[perm], labels[perm]
    return np.asfortranarray(X), labels, theta, w
