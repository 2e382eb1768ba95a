"""Shapes and summaries of the batch-tolerance fixtures (tests/golden/tolerance_*.json): shared by the generator
(tests/golden/make_tolerance_fixtures.py, oracle at batch = 1), the CPU preview (tests/tools/tolerance_eval.py, oracle at
any batch) and the GPU tests (tests/test_gpu_tolerance_fixtures.py, HIP path at the default batch)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

CASES = {
    # name: sampler, K (or maxK), K_true, N, P, data seed, burnin, kept, inits
    "ns": ("collapsed", 20, 20, 1_000_000, 50, 22, 40, 100, ("truth", "random")),
    "nsq": ("collapsed", 20, 20, 262_144, 50, 24, 40, 100, ("truth", "random")),
    "nsb": ("collapsed", 20, 20, 524_288, 50, 23, 40, 100, ("truth", "random")),
    "nse5": ("collapsed", 20, 20, 100_000, 50, 25, 40, 100, ("truth", "random")),
    "nse16": ("collapsed", 20, 20, 65_536, 50, 26, 40, 100, ("truth", "random")),   # the smallest N whose default batch is N/4
    "c2": ("collapsed", 3, 3, 100_000, 20, 18, 100, 200, ("truth", "random")),
    "c5s": ("collapsed", 20, 20, 2_000_000, 100, 21, 40, 100, ("truth", "random")),
    "c5": ("collapsed", 20, 20, 10_000_000, 100, 21, 40, 100, ("truth", "random")),
    "dp5_e16": ("dp", 30, 5, 65_536, 50, 79, 150, 100, ("empty",)),     # the smallest N whose default batch is N/4
    "dp10_e16": ("dp", 30, 10, 65_536, 50, 80, 150, 100, ("empty",)),
    "dp5_1e5": ("dp", 30, 5, 100_000, 50, 77, 150, 100, ("empty",)),
    "dp5_4e5": ("dp", 30, 5, 400_000, 50, 77, 150, 100, ("empty",)),
    "dp10_1e5": ("dp", 30, 10, 100_000, 50, 78, 150, 100, ("empty",)),
    "dp10_4e5": ("dp", 30, 10, 400_000, 50, 78, 150, 100, ("empty",)),
    "c3": ("dp", 30, 10, 1_000_000, 50, 19, 100, 100, ("empty",)),
}
CHAIN_SEEDS = (1000, 1001, 1002)
TOP = 12  # DP: rows of the size-ordered summaries that are kept


def initial_labels(init, labels, K, seed):
    if init == "truth":
        return (labels + 1).astype(np.int32)
    if init == "random":
        return np.random.default_rng(seed).integers(1, K + 1, labels.size).astype(np.int32)
    return None


def summarise(sampler, r, N, K, K_true, labels):
    """label-switching-invariant summaries of one chain (r = oracle.counts_summary(...) or the same fields
    from the HIP path): shared with tests/test_gpu_tolerance_fixtures.py"""
    nk = r["nk"].astype(np.float64)                       # (S, K)
    S = nk.shape[0]
    order = np.argsort(-nk, axis=1, kind="stable")        # clusters of every sweep by size
    props = np.take_along_axis(nk, order, axis=1) / N
    th = np.stack([r["theta"][order[s], :, s] for s in range(S)])   # (S, K, P)
    keep = K if sampler == "collapsed" else TOP
    out = {
        "props_mean": props.mean(axis=0)[:keep].round(7).tolist(),
        "props_sd": props.std(axis=0)[:keep].round(7).tolist(),
        "theta_by_size": np.nanmean(th, axis=0)[:keep].round(6).tolist(),
        "alpha_mean": float(np.mean(r["alpha"])),
    }
    if sampler == "dp":
        used = (nk > 0).sum(axis=1)
        big = (nk > N // 1000).sum(axis=1)
        out["k_used_hist"] = np.bincount(used, minlength=K + 1).tolist()
        out["k_big_hist"] = np.bincount(big, minlength=K + 1).tolist()
        out["k_used_mean"] = float(used.mean())
        out["k_big_mean"] = float(big.mean())
    # the final allocation against the generating one: members of each final cluster by generating component
    z = r["z_last"] - 1
    tab = np.zeros((K, K_true), dtype=np.int64)
    np.add.at(tab, (z, labels), 1)
    rows = np.argsort(-tab.sum(axis=1), kind="stable")
    out["final_crosstab_by_size"] = tab[rows][: (K if sampler == "collapsed" else TOP)].tolist()
    out["final_agreement"] = float(tab.max(axis=1).sum() / N)
    # per generating component: how many final clusters above N/1000 draw most of their members from it
    owner = tab.argmax(axis=1)
    bigc = tab.sum(axis=1) > N // 1000
    out["final_clusters_per_component"] = [int((bigc & (owner == c)).sum()) for c in range(K_true)]
    # posterior-mean share of the observations per GENERATING component: every label counts towards the
    # component most of its final members come from (a cluster keeps its label while it lives).  For the DP
    # sampler this is the summary that does not depend on which component happened to be seated as two
    # clusters (the sequential scan does that too, see the fixtures' final_clusters_per_component).
    live = tab.sum(axis=1) > 0
    out["props_by_component"] = [float(nk[:, live & (owner == c)].sum(axis=1).mean() / N) for c in range(K_true)]
    return out




def load_fixture(name):
    with open(os.path.join(GOLDEN, "tolerance_%s.json" % name)) as f:
        return json.load(f)


def compare(doc, init, chains):
    """max |mean over seeds of `chains` - mean over seeds of the fixture's batch-1 chains| for the chains of
    one initialisation: (proportions per component, theta-hat per cell, and for the DP sampler the mean number
    of clusters above N/1000 on both sides)"""
    ref = [c for c in doc["chains"] if c["init"] == init]
    got = [c for c in chains if c["init"] == init]
    assert len(ref) == len(got) > 0
    mean = lambda cs, key: np.mean([np.array(c[key], dtype=np.float64) for c in cs], axis=0)
    dp = np.abs(mean(got, "props_mean") - mean(ref, "props_mean"))
    dt = np.abs(mean(got, "theta_by_size") - mean(ref, "theta_by_size"))
    out = {"props": dp, "theta": dt}
    if "props_by_component" in ref[0]:
        out["by_component"] = np.abs(mean(got, "props_by_component") - mean(ref, "props_by_component"))
        out["by_component_vs_truth"] = np.abs(mean(got, "props_by_component") - np.array(doc["true_weights"]))
    if doc["sampler"] == "dp":
        out["k_big"] = (float(np.mean([c["k_big_mean"] for c in got])), float(np.mean([c["k_big_mean"] for c in ref])))
        out["k_used"] = (float(np.mean([c["k_used_mean"] for c in got])), float(np.mean([c["k_used_mean"] for c in ref])))
    return out
