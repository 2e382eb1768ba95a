#!/usr/bin/env python3
"""From a uniformly random allocation, how often does a chain end in the generating mode (every generating
component held by exactly one cluster) -- the sequential scan (batch 1) against the default batch (N/8)?
Oracle on CPU (the HIP path equals it bit for bit at equal batch and seed).  tests/tools/mode_trap_scan.py N seeds sweeps
Answers whether the batch makes the burn-in from a random start more trap-prone (tests/test_gpu_tolerance_fixtures.py
sees 6 of 12 against 8 of 12 over its four shapes x three seeds: too few to tell)."""
import json
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from bmm_mcmc_amd import synth  # noqa: E402
from oracle import oracle  # noqa: E402
from tolerance_cases import summarise  # noqa: E402

N, nseeds, sweeps = int(float(sys.argv[1])), int(sys.argv[2]), int(sys.argv[3])
K, P = 20, 50
X, labels, _, _ = synth.host_matrix(N, P, K, 22)
oracle.build()


def one(job):
    batch, seed = job
    z0 = np.random.default_rng(seed).integers(1, K + 1, N).astype(np.int32)
    r = oracle.counts_summary("collapsed", X, z0, sweeps + 1, K, 0.0, 0.5, 0.5, 1.0, 1.0, sweeps, seed=seed, batch=batch)
    s = summarise("collapsed", r, N, K, K, labels)
    return batch, seed, s["final_clusters_per_component"], s["final_agreement"]


BATCHES = [int(b) for b in os.environ.get("BATCHES", "1,%d" % max(1, N // 8)).split(",")]
jobs = [(b, 2000 + s) for s in range(nseeds) for b in BATCHES]
with ThreadPoolExecutor(max_workers=int(os.environ.get("THREADS", "6"))) as ex:
    res = list(ex.map(one, jobs))
out = {"N": N, "K": K, "P": P, "sweeps": sweeps, "seeds": nseeds, "chains": []}
for b in BATCHES:
    whole = [all(v == 1 for v in comp) for bb, _, comp, _ in res if bb == b]
    agree = [a for bb, _, _, a in res if bb == b]
    out["batch_%d" % b] = {"in_generating_mode": int(sum(whole)), "of": len(whole), "mean_final_agreement": float(np.mean(agree))}
out["chains"] = [{"batch": b, "seed": s, "clusters_per_component": c, "agreement": round(a, 4)} for b, s, c, a in res]
print(json.dumps(out))
