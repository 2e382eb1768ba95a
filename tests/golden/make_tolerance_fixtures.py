"""Generates tests/golden/tolerance_*.json: posterior summaries of the ORACLE AT BATCH = 1 -- the reference's
sequential scan (collapsed_gibbs.cpp:86-182, collapsed_gibbs_dp.cpp:108-242) -- on SURVEY.md section 8(d)'s
synthetic data at the shapes the benchmark numbers are quoted on.  The -m gpu tests
(tests/test_gpu_tolerance_fixtures.py) hold the HIP path at its DEFAULT batch to the stated tolerance
(include/bmm_mcmc.h: BMM_TOL_PROPORTIONS, BMM_TOL_THETA) against these numbers.

Run in the build container (CPU only; about half an hour on 6 threads for every case):

    python tests/golden/make_tolerance_fixtures.py [--cases ns,c2,c5s,dp5_1e5,...] [--threads 6]

Cases (data = bmm_mcmc_amd.synth.host_matrix(N, P, K_true, data_seed): weights prop. to K..1, theta = 0.1 + 0.8 U,
rows shuffled; priors beta = gamma = 0.5, alpha sampled with a = b = 1, as the wrappers default):
  ns    gibbs_collapsed K = 20, N = 1e6, P = 50   (the north-star shape), data seed 22
  c2    gibbs_collapsed K = 3,  N = 1e5, P = 20   (BASELINE config 2),    data seed 18
  c5s   gibbs_collapsed K = 20, N = 2e6, P = 100  -- the C5 generator (data seed 21) at a fifth of C5's N:
        the sequential scan costs about 6 s per sweep there, 30 s at the full 1e7
  dp*   gibbs_dp maxK = 30, P = 50: K_true 5 and 10 at N = 1e5 and 4e5; c3 = BASELINE config 3 (N = 1e6, K_true 10)
Collapsed chains start from the generating allocation (so that mode trapping does not pass for batch bias)
and from a uniformly random one; three chain seeds each.

Per chain the fixture holds: posterior-mean cluster proportions sorted by size (label-switching invariant),
their standard deviation over the kept sweeps, theta-hat with the clusters of every sweep ordered by size,
and for the DP sampler the distribution of the number of clusters per sweep (all, and those above N/1000)
and which generating component each final cluster's members come from.
"""
import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from bmm_mcmc_amd import synth  # noqa: E402  (host-side generator only; no GPU, no HIP library)
from oracle import oracle  # noqa: E402

sys.path.insert(0, os.path.join(ROOT, "tests"))
from tolerance_cases import CASES, CHAIN_SEEDS, initial_labels, summarise  # noqa: E402

def run_case(name, threads):
    sampler, K, K_true, N, P, dseed, burn, keep, inits = CASES[name]
    X, labels, theta, w = synth.host_matrix(N, P, K_true, dseed)
    nsamples = burn + keep
    jobs = [(init, s) for init in inits for s in CHAIN_SEEDS]

    def one(job):
        init, seed = job
        t0 = time.time()
        z0 = initial_labels(init, labels, K, seed)
        r = oracle.counts_summary(sampler, X, z0, nsamples, K, 0.0, 0.5, 0.5, 1.0, 1.0, burn, seed=seed, batch=1)
        out = summarise(sampler, r, N, K, K_true, labels)
        out.update({"init": init, "seed": seed, "seconds": round(time.time() - t0, 1)})
        print("[%s] init=%s seed=%d done in %.0f s: props %s" % (name, init, seed, time.time() - t0,
                                                                 np.round(out["props_mean"][:4], 4)), flush=True)
        return out

    with ThreadPoolExecutor(max_workers=threads) as ex:  # ctypes drops the GIL inside the oracle
        chains = list(ex.map(one, jobs))
    doc = {
        "what": "oracle at batch = 1 (the reference's sequential scan); made by tests/golden/make_tolerance_fixtures.py",
        "case": name, "sampler": sampler, "K": K, "K_true": K_true, "N": N, "P": P, "data_seed": dseed,
        "generator": "bmm_mcmc_amd.synth.host_matrix(N, P, K_true, data_seed)",
        "burnin": burn, "kept": keep, "alpha": "sampled (a = b = 1)", "beta": 0.5, "gamma": 0.5,
        "true_weights": w.tolist(), "true_theta": theta.round(6).tolist(),
        "chains": chains,
    }
    path = os.path.join(HERE, "tolerance_%s.json" % name)
    with open(path, "w") as f:
        json.dump(doc, f, indent=None, separators=(",", ":"))
        f.write("\n")
    print("wrote", path, flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default=",".join(CASES))
    ap.add_argument("--threads", type=int, default=6)
    a = ap.parse_args()
    oracle.build()
    for name in a.cases.split(","):
        run_case(name, a.threads)
