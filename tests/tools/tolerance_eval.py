"""CPU preview of tests/test_gpu_tolerance_fixtures.py: the ORACLE at a given batch against the committed
batch-1 fixtures (tests/golden/tolerance_*.json).  The HIP path equals the oracle bit for bit at equal batch
and seed (tests/test_gpu_parity.py), so what this prints for batch = "default" is what the GPU test will see.

    python tests/tools/tolerance_eval.py --cases c2,ns --batch default [--threads 6]
    python tests/tools/tolerance_eval.py --cases c5s --batch N/4        # the worst case of the default's rounding

--batch: "default" (bmm_default_batch's rule, restated below so that this tool needs no GPU library),
"N/<d>", or a number.
"""
import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from bmm_mcmc_amd import synth  # noqa: E402
from oracle import oracle  # noqa: E402
from tolerance_cases import CASES, CHAIN_SEEDS, compare, initial_labels, load_fixture, summarise  # noqa: E402


def default_batch(sampler, N):
    """include/bmm_mcmc.h bmm_default_batch (tests/test_capi_cpu.py holds the library to the same rule)"""
    div = 4 if N >= 2 ** 16 else (16 if sampler == "dp" else 8)
    return max(1, N // div)


def parse_batch(s, sampler, N):
    if s == "default":
        return default_batch(sampler, N)
    if s.startswith("N/"):
        return max(1, int(N / float(s[2:])))
    return int(s)


def run(name, batch_spec, threads, out_dir):
    sampler, K, K_true, N, P, dseed, burn, keep, inits = CASES[name]
    doc = load_fixture(name)
    X, labels, theta, w = synth.host_matrix(N, P, K_true, dseed)
    batch = parse_batch(batch_spec, sampler, N)
    jobs = [(init, s) for init in inits for s in CHAIN_SEEDS]

    def one(job):
        init, seed = job
        z0 = initial_labels(init, labels, K, seed)
        r = oracle.counts_summary(sampler, X, z0, burn + keep, K, 0.0, 0.5, 0.5, 1.0, 1.0, burn, seed=seed, batch=batch)
        out = summarise(sampler, r, N, K, K_true, labels)
        out.update({"init": init, "seed": seed})
        return out

    t0 = time.time()
    with ThreadPoolExecutor(max_workers=threads) as ex:
        chains = list(ex.map(one, jobs))
    print("== %s  batch %d (= N/%.2f)  %.0f s" % (name, batch, N / batch, time.time() - t0))
    for init in inits:
        c = compare(doc, init, chains)
        line = "   init=%-6s props max %.5f (worst component %d)  theta max %.5f" % (
            init, c["props"].max(), int(c["props"].argmax()), c["theta"].max())
        if "by_component" in c:
            line += "  per generating component max %.5f (vs truth %.5f)" % (c["by_component"].max(), c["by_component_vs_truth"].max())
        if sampler == "dp":
            line += "  clusters>N/1000: %.2f vs batch-1 %.2f; in use: %.2f vs %.2f" % (c["k_big"] + c["k_used"])
        print(line)
        for side, cs in (("this ", chains), ("b = 1", doc["chains"])):
            for ch in cs:
                if ch["init"] == init:
                    print("      %s seed %d agreement %.4f clusters/component %s" % (
                        side, ch["seed"], ch["final_agreement"], ch["final_clusters_per_component"]))
    if out_dir:
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, "%s_batch%d.json" % (name, batch)), "w") as f:
            json.dump({"case": name, "batch": batch, "chains": chains}, f)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default="c2")
    ap.add_argument("--batch", default="default")
    ap.add_argument("--threads", type=int, default=6)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    oracle.build()
    for name in a.cases.split(","):
        run(name, a.batch, a.threads, a.out)
