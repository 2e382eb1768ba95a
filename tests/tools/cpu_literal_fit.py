#!/usr/bin/env python3
"""SURVEY.md section 8(d) baseline (i): the reference algorithm as written (per-cluster member
lists, sums recomputed for every (i,k,d); oracle `*_literal`), one thread, timed at small N on this
host, with a c*N^2*P fit and its extrapolation to the benchmark shapes -- reported as an
extrapolation, never as a measurement.  Also times baseline (ii), the sufficient-statistics port,
on one thread.  Writes JSON to stdout."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle  # noqa: E402

sys.path.insert(0, os.path.join(ROOT, "bmm-mcmc_amd"))
import synth  # noqa: E402

K, P = 20, 50
rows = []
for N in (1000, 2000, 4000, 8000):
    X, _, _, _ = synth.host_matrix(N, P, K, 22)
    z0 = np.random.default_rng(1).integers(1, K + 1, N).astype(np.int32)
    sweeps = 3 if N <= 4000 else 2
    t = time.perf_counter()
    oracle.collapsed(X, z0, sweeps + 1, K, 1.0, 0.5, 0.5, 1, 1, sweeps, seed=1, literal=True)
    dt = (time.perf_counter() - t) / sweeps
    rows.append({"N": N, "s_per_sweep": dt, "c": dt / (N * N * P)})
c = float(np.median([r["c"] for r in rows]))
out = {"literal": {"K": K, "P": P, "threads": 1, "runs": rows, "c_seconds_per_N2P": c,
                   "extrapolated_s_per_sweep": {"ns K=20 N=1e6 P=50": c * 1e12 * 50, "c5 K=20 N=1e7 P=100": c * 1e14 * 100}}}
X, _, _, _ = synth.host_matrix(200000, P, K, 22)
t1 = oracle.time_sweeps("collapsed", X, K, 5, 25000, 1, 1)
out["suffstat_one_thread"] = {"K": K, "P": P, "N": 200000, "s_per_sweep": t1 / 5,
                              "allocations_per_s": 200000 * 5 / t1}
out["host"] = {"nproc": os.cpu_count()}
print(json.dumps(out, indent=1))
