#!/usr/bin/env python3
"""PCIe-inclusive rate of the drop-in entry points (host matrix in, S x N trace out), next to the
resident-chain rate bench.py reports.  Usage: python tools/run_api_timing.py [N P K nsamples]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bmm_mcmc_amd as bm
from bmm_mcmc_amd import synth

N, P, K, S = (int(v) for v in (sys.argv[1:5] + ["1000000", "50", "20", "220"][len(sys.argv) - 1:]))
X, _, _, _ = synth.host_matrix(N, P, min(K, 20), 22)
z0 = np.random.default_rng(0).integers(1, K + 1, N).astype(np.int32)
bm.gibbs_collapsed(X[:1000], 3, K, seed=1, initial_K=z0[:1000])  # load the library, warm the device
for burnin in (S - 20, 20):  # few kept sweeps / most sweeps kept
    t0 = time.perf_counter()
    out = bm.gibbs_collapsed(X, S, K, burnin=burnin, seed=1, initial_K=z0)
    dt = time.perf_counter() - t0
    kept = out["z"].shape[0]
    print("gibbs_collapsed N=%d P=%d K=%d: %d sweeps, %d kept (trace %.0f MB): %.3f s -> %.0f sweeps/s end to end"
          % (N, P, K, S, kept, kept * N * 4 / 1e6, dt, S / dt))
