#!/usr/bin/env python3
"""Where the drop-in call's time goes at the north-star shape (host matrix in, trace out): the same steps
through the resident-chain API, timed one by one, then gibbs_collapsed itself.  tools/e2e_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bmm_mcmc_amd as bm
from bmm_mcmc_amd import synth
N, P, K = 1_000_000, 50, 20
X, _, _, _ = synth.host_matrix(N, P, 20, 22)
z0 = np.random.default_rng(0).integers(1, K + 1, N).astype(np.int32)
bm.gibbs_collapsed(X[:1000], 3, K, seed=1, initial_K=z0[:1000])
for rep in range(2):
    t0 = time.perf_counter(); c = bm.Chain("collapsed", N, P, K, seed=1); t1 = time.perf_counter()
    c.set_data(X); t2 = time.perf_counter()
    c.set_initial_labels(z0); t3 = time.perf_counter()
    c.sweeps(220); c.sync(); t4 = time.perf_counter()
    z = c.labels(); t5 = time.perf_counter(); c.close()
    print("create %.1f ms, set_data(host 200 MB) %.1f ms, labels in %.1f ms, 220 sweeps %.1f ms, labels out %.1f ms" % tuple(1e3 * v for v in (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)))
    t0 = time.perf_counter(); out = bm.gibbs_collapsed(X, 220, K, burnin=200, seed=1, initial_K=z0); print("gibbs_collapsed total %.1f ms" % (1e3 * (time.perf_counter() - t0)))
