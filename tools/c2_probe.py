#!/usr/bin/env python3
"""c2-like shapes under the library named by BMM_LIB_PATH: sweeps/s, kernel time and movers per sweep for a few
chain seeds (is a timing difference between two builds the arithmetic or the chain's trajectory?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bmm_mcmc_amd as bm
from bmm_mcmc_amd import synth
dev = torch.device("cuda", 0)
for wl in sys.argv[1:] or ["c2"]:
    sampler, K, K_true, N, P, dseed = synth.WORKLOADS[wl]
    X, _ = synth.device_matrix(N, P, K_true, dseed, dev)
    for seed in (1000, 1001, 1002):
        c = bm.Chain(sampler, N, P, K, seed=seed)
        c.set_data_device(X.data_ptr())
        c.set_initial_labels(np.random.default_rng(seed).integers(1, K + 1, N).astype(np.int32))
        c.sweeps(35); c.sync()
        z0 = c.labels()
        c.sweeps(1); c.sync()
        movers = int((c.labels() != z0).sum())
        c.profile(1)
        t0 = time.perf_counter(); c.sweeps(50); c.sync(); dt = time.perf_counter() - t0
        ms, n = c.profile_read()
        nk = c.counts()[0]
        print(wl, "seed", seed, "sweeps/s %.0f" % (50 / dt), "kernel ms/launch %.4f" % (ms / max(n, 1)), "movers/sweep", movers,
              "sizes", sorted(nk.tolist()), flush=True)
        c.close()
