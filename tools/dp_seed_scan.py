#!/usr/bin/env python3
"""How often does a DP chain at the default batch hold exactly the generating number of clusters?
(A cluster split in the first sweeps never merges again under incremental Gibbs at N = 4e5: two
sub-clusters with the same theta trade observations in proportion to their sizes, a martingale.)
Usage: python tools/dp_seed_scan.py [first_seed n_seeds]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bmm_mcmc_amd as bm
from bmm_mcmc_amd import synth

s0, n = (int(v) for v in (sys.argv[1:3] + ["11", "12"][len(sys.argv) - 1:]))
dev = torch.device("cuda", 0)
K, K_true, N, P = 30, 5, 400_000, 50
X, truth = synth.device_matrix(N, P, K_true, 77, dev)
w, theta = synth.truth(K_true, P, 77)
for seed in range(s0, s0 + n):
    for batch in (None, 1 << 30):
        ch = bm.Chain("dp", N, P, K, seed=seed, batch=batch)
        ch.set_data_device(X.data_ptr(), keepalive=X)
        ch.sweeps(150)
        counts = ch.sweeps_counts(60)
        ch.close()
        props = np.sort(counts / N, axis=1)[:, ::-1].mean(axis=0)
        big = int((counts[-1] > N / 1000).sum())
        print("seed %d batch %s: clusters > N/1000: %d  props %s" % (seed, "default" if batch is None else "N", big, np.round(props[:7], 3)))
