#!/usr/bin/env python3
"""Do chains that share a GPU really overlap?  Aggregate sweeps/s of 4 (and 8) resident chains of the
north-star shape, fresh and after a sequence of other chains has come and gone in the same process
(what bench.py's other_workloads does), plus the cost of creating a chain.  tools/chains_seq_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bmm_mcmc_amd as bm
from bmm_mcmc_amd import synth

dev = torch.device("cuda", 0)


def run(nch, sampler="collapsed", K=20, N=1_000_000, P=50):
    X, _ = synth.device_matrix(N, P, min(K, 20), 22, dev)
    t0 = time.perf_counter()
    chains = [bm.Chain(sampler, N, P, K, seed=1000 + c) for c in range(nch)]
    t_create = (time.perf_counter() - t0) / nch
    chains[0].set_data_device(X.data_ptr())
    for c in chains[1:]:
        c.share_data(chains[0])
    for i, c in enumerate(chains):
        if sampler == "collapsed":
            c.set_initial_labels(np.random.default_rng(i).integers(1, K + 1, N).astype(np.int32))
    bm.sweep_chains(chains, 35)
    for c in chains:
        c.sync()
    t0 = time.perf_counter()
    bm.sweep_chains(chains, 50)
    for c in chains:
        c.sync()
    dt = time.perf_counter() - t0
    for c in chains:
        c.close()
    del X
    torch.cuda.empty_cache()
    return round(nch * 50 / dt), round(1e3 * t_create, 2)


print("fresh 4 chains (sweeps/s, ms per chain creation):", run(4))
for shape in ((20, 10_000_000, 100), (3, 100_000, 20), (20, 1_000_000, 50)):
    run(1, "collapsed", *shape)
run(1, "dp", 30, 1_000_000, 50)
a = torch.empty(1 << 26, device=dev); b = torch.empty_like(a); b.copy_(a); torch.cuda.synchronize(); del a, b
print("after single chains of four shapes and a torch copy, 4 chains:", run(4))
print("again 4 chains:", run(4))
print("1 chain:", run(1))
print("8 chains:", run(8))
