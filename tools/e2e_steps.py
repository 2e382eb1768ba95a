#!/usr/bin/env python3
"""The steps of the drop-in call through the resident-chain API, timed one by one.  tools/e2e_steps.py [ns|c5]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bmm_mcmc_amd as bm
from bmm_mcmc_amd import synth
SHAPES = {"ns": (1_000_000, 50, 20, 22, 220), "c5": (10_000_000, 100, 20, 21, 30)}
bm.gibbs_collapsed(np.zeros((1000, 4), np.int32), 3, 2, seed=1)
for name in (sys.argv[1:] or ["ns"]):
    N, P, K, dseed, ns = SHAPES[name]
    X, _, _, _ = synth.host_matrix(N, P, K, dseed)
    z0 = np.random.default_rng(0).integers(1, K + 1, N).astype(np.int32)
    for rep in range(3):
        t = [time.perf_counter()]
        c = bm.Chain("collapsed", N, P, K, seed=1); t.append(time.perf_counter())
        c.set_data(X); t.append(time.perf_counter())
        c.set_initial_labels(z0); t.append(time.perf_counter())
        c.sweeps(ns - 1); t.append(time.perf_counter())
        c.sync(); t.append(time.perf_counter())
        c.close(); t.append(time.perf_counter())
        d = [1e3 * (b - a) for a, b in zip(t, t[1:])]
        print("%s rep %d: create %.2f, set_data %.2f, labels in %.2f, enqueue %.2f, sync %.2f, close %.2f ms" % ((name, rep) + tuple(d)))
