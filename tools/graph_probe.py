#!/usr/bin/env python3
"""What would a hipGraph of a sweep buy?  Two consecutive sweeps of a resident chain (labels ping-pong
between two rows) are captured from the chain's stream and replayed; the replay re-uses the two sweep
indices (so the same uniforms every second sweep: a timing experiment, not a chain to keep) but every
launch sees consistent labels and statistics.  Compared with the same number of sweeps enqueued launch
by launch.  tools/graph_probe.py [shape ...]   shapes: ns c2 c3 c4 c5"""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bmm_mcmc_amd as bm
from bmm_mcmc_amd import synth

hip = ctypes.CDLL("libamdhip64.so")
dev = torch.device("cuda", 0)


def chk(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed: hip error %d" % (what, rc))


def probe(workload, reps=100):
    sampler, K, K_true, N, P, dseed = synth.WORKLOADS[workload]
    X, _ = synth.device_matrix(N, P, K_true, dseed, dev)
    c = bm.Chain(sampler, N, P, K, seed=1000)
    c.set_data_device(X.data_ptr())
    rng = np.random.default_rng(1)
    if sampler == "collapsed":
        c.set_initial_labels(rng.integers(1, K + 1, N).astype(np.int32))
    elif sampler in ("stickbreaking", "full"):
        pi0 = np.exp(rng.random(K))
        c.set_initial_params(pi0 / pi0.sum(), rng.random((K, P)))
    c.sweeps(30)
    c.sync()
    t0 = time.perf_counter()
    c.sweeps(2 * reps)
    c.sync()
    plain = 2 * reps / (time.perf_counter() - t0)

    stream = ctypes.c_void_p(c.stream())
    graph, gexec = ctypes.c_void_p(), ctypes.c_void_p()
    chk(hip.hipStreamBeginCapture(stream, 2), "begin capture")  # hipStreamCaptureModeRelaxed
    c.sweeps(2)
    chk(hip.hipStreamEndCapture(stream, ctypes.byref(graph)), "end capture")
    chk(hip.hipGraphInstantiate(ctypes.byref(gexec), graph, None, None, ctypes.c_size_t(0)), "instantiate")
    for _ in range(5):
        chk(hip.hipGraphLaunch(gexec, stream), "graph launch")
    c.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        chk(hip.hipGraphLaunch(gexec, stream), "graph launch")
    c.sync()
    graphed = 2 * reps / (time.perf_counter() - t0)
    nk = c.counts()[0]
    assert int(nk.sum()) == N and int(nk.min()) >= 0, "statistics inconsistent after the replay"
    hip.hipGraphExecDestroy(gexec)
    hip.hipGraphDestroy(graph)
    c.close()
    print("%s: launch by launch %.0f sweeps/s, graph of two sweeps replayed %.0f sweeps/s (%+.1f %%)"
          % (workload, plain, graphed, 100 * (graphed / plain - 1)), flush=True)


for w in (sys.argv[1:] or ["ns", "c2", "c3", "c4", "c5"]):
    probe(w)
