#!/usr/bin/env python3
"""Where the drop-in call's time goes (host matrix in, S x N trace out): bmm_collapsed_run through ctypes with the
library's own phase clock (bmm_last_run_phases), at the north-star shape and at C5.  tools/e2e_probe.py [ns|c5 ...]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bmm_mcmc_amd as bm
from bmm_mcmc_amd import _capi, synth

SHAPES = {"ns": (1_000_000, 50, 20, 22, 220, 200), "c5": (10_000_000, 100, 20, 21, 30, 3),
          "c2": (100_000, 20, 3, 18, 1000, 100)}
PH = ("pack left+upload", "create+start state", "enqueue", "device wait", "trace out", "release")


def probe(name, reps=int(os.environ.get("REPS", "6"))):
    N, P, K, dseed, ns, burn = SHAPES[name]
    X, _, _, _ = synth.host_matrix(N, P, K, dseed)
    z0 = np.random.default_rng(0).integers(1, K + 1, N).astype(np.int32)
    S = ns - burn
    L = _capi.lib()
    print("%s: N=%d P=%d K=%d nsamples=%d kept=%d, host threads %d" % (name, N, P, K, ns, S, L.bmm_host_threads()))
    keep = []
    for rep in range(reps):
        if os.environ.get("KEEP"):
            keep.append((locals().get("z"), locals().get("th"), locals().get("al"), locals().get("out")))
        z = np.empty((S, N), dtype=np.int32, order="F")   # as R's allocMatrix: untouched pages
        th = np.empty((K, P, S), order="F")
        al = np.empty((S, 1), order="F")
        t0 = time.perf_counter()
        rc = L.bmm_collapsed_run(_capi.vp(X), C.c_int64(N), C.c_int(P), _capi.vp(z0), C.c_int(ns), C.c_int(K),
                                 C.c_double(0), C.c_double(.5), C.c_double(.5), C.c_double(1), C.c_double(1),
                                 C.c_int(burn), C.c_int64(0), C.c_uint64(1), C.c_int(0), _capi.vp(z), _capi.vp(th), _capi.vp(al))
        dt = time.perf_counter() - t0
        _capi.check(rc)
        ms = (C.c_double * 6)()
        L.bmm_last_run_phases(ms)
        print("  rep %d: %.1f ms = %.0f sweeps/s | " % (rep, 1e3 * dt, ns / dt) + ", ".join("%s %.1f" % (n, v) for n, v in zip(PH, ms)))
        if os.environ.get("NO_WRAPPER"):
            continue
        out = None   # the previous result is released outside the clock
        t0 = time.perf_counter()
        out = bm.gibbs_collapsed(X, ns, K, burnin=burn, seed=1, initial_K=z0)
        print("         through the Python wrapper: %.1f ms" % (1e3 * (time.perf_counter() - t0)))
        assert np.array_equal(out["z"], z)


if __name__ == "__main__":
    bm.gibbs_collapsed(np.zeros((1000, 4), np.int32), 3, 2, seed=1)   # context, first-use costs
    for n in (sys.argv[1:] or ["ns"]):
        probe(n)
