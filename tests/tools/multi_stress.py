#!/usr/bin/env python3
"""Stress loop over the multi-chain paths on ONE GPU (the test variant's BMM_DEBUG_FAKE_DEVICES=2): bmm_multi_run with
four chains on two "devices", then three resident chains of which one runs alone beside two that share planes -- every
chain checked against its oracle chain, hundreds of times in one process.  Round 3 found two things with it that the
single-shot tests had passed over: a workgroup of a table-building launch that starts late read part of its own
launch's counts (one mismatch in twenty iterations; tests/test_gpu_layouts.py now holds a workgroup back on purpose),
and a process that creates and destroys four hardware-queue streams per call can hang inside the runtime's queue
creation (the streams are pooled since).  ITERS=250 WATCHDOG=110 timeout -k 5 130 python tests/tools/multi_stress.py full
prints one line per iteration, the mismatch counts of the three resident chains; the watchdog dumps the Python stack
and exits when the loop stops."""
import os, sys, faulthandler
faulthandler.dump_traceback_later(int(os.environ.get("WATCHDOG", "100")), exit=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["BMM_DEBUG_FAKE_DEVICES"] = "2"
import numpy as np
from bmm_mcmc_amd import _capi, build
_capi._LIB = _capi.load(os.environ.get("EXP_LIB") or build.LIB_DBG)
import bmm_mcmc_amd as bm
from bmm_mcmc_amd.synth import host_matrix as synth
from oracle import oracle
oracle.build()
X, _, _, _ = synth(5000, 24, 3, 6)
z0 = np.random.default_rng(10).integers(1, 4, 5000).astype(np.int32)
wants = {s: oracle.collapsed(X, z0, 5, 3, 0.0, 0.5, 0.5, 1, 1, 4, seed=s, batch=512)["z"][0] for s in (7, 8, 9)}
mode = sys.argv[1] if len(sys.argv) > 1 else "full"
for it in range(int(os.environ.get("ITERS", "12"))):
    if mode == "full":
        bm.gibbs_collapsed(X, 8, 3, burnin=2, seed=40, batch=700, chains=4, devices=[0, 1, 0, 1])
    a, b = bm.Chain("collapsed", 5000, 24, 3, seed=7, batch=512, device=0), bm.Chain("collapsed", 5000, 24, 3, seed=8, batch=512, device=1)
    third = bm.Chain("collapsed", 5000, 24, 3, seed=9, batch=512, device=1)
    a.set_data(X)
    bm.broadcast_planes([a, b])
    third.share_data(b)
    for ch in (a, b, third):
        ch.set_initial_labels(z0)
    bm.sweep_chains([a, b, third], 4)
    print(it, [int((ch.labels() != wants[s]).sum()) for ch, s in ((a, 7), (b, 8), (third, 9))], flush=True)
    for ch in (third, b, a):
        ch.close()
