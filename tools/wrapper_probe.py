import sys, time, cProfile, pstats, io
sys.path.insert(0, '/root/repo')
import numpy as np
import bmm_mcmc_amd as bm
from bmm_mcmc_amd import synth, _capi
import ctypes as C
X, _, _, _ = synth.host_matrix(1_000_000, 50, 20, 22)
z0 = np.random.default_rng(0).integers(1, 21, 1_000_000).astype(np.int32)
bm.gibbs_collapsed(X[:1000], 3, 20, seed=1, initial_K=z0[:1000])
for rep in range(3):
    t0 = time.perf_counter(); out = bm.gibbs_collapsed(X, 220, 20, burnin=200, seed=1, initial_K=z0); t1 = time.perf_counter()
    ms = (C.c_double * 6)(); _capi.lib().bmm_last_run_phases(ms)
    print("total %.2f ms, phases sum %.2f" % (1e3 * (t1 - t0), sum(ms)), [round(v, 2) for v in ms])
    del out
pr = cProfile.Profile(); pr.enable(); out = bm.gibbs_collapsed(X, 220, 20, burnin=200, seed=1, initial_K=z0); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(12); print(s.getvalue())
