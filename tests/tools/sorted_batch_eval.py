#!/usr/bin/env python3
"""The default batch on data SORTED by generating cluster -- the worst case for batches (the sets the reference bundles
are sorted; SURVEY.md section 8(d) asks for sorted data as a stress variant): the oracle at N/8 and N/4 against the
oracle at batch 1 (the sequential scan) from the generating allocation, at the smallest N that gets the N/4 default
(2^16, K = 20), at C2's shape and at N = 20 000.  CPU only, about a minute: python tests/tools/sorted_batch_eval.py
(output of the round-3 run: profiles/r03/sorted_batch_eval.log)."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from bmm_mcmc_amd import synth
from oracle import oracle
from tolerance_cases import summarise
from concurrent.futures import ThreadPoolExecutor
oracle.build()
def run(N,K,P,dseed,sorted_,init='truth'):
    X, labels, theta, w = synth.host_matrix(N,P,K,dseed)
    if sorted_:
        o=np.argsort(labels,kind='stable'); X=np.asfortranarray(X[o]); labels=labels[o]
    res={}
    def one(args):
        b,seed=args
        z0=(labels+1).astype(np.int32) if init=='truth' else np.random.default_rng(seed).integers(1,K+1,N).astype(np.int32)
        r=oracle.counts_summary('collapsed',X,z0,140,K,0.0,0.5,0.5,1.0,1.0,40,seed=seed,batch=b)
        return b,seed,summarise('collapsed',r,N,K,K,labels)
    jobs=[(b,s) for b in (1,N//8,N//4) for s in (1000,1001)]
    with ThreadPoolExecutor(6) as ex: out=list(ex.map(one,jobs))
    ref={s:o for b,s,o in out if b==1}
    for b,s,o in out:
        if b==1: continue
        dp=np.abs(np.array(o['props_by_component'])-np.array(ref[s]['props_by_component'])).max()
        dt=np.nanmax(np.abs(np.array(o['theta_by_size'])-np.array(ref[s]['theta_by_size'])))
        print(f"N={N} K={K} sorted={sorted_} init={init} batch=N/{N//b} seed={s}: props {dp:.5f} theta {dt:.5f} agreement {o['final_agreement']:.4f} vs {ref[s]['final_agreement']:.4f}", flush=True)
for N,K,P,ds in [(65536,20,50,26),(100000,3,20,18),(20000,4,12,5)]:
    run(N,K,P,ds,True)
