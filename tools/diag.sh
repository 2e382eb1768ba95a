#!/bin/bash
# Diagnostic-build run on the GPU box: swaps in the -DBMM_DIAG library for one bench run
# and prints the per-phase cycle shares each chain reports on destroy.  Never timed.
set -e
L=bmm-mcmc_amd/lib
cp $L/libbmmmcmc_hip.so /tmp/keep.so
cp $L/libbmmmcmc_hip_diag.so $L/libbmmmcmc_hip.so
python bench.py --no-cpu "$@" 2>&1 >/dev/null | grep "bmm diag" || true
cp /tmp/keep.so $L/libbmmmcmc_hip.so
