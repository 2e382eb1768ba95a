#!/bin/bash
# timing experiment with an alternative build of the library (bmm-mcmc_amd/lib/libbmmmcmc_hip_exp.so);
# chain results may be meaningless, only the timings are read
set -e
cd "$(dirname "$0")/.."
L=bmm-mcmc_amd/lib
cp $L/libbmmmcmc_hip.so /tmp/keep.so
trap 'cp /tmp/keep.so $L/libbmmmcmc_hip.so' EXIT
cp $L/libbmmmcmc_hip_exp.so $L/libbmmmcmc_hip.so
python bench.py --no-cpu --no-extra "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('exp lib:', round(d['value'],1), 'sweeps/s', round(d['ms_per_step'],4), 'ms/sweep, kernel', round(r['kernel_ms_per_sweep'],4))"
