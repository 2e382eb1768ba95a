#!/bin/bash
# timing experiment with an alternative build of the library (default lib/libbmmmcmc_hip_exp.so, or
# EXP_LIB=<path>), loaded through BMM_LIB_PATH: the product library is never overwritten.  Chain results
# may be meaningless, only the timings are read.
set -e
cd "$(dirname "$0")/.."
E=${EXP_LIB:-$(pwd)/bmm-mcmc_amd/lib/libbmmmcmc_hip_exp.so}
[ -f "$E" ] || { echo "$E not built" >&2; exit 1; }
BMM_LIB_PATH=$E python bench.py --no-cpu --no-extra "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('exp lib:', round(d['value'],1), 'sweeps/s', round(d['ms_per_step'],4), 'ms/sweep, kernel', round(r['kernel_ms_per_sweep'],4))"
