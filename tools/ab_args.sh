#!/bin/bash
# same-box timing of alternative builds over one bench.py configuration: tools/ab_args.sh "<bench args>" exp prod ...
cd "$(dirname "$0")/.."
A=$1; shift
for lib in "$@"; do
  export BMM_LIB_PATH=$(pwd)/bmm-mcmc_amd/lib/libbmmmcmc_hip_$lib.so
  timeout -k 10 150 python bench.py $A --no-cpu --no-extra 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$A | $lib', round(d['value'],1), 'sweeps/s  ms/sweep', round(d['ms_per_step'],4), 'kernel', round(r['kernel_ms_per_sweep'],4), 'lds', r['lds_bytes'])" || exit 1
done
