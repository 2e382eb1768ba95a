#!/bin/bash
# same-box timing of several alternative builds lib/libbmmmcmc_hip_<name>.so against each other over some workloads:
# tools/ab_many.sh "c5 ns c3" exp_base exp_NOFLUSH ...   (timing experiments: results of a non-product build may be wrong)
cd "$(dirname "$0")/.."
WL=$1; shift
for w in $WL; do
  for lib in "$@"; do
    export BMM_LIB_PATH=$(pwd)/bmm-mcmc_amd/lib/libbmmmcmc_hip_$lib.so
    timeout -k 10 150 python bench.py --workload $w --no-cpu --no-extra 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$w $lib', round(d['value'],1), 'sweeps/s  ms/sweep', round(d['ms_per_step'],4), 'kernel', round(r['kernel_ms_per_sweep'],4))" || exit 1
  done
done
