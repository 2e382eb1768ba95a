#!/bin/bash
# workgroup-size experiment on a workload: tools/try_threads.sh c5 1024 768 512
# (BMM_DEBUG_THREADS is read by the -DBMM_DEBUG_HOOKS test variant of the library only, loaded here
# through BMM_LIB_PATH)
cd "$(dirname "$0")/.."
export BMM_LIB_PATH=$(pwd)/bmm-mcmc_amd/lib/libbmmmcmc_hip_dbg.so
w=$1; shift
for nt in default "$@"; do
  if [ $nt = default ]; then unset BMM_DEBUG_THREADS; else export BMM_DEBUG_THREADS=$nt; fi
  timeout -k 10 150 python bench.py --workload $w --no-cpu --no-extra 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$w threads','$nt','->',r['threads'],'ms/sweep',round(d['ms_per_step'],4),'kernel ms',round(r['kernel_ms_per_sweep'],4),'batch',d['config']['batch'])" || exit 1
done
