#!/bin/bash
# same-box A/B over the workloads: the product library against an alternative build
# lib/libbmmmcmc_hip<suffix>.so (tools/ab_libs.sh _exp), loaded through BMM_LIB_PATH
ALT=${1:-_exp}
cd "$(dirname "$0")/.."
for w in c5 ns c2 c3 c4; do
  for lib in "" $ALT; do
    if [ -z "$lib" ]; then unset BMM_LIB_PATH; else export BMM_LIB_PATH=$(pwd)/bmm-mcmc_amd/lib/libbmmmcmc_hip$lib.so; fi
    timeout -k 10 150 python bench.py --workload $w --no-cpu --no-extra 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$w lib$lib', round(d['value'],1), 'sweeps/s kernel', round(r['kernel_ms_per_sweep'],4), 'lds', r.get('lds_bytes'), 'thr', r.get('threads'))" || exit 1
  done
done
unset BMM_LIB_PATH
timeout -k 10 150 python bench.py --workload c5 --x-layout int32 --no-cpu --no-extra 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('c5 int32 lib', round(d['value'],1), 'frac', round(r['frac'],3))"
BMM_LIB_PATH=$(pwd)/bmm-mcmc_amd/lib/libbmmmcmc_hip$ALT.so timeout -k 10 150 python bench.py --workload c5 --x-layout int32 --no-cpu --no-extra 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('c5 int32 lib$ALT', round(d['value'],1), 'frac', round(r['frac'],3))"
