#!/bin/bash
# Diagnostic-build run: `tools/diag.sh build` (in the build container) compiles the -DBMM_DIAG
# library next to the product one (lib/libbmmmcmc_hip_diag.so); `tools/diag.sh [bench.py flags]` (on
# the GPU box) loads it for one bench run through BMM_LIB_PATH -- the product library is never
# touched -- and prints the per-phase tick shares each chain reports on destroy.  Never timed, never shipped.
set -e
cd "$(dirname "$0")/.."
L=$(pwd)/bmm-mcmc_amd/lib
if [ "$1" = build ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -amdgpu-sched-strategy=iterative-ilp -fPIC -shared -DBMM_DIAG \
    -Wl,-rpath,/opt/rocm/lib -o $L/libbmmmcmc_hip_diag.so bmm-mcmc_amd/csrc/chain.hip
  exit 0
fi
[ -f $L/libbmmmcmc_hip_diag.so ] || { echo "run tools/diag.sh build first" >&2; exit 1; }
BMM_LIB_PATH=$L/libbmmmcmc_hip_diag.so python bench.py --no-cpu --no-extra "$@" 2>&1 >/dev/null | grep "bmm diag" || true
