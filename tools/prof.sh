#!/bin/bash
# rocprofv3 passes over bench.py on the GPU box: kernel trace + stats, then PMC counters in
# separate runs (never combined with other trace domains).  Usage: tools/prof.sh <tag> [bench args]
# Output under gpurun_out/prof_<tag>/; tools/prof_summary.py condenses it for profiles/.
set -o pipefail
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
[ -z "$GRAFT_REPO_ROOT" ] && OUT=$(pwd)/gpurun_out/prof_$TAG
REPO=$(dirname $(dirname $(readlink -f $0)))
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { # name, rocprof args...
  local name=$1; shift
  timeout -k 10 240 rocprofv3 "$@" --output-format csv -d $OUT/$name -o $name -- python3 $REPO/bench.py --no-cpu $BENCH_ARGS > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $OUT/$name.log; return 1; }
}
BENCH_ARGS="$*"
run trace --kernel-trace --stats &&
run pmc1 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS &&
run pmc2 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU &&
run pmc3 --kernel-trace --pmc FETCH_SIZE &&
run pmc4 --kernel-trace --pmc WRITE_SIZE GRBM_GUI_ACTIVE
find $OUT -name "*.csv" | head -30
