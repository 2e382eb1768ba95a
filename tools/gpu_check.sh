#!/bin/bash
# On the GPU box: parity tests, then a short bench of every workload (one line each).
# Usage: tools/gpu_check.sh [extra bench args]
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q --timeout 300 > gpurun_out/gpu_tests.log 2>&1; rc=$?
tail -3 gpurun_out/gpu_tests.log
[ $rc -le 1 ] || exit $rc
show() { python3 - "$1" "$2" <<'PY'
import json, sys
r = json.load(open(sys.argv[2])); f = r["roofline"]
print("%-8s %8.3f ms/sweep  kern %7.3f ms  bound %s frac %s  hbm %.0f GB/s  thr %d lds %d" % (
    sys.argv[1], r["ms_per_step"], f["kernel_ms_per_sweep"], f["bound"], f.get("frac"), f.get("hbm_GBps", f.get("achieved") or 0),
    f["threads"], f["lds_bytes"]))
PY
}
for w in c5 ns c3 c4 c2; do
  timeout -k 10 300 python bench.py --workload $w --steps 10 --warmup 30 --no-cpu "$@" > gpurun_out/bench_$w.json 2> gpurun_out/bench_$w.err || { tail -5 gpurun_out/bench_$w.err; exit 1; }
  show $w gpurun_out/bench_$w.json
done
timeout -k 10 300 python bench.py --workload c5 --batch 10000000 --steps 10 --warmup 30 --no-cpu > gpurun_out/bench_c5N.json 2> gpurun_out/bench_c5N.err && show "c5 B=N" gpurun_out/bench_c5N.json
exit $rc
