#!/bin/bash
# Do bench.py's own kernel timings (start/stop events attached to the launches) agree with rocprofv3's
# kernel trace of the same run?  tools/events_vs_trace.sh [bench args]; prints both per-launch averages.
set -o pipefail
REPO=$(dirname $(dirname $(readlink -f $0)))
OUT=$REPO/gpurun_out/evt_$$
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o trace -- python3 $REPO/bench.py --no-cpu --no-extra "$@" > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, json, sys
out = sys.argv[1]
d = json.loads(open(out + "/bench.json").read().strip().splitlines()[-1])
r = d["roofline"]
lps, steps = r["launches_per_sweep"], d["steps"]
f = glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True)[0]
dur = [(int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) / 1e3 for x in csv.DictReader(open(f)) if "k_resample" in x["Kernel_Name"]]
timed = dur[-lps * steps:]
print("%s: bench.py events %.2f us per launch, rocprofv3 kernel trace %.2f us over the same %d launches (ratio %.3f)"
      % (d["config"]["workload"][:40], 1e3 * r["kernel_ms_per_sweep"] / lps, sum(timed) / len(timed), len(timed),
         1e3 * r["kernel_ms_per_sweep"] / lps / (sum(timed) / len(timed))))
PY
