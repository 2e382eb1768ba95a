#!/usr/bin/env python3
"""Condense a tools/prof.sh output directory into one text summary (for profiles/)."""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    name = name.split("(")[0]
    return name.replace("void ", "").replace("bmm::", "")[:48]


def main(d):
    out = []
    for f in glob.glob(os.path.join(d, "trace", "**", "*kernel_stats.csv"), recursive=True):
        out.append("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
        rows = list(csv.DictReader(open(f)))
        for r in rows[:12]:
            out.append("%-50s calls=%6s total_ms=%10.3f avg_us=%10.2f pct=%6s" % (
                short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3,
                r["Percentage"]))
    # the stats above average over the whole process, burn-in sweeps included; bench.py times only the
    # last `steps` sweeps, so the same split is made here from the per-dispatch trace
    timed = int(os.environ.get("PROF_TIMED_LAUNCHES", "0"))
    for f in glob.glob(os.path.join(d, "trace", "**", "*kernel_trace.csv"), recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "k_resample" in r["Kernel_Name"]]
        if timed and len(rows) >= timed:
            dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
            out.append("== k_resample by region of the run (kernel trace): all %d launches avg %.2f us; last %d "
                       "(bench.py's timed sweeps) avg %.2f us; the %d before them avg %.2f us ==" % (
                           len(dur), sum(dur) / len(dur), timed, sum(dur[-timed:]) / timed, len(dur) - timed,
                           sum(dur[:-timed]) / max(1, len(dur) - timed)))
    agg = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    for f in glob.glob(os.path.join(d, "pmc*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k][r["Counter_Name"]] += 1
    for k in sorted(agg, key=lambda k: -agg[k].get("SQ_WAVE_CYCLES", 0)):
        if "resample" not in k and agg[k].get("SQ_WAVE_CYCLES", 0) < 1e6:
            continue
        out.append("== PMC totals per dispatch (mean over dispatches): %s ==" % k)
        for c in sorted(agg[k]):
            out.append("  %-26s %16.1f  (n=%d)" % (c, agg[k][c] / cnt[k][c], cnt[k][c]))
    print("\n".join(out))


if __name__ == "__main__":
    main(sys.argv[1])
