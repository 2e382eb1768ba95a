#!/usr/bin/env python3
"""Per-launch constants of k_resample from a tools/prof.sh output directory, over the launches of bench.py's
timed sweeps only (the last `timed` dispatches): what profiles/traffic.json holds and DESIGN.md quotes.
tools/prof_constants.py <dir> <timed launches> <observations per sweep> <launches per sweep>"""
import collections
import csv
import glob
import json
import sys


def timed_mean(d, counter, timed, kern="k_resample"):
    for f in glob.glob(d + "/pmc*/**/*counter_collection.csv", recursive=True):
        per = collections.OrderedDict()
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"] and r["Counter_Name"] == counter:
                per[r["Dispatch_Id"]] = per.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
        if per:
            v = list(per.values())[-timed:]
            return sum(v) / len(v)
    return None


def main(d, timed, N, lps):
    obs = N / lps
    c = {k: timed_mean(d, k, timed) for k in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "SQ_INSTS_LDS",
                                              "SQ_ACTIVE_INST_VALU", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT",
                                              "SQ_WAIT_INST_LDS", "SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE")}
    dur = []
    for f in glob.glob(d + "/trace/**/*kernel_trace.csv", recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "k_resample" in r["Kernel_Name"]]
        dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows][-timed:]
    us = sum(dur) / len(dur)
    cyc = us * 2400.0
    print(json.dumps({
        "launch_us": us,
        "hbm_bytes_per_launch": (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024,   # gfx950 FETCH_SIZE correction
        "algorithmic_bytes_per_launch_bits": obs * 24, "algorithmic_bytes_per_launch_int32": obs * 408,
        "valu_inst_per_64_obs": c["SQ_INSTS_VALU"] / (obs / 64), "lds_inst_per_64_obs": c["SQ_INSTS_LDS"] / (obs / 64),
        "valu_busy_frac": c["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / cyc,     # quad-cycles over 1024 SIMDs
        "lds_busy_frac": c["SQ_LDS_IDX_ACTIVE"] / 256 / cyc,             # LDS-array cycles over 256 CUs
        "lds_conflict_frac": c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"],
        "wait_inst_lds_frac_of_wave_cycles": c["SQ_WAIT_INST_LDS"] / c["SQ_WAVE_CYCLES"]}, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), float(sys.argv[3]), float(sys.argv[4]))
