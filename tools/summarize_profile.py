#!/usr/bin/env python
"""Condense a tools/gpu_profile.sh output directory into one JSON: per kernel, launches, average
duration (kernel-trace) and HBM bytes per launch from the PMC passes, corrected as
MI355X_MICROARCH.md (HBM section) prescribes for gfx950: FETCH_SIZE and WRITE_SIZE are in KiB;
FETCH_SIZE reports half the bytes of wide coalesced streaming reads, so it is doubled;
WRITE_SIZE is exact for 16-byte-per-lane streaming stores."""
import collections
import csv
import glob
import json
import os
import sys


def short(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    return name.split("(")[0]


def load_counter(d, counter):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                a = acc[short(r["Kernel_Name"])]
                a[0] += float(r["Counter_Value"])
                a[1] += 1
    return acc


def main():
    root = sys.argv[1]
    stats = {}
    for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            stats[short(r["Name"])] = dict(calls=int(r["Calls"]), avg_us=float(r["AverageNs"]) / 1e3,
                                           total_ms=float(r["TotalDurationNs"]) / 1e6, pct=float(r["Percentage"]))
    fetch = load_counter(os.path.join(root, "pmc_fetch"), "FETCH_SIZE")
    write = load_counter(os.path.join(root, "pmc_write"), "WRITE_SIZE")
    out = {}
    for k, s in sorted(stats.items(), key=lambda kv: -kv[1]["total_ms"]):
        if s["pct"] < 0.05:
            continue
        e = dict(s)
        if k in fetch and fetch[k][1]:
            e["fetch_bytes_per_launch_corrected"] = 2.0 * fetch[k][0] * 1024 / fetch[k][1]
        if k in write and write[k][1]:
            e["write_bytes_per_launch"] = write[k][0] * 1024 / write[k][1]
        if "fetch_bytes_per_launch_corrected" in e and "write_bytes_per_launch" in e:
            e["hbm_bytes_per_launch"] = e["fetch_bytes_per_launch_corrected"] + e["write_bytes_per_launch"]
        out[k] = e
    json.dump({"note": "FETCH_SIZE x2 (gfx950 wide-read correction), KiB -> bytes; per launch = sum / dispatches",
               "kernels": out}, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
