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


SQ_COUNTERS = ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
               "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "GRBM_GUI_ACTIVE")


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
    sq = {c: load_counter(os.path.join(root, "pmc_sq"), c) for c in SQ_COUNTERS}
    l2 = {c: load_counter(os.path.join(root, "pmc_l2"), c) for c in ("TCC_REQ_sum", "TCC_HIT_sum", "TCC_MISS_sum")}
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
        # L2 pass (own run).  TCC_REQ_sum counts the 128-byte requests the CUs send to the XCDs' L2s: for an LDS-DMA-fed
        # kernel that is its L2 -> LDS staging traffic plus the epilogue's loads and stores (checked on the 256x256
        # implicit-GEMM tile: 1 444 tiles x 1 180 KB staged + 378 MB of epilogue = 2.08 GB against 16.2 M requests).
        r = l2["TCC_REQ_sum"].get(k)
        if r and r[1]:
            e["l2_request_bytes_per_launch"] = 128.0 * r[0] / r[1]
            e["l2_request_tbps"] = e["l2_request_bytes_per_launch"] / (s["avg_us"] * 1e-6) / 1e12
            h, m = l2["TCC_HIT_sum"].get(k), l2["TCC_MISS_sum"].get(k)
            if h and m and h[0] + m[0] > 0:
                e["l2_hit_pct"] = 100.0 * h[0] / (h[0] + m[0])
        # SQ pass (own run): MFMA-pipe busy share and where wave time goes.  GRBM_GUI_ACTIVE is summed over the
        # 8 XCDs (MI355X_MICROARCH.md, DVFS give-back), SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs;
        # SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_ANY are in quad-cycles and disjoint shares of wave time.
        g = sq["GRBM_GUI_ACTIVE"].get(k)
        if g and g[1]:
            cyc = g[0] / g[1] / 8.0
            e["gpu_cycles_per_launch"] = cyc
            m = sq["SQ_VALU_MFMA_BUSY_CYCLES"].get(k)
            if m and m[1] and cyc > 0:
                e["mfma_busy_pct"] = 100.0 * (m[0] / m[1]) / (cyc * 1024.0)
            w = sq["SQ_WAVE_CYCLES"].get(k)
            if w and w[0] > 0:
                for name, key in (("SQ_WAIT_ANY", "wave_wait_any_pct"), ("SQ_WAIT_INST_ANY", "wave_wait_inst_pct"),
                                  ("SQ_ACTIVE_INST_ANY", "wave_active_inst_pct")):
                    c = sq[name].get(k)
                    if c:
                        e[key] = 100.0 * c[0] / w[0]
            a, b = sq["SQ_LDS_IDX_ACTIVE"].get(k), sq["SQ_LDS_BANK_CONFLICT"].get(k)
            if a and b and a[0] > 0:
                e["lds_bank_conflict_pct"] = 100.0 * b[0] / a[0]
        out[k] = e
    json.dump({"note": "FETCH_SIZE x2 (gfx950 wide-read correction), KiB -> bytes; per launch = sum / dispatches; mfma_busy_pct = "
                       "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs); wave_* = share of SQ_WAVE_CYCLES",
               "kernels": out}, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
