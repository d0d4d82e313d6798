#!/usr/bin/env python
"""Condenses rocprofv3 outputs (gpurun_out/prof_<tag>_{kt,fetch,write,sq}) into profiles/:
  <tag>_kernel_stats.csv   -- rocprofv3 --kernel-trace --stats summary, verbatim
  <tag>_pmc_summary.json   -- per-kernel means of the PMC counters, with the gfx950 FETCH_SIZE x2
                              correction (MI355X_MICROARCH.md, HBM section) applied to `hbm_read_bytes`
Usage: python tools/summarize_profiles.py r01 gpurun_out/prof_r1
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag, prefix = sys.argv[1], sys.argv[2]
os.makedirs("profiles", exist_ok=True)
ks = glob.glob(prefix + "_kt/*/*_kernel_stats.csv")
if ks:
    shutil.copy(ks[0], f"profiles/{tag}_kernel_stats.csv")
summary = collections.defaultdict(dict)
for sub in ("fetch", "write", "sq"):
    for f in glob.glob(f"{prefix}_{sub}/*/*_counter_collection.csv"):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
            agg[name]["_dur_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        for k, v in agg.items():
            for c, x in v.items():
                if c == "_dur_ns":
                    summary[k].setdefault("dispatches_" + sub, len(x))
                    summary[k]["mean_duration_us_" + sub] = sum(x) / len(x) / 1e3
                else:
                    summary[k][c] = sum(x) / len(x)
for k, v in summary.items():
    if "FETCH_SIZE" in v:   # KB, reports 1/2 of wide coalesced reads on gfx950
        v["hbm_read_bytes_per_launch"] = v["FETCH_SIZE"] * 1024 * 2
    if "WRITE_SIZE" in v:
        v["hbm_write_bytes_per_launch"] = v["WRITE_SIZE"] * 1024
    if "hbm_read_bytes_per_launch" in v and "hbm_write_bytes_per_launch" in v:
        v["hbm_traffic_bytes_per_launch"] = v["hbm_read_bytes_per_launch"] + v["hbm_write_bytes_per_launch"]
json.dump(summary, open(f"profiles/{tag}_pmc_summary.json", "w"), indent=1, sort_keys=True)
for k, v in summary.items():
    print(k[:60], {c: round(x, 1) for c, x in v.items() if "bytes" in c or "duration" in c})
