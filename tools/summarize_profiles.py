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
# per-kernel durations of the one-stage launches from the kernel trace (the stats file averages all launches)
for f in glob.glob(prefix + "_kt/*/*_kernel_trace.csv"):
    rows = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "slab_stage_" in r["Kernel_Name"] or "symm_" in r["Kernel_Name"]:
            rows[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    out = {}
    for k, v in rows.items():
        full = [x for x in v if x >= 52000]
        small = [x for x in v if x < 20000]
        half = [x for x in v if 20000 <= x < 52000]
        if "symm_sweep_kernel" in k:     # whole-triangle sweeps (one-stage iterations) against half sweeps (two-stage ones)
            full = [x for x in v if x >= 38000]
            half = [x for x in v if 10000 <= x < 38000]
            small = [x for x in v if x < 10000]
        mean = lambda xs: (sum(xs) / len(xs) / 1e3) if xs else None
        # the trace covers whole rotations of the job, so the mean over ALL launches of the plain instance is the mean over
        # the job's mix of sixteen-, two- and one-stage launches: what bench.py's by_kind.stage_kernel.avg_launch_us prices
        job_mix = None
        if small and half and k.rstrip(">").endswith("false, false"):
            job_mix = sum(v) / len(v) / 1e3
        out[k] = dict(launches=len(v), mean_us=sum(v) / len(v) / 1e3, sixteen_stage_launches=len(small),
                      sixteen_stage_mean_us=mean(small), two_stage_launches=len(half), two_stage_mean_us=mean(half),
                      job_mix_mean_us=job_mix,
                      classes=("whole-triangle sweeps >= 38 us / half sweeps 10-38 us / early exits < 10 us" if "symm_sweep_kernel" in k
                               else "by duration: one-stage >= 52 us, two-stage 20-52 us, sixteen-stage < 20 us"),
                      one_stage_launches=len(full), one_stage_mean_us=(sum(full) / len(full) / 1e3) if full else None,
                      one_stage_min_us=min(full) / 1e3 if full else None, one_stage_max_us=max(full) / 1e3 if full else None)
    json.dump(out, open(f"profiles/{tag}_stage_kernel_trace.json", "w"), indent=1, sort_keys=True)
    for k, v in out.items():
        print("trace", k[-40:], v)
summary = collections.defaultdict(dict)
for sub in ("fetch", "write", "sq"):
    for f in glob.glob(f"{prefix}_{sub}/*/*_counter_collection.csv"):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            names = [name]
            # the stage kernel is launched with 16, 2 or 1 stages per iteration (same grid, different slab
            # width): the one-stage launches -- the full 4 N^2-byte sweeps -- are those that take >= 52 us at
            # config 3 (two-stage launches take ~40 us, sixteen-stage ones ~9 us)
            if "slab_stage_" in name and dur >= 52000:
                names.append(name + " [one-stage launches]")
            # the symmetric sweep is also launched on half the tiles (the two stages of a two-stage iteration, ~25 us
            # at config 3): the whole-triangle sweeps of one-stage iterations are those of >= 38 us
            if "symm_sweep_kernel" in name:
                names.append(name + (" [whole-triangle sweeps]" if dur >= 38000 else " [half sweeps]"))
            for nm in names:
                agg[nm][r["Counter_Name"]].append(float(r["Counter_Value"]))
                agg[nm]["_dur_ns"].append(dur)
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
# one symmetric one-stage iteration = its sweep (plain instance) + its apply: the entry bench.py's roofline.traffic reads
sw = [k for k in summary if "symm_sweep_kernel<5, false, false>" in k and "[whole-triangle sweeps]" in k]
ap = [k for k in summary if "symm_apply_kernel<5>" in k]
if sw and ap and "hbm_traffic_bytes_per_launch" in summary[sw[0]] and "hbm_traffic_bytes_per_launch" in summary[ap[0]]:
    a, b = summary[sw[0]], summary[ap[0]]
    both = {}
    for key in ("hbm_read_bytes_per_launch", "hbm_write_bytes_per_launch", "hbm_traffic_bytes_per_launch",
                "mean_duration_us_fetch", "mean_duration_us_write", "mean_duration_us_sq"):
        if key in a and key in b:
            both[key] = a[key] + b[key]
    both["parts"] = [sw[0], ap[0]]
    summary["symmetric one-stage iteration (sweep + apply)"] = both
json.dump(summary, open(f"profiles/{tag}_pmc_summary.json", "w"), indent=1, sort_keys=True)
for k, v in summary.items():
    print(k[:60], {c: round(x, 1) for c, x in v.items() if ("bytes" in c or "duration" in c) and not isinstance(x, list)})
