#!/usr/bin/env python3
"""Copies the rocprofv3 / bench outputs of a gpurun profile session (gpurun_out/<prefix>_*) into
profiles/<round>/ and derives the per-kernel HBM traffic file bench.py reads.

    python tools/collect_profiles.py r1 r01
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

prefix, rnd = sys.argv[1], sys.argv[2]
G = "gpurun_out"
P = os.path.join("profiles", rnd)
os.makedirs(P, exist_ok=True)


def cp(src, dst):
    if os.path.exists(src):
        shutil.copy(src, os.path.join(P, dst))


ks = glob.glob(f"{G}/{prefix}_kt/*/*_kernel_stats.csv")
if ks:
    cp(ks[0], "final_kernel_stats.csv")
cp(f"{G}/{prefix}_bench_default.log", "bench_default.json.log")
cp(f"{G}/{prefix}_c3.log", "final_c3.json.log")
cp(f"{G}/{prefix}_c5.log", "final_c5.json.log")
cp(f"{G}/{prefix}_mega.log", "final_mega.json.log")
cp(f"{G}/{prefix}_4k.log", "final_4k.json.log")
cp(f"{G}/{prefix}_pmc/summary.txt", "final_pmc_summary.txt")

KERNELS = ("k_wf_trace<false>", "k_wf_logic<false, false>", "k_wf_logic<false, true>", "k_wf_gen<true, false>", "k_wf_resolve")
out = {}
for tag, name in (("", "c2_70k"), ("_c3", "c3_1M")):
    res = {}
    for d, c in ((f"{prefix}_fetch{tag}", "FETCH_SIZE"), (f"{prefix}_write{tag}", "WRITE_SIZE")):
        agg = collections.defaultdict(list)
        for f in glob.glob(f"{G}/{d}/*/*_counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                for k in KERNELS:
                    if k in r["Kernel_Name"]:
                        agg[k].append(float(r["Counter_Value"]))
        res[c] = agg
    o = {}
    for k in res.get("FETCH_SIZE", {}):
        f, w = res["FETCH_SIZE"][k], res["WRITE_SIZE"].get(k, [])
        if not f or not w:
            continue
        o[k] = {"launches": len(f), "FETCH_SIZE_KB_sum": sum(f), "WRITE_SIZE_KB_sum": sum(w),
                "hbm_bytes_per_launch_corrected": (2 * sum(f) / len(f) + sum(w) / len(w)) * 1024,
                "note": "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per MI355X_MICROARCH.md (gfx950 FETCH_SIZE counts 64 B per "
                        "128-B request); separate --pmc passes over `bench.py --warmup 0 --no-cpu-baseline` (one 64-tick batch = the timed region; batch 64)"}
    if o:
        out[name] = o
json.dump(out, open(os.path.join(P, "final_hbm_traffic.json"), "w"), indent=1)
for name, o in out.items():
    for k, v in o.items():
        print(name, k, "launches", v["launches"], "GB/launch", round(v["hbm_bytes_per_launch_corrected"] / 1e9, 3))
