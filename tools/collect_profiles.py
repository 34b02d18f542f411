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
cp(f"{G}/{prefix}_lanes2.log", "final_lanes2.json.log")
cp(f"{G}/{prefix}_l1/summary_b.txt", "l1_pmc_bench.txt")
cp(f"{G}/{prefix}_l1/summary_m.txt", "l1_pmc_microbench.txt")
cp(f"{G}/{prefix}_pmc/summary.txt", "final_pmc_summary.txt")

KERNELS = ("k_wf_trace<false>", "k_wf_logic<false, false, true>", "k_wf_logic<false, true, true>", "k_wf_resolve")
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
                        "128-B request); separate --pmc passes over `bench.py --warmup 0 --no-cpu-baseline` (one 128-tick batch = the timed region)"}
    if o:
        out[name] = o
json.dump(out, open(os.path.join(P, "final_hbm_traffic.json"), "w"), indent=1)
for name, o in out.items():
    for k, v in o.items():
        print(name, k, "launches", v["launches"], "GB/launch", round(v["hbm_bytes_per_launch_corrected"] / 1e9, 3))


# ---- vector-memory pipeline busy fractions (tools/pmc_l1.sh) -> l1_pipe.json (bench.py: roofline.vmem_pipe) ----
import re


def load(path):
    d = collections.OrderedDict()
    for l in open(path):
        m = re.match(r"(.{62}) (\S+)\s+launches=\s*(\d+) sum=(\S+) max=(\S+)", l)
        if m:
            d.setdefault(m.group(1).strip(), {})[m.group(2)] = float(m.group(4))
    return d


pb, pm = os.path.join(P, "l1_pmc_bench.txt"), os.path.join(P, "l1_pmc_microbench.txt")
if os.path.exists(pb) and os.path.exists(pm):
    b, m = load(pb), load(pm)
    cal = [v["TCP_GATE_EN1_sum"] / v["GRBM_GUI_ACTIVE"] for k, v in m.items() if k.startswith("k<0>")][0]
    res = {"note": "rocprofv3 --pmc (tools/pmc_l1.sh) over `bench.py --warmup 0 --no-cpu-baseline` and "
                   "tools/microbench/gather2; busy = *_BUSY_sum / GRBM_GUI_ACTIVE / instances, instances = "
                   "TCP_GATE_EN1_sum/GRBM_GUI_ACTIVE of the saturated microbenchmark (%.2f)" % cal,
           "instances": cal, "kernels": {}, "microbench": {}}

    def row(v):
        g = v["GRBM_GUI_ACTIVE"]
        return {"TA_busy": round(v["TA_TA_BUSY_sum"] / g / cal, 3), "TD_busy": round(v["TD_TD_BUSY_sum"] / g / cal, 3),
                "tcp_lane_accesses_per_cycle_per_cu": round(v["TCP_TOTAL_ACCESSES_sum"] / g / cal, 3),
                "tcp_cache_accesses_per_cycle_per_cu": round(v["TCP_TOTAL_CACHE_ACCESSES_sum"] / g / cal, 3)}
    for k, v in b.items():
        res["kernels"][k.replace("(fspt::WfP)", "").replace("fspt::", "")] = row(v)
    names = {"k<0>": "own4 (64 lanes x own 64-B record, 4 x dwordx4)", "k<6>": "coal (4 x fully coalesced dwordx4)",
             "k<1>": "quad (4 lanes share a record)", "k<9>": "dman (quad pattern via global_load_lds_dwordx4)"}
    for k, v in m.items():
        for pfx, n in names.items():
            if k.startswith(pfx):
                res["microbench"][n] = row(v)
    res["l1_gather_GBps"] = {"divergent_64B_records": round(22.5 * 256 * 2.4, 0), "coalesced_dwordx4": round(31.0 * 256 * 2.4, 0),
                             "assumed_clock_GHz": 2.4, "source": "l1_gather_modes.log, l1_gather_dma_chain.log"}
    json.dump(res, open(os.path.join(P, "l1_pipe.json"), "w"), indent=1)
    for k in ("k_wf_trace<false>", "k_wf_logic<false, false, true>", "k_wf_logic<false, true, true>"):
        print(k, res["kernels"].get(k))
