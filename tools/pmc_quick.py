#!/usr/bin/env python3
"""Per-kernel sums of the counters in rocprofv3 --pmc output directories:  python tools/pmc_quick.py <dir> [<dir> ...]"""
import collections, csv, glob, re, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(set)
for d in sys.argv[1:]:
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"(k_wf_\w+|k_trace)(<[^>]*>)?", r["Kernel_Name"])
            if m:
                agg[m.group(0)][r["Counter_Name"]] += float(r["Counter_Value"])
                calls[m.group(0)].add(r["Dispatch_Id"])
for k in sorted(agg):
    c = agg[k]
    line = {n: f"{v:.4g}" for n, v in sorted(c.items())}
    der = {}
    if c.get("SQ_WAVE_CYCLES"):
        der["wait_any/wave_cyc"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
        der["wait_inst_any/wave_cyc"] = c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"]
        der["active_valu/busy_cyc"] = c["SQ_ACTIVE_INST_VALU"] / max(c["SQ_BUSY_CYCLES"], 1)
        der["lane_util"] = c["SQ_THREAD_CYCLES_VALU"] / max(c["SQ_ACTIVE_INST_VALU"] * 64, 1)
    if c.get("GRBM_GUI_ACTIVE"):
        der["TA_busy"] = c["TA_TA_BUSY_sum"] / c["GRBM_GUI_ACTIVE"] / 256 / 1  # per CU (sum over the TAs)
        der["TD_busy"] = c["TD_TD_BUSY_sum"] / c["GRBM_GUI_ACTIVE"] / 256
    print(k, "dispatches", len(calls[k]), line, {a: round(b, 3) for a, b in der.items()})
