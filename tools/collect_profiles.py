#!/usr/bin/env python3
"""Derive the committed profile files from a tools/prof_session.sh session.
    python tools/collect_profiles.py <tag> <workload-key> [<tag> <workload-key> ...]
 -> profiles/<round>/<tag>_kernel_stats.csv, <tag>_bench.json.log, <tag>_pmc_summary.txt and profiles/hbm_traffic.json
    (bytes per sample per kernel, stamped with the hash of the kernel sources the numbers were measured on)."""
import collections, csv, glob, json, os, re, shutil, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

G, P = "gpurun_out", os.path.join("profiles", os.environ.get("FSPT_PROFILE_ROUND", "r05"))
os.makedirs(P, exist_ok=True)
pairs = list(zip(sys.argv[1::2], sys.argv[2::2]))
out = {"source_sha": bench.source_sha(),
       "how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate runs) over `bench.py --warmup 0 --reps 1 --no-cpu-baseline"
              " <the timed configuration>` (the warm-up and the timed regions of the driver's 20-step form + the counting ticks, which run other kernel variants); bytes ="
              " (2*FETCH_SIZE + WRITE_SIZE)*1024.  The factor 2 is calibrated for gathers too (profiles/r03/fetch_calib.json,"
              " tools/fetch_calib.sh): on gfx950 EVERY L2 read miss is one 128-byte request (TCC_EA0_RDREQ_128B == TCC_EA0_RDREQ,"
              " for 4-byte random reads and 16-byte streams alike) and FETCH_SIZE tallies it as 64 bytes; the kernels' own"
              " request-size split is in rdreq_by_size.  These are bytes between L2 and the fabric: Infinity-Cache hits are"
              " included (a 64 MiB table read twice reports the same bytes), they are an upper bound on DRAM bytes", "workloads": {}}


def newest(paths):
    """gpurun merges a session's files into directories that may still hold an older session's: keep, per directory,
    the files written within a minute of its newest one."""
    by_dir = collections.defaultdict(list)
    for f in paths:
        by_dir[os.path.dirname(f)].append(f)
    out = []
    for d, fs in by_dir.items():
        t = max(os.path.getmtime(f) for f in fs)
        out += [f for f in fs if t - os.path.getmtime(f) < 60]
    return sorted(out)


def kname(n):
    m = re.search(r"(k_wf_\w+|k_trace)(<[^>]*>)?", n)
    return m.group(0) if m else None


for tag, wl in pairs:
    base = os.path.basename(tag)  # the session may live in a sub-directory of gpurun_out
    ks = newest(glob.glob(f"{G}/{tag}_kt/**/*kernel_stats.csv", recursive=True))
    if ks:
        shutil.copy(ks[0], os.path.join(P, f"{base}_kernel_stats.csv"))
    log = f"{G}/{tag}_kt.log"
    samples = None
    if os.path.exists(log):
        for l in open(log):
            if l.startswith("{"):
                open(os.path.join(P, f"{base}_bench.json.log"), "w").write(l)
                j = json.loads(l)
                w, h = re.search(r"(\d+)x(\d+)", j["metric"]).groups()
                samples = int(w) * int(h) * j["steps"]  # samples of the one timed batch
    res = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        agg = collections.defaultdict(list)
        for f in newest(glob.glob(f"{G}/{tag}_{c}/**/*counter_collection.csv", recursive=True)):
            for r in csv.DictReader(open(f)):
                k = kname(r["Kernel_Name"])
                if k and r["Counter_Name"] == c:
                    agg[k].append(float(r["Counter_Value"]))
        res[c] = agg
    rdreq = collections.defaultdict(dict)
    for f in newest(glob.glob(f"{G}/{tag}_rdreq/**/*counter_collection.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            k = kname(r["Kernel_Name"])
            if k:
                rdreq[k][r["Counter_Name"]] = rdreq[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    # the traffic passes time one region of the timed configuration: its samples
    tlog = f"{G}/{tag}_FETCH_SIZE.log"
    if os.path.exists(tlog):
        for l in open(tlog):
            if l.startswith("{"):
                j = json.loads(l)
                w, h = re.search(r"(\d+)x(\d+)", j["metric"]).groups()
                samples = int(w) * int(h) * (j["steps"] * j.get("reps", 1) + j["warmup"])  # every tick the counters saw
    kern = {}
    for k, f in res["FETCH_SIZE"].items():
        w = res["WRITE_SIZE"].get(k, [])
        if not w or not samples or "<true" in k:
            continue  # counting variants are not the timed kernels
        total = (2 * sum(f) + sum(w)) * 1024
        kern[k] = {"launches": len(f), "FETCH_SIZE_KB_sum": sum(f), "WRITE_SIZE_KB_sum": sum(w),
                   "hbm_bytes_per_launch": total / len(f), "hbm_bytes_per_sample": total / samples,
                   "read_bytes_per_sample": 2 * sum(f) * 1024 / samples, "write_bytes_per_sample": sum(w) * 1024 / samples}
        rq = rdreq.get(k)
        if rq:
            kern[k]["rdreq_by_size"] = {n: rq.get(n, 0.0) for n in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum")}
    out["workloads"][wl] = {"samples_counted": samples, "kernels": kern}
    # SQ / TA / TD summaries
    agg = collections.OrderedDict()
    for f in newest(glob.glob(f"{G}/{tag}_sq*/**/*counter_collection.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            k = kname(r["Kernel_Name"])
            if k and "<true" not in k:
                agg.setdefault((k, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    if agg:
        with open(os.path.join(P, f"{base}_pmc_summary.txt"), "w") as o:
            for (k, c), v in agg.items():
                o.write(f"{k:32s} {c:40s} launches={len(v):4d} sum={sum(v):.6g} mean={sum(v) / len(v):.6g}\n")
    # derived per-kernel figures: vector-memory pipeline busy fractions (instances calibrated on the saturated
    # microbenchmark, profiles/r01/l1_pipe.json), VALU lane utilisation, share of wave-cycles spent waiting, L2 hit rate
    INST = 31.334512006803482
    c = collections.defaultdict(dict)
    for (k, name), v in agg.items():
        c[k][name] = sum(v)
    derived = {}
    for k, v in c.items():
        d = {}
        if "GRBM_GUI_ACTIVE" in v and v["GRBM_GUI_ACTIVE"]:
            d["TA_busy"] = round(v.get("TA_TA_BUSY_sum", 0) / v["GRBM_GUI_ACTIVE"] / INST, 3)
            d["TD_busy"] = round(v.get("TD_TD_BUSY_sum", 0) / v["GRBM_GUI_ACTIVE"] / INST, 3)
        if v.get("SQ_ACTIVE_INST_VALU"):
            d["valu_lane_utilisation"] = round(v["SQ_THREAD_CYCLES_VALU"] / (64 * v["SQ_ACTIVE_INST_VALU"]), 3)
        if v.get("SQ_WAVE_CYCLES"):
            d["wait_any_share"] = round(v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], 3)
            d["valu_active_share"] = round(v["SQ_ACTIVE_INST_VALU"] / v["SQ_WAVE_CYCLES"], 3)
        if v.get("TCC_REQ_sum"):
            d["l2_hit_rate"] = round(v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"]), 3)
        derived[k] = d
    # VALU wave-instructions per sample (SQ_INSTS_VALU over the one 32-tick batch of the SQ pass)
    sq_samples = None
    for l in (open(f"{G}/{tag}_sq1.log") if os.path.exists(f"{G}/{tag}_sq1.log") else []):
        if l.startswith("{"):
            j = json.loads(l)
            w, h = re.search(r"(\d+)x(\d+)", j["metric"]).groups()
            sq_samples = int(w) * int(h) * j["steps"]
    if sq_samples:
        for k, v in c.items():
            if "SQ_INSTS_VALU" in v and k in kern:
                kern[k]["valu_wave_instr_per_sample"] = v["SQ_INSTS_VALU"] / sq_samples
                derived[k]["valu_wave_instr_per_sample"] = round(v["SQ_INSTS_VALU"] / sq_samples, 2)
    # the counters bench.py attaches to its kernel classes (roofline.kernels.*.counters, roofline.dominant_kernel_counters)
    for k, d in derived.items():
        if k in kern:
            for src, dst in (("TA_busy", "ta_busy"), ("TD_busy", "td_busy"), ("valu_active_share", "valu_active_share"),
                             ("wait_any_share", "wait_share"), ("l2_hit_rate", "l2_hit"), ("valu_lane_utilisation", "valu_lane_utilisation")):
                if src in d:
                    kern[k][dst] = d[src]
    if derived:
        json.dump({"note": "from <tag>_pmc_summary.txt: one timed region of bench.py (SQ_ARGS of tools/prof_session.sh, default --steps 20); busy = *_BUSY_sum / "
                           "GRBM_GUI_ACTIVE / 31.33 instances (profiles/r01/l1_pipe.json)", "kernels": derived},
                  open(os.path.join(P, f"{base}_derived.json"), "w"), indent=1)
        for k, d in derived.items():
            print(wl, k, d)
    for k, v in kern.items():
        print(wl, k, "launches", v["launches"], "GB/launch", round(v["hbm_bytes_per_launch"] / 1e9, 3), "B/sample", round(v["hbm_bytes_per_sample"], 1))
json.dump(out, open(os.path.join("profiles", "hbm_traffic.json"), "w"), indent=1)
