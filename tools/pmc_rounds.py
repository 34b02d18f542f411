#!/usr/bin/env python3
"""Per-launch counter table of the FIRST batch of a tools/pmc_rounds.sh session:  python tools/pmc_rounds.py <dir>
A batch's launches in dispatch order are its rounds (primary = round 1; trace / logic launches = rounds 2, 3, ...)."""
import collections, csv, glob, re, sys

d = sys.argv[1]
rows = collections.OrderedDict()  # (pass, dispatch id) -> {kernel, counters}
for f in sorted(glob.glob(f"{d}/p*/**/*counter_collection.csv", recursive=True)):
    p = re.search(r"/(p\d+)/", f).group(1)
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_wf_\w+)(<[^>]*>)?", r["Kernel_Name"])
        if not m:
            continue
        e = rows.setdefault((p, int(r["Dispatch_Id"])), {"k": m.group(1), "c": {}})
        e["c"][r["Counter_Name"]] = e["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
# per pass: the launches of the first batch (up to and including the first resolve), numbered per kernel class
per = collections.defaultdict(dict)  # (kernel, nth launch) -> counters merged over the passes
for p in sorted({k[0] for k in rows}):
    nth = collections.Counter()
    for (pp, did), e in sorted(rows.items()):
        if pp != p:
            continue
        nth[e["k"]] += 1
        per[(e["k"], nth[e["k"]])].update(e["c"])
        if e["k"] == "k_wf_resolve":
            break


def ratio(c, a, b):
    return c[a] / c[b] if a in c and c.get(b) else float("nan")


print(f"{'kernel':14s} {'#':>2s} {'L2 hit':>7s} {'L2 req':>10s} {'L1 acc':>10s} {'L1->L2 rd':>10s} {'L1 miss':>7s} {'lane use':>8s} {'VMEM rd':>9s} {'TA busy':>7s} {'TD busy':>7s}")
for (k, n), c in sorted(per.items(), key=lambda x: (x[0][0], x[0][1])):
    hit = c.get("TCC_HIT_sum", 0) / max(c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0), 1)
    l1m = ratio(c, "TCP_TCC_READ_REQ_sum", "TCP_TOTAL_CACHE_ACCESSES_sum")
    lane = c.get("SQ_THREAD_CYCLES_VALU", 0) / max(c.get("SQ_ACTIVE_INST_VALU", 0) * 64, 1)
    ta = c.get("TA_TA_BUSY_sum", 0) / max(c.get("GRBM_GUI_ACTIVE", 0) * 256, 1)
    td = c.get("TD_TD_BUSY_sum", 0) / max(c.get("GRBM_GUI_ACTIVE", 0) * 256, 1)
    print(f"{k:14s} {n:2d} {hit:7.3f} {c.get('TCC_REQ_sum', 0):10.4g} {c.get('TCP_TOTAL_CACHE_ACCESSES_sum', 0):10.4g} {c.get('TCP_TCC_READ_REQ_sum', 0):10.4g} {l1m:7.3f} {lane:8.3f} {c.get('SQ_INSTS_VMEM_RD', 0):9.4g} {ta:7.3f} {td:7.3f}")
