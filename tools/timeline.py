#!/usr/bin/env python3
"""Overlap analysis of a rocprofv3 --kernel-trace run: per kernel class the summed duration, and for the whole trace the
time at least one kernel was running (union), the time two or more were (overlap) and the idle gaps between the first
and the last kernel of every burst.   usage: tools/timeline.py <dir-with-*_kernel_trace.csv> [min_gap_us]"""
import csv, glob, re, sys, collections
d = sys.argv[1]
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_wf_\w+|k_trace|k_draw|k_camera)", r["Kernel_Name"])
        if not m or "true>" in r["Kernel_Name"].split("<")[-1][:5]:
            if not m: continue
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1), r["Kernel_Name"]))
rows.sort()
if not rows: sys.exit("no kernels")
# bursts: split where nothing runs for > 200 us (between bench regions)
bursts, cur, end = [], [rows[0]], rows[0][1]
for r in rows[1:]:
    if r[0] > end + 200000:
        bursts.append(cur); cur = []
    cur.append(r); end = max(end, r[1])
bursts.append(cur)
big = sorted(bursts, key=lambda b: -(max(x[1] for x in b) - b[0][0]))[:8]
for b in sorted(big, key=lambda b: b[0][0]):
    t0, t1 = b[0][0], max(x[1] for x in b)
    ev = sorted([(x[0], 1) for x in b] + [(x[1], -1) for x in b])
    run = busy = over = 0; last = t0
    for t, dlt in ev:
        if run >= 1: busy += t - last
        if run >= 2: over += t - last
        run += dlt; last = t
    per = collections.Counter(); cnt = collections.Counter()
    for x in b: per[x[2]] += x[1] - x[0]; cnt[x[2]] += 1
    print(f"burst {(t1 - t0) / 1e6:8.3f} ms  kernels {len(b):4d}  busy {busy / 1e6:8.3f}  idle {(t1 - t0 - busy) / 1e6:7.3f}  >=2 running {over / 1e6:7.3f}  sum of durations {sum(per.values()) / 1e6:8.3f}  "
          + "  ".join(f"{k}:{v / 1e6:.3f}/{cnt[k]}" for k, v in sorted(per.items())))
