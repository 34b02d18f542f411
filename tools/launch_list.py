#!/usr/bin/env python3
"""The launches of one timed batch, in order: start offset, duration, gap to the previous kernel's end (us), kernel.
usage: tools/launch_list.py <dir-with-*_kernel_trace.csv> [which-burst-from-the-end, default 1 = the last timed region]"""
import csv, glob, re, sys
d = sys.argv[1]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows = []
import os
files = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
newest = max(os.path.getmtime(f) for f in files)  # (a directory merged over several sessions holds older runs' traces too)
for f in files:
    if newest - os.path.getmtime(f) > 60:
        continue
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
bursts, cur, end = [], [rows[0]], rows[0][1]
for r in rows[1:]:
    if r[0] > end + 200000:
        bursts.append(cur); cur = []
    cur.append(r); end = max(end, r[1])
bursts.append(cur)
big = [b for b in bursts if any("k_wf_primary<false" in x[2] for x in b)]
b = big[-which]
t0 = b[0][0]
prev = t0
tot = {}
for s, e, n in b:
    short = re.sub(r"\(.*", "", n).replace("void fspt::", "").replace("fspt::", "")
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap {(s - prev) / 1e3:6.1f}  {short[:70]}")
    prev = max(prev, e)
    k = re.sub(r"<.*", "", short)
    tot[k] = tot.get(k, 0) + (e - s)
print("burst", (prev - t0) / 1e3, "us;", "  ".join(f"{k}:{v / 1e3:.1f}" for k, v in tot.items()), "; sum", sum(tot.values()) / 1e3)
