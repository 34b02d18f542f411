"""Print the kernel timeline of a rocprofv3 --kernel-trace run (csv or rocpd .db): start offset, duration, gap.
    python tools/timeline.py <dir> [last_n]"""
import csv, glob, os, sqlite3, sys
d = sys.argv[1]
last = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
for f in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
    c = sqlite3.connect(f)
    rows += list(c.execute("select name, start, end from kernels"))
rows.sort(key=lambda r: r[1])
rows = rows[-last:]
t0, prev_end = rows[0][1], rows[0][1]
for name, s, e in rows:
    short = name.replace("void fspt::", "").split("(")[0][:34]
    print(f"{short:34s} start {(s - t0) / 1e3:9.1f}  dur {(e - s) / 1e3:8.1f}  gap {(s - prev_end) / 1e3:6.1f}")
    prev_end = e
