#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration on known byte counts (tools/microbench/fetch_calib.hip); separate --pmc passes,
# no trace domains.   usage: tools/fetch_calib.sh <outdir-under-gpurun_out>
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1
mkdir -p $O
B=$R/tools/microbench/fetch_calib
[ -x $B ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $B $R/tools/microbench/fetch_calib.hip
cd /tmp
rocprofv3 -L 2>/dev/null | grep -o "TCC_EA0_R[A-Z0-9_]*\|TCC_EA0_WR[A-Z0-9_]*\|FETCH_SIZE\|WRITE_SIZE\|TCC_BUBBLE[A-Z_]*" | sort -u > $O/counters_available.txt
for tab in 2048 64; do
  $B $tab 64 > $O/alg_$tab.jsonl 2>&1
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $set --output-format csv -d $O/t${tab}_p$i -- $B $tab 64 > $O/t${tab}_p$i.log 2>&1
  done
done
python3 - "$O" <<'PY'
import csv, glob, json, sys, collections, re
O = sys.argv[1]
res = {}
for tab in (2048, 64):
    alg = {}
    for l in open(f"{O}/alg_{tab}.jsonl"):
        if l.startswith("{"):
            j = json.loads(l); alg[j["kernel"]] = j
    cnt = collections.defaultdict(dict)
    for f in sorted(glob.glob(f"{O}/t{tab}_p*/*/*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            m = re.match(r"(k_\w+(<\d+>)?)", r["Kernel_Name"].replace("void ", ""))
            if not m: continue
            cnt[m.group(1)][r["Counter_Name"]] = cnt[m.group(1)].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    rows = {}
    for k, a in alg.items():
        c = cnt.get(k, {})
        row = {"alg_bytes": a["alg_bytes"], "ms": a["ms"], "alg_GBps": a["alg_GBps"]}
        row.update({n: v for n, v in c.items()})
        if "FETCH_SIZE" in c and not k.startswith("k_write"):
            row["fetch_bytes_reported"] = c["FETCH_SIZE"] * 1024.0
            row["true_over_reported"] = round(a["alg_bytes"] / max(1.0, c["FETCH_SIZE"] * 1024.0), 4)
        if "WRITE_SIZE" in c and k.startswith("k_write"):
            row["write_bytes_reported"] = c["WRITE_SIZE"] * 1024.0
            row["true_over_reported"] = round(a["alg_bytes"] / max(1.0, c["WRITE_SIZE"] * 1024.0), 4)
        if "TCC_EA0_RDREQ_sum" in c and not k.startswith("k_write"):
            # request sizes: FETCH_SIZE tallies every request as 64 bytes; the per-size counters give the bytes
            r32, r64, r128 = (c.get(f"TCC_EA0_RDREQ_{n}B_sum", 0.0) for n in (32, 64, 128))
            rest = c["TCC_EA0_RDREQ_sum"] - r32 - r64 - r128
            row["rdreq_bytes_by_size"] = 32.0 * r32 + 64.0 * (r64 + max(rest, 0.0)) + 128.0 * r128
            row["true_over_by_size"] = round(a["alg_bytes"] / max(1.0, row["rdreq_bytes_by_size"]), 4)
        rows[k] = row
    res[f"table_{tab}MiB"] = rows
json.dump(res, open(O + "/fetch_calib.json", "w"), indent=1)
for t, rows in res.items():
    print(t)
    for k, r in rows.items():
        print(f"  {k:18s} alg {r['alg_bytes']/1e9:8.3f} GB  {r['alg_GBps']:8.1f} GB/s  reported {r.get('fetch_bytes_reported', r.get('write_bytes_reported', 0))/1e9:8.3f} GB  true/reported {r.get('true_over_reported')}  rdreq {r.get('TCC_EA0_RDREQ_sum')}  32B {r.get('TCC_EA0_RDREQ_32B_sum')} 64B {r.get('TCC_EA0_RDREQ_64B_sum')} 128B {r.get('TCC_EA0_RDREQ_128B_sum')} dram {r.get('TCC_EA0_RDREQ_DRAM_sum')}  true/by-size {r.get('true_over_by_size')}")
PY
