#!/bin/bash
# Per-kernel register / scratch / occupancy table of fspt_kernels.hip (hipcc -Rpass-analysis=kernel-resource-usage), demangled.
# usage: tools/kernel_resources.sh [extra -D flags]
cd "$(dirname "$0")/.." || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off ${AB_SLP:--fno-slp-vectorize} -std=c++17 -Wno-unused-value -c --cuda-device-only \
  -Rpass-analysis=kernel-resource-usage "$@" fspt_amd/csrc/fspt_kernels.hip -o /tmp/fspt_kernels_res.o 2>&1 |
python3 -c '
import re, sys, subprocess
rows, cur = [], None
for line in sys.stdin:
    m = re.search(r"Function Name: (\S+)", line)
    if m: cur = {"name": m.group(1)}; rows.append(cur); continue
    for key, pat in (("sgpr", r" SGPRs: (\d+)"), ("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                     ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("sspill", r"SGPRs Spill: (\d+)"), ("vspill", r"VGPRs Spill: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
        m = re.search(pat, line)
        if m and cur is not None: cur[key] = int(m.group(1))
names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.split("\n")
print("%-62s %5s %5s %7s %6s %4s %6s" % ("kernel", "VGPR", "SGPR", "scratch", "vspill", "occ", "LDS"))
for r, n in zip(rows, names):
    n = re.sub(r"^void fspt::", "", n); n = re.sub(r"\(.*$", "", n)
    print("%-62s %5d %5d %7d %6d %4d %6d" % (n, r.get("vgpr", -1), r.get("sgpr", -1), r.get("scratch", -1), r.get("vspill", -1), r.get("occ", -1), r.get("lds", -1)))
'
