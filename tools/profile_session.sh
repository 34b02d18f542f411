#!/bin/bash
# One GPU call that regenerates everything under profiles/<round> (then: python tools/collect_profiles.py <prefix> <round>).
# usage (on the GPU box, from the repo root): bash tools/profile_session.sh <prefix>
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
P=$1
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 600 python3 bench.py > $O/${P}_bench_default.log 2>&1
timeout 600 python3 bench.py --mesh-n 289 --no-cpu-baseline > $O/${P}_c3.log 2>&1
timeout 300 python3 bench.py --aperture 0.1 --no-cpu-baseline > $O/${P}_c5.log 2>&1
timeout 300 python3 bench.py --width 3840 --height 2160 --steps 64 --no-cpu-baseline > $O/${P}_4k.log 2>&1
timeout 300 python3 bench.py --pipeline megakernel --steps 16 --no-cpu-baseline > $O/${P}_mega.log 2>&1
timeout 300 python3 bench.py --pipeline wavefront2 --steps 512 --warmup 256 --batch 256 --no-cpu-baseline > $O/${P}_lanes2.log 2>&1
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${P}_kt -- python3 $R/bench.py --no-cpu-baseline > $O/${P}_kt.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  d=$(echo $c | tr 'A-Z' 'a-z' | sed 's/_size//')
  timeout 600 rocprofv3 --pmc $c --output-format csv -d $O/${P}_$d -- python3 $R/bench.py --warmup 0 --no-cpu-baseline > $O/${P}_$d.log 2>&1
  timeout 900 rocprofv3 --pmc $c --output-format csv -d $O/${P}_${d}_c3 -- python3 $R/bench.py --warmup 0 --mesh-n 289 --no-cpu-baseline > $O/${P}_${d}_c3.log 2>&1
done
bash $R/tools/pmc_passes.sh ${P}_pmc > $O/${P}_pmc.log 2>&1
bash $R/tools/pmc_l1.sh ${P}_l1 > $O/${P}_l1.log 2>&1
echo session done
