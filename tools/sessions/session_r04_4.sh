O=gpurun_out/r04; mkdir -p $O
for n in tprof tprof1e9; do
echo "== $n tick1"; FSPT_LIB=$PWD/ab_libs/$n.so timeout 300 python3 bench.py --steps 1 --warmup 3 --reps 1 --no-cpu-baseline --no-l1-microbench --no-parity-check 2>&1 | grep -E "tailprof" | tail -12
echo "== $n c3 20"; FSPT_LIB=$PWD/ab_libs/$n.so timeout 300 python3 bench.py --steps 20 --warmup 5 --reps 1 --config c3 --no-cpu-baseline --no-l1-microbench --no-parity-check 2>&1 | grep -E "tailprof" | tail -10
done > $O/tail_prof.log 2>&1
cat $O/tail_prof.log
