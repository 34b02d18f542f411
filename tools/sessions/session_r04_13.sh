# (experiment of round 4 that was NOT adopted: the sources it ran - primary form 3 / the tail kernel with an LDS copy of the top of the tree - are not in the tree; results: profiles/r04/README.md)
O=$GRAFT_REPO_ROOT/gpurun_out/r04; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for f in 3 1; do
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_f$f -- python3 $R/bench.py --steps 20 --warmup 5 --primary-form $f --no-cpu-baseline --no-l1-microbench --no-parity-check > /tmp/prof_f$f.log 2>&1
find /tmp/prof_f$f -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/s13_form${f}_kernel_stats.csv
tail -3 /tmp/prof_f$f.log | cut -c1-300
done
head -8 $O/s13_form3_kernel_stats.csv; head -6 $O/s13_form1_kernel_stats.csv
