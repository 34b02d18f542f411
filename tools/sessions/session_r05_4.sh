O=gpurun_out/r05; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "js_bench or halved or stream_pool or memory_limit or path_state" > $O/gpu_new_tests_2.log 2>&1; tail -5 $O/gpu_new_tests_2.log
cd /tmp
for c in c2 c3; do
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/kt_$c -- python3 $R/bench.py --config $c --steps 20 --warmup 5 --reps 3 --no-cpu-baseline --no-extra-configs --no-l1-microbench --no-parity-check > $R/$O/kt_$c.log 2>&1
  python3 $R/tools/launch_list.py $R/$O/kt_$c > $R/$O/launch_list_$c.txt 2>&1; tail -40 $R/$O/launch_list_$c.txt
  find $R/$O/kt_$c -name "*kernel_trace.csv" -size +4M -delete
done
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/kt_tick1 -- python3 $R/bench.py --steps 1 --warmup 3 --reps 3 --no-cpu-baseline --no-extra-configs --no-l1-microbench --no-parity-check > $R/$O/kt_tick1.log 2>&1
python3 $R/tools/launch_list.py $R/$O/kt_tick1 > $R/$O/launch_list_tick1.txt 2>&1; tail -30 $R/$O/launch_list_tick1.txt
