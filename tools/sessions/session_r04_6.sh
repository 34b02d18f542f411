O=gpurun_out/r04; mkdir -p $O
tools/ab_run.sh "--steps 20 --warmup 5 --no-parity-check" e0 e1 e2 > $O/exp_noenv.log 2>&1; cat $O/exp_noenv.log
