O=gpurun_out/r04; mkdir -p $O
tools/ab_run.sh "--steps 20 --warmup 5" cur im8 im24 im32 lu4 lu16 top15 top63 > $O/ab_tunables_r04.log 2>&1; cat $O/ab_tunables_r04.log
