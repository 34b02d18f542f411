cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g5
python bench.py --steps 20 --warmup 5 > gpurun_out/g5/bench20.json 2> gpurun_out/g5/bench20.err
cut -c1-1500 gpurun_out/g5/bench20.json; tail -3 gpurun_out/g5/bench20.err
tools/microbench/l1_peak
tools/prof_r03.sh g5/c2 > gpurun_out/g5/prof.log 2>&1
ls gpurun_out/g5
