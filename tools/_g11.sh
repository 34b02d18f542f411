cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g11
tools/prof_r03.sh g11/final > gpurun_out/g11/prof_final.log 2>&1
tools/prof_r03.sh g11/textured --textured > gpurun_out/g11/prof_textured.log 2>&1
tools/prof_r03.sh g11/c3 --config c3 > gpurun_out/g11/prof_c3.log 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/g11/bench20_pre.json 2>gpurun_out/g11/bench20_pre.err
python bench.py > gpurun_out/g11/bench128_pre.json 2>gpurun_out/g11/bench128_pre.err
find gpurun_out/g11 -name "*kernel_trace.csv" -size +2M -delete
find gpurun_out/g11 -name "*agent_info.csv" -delete
du -sh gpurun_out/g11
cut -c1-300 gpurun_out/g11/bench20_pre.json
