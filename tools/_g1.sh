cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g1
python -m pytest tests -m gpu -x -q > gpurun_out/g1/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/g1/pytest.log
python bench.py --steps 20 --warmup 5 > gpurun_out/g1/bench20.json 2> gpurun_out/g1/bench20.err
python bench.py > gpurun_out/g1/bench128.json 2> gpurun_out/g1/bench128.err
tools/fetch_calib.sh g1/calib > gpurun_out/g1/calib.log 2>&1
tools/microbench/gather4 > gpurun_out/g1/gather4.log 2>&1
tail -3 gpurun_out/g1/pytest.log; cut -c1-400 gpurun_out/g1/bench20.json; tail -30 gpurun_out/g1/calib.log; cat gpurun_out/g1/gather4.log
