# Bench lines of the BASELINE configs other than the default, one GPU:  tools/configs_r02.sh <tag>
cd ${GRAFT_REPO_ROOT:-/root/repo}
T=$1
timeout 600 python3 bench.py --config c3 --no-cpu-baseline > gpurun_out/${T}_c3.log 2>&1
timeout 600 python3 bench.py --config c5 --no-cpu-baseline > gpurun_out/${T}_c5.log 2>&1
timeout 600 python3 bench.py --config c4 --gpus 1 --no-cpu-baseline > gpurun_out/${T}_c4_1gpu.log 2>&1
timeout 600 python3 bench.py --textured --no-cpu-baseline > gpurun_out/${T}_textured.log 2>&1
timeout 600 python3 bench.py --pipeline megakernel --steps 32 --warmup 8 --no-cpu-baseline > gpurun_out/${T}_mega.log 2>&1
timeout 600 python3 bench.py --pipeline wavefront2 --steps 256 --warmup 256 --batch 256 --no-cpu-baseline > gpurun_out/${T}_lanes2.log 2>&1
echo configs done
