# what the driver runs at round end, on the final tree: smoke(), the GPU suite, the bench command
O=gpurun_out/r04; mkdir -p $O
{
timeout 600 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
timeout 900 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -2
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | cut -c1-400
} > $O/s29_driver_like.log 2>&1
cat $O/s29_driver_like.log
