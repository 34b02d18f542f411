# the 4- and 8-rank programs on ONE GPU (--share-gpu: gloo through host memory): plumbing of the weak-scaled frames
# (3840x2160, 5432x3056), tile dealing over 4 / 8 ranks, the gather of 4 / 8 pieces, parity of the assembled frame
O=gpurun_out/r04; mkdir -p $O
{
for n in 4 8; do
  echo "== bench.py --gpus $n --share-gpu --steps 8 --warmup 2 --reps 2"
  timeout 1200 python3 bench.py --gpus $n --share-gpu --steps 8 --warmup 2 --reps 2 --no-l1-microbench --rendezvous-timeout 300 2>$O/s30_share_gpu_n$n.err | python3 -c "
import sys,json
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line)
        print(d['metric'], d['value'], d['n_gpus'], d['config']['world_size_seen'], d['config']['sharding'], 'exchange_ms', d.get('exchange_ms'), 'parity', d['parity_check'])
"
  grep -c "bench rank" $O/s30_share_gpu_n$n.err
done
} > $O/s30_share_gpu_4_8.log 2>&1
cat $O/s30_share_gpu_4_8.log
