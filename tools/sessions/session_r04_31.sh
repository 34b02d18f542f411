# BASELINE configs[3] (C4: 3840x2160, depth 8, 8 ranks, strong scaling) as an 8-rank program on ONE GPU, both exchanges
O=gpurun_out/r04; mkdir -p $O
{
for ex in gather reduce; do
  echo "== bench.py --gpus 8 --share-gpu --config c4 --exchange $ex --steps 8 --warmup 2 --reps 2"
  timeout 1200 python3 bench.py --gpus 8 --share-gpu --config c4 --exchange $ex --steps 8 --warmup 2 --reps 2 --no-l1-microbench --rendezvous-timeout 300 2>$O/s31_c4_$ex.err | python3 -c "
import sys,json
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line)
        print(d['metric'], d['value'], d['scaling'], d['n_gpus'], d['config']['world_size_seen'], d['config']['sharding'], d['config']['exchange'][:20], 'exchange_ms', d.get('exchange_ms'), 'parity', d['parity_check'])
"
  grep -c "bench rank" $O/s31_c4_$ex.err
done
} > $O/s31_share_gpu_c4.log 2>&1
cat $O/s31_share_gpu_c4.log
