# the suspension budget of the trace launches (steps a starved wave walks on before it parks its rays) on the final
# kernels: the default against a few values, C2 and the 1 M-triangle scene at 20 steps
O=gpurun_out/r04; mkdir -p $O
{
for rep in 1 2; do
for cfg in "--steps 20 --warmup 5" "--steps 20 --warmup 5 --config c3"; do
for tb in -1 0 8 16 32 64 128; do
  echo -n "== $cfg budget $tb: "
  timeout 900 python3 bench.py $cfg --trace-budget $tb --no-cpu-baseline --no-l1-microbench 2>/dev/null | python3 -c "
import sys,json
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); k=d['roofline'].get('kernels',{})
        print(d['value'], d['ms_per_step'], {c:v['ms_per_step'] for c,v in k.items()}, 'parity', (d.get('parity_check') or {}).get('equal'))
"
done; done; done
} > $O/s32_trace_budget.log 2>&1
cat $O/s32_trace_budget.log
