O=gpurun_out/r05; mkdir -p $O
tools/ab_args.sh "--config c3 --steps 20 --warmup 5" "--primary-form 0" "--primary-form 1" "--primary-form 2" > $O/ab_primary_form_tickets_c3.log 2>&1; cat $O/ab_primary_form_tickets_c3.log
tools/ab_args.sh "--steps 20 --warmup 5" "--primary-form 0" "--primary-form 1" "--primary-form 2" > $O/ab_primary_form_tickets_c2_20.log 2>&1; cat $O/ab_primary_form_tickets_c2_20.log
timeout 300 python3 bench.py --config c3 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs --no-l1-microbench 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): print(json.loads(l)['config'].get('primary_form'))"
