# usage (GPU box): bash tools/sessions/session_r06_9.sh
# TIMING EXPERIMENT, results discarded (the images are wrong): what the logic launches would cost if the environment lookups
# of the blocks on one XCD fell into 1/2, 1/4, 1/8, 1/32 of the map (FSPT_EXP_ENV_SECTOR folds every lookup's column into the
# first 1/n of the columns) - the upper bound of what sorting the survivors GLOBALLY by the azimuth of their new direction
# (each XCD one sector of the map, its lines resident in that XCD's L2) could win for k_wf_logic
O=gpurun_out/r06; mkdir -p $O
for rep in 1 2; do for n in t_base e_sec2 e_sec4 e_sec8 e_sec32; do
echo -n "== $n (rep $rep): "
FSPT_LIB=$PWD/ab_libs/$n.so timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs --no-l1-microbench --no-parity-check 2>/dev/null | python3 -c "
import sys,json
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); k=d['roofline'].get('kernels',{})
        print(d['value'], d['ms_per_step'], {c:v['ms_per_step'] for c,v in k.items()})
"
done; done > $O/exp_env_sector_c2_20.log 2>&1; cat $O/exp_env_sector_c2_20.log
FSPT_LIB=$PWD/ab_libs/e_sec8.so bash tools/pmc_rounds.sh r06/rounds_c2_env_sector8 --no-parity-check > /dev/null 2>&1; grep -E "kernel|logic  *[12] " $O/rounds_c2_env_sector8/rounds.txt
# ... and the two constants session 8 found worth a second look (1 M triangles: +2.5 % each), alone and together, on every form
bash tools/ab_run.sh "--steps 20 --warmup 5 --config c3" t_base t_im24 t_im32 t_im20 t_ts16 t_ts8 t_ts24 t_im24ts16 > $O/scan_constants2_c3.log 2>&1; cat $O/scan_constants2_c3.log
bash tools/ab_run.sh "--steps 20 --warmup 5" t_base t_im24 t_ts16 t_im24ts16 > $O/scan_constants2_c2_20.log 2>&1; cat $O/scan_constants2_c2_20.log
bash tools/ab_run.sh "--steps 1 --warmup 3 --reps 15" t_base t_im24 t_ts16 t_im24ts16 > $O/scan_constants2_tick1.log 2>&1; cat $O/scan_constants2_tick1.log
bash tools/ab_run.sh "--steps 128 --warmup 128 --reps 3" t_base t_im24ts16 > $O/scan_constants2_c2_128.log 2>&1; cat $O/scan_constants2_c2_128.log
bash tools/ab_run.sh "--steps 20 --warmup 5 --textured" t_base t_im24ts16 > $O/scan_constants2_tex.log 2>&1; cat $O/scan_constants2_tex.log
