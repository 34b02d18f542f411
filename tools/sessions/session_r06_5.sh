# usage (GPU box): bash tools/sessions/session_r06_5.sh
# the committed evidence of the round: GPU suite, kernel trace + PMC passes of the timed configuration (c2) and of c3 / c5 /
# the image-mapped scene, the driver's bench form twice; then on the build container:
#   FSPT_PROFILE_ROUND=r06 python tools/collect_profiles.py r06/final c2 r06/c3 c3 r06/c5 c5 r06/tex c2_tex
O=gpurun_out/r06; mkdir -p $O
timeout 1500 python3 -m pytest tests -q -m gpu > $O/gpu_suite_3.log 2>&1; grep -E "passed|failed" $O/gpu_suite_3.log | tail -1
bash tools/prof_session.sh r06/final > $O/final_prof.log 2>&1; tail -1 $O/final_prof.log
bash tools/prof_session.sh r06/c3 --config c3 > $O/c3_prof.log 2>&1; tail -1 $O/c3_prof.log
bash tools/prof_session.sh r06/c5 --config c5 > $O/c5_prof.log 2>&1; tail -1 $O/c5_prof.log
bash tools/prof_session.sh r06/tex --textured > $O/tex_prof.log 2>&1; tail -1 $O/tex_prof.log
bash tools/pmc_rounds.sh r06/rounds_c2_final > /dev/null 2>&1; grep -E "kernel|trace  *[12] |logic  *[12] |primary|tail" $O/rounds_c2_final/rounds.txt
( time timeout 900 python3 bench.py --steps 20 --warmup 5 ) > $O/bench_driver_form_2.json.log 2> $O/bench_driver_form_2.time; tail -3 $O/bench_driver_form_2.time
