# flakiness check: the whole GPU suite five times over on one box
O=gpurun_out/r04; mkdir -p $O
for i in 1 2 3 4 5; do timeout 900 python3 -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -2; done > $O/s21_gpu_suite_x5.log 2>&1
cat $O/s21_gpu_suite_x5.log
