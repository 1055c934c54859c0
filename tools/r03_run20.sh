mkdir -p gpurun_out/r03
timeout -k 10 300 python -m pytest tests/test_gpu_tuning.py -m gpu -x -q -k "gi_part2 or range_checked" > gpurun_out/r03/t20.log 2>&1; echo "tests rc=$?"; tail -n 5 gpurun_out/r03/t20.log
t() { echo "$1: $(FYPRT_TUNING=$2 timeout -k 10 120 python tools/bench_configs.py 5 2>/dev/null | grep -o '"kernel_ms_per_frame": [0-9.]*')"; }
for rep in 1 2; do
t staged 19=0
t p2 19=2
t p1 21=1
t p1p2 19=2,21=1
t p1p2_40 19=2,21=1,20=40
t p1p2_56 19=2,21=1,20=56
done
