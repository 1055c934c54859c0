V=fypraytracer_amd/csrc/variants
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_counters.py tests/test_gpu_tuning.py -m gpu -x -q 2>&1 | tail -n 2
b() { echo "$1 bench: $(FYPRT_LIB=${3:-fypraytracer_amd/csrc/libfyprt.so} timeout -k 10 200 python bench.py --no-cpu-baseline --set $2 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')"; }
t() { echo "$1 cfg$3: $(FYPRT_LIB=${4:-fypraytracer_amd/csrc/libfyprt.so} FYPRT_TUNING=$2 timeout -k 10 120 python tools/bench_configs.py $3 2>/dev/null | grep -o '"kernel_ms_per_frame": [0-9.]*')"; }
for rep in 1 2; do
b base 7=32 $V/libfyprt_base5.so
b rel 7=32
b rel_q6_32 6=32
b rel_q6_40 6=40
b rel_q7_40 7=40
b rel_q7_48 7=48
b rel_q6_32_q7_48 "6=32 7=48"
for c in 1 2 3 5; do
t base "" $c $V/libfyprt_base5.so
t rel "" $c
done
t rel_q6_32 6=32 3
t rel_q6_32 6=32 5
done
