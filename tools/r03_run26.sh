mkdir -p gpurun_out/r03
V=fypraytracer_amd/csrc/variants
timeout -k 10 300 python -m pytest tests/test_gpu_tuning.py -m gpu -x -q -k "gi_part2" 2>&1 | tail -n 2
t() { echo "$1: $(FYPRT_TUNING=$2 FYPRT_LIB=${3:-fypraytracer_amd/csrc/libfyprt.so} timeout -k 10 120 python tools/bench_configs.py 5 2>/dev/null | grep -o '"kernel_ms_per_frame": [0-9.]*')"; }
for rep in 1 2; do
t base_48 "" $V/libfyprt_base3.so
t new_48 20=48
t new_40 20=40
t new_32 20=32
t new_24 20=24
t new_16 20=16
done
