V=fypraytracer_amd/csrc/variants
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_counters.py tests/test_gpu_tuning.py tests/test_gpu_edge_scenes.py tests/test_gpu_fullsize.py tests/test_gpu_convergence.py -m gpu -x -q 2>&1 | tail -n 2
t() { echo "$1 cfg$3: $(FYPRT_LIB=${2:-fypraytracer_amd/csrc/libfyprt.so} timeout -k 10 120 python tools/bench_configs.py $3 2>/dev/null | grep -o '"kernel_ms_per_frame": [0-9.]*')"; }
for rep in 1 2 3; do
t base $V/libfyprt_base6.so 3
t flat "" 3
done
