V=fypraytracer_amd/csrc/variants
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "NEE or nee or LIGHT or light" 2>&1 | tail -n 2
t() { echo "$1 cfg$3: $(FYPRT_LIB=${2:-fypraytracer_amd/csrc/libfyprt.so} timeout -k 10 120 python tools/bench_configs.py $3 2>/dev/null | grep -o '"kernel_ms_per_frame": [0-9.]*')"; }
for rep in 1 2 3; do
t base $V/libfyprt_base6.so 3
t flat5w "" 3
done
