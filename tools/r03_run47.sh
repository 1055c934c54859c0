V=fypraytracer_amd/csrc/variants
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_counters.py tests/test_gpu_tuning.py tests/test_gpu_edge_scenes.py -m gpu -x -q 2>&1 | tail -n 2
b() { echo "$1 bench: $(FYPRT_LIB=${2:-fypraytracer_amd/csrc/libfyprt.so} timeout -k 10 200 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']; print(d['ms_per_step'], k['k_di_part1']['avg_ms_alone'], k['k_di_part2_trace']['avg_ms_alone'])")"; }
t() { echo "$1 cfg$3: $(FYPRT_LIB=${2:-fypraytracer_amd/csrc/libfyprt.so} timeout -k 10 120 python tools/bench_configs.py $3 2>/dev/null | grep -o '"kernel_ms_per_frame": [0-9.]*')"; }
for rep in 1 2 3; do
b base $V/libfyprt_notos.so
b tos ""
done
for rep in 1 2; do
for c in 1 2 3 5; do
t base $V/libfyprt_notos.so $c
t tos "" $c
done
done
