mkdir -p gpurun_out/r03
V=fypraytracer_amd/csrc/variants
t() { echo "$1: $(FYPRT_TUNING=$2 FYPRT_LIB=${3:-fypraytracer_amd/csrc/libfyprt.so} timeout -k 10 120 python tools/bench_configs.py 5 2>/dev/null | grep -o '"kernel_ms_per_frame": [0-9.]*')"; }
for rep in 1 2; do
t staged 19=0
t p5w_48 19=2,20=48
t p6w_32 19=2,20=32 $V/libfyprt_gi2w6.so
t p6w_40 19=2,20=40 $V/libfyprt_gi2w6.so
t p6w_48 19=2,20=48 $V/libfyprt_gi2w6.so
t p6w_56 19=2,20=56 $V/libfyprt_gi2w6.so
done
FYPRT_LIB=$V/libfyprt_gi2w6.so timeout -k 10 300 python -m pytest tests/test_gpu_tuning.py -m gpu -x -q -k "gi_part2" 2>&1 | tail -n 2
