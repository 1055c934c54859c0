mkdir -p gpurun_out/r03
V=fypraytracer_amd/csrc/variants
t() { echo "$1: $(FYPRT_TUNING=$2 FYPRT_LIB=${3:-fypraytracer_amd/csrc/libfyprt.so} timeout -k 10 120 python tools/bench_configs.py 5 2>/dev/null | grep -o '"kernel_ms_per_frame": [0-9.]*')"; }
for rep in 1 2; do
t staged 19=0
t p5w_32 19=2
t p5w_24 19=2,20=24
t p5w_40 19=2,20=40
t p5w_48 19=2,20=48
t p5w_56 19=2,20=56
t p4w_32 19=2,5=32 $V/libfyprt_gi2w4.so
t p5w_32_occ4 19=2,2=4
t p5w_32_occ3 19=2,2=3
done
