mkdir -p gpurun_out/r03
V=fypraytracer_amd/csrc/variants
t() { echo "$1 cfg$3: $(FYPRT_LIB=${2:-fypraytracer_amd/csrc/libfyprt.so} timeout -k 10 120 python tools/bench_configs.py $3 2>/dev/null | grep -o '"kernel_ms_per_frame": [0-9.]*')"; }
for rep in 1 2; do
for c in 5 3; do
t base "" $c
t shade5 $V/libfyprt_shade5.so $c
t shade6 $V/libfyprt_shade6.so $c
done
done
