mkdir -p gpurun_out/r03
t() { echo "$1 cfg$3: $(FYPRT_TUNING=$2 timeout -k 10 120 python tools/bench_configs.py $3 2>/dev/null | grep -o '"kernel_ms_per_frame": [0-9.]*')"; }
for rep in 1 2; do
for c in 5 4; do
t base "" $c
t q7_8 7=8 $c
t q7_16 7=16 $c
t q7_24 7=24 $c
t tile0 0=0 $c
t tile1 0=1 $c
done
done
