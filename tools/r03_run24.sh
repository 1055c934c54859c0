mkdir -p gpurun_out/r03
t() { echo "$1 cfg$3: $(FYPRT_TUNING=$2 timeout -k 10 120 python tools/bench_configs.py $3 2>/dev/null | grep -o '"kernel_ms_per_frame": [0-9.]*')"; }
for rep in 1 2; do
for c in 5 3; do
t base "" $c
t r16 5=16 $c
t r32 5=32 $c
t r40 5=40 $c
t r48 5=48 $c
t s2 9=2 $c
t r32s2 5=32,9=2 $c
done
done
