b() { echo "$1 bench: $(timeout -k 10 200 python bench.py --no-cpu-baseline --set $2 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')"; }
t() { echo "$1 cfg$3: $(FYPRT_TUNING=$2 timeout -k 10 120 python tools/bench_configs.py $3 2>/dev/null | grep -o '"kernel_ms_per_frame": [0-9.]*')"; }
for rep in 1 2 3; do
b q7_32 7=32
b q7_48 7=48
b q7_56 7=56
b q7_64 7=64
b q7_48_q6_32 "7=48 6=32"
done
for rep in 1 2; do
for c in 1 2 3 5; do
t q7_32 "" $c
t q7_48 7=48 $c
t q7_64 7=64 $c
done
done
