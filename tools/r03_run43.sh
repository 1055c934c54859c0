t() { echo "$1 cfg$3: $(FYPRT_TUNING=$2 timeout -k 10 120 python tools/bench_configs.py $3 2>/dev/null | grep -o '"kernel_ms_per_frame": [0-9.]*')"; }
for rep in 1 2; do
t base "" 5
t s40 20=40 5
t s56 20=56 5
t q32 6=32 5
t q40 6=40 5
t q32s40 6=32,20=40 5
t q16 6=16 5
done
