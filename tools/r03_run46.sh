t() { echo "$1 cfg$3: $(FYPRT_TUNING=$2 timeout -k 10 120 python tools/bench_configs.py $3 2>/dev/null | grep -o '"kernel_ms_per_frame": [0-9.]*')"; }
for rep in 1 2; do
t base "" 3
t g5 2=5 3
t g4 2=4 3
t base "" 5
t g5 2=5 5
done
