t() { echo "$1 cfg$3: $(FYPRT_TUNING=$2 timeout -k 10 120 python tools/bench_configs.py $3 2>/dev/null | grep -o '"kernel_ms_per_frame": [0-9.]*')"; }
for rep in 1 2; do
for c in 1 2; do
t q0 7=0 $c
t q32 "" $c
t q16 7=16 $c
done
done
