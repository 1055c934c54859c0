t() { echo "$1 cfg$3: $(FYPRT_TUNING=$2 timeout -k 10 120 python tools/bench_configs.py $3 2>/dev/null | grep -o '"kernel_ms_per_frame": [0-9.]*')"; }
for rep in 1 2; do
t base "" 3
t q6_16 6=16 3
t q6_32 6=32 3
t q6_40 6=40 3
t c64 4=64 3
t c256 4=256 3
t base "" 5
t q6_32 6=32 5
t q6_40 6=40 5
done
