b() { echo "$1 bench: $(timeout -k 10 200 python bench.py --no-cpu-baseline --set $2 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')"; }
t() { echo "$1 cfg$3: $(FYPRT_TUNING=$2 timeout -k 10 120 python tools/bench_configs.py $3 2>/dev/null | grep -o '"kernel_ms_per_frame": [0-9.]*')"; }
for rep in 1 2; do
b base 4=128
b c64 4=64
b c256 4=256
b m16 10=16
b m64 10=64
b s2 9=2
b sort 3=1
t base "" 3
t m16 10=16 3
t m64 10=64 3
t s3 9=3 3
done
