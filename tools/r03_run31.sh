mkdir -p gpurun_out/r03
t() { echo "$1 cfg$3: $(FYPRT_TUNING=$2 timeout -k 10 120 python tools/bench_configs.py $3 2>/dev/null | grep -o '"kernel_ms_per_frame": [0-9.]*')"; }
b() { echo "$1 bench: $(timeout -k 10 200 python bench.py --no-cpu-baseline --set $2 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')"; }
for rep in 1 2; do
t base "" 4
t q24 7=24 4
t q32 7=32 4
t q40 7=40 4
t q48 7=48 4
b base 7=0
b q24 7=24
b q32 7=32
b q40 7=40
t q32 7=32 5
t q40 7=40 5
t q32 7=32 3
t q24 7=24 3
t base "" 3
done
