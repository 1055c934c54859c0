timeout -k 10 300 python -m pytest tests/test_gpu_tuning.py -m gpu -x -q -k "deferred" 2>&1 | tail -n 15
t() { echo "$1: $(FYPRT_TUNING=$2 timeout -k 10 120 python tools/bench_configs.py 5 2>/dev/null | grep -o '"kernel_ms_per_frame": [0-9.]*\|"wall_ms_per_frame": [0-9.]*' | tr '\n' ' ')"; }
t base ""
t defer1 21=1
t pipe2 21=2
for tn in "" "21=2"; do
FYPRT_TUNING=$tn timeout -k 10 300 python tools/band_rate.py --technique 8 --width 3840 --height 2160 --frames 30 --mode recompute --n 1 8 2>&1 | grep '"speedup_vs_1"' | cut -c1-200
done
