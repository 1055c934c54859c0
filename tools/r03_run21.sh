mkdir -p gpurun_out/r03
for tn in "19=2" "19=2,21=1"; do
FYPRT_TUNING=$tn timeout -k 10 300 python tools/band_rate.py --technique 8 --width 3840 --height 2160 --frames 30 --mode recompute --n 1 8 > gpurun_out/r03/band_gi_$tn.jsonl 2>&1; echo "$tn"; grep '"speedup_vs_1"' gpurun_out/r03/band_gi_$tn.jsonl | cut -c1-260
done
t() { echo "$1: $(FYPRT_TUNING=$2 timeout -k 10 120 python tools/bench_configs.py 5 2>/dev/null | grep -o '"kernel_ms_per_frame": [0-9.]*')"; }
for rep in 1 2; do
t p2 19=2
t p2_q16 19=2,6=16
t p2_q32 19=2,6=32
t p2_q0 19=2,6=0
t p2_c64 19=2,4=64
t p2_c256 19=2,4=256
t p2_s2 19=2,9=2
done
