mkdir -p gpurun_out/r03
timeout -k 10 300 python -m pytest tests/test_gpu_tuning.py -m gpu -x -q -k "gi_part2" > gpurun_out/r03/t17.log 2>&1; echo "tests rc=$?"; tail -n 5 gpurun_out/r03/t17.log
for rep in 1 2; do
python tools/bench_configs.py 5 2>/dev/null | cut -c200-330
FYPRT_TUNING=19=2 timeout -k 10 120 python tools/bench_configs.py 5 2>/dev/null | cut -c200-330
FYPRT_TUNING=19=2,5=16 timeout -k 10 120 python tools/bench_configs.py 5 2>/dev/null | cut -c200-330
FYPRT_TUNING=19=2,5=32 timeout -k 10 120 python tools/bench_configs.py 5 2>/dev/null | cut -c200-330
done
FYPRT_TUNING=19=2 timeout -k 10 300 python tools/band_rate.py --technique 8 --width 3840 --height 2160 --frames 30 --mode recompute > gpurun_out/r03/band_rate_config5_gi_4k_recompute_persistent_part2.jsonl 2>&1; grep '"speedup_vs_1"' gpurun_out/r03/band_rate_config5_gi_4k_recompute_persistent_part2.jsonl | cut -c1-260
