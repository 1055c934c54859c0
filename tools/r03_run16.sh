mkdir -p gpurun_out/r03
timeout -k 10 300 tools/microbench chase > gpurun_out/r03/microbench_chase.jsonl 2> gpurun_out/r03/microbench_chase.err; echo "chase rc=$?"; grep -c chase gpurun_out/r03/microbench_chase.jsonl
FYPRT_TUNING=19=1 python tools/band_rate.py --technique 8 --width 3840 --height 2160 --frames 30 --mode recompute > gpurun_out/r03/band_rate_config5_gi_4k_recompute_fused_part2.jsonl 2>&1; grep '"speedup_vs_1"' gpurun_out/r03/band_rate_config5_gi_4k_recompute_fused_part2.jsonl | cut -c1-260
python tools/band_rate.py --technique 8 --width 3840 --height 2160 --frames 30 --mode recompute > gpurun_out/r03/band_rate_config5_gi_4k_recompute_b.jsonl 2>&1; grep '"speedup_vs_1"' gpurun_out/r03/band_rate_config5_gi_4k_recompute_b.jsonl | cut -c1-260
