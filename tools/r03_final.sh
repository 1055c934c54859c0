# ON THE GPU BOX: the round's evidence in one call (profiles/r03/): full GPU test suite, rocprofv3 passes of the bench workload, all five configurations, band-rate projections
mkdir -p gpurun_out/r03
python -m pytest tests -x -q -m gpu > gpurun_out/r03/t_final.log 2>&1; echo "all gpu tests rc=$?"; tail -n 3 gpurun_out/r03/t_final.log
bash tools/profile_round.sh prof_r03 > gpurun_out/r03/profile_round.log 2>&1; echo "profile rc=$?"; tail -n 3 gpurun_out/r03/profile_round.log
python tools/bench_configs.py > gpurun_out/r03/configs_single_gpu.jsonl 2> gpurun_out/r03/configs.err; echo "configs rc=$?"; cut -c1-200 gpurun_out/r03/configs_single_gpu.jsonl
bash tools/profile_config5.sh cfg5_r03 > gpurun_out/r03/profile_config5.log 2>&1; echo "config5 profile rc=$?"
python tools/band_rate.py --technique 7 --mode recompute > gpurun_out/r03/band_rate_config4_di_1080p_recompute.jsonl 2>&1; tail -n 1 gpurun_out/r03/band_rate_config4_di_1080p_recompute.jsonl | cut -c1-300
python tools/band_rate.py --technique 7 --mode exchange --balance 2 > gpurun_out/r03/band_rate_config4_di_1080p_exchange_balanced.jsonl 2>&1; tail -n 1 gpurun_out/r03/band_rate_config4_di_1080p_exchange_balanced.jsonl | cut -c1-300
python tools/band_rate.py --technique 8 --width 3840 --height 2160 --frames 30 --mode recompute > gpurun_out/r03/band_rate_config5_gi_4k_recompute.jsonl 2>&1; tail -n 1 gpurun_out/r03/band_rate_config5_gi_4k_recompute.jsonl | cut -c1-300
python tools/band_rate.py --technique 8 --width 3840 --height 2160 --frames 30 --mode exchange --balance 2 > gpurun_out/r03/band_rate_config5_gi_4k_exchange_balanced.jsonl 2>&1; tail -n 1 gpurun_out/r03/band_rate_config5_gi_4k_exchange_balanced.jsonl | cut -c1-300
