mkdir -p gpurun_out/r03
python -m pytest tests/test_gpu_tuning.py tests/test_gpu_parity.py tests/test_gpu_counters.py tests/test_gpu_fullsize.py tests/test_gpu_group.py tests/test_gpu_edge_scenes.py -x -q -m gpu > gpurun_out/r03/t11.log 2>&1; echo "tests rc=$?"; tail -n 4 gpurun_out/r03/t11.log
FYPRT_TUNING=18=0 python tools/bench_configs.py 3 | cut -c1-400
python tools/bench_configs.py 3 | cut -c1-400
FYPRT_TUNING=18=0 python tools/bench_configs.py 3 | cut -c1-400
python tools/bench_configs.py 3 | cut -c1-400
