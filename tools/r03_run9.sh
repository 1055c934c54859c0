mkdir -p gpurun_out/r03
python -m pytest tests/test_gpu_tuning.py tests/test_gpu_parity.py tests/test_gpu_counters.py tests/test_gpu_fullsize.py tests/test_golden.py tests/test_gpu_moving_camera.py -x -q -m gpu > gpurun_out/r03/t9.log 2>&1; echo "tests rc=$?"; tail -n 4 gpurun_out/r03/t9.log
for rep in 1 2; do
python tools/perf_ab.py --key 18 --values 0 1 --rounds 3 2>&1 | tail -n 2 | cut -c1-210
python tools/perf_ab.py --key 18 --values 0 1 --rounds 3 --async-frames 60 2>&1 | tail -n 2
done
python tools/band_probe.py 8 3 80 recompute 2>&1 | tail -n 1
python tools/band_probe.py 8 3 80 exchange 2>&1 | tail -n 1
