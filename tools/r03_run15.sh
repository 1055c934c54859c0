mkdir -p gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_gpu_tuning.py -m gpu -x -q -k "gi_part2" > gpurun_out/r03/t15.log 2>&1; echo "tests rc=$?"; tail -n 5 gpurun_out/r03/t15.log
for rep in 1 2; do
python tools/bench_configs.py 5 2>/dev/null | cut -c1-420
FYPRT_TUNING=19=1 python tools/bench_configs.py 5 2>/dev/null | cut -c1-420
done
