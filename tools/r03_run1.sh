mkdir -p gpurun_out/r03
tools/microbench > gpurun_out/r03/microbench.jsonl 2> gpurun_out/r03/microbench.err
echo "microbench done rc=$?"
python -m pytest tests/test_gpu_parity.py tests/test_gpu_tuning.py tests/test_gpu_group.py -x -q -m gpu > gpurun_out/r03/t_parity_base.log 2>&1; echo "parity base rc=$?"
FYPRT_TOP_NODES=128 python -m pytest tests/test_gpu_parity.py tests/test_gpu_counters.py -x -q -m gpu > gpurun_out/r03/t_parity_top128.log 2>&1; echo "parity top128 rc=$?"
python tools/perf_ab.py --key 16 --values 0 64 128 256 --rounds 4 > gpurun_out/r03/ab_top_b22.jsonl 2>&1
python tools/perf_ab.py --key 16 --values 0 64 128 256 --rounds 4 --set 8=18 > gpurun_out/r03/ab_top_b18.jsonl 2>&1
python tools/perf_ab.py --key 16 --values 0 64 128 256 --rounds 4 --set 8=15 > gpurun_out/r03/ab_top_b15.jsonl 2>&1
python tools/perf_ab.py --key 16 --values 0 128 --rounds 3 --async-frames 50 > gpurun_out/r03/ab_top_async.jsonl 2>&1
tail -n 5 gpurun_out/r03/ab_top_*.jsonl
