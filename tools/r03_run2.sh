mkdir -p gpurun_out/r03
python -m pytest tests/test_gpu_parity.py tests/test_gpu_counters.py tests/test_gpu_tuning.py -x -q -m gpu > gpurun_out/r03/t2_parity.log 2>&1; echo "parity rc=$?"
for i in 1 2; do
python tools/perf_ab.py --key 16 --values 0 --rounds 3 --lib fypraytracer_amd/csrc/variants/libfyprt_r02.so --key 7 > gpurun_out/r03/ab2_r02_$i.jsonl 2>&1
python tools/perf_ab.py --key 16 --values 0 64 --rounds 3 > gpurun_out/r03/ab2_new_$i.jsonl 2>&1
done
python tools/perf_ab.py --key 7 --values 0 --rounds 3 --async-frames 50 --lib fypraytracer_amd/csrc/variants/libfyprt_r02.so > gpurun_out/r03/ab2_async_r02.jsonl 2>&1
python tools/perf_ab.py --key 16 --values 0 64 --rounds 3 --async-frames 50 > gpurun_out/r03/ab2_async_new.jsonl 2>&1
tail -n 3 gpurun_out/r03/ab2_*.jsonl
python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_group.py -x -q -m gpu > gpurun_out/r03/t2_fullsize.log 2>&1; echo "fullsize+group rc=$?"
tail -n 5 gpurun_out/r03/t2_fullsize.log
cat gpurun_out/r03/reference_order_fullsize.jsonl
