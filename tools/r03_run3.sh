mkdir -p gpurun_out/r03
python -m pytest tests/test_gpu_tuning.py tests/test_gpu_parity.py tests/test_gpu_edge_scenes.py -x -q -m gpu > gpurun_out/r03/t3_fused.log 2>&1; echo "fused tests rc=$?"; tail -n 3 gpurun_out/r03/t3_fused.log
for i in 1 2 3; do
python tools/perf_ab.py --key 7 --values 0 --rounds 3 --lib fypraytracer_amd/csrc/variants/libfyprt_r02.so > gpurun_out/r03/ab3_r02_$i.jsonl 2>&1
python tools/perf_ab.py --key 16 --values 0 64 --rounds 3 > gpurun_out/r03/ab3_new_$i.jsonl 2>&1
done
python tools/perf_ab.py --key 7 --values 0 --rounds 3 --async-frames 50 --lib fypraytracer_amd/csrc/variants/libfyprt_r02.so > gpurun_out/r03/ab3_async_r02.jsonl 2>&1
python tools/perf_ab.py --key 16 --values 0 64 --rounds 3 --async-frames 50 > gpurun_out/r03/ab3_async_new.jsonl 2>&1
tail -n 3 gpurun_out/r03/ab3_*.jsonl
FYPRT_TUNING=17=1 python tools/bench_configs.py 1 2 > gpurun_out/r03/cfg12_stages.jsonl 2>&1
python tools/bench_configs.py 1 2 > gpurun_out/r03/cfg12_fused.jsonl 2>&1
FYPRT_LIB=fypraytracer_amd/csrc/variants/libfyprt_r02.so python tools/bench_configs.py 1 2 3 > gpurun_out/r03/cfg123_r02.jsonl 2>&1
python tools/bench_configs.py 3 > gpurun_out/r03/cfg3_new.jsonl 2>&1
cat gpurun_out/r03/cfg12_stages.jsonl gpurun_out/r03/cfg12_fused.jsonl gpurun_out/r03/cfg123_r02.jsonl gpurun_out/r03/cfg3_new.jsonl | cut -c1-330
