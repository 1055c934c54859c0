mkdir -p gpurun_out/r03
for i in 1 2 3; do
python tools/perf_ab.py --key 7 --values 0 --rounds 3 --lib fypraytracer_amd/csrc/variants/libfyprt_r02.so > gpurun_out/r03/ab4_r02_$i.jsonl 2>&1
python tools/perf_ab.py --key 16 --values 0 64 --rounds 3 > gpurun_out/r03/ab4_new_$i.jsonl 2>&1
done
python tools/perf_ab.py --key 7 --values 0 --rounds 3 --async-frames 50 --lib fypraytracer_amd/csrc/variants/libfyprt_r02.so > gpurun_out/r03/ab4_async_r02.jsonl 2>&1
python tools/perf_ab.py --key 16 --values 0 64 --rounds 3 --async-frames 50 > gpurun_out/r03/ab4_async_new.jsonl 2>&1
tail -n 2 gpurun_out/r03/ab4_*.jsonl | cut -c1-260
FYPRT_LIB=fypraytracer_amd/csrc/variants/libfyprt_r02.so python tools/bench_configs.py 3 5 > gpurun_out/r03/cfg35_r02.jsonl 2>&1
python tools/bench_configs.py 3 5 > gpurun_out/r03/cfg35_new.jsonl 2>&1
cat gpurun_out/r03/cfg35_r02.jsonl gpurun_out/r03/cfg35_new.jsonl | cut -c1-330
python bench.py > gpurun_out/r03/bench_mid.json 2> gpurun_out/r03/bench_mid.err; echo "bench rc=$?"; cut -c1-600 gpurun_out/r03/bench_mid.json
python -m pytest tests -x -q -m gpu > gpurun_out/r03/t4_all.log 2>&1; echo "all gpu tests rc=$?"; tail -n 5 gpurun_out/r03/t4_all.log
