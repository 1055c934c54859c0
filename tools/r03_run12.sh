V=fypraytracer_amd/csrc/variants
for rep in 1 2 3; do
python tools/perf_ab.py --key 7 --values 0 --rounds 3 2>&1 | tail -n 1 | cut -c1-200
python tools/perf_ab.py --key 7 --values 0 --rounds 3 --lib $V/libfyprt_greedy.so 2>&1 | tail -n 1 | cut -c1-200
done
python tools/perf_ab.py --key 7 --values 0 --rounds 3 --async-frames 60 2>&1 | tail -n 1
python tools/perf_ab.py --key 7 --values 0 --rounds 3 --async-frames 60 --lib $V/libfyprt_greedy.so 2>&1 | tail -n 1
python tools/perf_ab.py --key 7 --values 0 --rounds 3 --async-frames 60 2>&1 | tail -n 1
python tools/perf_ab.py --key 7 --values 0 --rounds 3 --async-frames 60 --lib $V/libfyprt_greedy.so 2>&1 | tail -n 1
FYPRT_LIB=$V/libfyprt_greedy.so python -m pytest tests/test_gpu_parity.py tests/test_gpu_counters.py -x -q -m gpu 2>&1 | tail -n 2
python tools/bench_configs.py 3 5 | cut -c1-330
