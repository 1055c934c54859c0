mkdir -p gpurun_out/r03
V=fypraytracer_amd/csrc/variants
for rep in 1 2; do
python tools/perf_ab.py --key 7 --values 0 --rounds 3 --lib $V/libfyprt_notop.so > gpurun_out/r03/ab6_notop_$rep.jsonl 2>&1; echo "notop(before restructure) $(tail -n 1 gpurun_out/r03/ab6_notop_$rep.jsonl | cut -c1-200)"
python tools/perf_ab.py --key 7 --values 0 --rounds 3 > gpurun_out/r03/ab6_new_$rep.jsonl 2>&1; echo "new $(tail -n 1 gpurun_out/r03/ab6_new_$rep.jsonl | cut -c1-200)"
done
python tools/perf_ab.py --key 7 --values 0 --rounds 3 --async-frames 50 --lib $V/libfyprt_notop.so | tail -n 1
python tools/perf_ab.py --key 7 --values 0 --rounds 3 --async-frames 50 | tail -n 1
python -m pytest tests/test_gpu_parity.py tests/test_gpu_counters.py tests/test_gpu_tuning.py -x -q -m gpu 2>&1 | tail -n 2
python tools/band_probe.py 8 3 60 recompute
python tools/band_probe.py 8 3 60 exchange
python tools/band_probe.py 8 0 60 recompute
python tools/band_probe.py 2 0 60 recompute
bash tools/timeline_run.sh k_di_part1 /root/repo/tools/band_probe.py 8 3 40 recompute > gpurun_out/r03/band8_timeline.txt 2>&1; cat gpurun_out/r03/band8_timeline.txt | tail -n 12
bash tools/timeline_run.sh k_di_part1 /root/repo/tools/band_probe.py 8 3 40 exchange > gpurun_out/r03/band8x_timeline.txt 2>&1; cat gpurun_out/r03/band8x_timeline.txt | tail -n 12
