mkdir -p gpurun_out/r03
V=fypraytracer_amd/csrc/variants
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_counters.py tests/test_gpu_refit.py tests/test_gpu_lbvh.py tests/test_gpu_tuning.py tests/test_gpu_edge_scenes.py tests/test_gpu_transforms.py -m gpu -x -q > gpurun_out/r03/t14.log 2>&1; echo "tests rc=$?"; tail -n 3 gpurun_out/r03/t14.log
run() { # name lib env
  local name=$1 lib=$2; shift 2
  env "$@" python tools/perf_ab.py --key 7 --values 0 --rounds 3 ${lib:+--lib $lib} > gpurun_out/r03/ab14_$name.jsonl 2>&1
  echo "$name $(tail -n 1 gpurun_out/r03/ab14_$name.jsonl | cut -c1-260)"
}
for rep in 1 2 3; do
run base_$rep $V/libfyprt_base2.so A=1
run new_$rep "" A=1
done
