mkdir -p gpurun_out/r03
V=fypraytracer_amd/csrc/variants
run() { # name lib env
  local name=$1 lib=$2; shift 2
  env "$@" python tools/perf_ab.py --key 7 --values 0 --rounds 3 ${lib:+--lib $lib} > gpurun_out/r03/ab5_$name.jsonl 2>&1
  echo "$name $(tail -n 1 gpurun_out/r03/ab5_$name.jsonl | cut -c1-230)"
}
for rep in 1 2; do
run r02_$rep $V/libfyprt_r02.so A=1
run new_$rep "" A=1
run new_preorder_$rep "" FYPRT_BVH_PREORDER=1
run notop_$rep $V/libfyprt_notop.so A=1
run notop_preorder_$rep $V/libfyprt_notop.so FYPRT_BVH_PREORDER=1
run cnd_preorder_$rep $V/libfyprt_cnd.so FYPRT_BVH_PREORDER=1
run notop_cnd_preorder_$rep $V/libfyprt_notop_cnd.so FYPRT_BVH_PREORDER=1
run notop_cnd_$rep $V/libfyprt_notop_cnd.so A=1
done
