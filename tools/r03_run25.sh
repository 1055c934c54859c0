mkdir -p gpurun_out/r03
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf "$OUT/cfg5sq_$tag"
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/cfg5sq_$tag" -- python3 "$R/tools/bench_configs.py" 5 > /dev/null 2> "$OUT/cfg5sq_$tag.err" || echo "pass $tag FAILED"
  echo "$tag pass done"
done
find "$OUT" -path "*cfg5sq_*" -type f ! -name "*counter_collection.csv" ! -name "*.err" -delete
du -sh "$OUT"/cfg5sq_* | tail -6
