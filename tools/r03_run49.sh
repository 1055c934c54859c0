R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_WAIT_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf "$OUT/cfg3sq_$tag"
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/cfg3sq_$tag" -- python3 "$R/tools/bench_configs.py" 3 > /dev/null 2> "$OUT/cfg3sq_$tag.err" || echo "pass $tag FAILED"
  echo "$tag pass done"
done
find "$OUT" -path "*cfg3sq_*" -type f ! -name "*counter_collection.csv" ! -name "*.err" -delete
