#!/usr/bin/env bash
# Runs ON THE GPU BOX: a few more counter passes for the final kernels, frames NOT pipelined so that every dispatch runs alone.
set -euo pipefail
P=$1
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for set in "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" "SQ_WAIT_ANY SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/${P}_$tag" -- python3 "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --set 11=0 > /dev/null 2> "$OUT/${P}_$tag.err" || echo "pass $tag failed"
  echo "$tag done"
done
find "$OUT" -path "*${P}_*" -type f ! -name "*counter_collection.csv" ! -name "*.err" -delete
