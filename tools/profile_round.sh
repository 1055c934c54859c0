#!/usr/bin/env bash
# Runs ON THE GPU BOX (through gpurun): the rocprofv3 passes behind profiles/<round>/ for the bench workload.
#   usage: bash tools/profile_round.sh <prefix>     ->  gpurun_out/<prefix>_{stats,FETCH_SIZE,...}/ + gpurun_out/<prefix>_bench.json
# Then, back in the container:  python tools/summarize_profile.py <prefix> <round, e.g. r02>
# Counters are collected in their own passes with --kernel-trace only (MI355X_MICROARCH.md §HBM, §rocprofv3 PMC slots: FETCH_SIZE and
# WRITE_SIZE do not fit one pass; at most two SQ counters per pass here), frames NOT pipelined (--set 11=0) so that every dispatch
# runs alone; the program itself follows `--`.  The --stats pass runs the default (pipelined) bench: its timed region is what
# BENCH_rNN.json measures.
set -euo pipefail
P=$1
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${P}_stats" -- python3 "$R/bench.py" --steps 50 --warmup 10 --no-cpu-baseline > "$OUT/${P}_bench.json" 2> "$OUT/${P}_stats.err"
echo "stats pass done"
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE SQ_WAVES" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/${P}_$tag" -- python3 "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --set 11=0 > /dev/null 2> "$OUT/${P}_$tag.err" || echo "pass $tag FAILED"
  echo "$tag pass done"
done
# keep only what the summary needs (kernel stats / trace / counter csv): the merge back is capped at 64 MiB
find "$OUT" -path "*${P}_*" -type f ! -name "*kernel_stats.csv" ! -name "*kernel_trace.csv" ! -name "*counter_collection.csv" ! -name "*.json" ! -name "*.err" -delete
du -sh "$OUT"/${P}_* | tail -12
