#!/usr/bin/env bash
# Runs ON THE GPU BOX (through gpurun): the four rocprofv3 passes behind profiles/<round>/ for the bench workload.
#   usage: bash tools/profile_round.sh <prefix>      e.g. prof_r01f  ->  gpurun_out/<prefix>_{stats,FETCH_SI,WRITE_SI,TCC_HIT_}/ + gpurun_out/<prefix>_bench.json
# Then, back in the container:  python tools/summarize_profile.py <prefix> r01 <prefix>_bench.json
# Counters are collected in their own passes with --kernel-trace only (MI355X_MICROARCH.md §HBM); the program itself follows `--`.
set -euo pipefail
P=$1
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${P}_stats" -- python3 "$R/bench.py" --steps 50 --warmup 10 --no-cpu-baseline > "$OUT/${P}_bench.json" 2> "$OUT/${P}_stats.err"
echo "stats pass done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/${P}_FETCH_SI" -- python3 "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> "$OUT/${P}_fetch.err"
echo "FETCH_SIZE pass done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/${P}_WRITE_SI" -- python3 "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> "$OUT/${P}_write.err"
echo "WRITE_SIZE pass done"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d "$OUT/${P}_TCC_HIT_" -- python3 "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> "$OUT/${P}_tcc.err"
echo "TCC pass done"
# keep only what the summary needs (kernel stats / trace / counter csv): the merge back is capped at 64 MiB
find "$OUT" -path "*${P}_*" -type f ! -name "*kernel_stats.csv" ! -name "*kernel_trace.csv" ! -name "*counter_collection.csv" ! -name "*.json" ! -name "*.err" -delete
du -sh "$OUT"/${P}_* | tail -8
