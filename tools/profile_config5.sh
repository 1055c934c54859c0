#!/usr/bin/env bash
# Runs ON THE GPU BOX: per-kernel times and HBM traffic of BASELINE config 5 on one GPU (4K ReSTIR GI, hall 1 M): one --stats pass and the
# FETCH_SIZE / WRITE_SIZE passes (separate, --kernel-trace only, the program itself after `--`) of tools/bench_configs.py 5.
#   usage: bash tools/profile_config5.sh <prefix> [FYPRT_TUNING value]   ->  gpurun_out/<prefix>_{stats,FETCH_SIZE,WRITE_SIZE}/
# Then, in the container:  python tools/summarize_config5.py <prefix> <round>
set -euo pipefail
P=$1
export FYPRT_TUNING=${2:-}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p "$OUT"
rm -rf "$OUT/${P}_stats" "$OUT/${P}_FETCH_SIZE" "$OUT/${P}_WRITE_SIZE"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${P}_stats" -- python3 "$R/tools/bench_configs.py" 5 > "$OUT/${P}_config5.jsonl" 2> "$OUT/${P}_stats.err"
echo "stats pass done"
for tag in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $tag --kernel-trace --output-format csv -d "$OUT/${P}_$tag" -- python3 "$R/tools/bench_configs.py" 5 > /dev/null 2> "$OUT/${P}_$tag.err" || echo "pass $tag FAILED"
  echo "$tag pass done"
done
find "$OUT" -path "*${P}_*" -type f ! -name "*kernel_stats.csv" ! -name "*counter_collection.csv" ! -name "*.jsonl" ! -name "*.err" -delete
du -sh "$OUT"/${P}_* | tail -6
