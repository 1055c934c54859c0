#!/usr/bin/env bash
# Runs ON THE GPU BOX: rocprofv3 kernel stats of tools/bench_configs.py for the given configs.
#   usage: bash tools/prof_configs.sh <tag> <config numbers...>   ->  gpurun_out/<tag>_stats.csv (per-kernel totals) + <tag>.jsonl
set -euo pipefail
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_prof" -- python3 "$R/tools/bench_configs.py" "$@" > "$OUT/${TAG}.jsonl" 2> "$OUT/${TAG}.err"
cp "$(find "$OUT/${TAG}_prof" -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_stats.csv"
rm -rf "$OUT/${TAG}_prof"
cut -c1-150 "$OUT/${TAG}_stats.csv" | head -20
