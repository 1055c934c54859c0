#!/usr/bin/env bash
# ON THE GPU BOX: kernel trace of one config + idle-gap analysis.  usage: bash tools/gap_run.sh <config number>
set -euo pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT/gap_$1" -- python3 "$R/tools/bench_configs.py" "$1" > /dev/null 2>&1
python3 "$R/tools/gap_analysis.py" "$(find "$OUT/gap_$1" -name '*kernel_trace.csv' | head -1)"
rm -rf "$OUT/gap_$1"
