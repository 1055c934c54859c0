#!/usr/bin/env bash
# ON THE GPU BOX: bash tools/timeline_run.sh <python script + args...> -- <first kernel name>
set -euo pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; cd /tmp && export TMPDIR=/tmp
FIRST=$1; shift
rocprofv3 --kernel-trace --output-format csv -d "$OUT/tl" -- python3 "$@" > /dev/null 2>&1
python3 "$R/tools/frame_timeline.py" "$(find "$OUT/tl" -name '*kernel_trace.csv' | head -1)" "$FIRST"
rm -rf "$OUT/tl"
