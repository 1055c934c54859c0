#!/usr/bin/env bash
# Runs ON THE GPU BOX: SQ counter passes (own runs, --kernel-trace only) of tools/bench_configs.py for the given configs.
#   usage: bash tools/pmc_configs.sh <tag> <config numbers...>   ->  gpurun_out/<tag>_pmc.csv  (kernel, counter, mean per dispatch, dispatches)
set -euo pipefail
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
: > "$OUT/${TAG}_pmc.csv"
# counter sets, one pass each; override with PMC_SETS="A B;C;D E" (';' separates passes)
IFS=';' read -r -a SETS <<< "${PMC_SETS:-SQ_INSTS_VALU SQ_ACTIVE_INST_VALU;SQ_WAVE_CYCLES SQ_WAIT_ANY;SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU;GRBM_GUI_ACTIVE SQ_WAVES}"
for set in "${SETS[@]}"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/${TAG}_$tag" -- python3 "$R/tools/bench_configs.py" "$@" > /dev/null 2> "$OUT/${TAG}_$tag.err" || echo "pass $tag failed"
  f=$(find "$OUT/${TAG}_$tag" -name '*counter_collection.csv' | head -1)
  python3 - "$f" >> "$OUT/${TAG}_pmc.csv" <<'PY'
import collections, csv, sys
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Kernel_Name"].startswith("void rt::k_") or r["Kernel_Name"].startswith("rt::k_"):
        agg[(r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(agg.items()):
    print(f"{k},{c},{sum(v) / len(v):.0f},{len(v)}")
PY
  rm -rf "$OUT/${TAG}_$tag"
done
cat "$OUT/${TAG}_pmc.csv"
