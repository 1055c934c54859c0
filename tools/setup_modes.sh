#!/usr/bin/env bash
# Runs ON THE GPU BOX: the three variants of the ReSTIR DI Part-2 setup kernel (tuning key 14: 0 dependent gathers, 1 speculative gathers,
# 2 neighbourhood hot fields in LDS) on the bench frame: stand-alone duration (frames not pipelined) and HBM / L2 counters, each in its own pass.
#   usage: bash tools/setup_modes.sh   ->  gpurun_out/setup_modes.csv  (mode, counter, mean per dispatch of k_di_part2_setup)
set -euo pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
: > "$OUT/setup_modes.csv"
for mode in 0 1 2; do
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY"; do
    tag=$(echo $set | cut -d' ' -f1)
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/sm_${mode}_$tag" -- python3 "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --set 11=0 14=$mode > /dev/null 2> "$OUT/sm_${mode}_$tag.err" || echo "pass $mode $tag failed"
    python3 - "$(find "$OUT/sm_${mode}_$tag" -name '*counter_collection.csv' | head -1)" "$mode" >> "$OUT/setup_modes.csv" <<'PY'
import collections, csv, sys
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "k_di_part2_setup" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        start_end = None
for c, v in sorted(agg.items()):
    v = v[2:] if len(v) > 4 else v
    print(f"{sys.argv[2]},{c},{sum(v) / len(v):.0f},{len(v)}")
PY
    rm -rf "$OUT/sm_${mode}_$tag" "$OUT/sm_${mode}_$tag.err"
  done
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/sm_${mode}_stats" -- python3 "$R/bench.py" --steps 30 --warmup 5 --no-cpu-baseline --set 11=0 14=$mode > /dev/null 2> /dev/null
  python3 - "$(find "$OUT/sm_${mode}_stats" -name '*kernel_stats.csv' | head -1)" "$mode" >> "$OUT/setup_modes.csv" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_di_part2_setup" in r["Name"]:
        print(f"{sys.argv[2]},avg_duration_ns,{float(r['AverageNs']):.0f},{r['Calls']}")
PY
  rm -rf "$OUT/sm_${mode}_stats"
done
cat "$OUT/setup_modes.csv"
