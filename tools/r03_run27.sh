mkdir -p gpurun_out/r03
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf "$OUT/cfg3_r03_stats"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/cfg3_r03_stats" -- python3 "$R/tools/bench_configs.py" 3 > "$OUT/cfg3_r03_config3.jsonl" 2> "$OUT/cfg3_r03_stats.err"
find "$OUT" -path "*cfg3_r03_*" -type f ! -name "*kernel_stats.csv" ! -name "*.jsonl" ! -name "*.err" -delete
cat "$OUT"/cfg3_r03_stats/*/*kernel_stats.csv | cut -c1-200
