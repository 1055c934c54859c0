#!/usr/bin/env python3
"""Turn the raw rocprofv3 output of a profiled bench run (gpurun_out/<prefix>_{stats,FETCH_SI,WRITE_SI,TCC_HIT_}) into the
tracked summaries under profiles/<round>/ and profiles/traffic.json.
  usage: summarize_profile.py <prefix e.g. prof_r01d> <round dir e.g. r01> <bench json under rocprof>"""
import collections
import csv
import glob
import json
import shutil
import statistics
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def main():
    prefix, rnd, bench_json = sys.argv[1], sys.argv[2], sys.argv[3]
    out_dir = ROOT / "profiles" / rnd
    out_dir.mkdir(parents=True, exist_ok=True)
    g = ROOT / "gpurun_out"
    tag = "restir_di_1080p_hall1M"
    shutil.copy(glob.glob(str(g / f"{prefix}_stats/runc/*_kernel_stats.csv"))[0], out_dir / f"{tag}_kernel_stats.csv")
    shutil.copy(g / bench_json, out_dir / f"{tag}_bench_under_rocprof.json")
    bench = json.loads((g / bench_json).read_text())
    # timed-region means from the kernel trace of the --stats run
    d = {}
    for r in csv.DictReader(open(glob.glob(str(g / f"{prefix}_stats/runc/*_kernel_trace.csv"))[0])):
        if r["Kernel_Name"].startswith("rt::k_"):
            d.setdefault(r["Kernel_Name"].split("(")[0].replace("rt::", ""), []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    timed = {}
    w, k = bench["warmup"], bench["steps"]
    for name, v in d.items():
        t = v[w:w + k]
        if not t:
            continue
        timed[name] = {"dispatches_total": len(v), "all_mean_us": round(statistics.mean(v) / 1e3, 2), "all_median_us": round(statistics.median(v) / 1e3, 2),
                       "timed_region_dispatches": len(t), "timed_region_mean_us": round(statistics.mean(t) / 1e3, 2),
                       "bench_hipEvent_avg_us_same_run": round(bench["roofline"]["kernels"].get(name, {}).get("avg_ms", float("nan")) * 1e3, 2)}
    (out_dir / f"{tag}_timed_region.json").write_text(json.dumps(timed, indent=1))
    # PMC passes
    out = {}
    for t in ("FETCH_SI", "WRITE_SI", "TCC_HIT_"):
        f = glob.glob(str(g / f"{prefix}_{t}/runc/*_counter_collection.csv"))[0]
        agg, meta = collections.defaultdict(list), {}
        for r in csv.DictReader(open(f)):
            if not r["Kernel_Name"].startswith("rt::k_") or "build_light" in r["Kernel_Name"]:
                continue
            kn = r["Kernel_Name"].split("(")[0].replace("rt::", "")
            agg[(kn, r["Counter_Name"])].append(float(r["Counter_Value"]))
            meta[kn] = {c: int(r[c]) for c in ("VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Grid_Size", "Workgroup_Size")}
        for (kn, c), v in agg.items():
            v = v[2:-1] if len(v) > 4 else v          # drop warm-up and the instrumented frame
            out.setdefault(kn, {"dispatch": meta[kn]})[c] = {"mean_per_dispatch": sum(v) / len(v), "n": len(v)}
    for kn, x in out.items():
        x["hbm_read_bytes_corrected"] = x["FETCH_SIZE"]["mean_per_dispatch"] * 1024 * 2      # MI355X_MICROARCH.md §HBM: x2 for 16-B/lane reads
        x["hbm_write_bytes"] = x["WRITE_SIZE"]["mean_per_dispatch"] * 1024
        x["hbm_bytes_per_launch"] = x["hbm_read_bytes_corrected"] + x["hbm_write_bytes"]
        x["l2_hit_rate"] = x["TCC_HIT_sum"]["mean_per_dispatch"] / (x["TCC_HIT_sum"]["mean_per_dispatch"] + x["TCC_MISS_sum"]["mean_per_dispatch"])
    (out_dir / f"{tag}_pmc_summary.json").write_text(json.dumps(out, indent=1))
    traffic = {f"{kn}@1920x1080@hall": {"hbm_bytes_per_launch": int(x["hbm_bytes_per_launch"]), "l2_hit_rate": round(x["l2_hit_rate"], 4),
                                        "source": f"profiles/{rnd}/{tag}_pmc_summary.json"} for kn, x in out.items()}
    (ROOT / "profiles" / "traffic.json").write_text(json.dumps(traffic, indent=1))
    print(json.dumps({"timed": timed, "traffic": traffic}, indent=1))


if __name__ == "__main__":
    main()
