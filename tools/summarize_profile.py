#!/usr/bin/env python3
"""Turn the raw rocprofv3 output of tools/profile_round.sh (gpurun_out/<prefix>_*) into the tracked summaries under
profiles/<round>/ and profiles/counters.json (what bench.py quotes: HBM bytes, VALU wave-instructions and lane utilisation per
launch — stamped with the kernel-source fingerprint and the commit they were measured on).
  usage: summarize_profile.py <prefix, e.g. prof_r02> <round dir, e.g. r02>"""
import collections
import csv
import glob
import json
import shutil
import statistics
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def kname(s):
    import re
    return re.sub(r"<[^>]*>", "", s.split("(")[0].replace("void ", "").replace("rt::", "")).strip()


def newest_run_only(pass_dir):
    """gpurun merges every call's files into the same local directories: keep the files of the newest profiler process only"""
    import os
    import re
    by_pid = collections.defaultdict(list)
    for f in glob.glob(str(pass_dir / "**" / "*.csv"), recursive=True):
        m = re.match(r"(\d+)_", os.path.basename(f))
        if m:
            by_pid[m.group(1)].append(f)
    if len(by_pid) > 1:
        newest = max(by_pid, key=lambda p: max(os.path.getmtime(f) for f in by_pid[p]))
        for p, fs in by_pid.items():
            if p != newest:
                for f in fs:
                    os.remove(f)


def main():
    from bench import kernel_source_sha
    prefix, rnd = sys.argv[1], sys.argv[2]
    out_dir = ROOT / "profiles" / rnd
    out_dir.mkdir(parents=True, exist_ok=True)
    g = ROOT / "gpurun_out"
    tag = "restir_di_1080p_hall1M"
    for d in g.glob(f"{prefix}_*"):
        if d.is_dir():
            newest_run_only(d)
    shutil.copy(glob.glob(str(g / f"{prefix}_stats/**/*_kernel_stats.csv"), recursive=True)[0], out_dir / f"{tag}_kernel_stats.csv")
    shutil.copy(g / f"{prefix}_bench.json", out_dir / f"{tag}_bench_under_rocprof.json")
    bench = json.loads((g / f"{prefix}_bench.json").read_text())
    # timed-region means from the kernel trace of the --stats run
    d = {}
    for r in csv.DictReader(open(glob.glob(str(g / f"{prefix}_stats/**/*_kernel_trace.csv"), recursive=True)[0])):
        if "rt::k_" in r["Kernel_Name"]:
            d.setdefault(kname(r["Kernel_Name"]), []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    timed = {}
    w, k = bench["warmup"], bench["steps"]
    for name, v in d.items():
        t = v[w:w + k]
        if not t:
            continue
        timed[name] = {"dispatches_total": len(v), "timed_region_dispatches": len(t), "timed_region_mean_us": round(statistics.mean(t) / 1e3, 2),
                       "bench_hipEvent_avg_us_same_run": round(bench["roofline"]["kernels"].get(name, {}).get("avg_ms_timed_region", float("nan")) * 1e3, 2),
                       "alone_mean_us_same_run (bench: frames not pipelined)": round(bench["roofline"]["kernels"].get(name, {}).get("avg_ms_alone", float("nan")) * 1e3, 2)}
    (out_dir / f"{tag}_timed_region.json").write_text(json.dumps(timed, indent=1))
    # PMC passes: mean per dispatch over the steady-state frames
    out = {}
    for f in glob.glob(str(g / f"{prefix}_*/**/*_counter_collection.csv"), recursive=True):
        agg, meta = collections.defaultdict(list), {}
        for r in csv.DictReader(open(f)):
            if "rt::k_" not in r["Kernel_Name"] or "build_light" in r["Kernel_Name"] or "<true>" in r["Kernel_Name"]:
                continue
            kn = kname(r["Kernel_Name"])
            agg[(kn, r["Counter_Name"])].append(float(r["Counter_Value"]))
            meta[kn] = {"arch_vgpr_count_reported (allocation granules, not the compiler's count: see tools/kernel_resources.py)": int(r["VGPR_Count"]),
                        "lds_block_size_reported (static LDS only; the traversal stack is dynamic LDS sized at launch)": int(r["LDS_Block_Size"]),
                        "scratch_size": int(r["Scratch_Size"]), "grid_size": int(r["Grid_Size"]), "workgroup_size": int(r["Workgroup_Size"])}
        for (kn, c), v in agg.items():
            v = v[2:] if len(v) > 4 else v            # drop the warm-up frames
            out.setdefault(kn, {"dispatch": meta[kn]})[c] = {"mean_per_dispatch": sum(v) / len(v), "n": len(v)}
    kernels = {}
    for kn, x in out.items():
        m = lambda c: x[c]["mean_per_dispatch"] if c in x else None   # noqa: E731
        y = {}
        if m("FETCH_SIZE") is not None and m("WRITE_SIZE") is not None:
            x["hbm_read_bytes_corrected"] = m("FETCH_SIZE") * 1024 * 2      # MI355X_MICROARCH.md §HBM: x2 for 16-B/lane reads on gfx950
            x["hbm_write_bytes"] = m("WRITE_SIZE") * 1024
            y["hbm_bytes_per_launch"] = int(x["hbm_read_bytes_corrected"] + x["hbm_write_bytes"])
        if m("TCC_HIT_sum") is not None:
            y["l2_hit_rate"] = round(m("TCC_HIT_sum") / (m("TCC_HIT_sum") + m("TCC_MISS_sum")), 4)
        if m("SQ_INSTS_VALU") is not None:
            y["valu_wave_instructions"] = int(m("SQ_INSTS_VALU"))
        if m("SQ_THREAD_CYCLES_VALU") is not None and m("SQ_ACTIVE_INST_VALU"):
            y["lane_utilisation"] = round(m("SQ_THREAD_CYCLES_VALU") / (64.0 * m("SQ_ACTIVE_INST_VALU")), 3)
        if m("TCP_TOTAL_CACHE_ACCESSES_sum") is not None:
            y["l1_lookups_per_launch"] = int(m("TCP_TOTAL_CACHE_ACCESSES_sum"))        # vector-L1 tag look-ups (what the microbenchmark's gather rate is measured in)
            if m("TCP_TCC_READ_REQ_sum") is not None and m("TCP_TOTAL_CACHE_ACCESSES_sum"):
                y["l1_hit_rate"] = round(1.0 - m("TCP_TCC_READ_REQ_sum") / m("TCP_TOTAL_CACHE_ACCESSES_sum"), 4)
        if m("SQ_INSTS_VMEM_RD") is not None:
            y["vmem_read_wave_instructions"] = int(m("SQ_INSTS_VMEM_RD"))
        if m("SQ_WAIT_ANY") is not None and m("SQ_WAVE_CYCLES"):
            y["wait_any_over_wave_cycles"] = round(m("SQ_WAIT_ANY") / m("SQ_WAVE_CYCLES"), 3)
        x["derived"] = y
        kernels[f"{kn}@1920x1080@hall"] = y
    (out_dir / f"{tag}_pmc_summary.json").write_text(json.dumps(out, indent=1))
    commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
    counters = {"round": rnd, "commit": commit, "kernel_source_sha": kernel_source_sha(), "source": f"profiles/{rnd}/{tag}_pmc_summary.json", "kernels": kernels}
    (ROOT / "profiles" / "counters.json").write_text(json.dumps(counters, indent=1))
    print(json.dumps({"timed": timed, "counters": counters}, indent=1))


if __name__ == "__main__":
    main()
