#!/usr/bin/env python3
"""Per-kernel table of BASELINE config 5 on one GPU from the rocprofv3 passes of tools/profile_config5.sh:
   python tools/summarize_config5.py <prefix> <round>   ->  profiles/<round>/config5_gi_4k_kernels.json (+ a markdown table on stdout)
Per kernel: launches per frame, mean duration, time per frame, HBM bytes per launch (FETCH_SIZE x 1024 x 2 + WRITE_SIZE x 1024, the
gfx950 correction of MI355X_MICROARCH.md §HBM for 16-byte-per-lane loads) and the rate that is.  The instrumented twins (<true>) of the
one counting frame are left out."""
import csv
import os
import glob
import json
import re
import sys
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def short(name):
    m = re.match(r"(?:void )?(?:rt::)?([A-Za-z0-9_]+(?:<[^>]*>)?)", name)
    return m.group(1) if m else name


def main():
    prefix, rnd = sys.argv[1], sys.argv[2]
    out = ROOT / "gpurun_out"
    stats = sorted(glob.glob(str(out / f"{prefix}_stats" / "*" / "*kernel_stats.csv")), key=os.path.getmtime)
    assert stats, "no kernel_stats.csv"
    rows = {}
    raw = list(csv.DictReader(open(stats[-1])))
    FRAMES = min(int(r["Calls"]) for r in raw if "k_gi_primary<false>" in r["Name"])      # one primary launch per production frame (the counting frame runs the <true> twins)
    for r in raw:
        n = short(r["Name"])
        if "true" in n or n.startswith("__amd") or not (n.startswith("k_") or "Kernel" in n):
            continue
        rows[n] = {"launches_per_frame": round(int(r["Calls"]) / FRAMES, 2), "avg_ms": round(float(r["AverageNs"]) / 1e6, 4),
                   "ms_per_frame": round(float(r["TotalDurationNs"]) / 1e6 / FRAMES, 4)}
    for tag in ("FETCH_SIZE", "WRITE_SIZE"):
        files = sorted(glob.glob(str(out / f"{prefix}_{tag}" / "*" / "*counter_collection.csv")), key=os.path.getmtime)
        acc, cnt = defaultdict(float), defaultdict(int)
        for r in csv.DictReader(open(files[-1])) if files else []:
            n = short(r["Kernel_Name"])
            if r["Counter_Name"] == tag:
                acc[n] += float(r["Counter_Value"]); cnt[n] += 1
        for n in rows:
            if cnt.get(n):
                rows[n][tag.lower() + "_kb_per_launch"] = round(acc[n] / cnt[n], 1)
    for n, x in rows.items():
        if "fetch_size_kb_per_launch" in x and "write_size_kb_per_launch" in x:
            b = x["fetch_size_kb_per_launch"] * 1024 * 2 + x["write_size_kb_per_launch"] * 1024
            x["hbm_mb_per_launch"] = round(b / 1e6, 1)
            x["hbm_tb_per_s"] = round(b / (x["avg_ms"] * 1e-3) / 1e12, 2) if x["avg_ms"] > 0 else None
    total = round(sum(x["ms_per_frame"] for x in rows.values()), 3)
    doc = {"workload": "config 5: hall 1M, 3840x2160, ReSTIR GI, 1 spp, 2 bounces, one GPU", "frames_profiled": FRAMES, "kernel_ms_per_frame_sum": total,
           "kernels": dict(sorted(rows.items(), key=lambda kv: -kv[1]["ms_per_frame"]))}
    dst = ROOT / "profiles" / rnd / "config5_gi_4k_kernels.json"
    dst.write_text(json.dumps(doc, indent=1))
    print(f"| kernel | launches / frame | mean ms | ms / frame | HBM MB / launch | TB/s |\n|---|---|---|---|---|---|")
    for n, x in doc["kernels"].items():
        print(f"| `{n}` | {x['launches_per_frame']} | {x['avg_ms']} | {x['ms_per_frame']} | {x.get('hbm_mb_per_launch', '')} | {x.get('hbm_tb_per_s', '')} |")
    print(f"| sum | | | {total} | | |")


if __name__ == "__main__":
    main()
