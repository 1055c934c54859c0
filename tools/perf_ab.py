#!/usr/bin/env python3
"""Interleaved A/B of result-neutral tuning knobs in ONE process (cdna guide §5.4 rule 24):
N variants x M rounds on the bench workload, median / min of the per-kernel hipEvent times."""
import argparse
import itertools
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from fypraytracer_amd import capi, scenes  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--key", type=int, default=0)
    ap.add_argument("--values", type=int, nargs="+", default=[0, 1, 2])
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--frames", type=int, default=12)
    ap.add_argument("--technique", type=int, default=7)
    ap.add_argument("--scene", default="hall")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--async-frames", type=int, default=0, help="measure wall ms/frame over this many asynchronous (pipelined) frames instead of blocking frames")
    ap.add_argument("--lib", default=None, help="alternative libfyprt build to load")
    ap.add_argument("--set", nargs="*", default=[], help="fixed knobs key=value")
    a = ap.parse_args()
    if a.lib:
        capi._lib = capi.load_library(a.lib)
    W, H = a.width, a.height
    sc = scenes.hall_scene() if a.scene == "hall" else scenes.hall_scene_small()
    cam = scenes.hall_camera(W, H)
    ctx = capi.Context(0)
    ctx.resize(W, H)
    ctx.upload_scene(sc)
    ctx.set_camera(cam)
    for kv in a.set:
        k, v = kv.split("=")
        ctx.set_tuning(int(k), int(v))
    st = capi.Settings(technique=a.technique, light_bounces=2 if a.technique != 7 else 1, sky_color=(0, 0, 0), use_temporal_reuse=1, use_spatial_reuse=1)
    res = {v: [] for v in a.values}
    allparts = {}
    import time
    wall = {v: [] for v in a.values}
    for r in range(a.rounds):
        for v in a.values:
            ctx.set_tuning(a.key, v)
            ctx.reset_frame_index()
            if a.async_frames:
                for f in range(10):
                    st.rand_seed = f + 1
                    ctx.render_async(st)
                ctx.synchronize()
                t0 = time.perf_counter()
                for f in range(a.async_frames):
                    st.rand_seed = 11 + f
                    ctx.render_async(st)
                ctx.synchronize()
                wall[v].append((time.perf_counter() - t0) / a.async_frames * 1e3)
                continue
            parts = []
            for f in range(a.frames):
                st.rand_seed = f + 1
                s = ctx.render(st)
                if f >= 2:
                    parts.append([s.kernel_ms_part[0], sum(list(s.kernel_ms_part)[1:])])
            res[v].append(np.median(np.array(parts), axis=0))
            allparts.setdefault(v, []).append(list(s.kernel_ms_part))
    if a.async_frames:
        for v in a.values:
            print(json.dumps({"key": a.key, "value": v, "async_wall_ms_per_frame_median": round(float(np.median(wall[v])), 4), "min": round(float(np.min(wall[v])), 4)}))
        return
    for v in a.values:
        m = np.array(res[v])
        print(json.dumps({"key": a.key, "value": v, "p1_ms_median": round(float(np.median(m[:, 0])), 4), "p2_ms_median": round(float(np.median(m[:, 1])), 4),
                          "p1_ms_min": round(float(m[:, 0].min()), 4), "p2_ms_min": round(float(m[:, 1].min()), 4),
                          "total_median": round(float(np.median(m.sum(1))), 4),
                          "last_frame_parts_ms": [round(x, 4) for x in np.median(np.array(allparts[v]), axis=0)]}))


if __name__ == "__main__":
    main()
