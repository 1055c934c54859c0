#!/usr/bin/env python3
"""One rank's share of an N-GPU ReSTIR DI frame on ONE GPU: rows of band `rank` of `n` (+ the Part-1 halo), frames enqueued
asynchronously, no gather.  Reports wall ms/frame, the kernels' hipEvent time and the host enqueue cost per frame — i.e. what
bounds strong scaling at small bands (kernel time vs launch/enqueue overhead)."""
import argparse
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from fypraytracer_amd import capi, multigpu, scenes  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, nargs="+", default=[1, 2, 4, 8])
    ap.add_argument("--frames", type=int, default=200)
    ap.add_argument("--technique", type=int, default=7)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--all-ranks", action="store_true", help="every band of each split instead of one interior band")
    a = ap.parse_args()
    W, H = a.width, a.height
    sc, cam = scenes.hall_scene(), scenes.hall_camera(W, H)
    ctx = capi.Context(0)
    ctx.resize(W, H)
    ctx.upload_scene(sc)
    ctx.set_camera(cam)
    st = capi.Settings(technique=a.technique, light_bounces=1 if a.technique == 7 else 2, sky_color=(0, 0, 0), use_temporal_reuse=1, use_spatial_reuse=1)
    for n, rank in [(n, r) for n in a.n for r in (range(n) if a.all_ranks else [n // 2 if n > 1 else 0])]:   # default: an interior band (two halos)
        y0, y1 = multigpu.band_rows(H, n, rank)
        ctx.set_rows(y0, y1, multigpu.halo_rows(st, a.technique, n))
        ctx.reset_frame_index()
        for f in range(10):
            st.rand_seed = f + 1
            ctx.render_async(st)
        ctx.synchronize()
        t0 = time.perf_counter()
        for f in range(a.frames):
            st.rand_seed = 11 + f
            ctx.render_async(st)
        t_enq = time.perf_counter() - t0
        ctx.synchronize()
        wall = time.perf_counter() - t0
        ks, parts = [], []
        for b in range(min(a.frames, 100)):
            ms, nl = ctx.frame_timings(b)
            ks.append(sum(ms[:nl])); parts.append(ms[:4])
        import numpy as np
        parts = np.median(np.array(parts), axis=0).round(4).tolist()
        print(json.dumps({"n": n, "rank": rank, "rows": [y0, y1], "wall_ms_per_frame": round(wall / a.frames * 1e3, 4), "kernel_ms_per_frame": round(sum(ks) / len(ks), 4),
                          "host_enqueue_ms_per_frame": round(t_enq / a.frames * 1e3, 4), "launch_ms": parts}), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
