#!/usr/bin/env python3
"""What bounds strong scaling, measured on ONE GPU: every band of an N-way row split rendered on its own (asynchronous frames, no
gather), so a split's frame time is its slowest band (+ the transfer estimate below).  Modes:
  recompute  each band runs ReSTIR Part 1 on its 30-row halo too (halo mode 0: what bench.py --gpus N does by default);
  exchange   each band runs Part 1 on its own rows only (halo mode 1; tuning key 13 emulates it for a lone context — the halo records
             are those of a preceding full-frame render, so Part 2 does realistic work); the exchange itself is priced at
             bytes / 50 GB/s per direction (one xGMI link: MI355X_MICROARCH.md) + 20 us, and reported separately;
  --balance K  K rounds of fyprt_balance_rows on the measured band times before the final measurement (cost-balanced bands).
  --stripes S  (per-pixel techniques 0-6) interleaved stripes of S rows instead of bands: part r of n = fyprt_set_row_stripes(S, n, r).
One JSON line per band and one summary line per split: {"n", "mode", "bounds", "slowest_band_ms", "speedup_vs_1"}."""
import argparse
import os
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from fypraytracer_amd import capi, multigpu, scenes  # noqa: E402

XGMI_GBS = 50.0        # one direction of one link, conservative (7 links x ~153 GB/s aggregate per GPU)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, nargs="+", default=[1, 2, 4, 8])
    ap.add_argument("--frames", type=int, default=100)
    ap.add_argument("--technique", type=int, default=7)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--mode", default="recompute", choices=["recompute", "exchange"])
    ap.add_argument("--balance", type=int, default=0)
    ap.add_argument("--stripes", type=int, default=0)
    a = ap.parse_args()
    W, H = a.width, a.height
    sc, cam = scenes.hall_scene(), scenes.hall_camera(W, H)
    ctx = capi.Context(0)
    ctx.resize(W, H)
    ctx.upload_scene(sc)
    ctx.set_camera(cam)
    for kv in filter(None, os.environ.get("FYPRT_TUNING", "").split(",")):      # e.g. FYPRT_TUNING=19=1 (experiments)
        k, v = kv.split("=")
        ctx.set_tuning(int(k), int(v))
    st = capi.Settings(technique=a.technique, light_bounces=1 if a.technique == 7 else 2, sky_color=(0, 0, 0), use_temporal_reuse=1, use_spatial_reuse=1)
    per_px_bytes = {7: (32, 32), 8: (96, 72)}.get(a.technique, (0, 0))     # Part-1 records, history (fyprt_multi.h)

    def time_band(y0, y1, halo, stripe_part=None):
        ctx.set_tuning(13, 0)
        ctx.set_row_stripes(0)
        ctx.set_rows(0, H, 0)
        ctx.reset_frame_index()
        for f in range(2):                                   # sane records in every row (the halo rows of exchange mode are read, not written)
            st.rand_seed = f + 1
            ctx.render_async(st)
        ctx.set_tuning(13, 1 if a.mode == "exchange" else 0)
        ctx.set_rows(y0, y1, halo)
        if stripe_part is not None:
            ctx.set_row_stripes(a.stripes, *stripe_part)
        for f in range(8):
            st.rand_seed = f + 3
            ctx.render_async(st)
        ctx.synchronize()
        t0 = time.perf_counter()
        for f in range(a.frames):
            st.rand_seed = 11 + f
            ctx.render_async(st)
        ctx.synchronize()
        wall = (time.perf_counter() - t0) / a.frames * 1e3
        parts = np.median(np.array([ctx.frame_timings(b)[0] for b in range(min(a.frames, 64))]), axis=0).round(4).tolist()
        return wall, parts

    base = None
    for n in a.n:
        halo = multigpu.halo_rows(st, a.technique, n)
        bounds = [multigpu.band_rows(H, n, r)[0] for r in range(n)] + [H]
        for it in range(a.balance + 1 if n > 1 else 1):
            rows = []
            for r in range(n):
                wall, parts = time_band(bounds[r], bounds[r + 1], halo, (n, r) if a.stripes and n > 1 else None)
                xfer = 0.0
                if a.mode == "exchange" and n > 1:
                    halo_rows = min(halo, bounds[r]) + min(halo, H - bounds[r + 1])
                    xfer = (halo_rows * W * sum(per_px_bytes)) / (XGMI_GBS * 1e9) * 1e3 / 2 + 0.02    # both neighbours in parallel
                rows.append({"n": n, "mode": a.mode if not a.stripes else f"stripes{a.stripes}", "balance_round": it, "rank": r, "rows": [bounds[r], bounds[r + 1]], "wall_ms_per_frame": round(wall, 4),
                             "launch_ms": parts, "exchange_ms_estimate": round(xfer, 4)})
            final = it == (a.balance if n > 1 else 0)
            if final:
                for row in rows:
                    print(json.dumps(row), flush=True)
            else:
                bounds = capi.balance_rows(bounds, [x["wall_ms_per_frame"] + x["exchange_ms_estimate"] for x in rows], min_rows=16)
        slow = max(x["wall_ms_per_frame"] + x["exchange_ms_estimate"] for x in rows)
        if n == 1:
            base = slow
        print(json.dumps({"n": n, "mode": a.mode if not a.stripes else f"stripes{a.stripes}", "balanced_rounds": a.balance if n > 1 else 0, "bounds": bounds, "slowest_band_ms": round(slow, 4),
                          "band_ms": [round(x["wall_ms_per_frame"], 4) for x in rows], "speedup_vs_1": round(base / slow, 2) if base else None,
                          "note": "projection from one GPU: each band timed alone, no fabric, exchange priced at 50 GB/s + 20 us"}), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
