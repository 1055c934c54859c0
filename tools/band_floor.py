#!/usr/bin/env python3
"""Stand-alone kernel times (frames not pipelined, blocking) of the ReSTIR DI frame for bands of different heights around the middle of
the 1080p bench frame: where the time of a narrow band goes (latency floor of each launch vs work).   usage: python tools/band_floor.py"""
import json
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from fypraytracer_amd import capi, scenes  # noqa: E402

W, H = 1920, 1080
sc, cam = scenes.hall_scene(), scenes.hall_camera(W, H)
ctx = capi.Context(0)
ctx.resize(W, H); ctx.upload_scene(sc); ctx.set_camera(cam)
for kv in sys.argv[1:]:
    k, v = kv.split("="); ctx.set_tuning(int(k), int(v))
st = capi.Settings(technique=7, light_bounces=1, sky_color=(0, 0, 0), use_temporal_reuse=1, use_spatial_reuse=1)
ctx.set_tuning(11, 0)
for f in range(3):
    st.rand_seed = f + 1; ctx.render(st)
for rows, halo in ((16, 0), (32, 0), (64, 0), (135, 0), (135, 30), (270, 30), (540, 30), (1080, 0)):
    y0 = max(0, 540 - rows // 2); y1 = min(H, y0 + rows)
    ctx.set_rows(y0, y1, halo)
    parts = []
    for f in range(14):
        st.rand_seed = 10 + f
        s = ctx.render(st)
        if f >= 4:
            parts.append(list(s.kernel_ms_part)[:3])
    m = np.median(np.array(parts), axis=0)
    print(json.dumps({"rows": rows, "halo": halo, "part1_ms": round(float(m[0]), 4), "setup_ms": round(float(m[1]), 4), "trace_ms": round(float(m[2]), 4), "sum_ms": round(float(m.sum()), 4),
                      "us_per_row_sum": round(float(m.sum()) / rows * 1e3, 3)}), flush=True)
