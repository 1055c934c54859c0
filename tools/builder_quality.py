#!/usr/bin/env python3
"""The three builders of the acceleration structure (tuning key 12) on the 1 M-triangle hall: time of fyprt_upload_scene, size and
depth of the tree, and what the ReSTIR DI bench frame costs on it (median kernel ms, node visits per ray).
  0 host binned SAH + SAH-optimal collapse    1 device: Morton sort + radix tree    2 device: Morton sort + PLOC (FYPRT_PLOC_RADIUS)"""
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from fypraytracer_amd import capi, scenes  # noqa: E402

W, H = 1920, 1080
sc = scenes.hall_scene()
cam = scenes.hall_camera(W, H)
builders = [int(x) for x in sys.argv[1:]] or [0, 1, 2]
for builder in builders:
    ctx = capi.Context(0)
    ctx.resize(W, H)
    ctx.set_tuning(12, builder)
    tb = []
    for _ in range(3 if builder else 1):
        t = time.perf_counter(); ctx.upload_scene(sc); tb.append(time.perf_counter() - t)
    ctx.set_camera(cam)
    out = {"builder": builder, "ploc_radius": int(os.environ.get("FYPRT_PLOC_RADIUS", 16)) if builder == 2 else None, "upload_scene_s": [round(x, 4) for x in tb]}
    for tech, name in ((capi.RESTIR_DI, "restir_di"), (capi.NEE, "nee")):
        st = capi.Settings(technique=tech, light_bounces=1 if tech == capi.RESTIR_DI else 2, sky_color=(0, 0, 0), use_temporal_reuse=1, use_spatial_reuse=1)
        ctx.reset_frame_index()
        ks = []
        for f in range(24):
            st.rand_seed = f + 1
            ks.append(ctx.render(st).kernel_ms)
        out[f"{name}_frame_ms"] = round(float(np.median(ks[6:])), 4)
        ctx.set_ray_counting(True)
        s = ctx.render(st)
        ctx.set_ray_counting(False)
        out[f"{name}_node_visits_per_ray"] = round(s.node_visits / max(1, s.rays), 2)
        out[f"{name}_tri_tests_per_ray"] = round(s.tri_tests / max(1, s.rays), 2)
    b = ctx.export_bvh()
    out.update({"nodes": int(len(b["nodes"])), "levels": int(b["max_stack"]), "mean_children": round(float((b["nodes"]["meta"] & 7).mean()), 2)})
    print(json.dumps(out), flush=True)
    ctx.close()
