#!/usr/bin/env python3
"""Time of fyprt_update_vertices (device refit) next to a full fyprt_upload_scene (host rebuild) on the 1M-triangle hall,
and what the refitted tree costs to trace after a sizeable edit (one drape-sized mesh moved by a metre)."""
import json
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
ctx = capi.Context(0)
ctx.resize(W, H)
t = time.perf_counter(); ctx.upload_scene(sc); up = time.perf_counter() - t
ctx.set_camera(cam)
st = capi.Settings(technique=capi.RESTIR_DI, sky_color=(0, 0, 0), use_temporal_reuse=1, use_spatial_reuse=1)


def frame_ms(n=20):
    ks = []
    for f in range(n):
        st.rand_seed = f + 1
        ks.append(ctx.render(st).kernel_ms)
    return float(np.median(ks[5:]))


base = frame_ms()
mgr = sc.manager(); mgr.perform_all_scene_updates(sc)
big = int(np.argmax([c for (_, c, _) in sc.meshes]))
mgr.set_mesh_transform(sc, big, pos=(1.0, 0.2, -0.5), rotation=(0, 10, 0))
mgr.perform_all_scene_updates(sc)
ts = []
for _ in range(5):
    t = time.perf_counter(); ctx.update_vertices(sc); ts.append(time.perf_counter() - t)
refit = frame_ms()
t = time.perf_counter(); ctx.upload_scene(sc); up2 = time.perf_counter() - t
rebuilt = frame_ms()
print(json.dumps({"upload_scene_s": round(up, 3), "upload_scene_again_s": round(up2, 3), "update_vertices_s_median": round(float(np.median(ts)), 4),
                  "moved_mesh_triangles": int(sc.meshes[big][1]), "frame_ms_before": round(base, 4), "frame_ms_after_refit": round(refit, 4),
                  "frame_ms_after_rebuild": round(rebuilt, 4)}))

# device builder (tuning key 12) on the same scene: build time and what its tree costs to trace
ctx2 = capi.Context(0)
ctx2.resize(W, H)
ctx2.set_tuning(12, 1)
tb = []
for _ in range(3):
    t = time.perf_counter(); ctx2.upload_scene(sc); tb.append(time.perf_counter() - t)
ctx2.set_camera(cam)
ctx = ctx2
lb = frame_ms()
b = ctx2.export_bvh()
print(json.dumps({"device_builder_upload_scene_s": [round(x, 4) for x in tb], "frame_ms_device_built_tree": round(lb, 4), "nodes": int(len(b["nodes"])),
                  "levels": int(b["max_stack"]), "mean_children": round(float((b["nodes"]["meta"] & 7).mean()), 2)}))
