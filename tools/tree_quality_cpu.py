#!/usr/bin/env python3
"""Quality of the host-built acceleration structure without a GPU: builds the 1 M-triangle hall's tree in a host-only context
(device -1), lets the oracle's twin of the product traversal walk it for one small ReSTIR DI frame and reports node visits and
triangle tests per ray — the two numbers the traversal kernels' time is proportional to.  Builder experiments are selected with the
FYPRT_BVH_* environment variables (bvh_build.cpp).  Test infrastructure (uses oracle/)."""
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from fypraytracer_amd import capi, scenes  # noqa: E402
from oraclelib import Oracle  # noqa: E402

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (320, 180)
sc = scenes.hall_scene()
cam = scenes.hall_camera(W, H)
ctx = capi.Context(-1)
t = time.perf_counter(); ctx.upload_scene(sc); build_s = time.perf_counter() - t
bvh = ctx.export_bvh()
orc = Oracle(sc, W, H)
orc.set_camera(cam)
orc.use_product_bvh(bvh)
out = {"env": {k: v for k, v in os.environ.items() if k.startswith("FYPRT_BVH")}, "build_s": round(build_s, 2), "nodes": len(bvh["nodes"]), "levels": bvh["max_stack"]}
for tech, name in ((capi.RESTIR_DI, "di"), (capi.NEE, "nee")):
    st = capi.Settings(technique=tech, light_bounces=1 if tech == capi.RESTIR_DI else 2, sky_color=(0, 0, 0), use_temporal_reuse=1, use_spatial_reuse=1)
    orc.reset_frame_index()
    c = orc.render(st)
    out[f"{name}_visits_per_ray"] = round(c["node_visits"] / c["rays"], 3)
    out[f"{name}_tris_per_ray"] = round(c["tri_tests"] / c["rays"], 3)
print(json.dumps(out))
