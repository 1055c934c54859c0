#!/usr/bin/env python3
"""Tree-quality lab (CPU only, no GPU): builds the bench scene's acceleration structure with the library's HOST builder (a host-only
context), hands it to the oracle's twin of the product traversal and renders ReSTIR DI frames at a reduced resolution, printing what the
builder's choices cost per ray: node visits, child-box tests, triangle tests.  Builder knobs are environment variables read by
bvh_build.cpp (FYPRT_BVH_*), so a variant is `FYPRT_BVH_BINS=32 python tools/tree_lab.py`.
usage: python tools/tree_lab.py [--width 480 --height 270 --frames 2 --technique 7 --scene hall|hall_small]"""
import argparse
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=480)
    ap.add_argument("--height", type=int, default=270)
    ap.add_argument("--frames", type=int, default=2)
    ap.add_argument("--technique", type=int, default=7)
    ap.add_argument("--bounces", type=int, default=1)
    ap.add_argument("--scene", default="hall")
    ap.add_argument("--label", default="")
    a = ap.parse_args()
    from fypraytracer_amd import capi, scenes
    from oraclelib import Oracle, lib as orc_lib
    sc = scenes.hall_scene() if a.scene == "hall" else scenes.hall_scene_small()
    cam = scenes.hall_camera(a.width, a.height)
    ctx = capi.Context(-1)
    t0 = time.perf_counter()
    ctx.upload_scene(sc)
    build_s = time.perf_counter() - t0
    bvh = ctx.export_bvh()
    orc_lib().orc_set_threads(8)
    orc = Oracle(sc, a.width, a.height)
    orc.set_camera(cam)
    orc.use_product_bvh(bvh)
    st = capi.Settings(technique=a.technique, light_bounces=a.bounces, sample_count=1, sky_color=(0.0, 0.0, 0.0), light_candidate_count=4,
                       use_temporal_reuse=1, use_spatial_reuse=1)
    tot = {"rays": 0, "box_tests": 0, "tri_tests": 0, "node_visits": 0}
    for f in range(a.frames):
        st.rand_seed = f + 1
        c = orc.render(st)
        for k in tot:
            tot[k] += c[k]
    r = max(tot["rays"], 1)
    print(json.dumps({"label": a.label, "scene": a.scene, "size": [a.width, a.height], "frames": a.frames, "technique": a.technique, "nodes": int(len(bvh["nodes"])),
                      "leaf_tris": int(len(bvh["tris"])), "levels": int(bvh["max_stack"]), "build_s": round(build_s, 2), "rays": tot["rays"],
                      "node_visits_per_ray": round(tot["node_visits"] / r, 3), "box_tests_per_ray": round(tot["box_tests"] / r, 3),
                      "tri_tests_per_ray": round(tot["tri_tests"] / r, 3)}), flush=True)
    orc.close()
    ctx.close()


if __name__ == "__main__":
    main()
