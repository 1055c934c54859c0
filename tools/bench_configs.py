#!/usr/bin/env python3
"""The five BASELINE.json configurations on ONE MI355X (kernel time by hipEvent + wall time of fyprt_render),
with exact ray counts from the device instrumentation.  Writes one JSON line per config.
(Config 1 is the reference's CPU case — it is run on the GPU here too; config 5's 8-GPU split is bench.py's job:
this tool reports its single-GPU 4K number.)"""
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from fypraytracer_amd import capi, scenes  # noqa: E402


def run(name, sc, cam, W, H, st, frames=30, warm=5):
    ctx = capi.Context(0)
    ctx.resize(W, H)
    ctx.upload_scene(sc)
    ctx.set_camera(cam)
    for kv in filter(None, os.environ.get("FYPRT_TUNING", "").split(",")):      # e.g. FYPRT_TUNING=15=2,6=16 (experiments)
        k, v = kv.split("=")
        ctx.set_tuning(int(k), int(v))
    for f in range(warm):
        st.rand_seed = f + 1
        ctx.render(st)
    ctx.synchronize()
    kms = []
    t0 = time.perf_counter()
    for f in range(frames):
        st.rand_seed = warm + f + 1
        kms.append(ctx.render(st).kernel_ms)
    wall = (time.perf_counter() - t0) / frames * 1e3
    ctx.set_ray_counting(True)
    st.rand_seed = warm + frames + 1
    cs = ctx.render(st)
    ctx.set_ray_counting(False)
    k = float(np.median(kms))
    out = {"config": name, "resolution": f"{W}x{H}", "technique": capi.TECHNIQUE_NAMES[st.technique], "triangles": int(len(sc.triangles)),
           "sample_count": st.sample_count, "light_bounces": st.light_bounces, "rays_per_frame": int(cs.rays),
           "kernel_ms_per_frame": round(k, 4), "wall_ms_per_frame": round(wall, 4), "Mrays_per_s_kernel": round(cs.rays / k / 1e3, 1),
           "box_tests_per_ray": round(cs.box_tests / max(1, cs.rays), 2), "tri_tests_per_ray": round(cs.tri_tests / max(1, cs.rays), 2)}
    print(json.dumps(out), flush=True)
    ctx.close()


def main():
    only = set(sys.argv[1:])                      # e.g. "3 5": only these configs (for a profiler run)
    global run
    _run = run

    def run(name, *a, **kw):
        if not only or name.split()[0] in only:
            _run(name, *a, **kw)
    hall = scenes.hall_scene() if (not only or only & {"3", "4", "5"}) else None
    run("1 cornell 512x512 brute force 1spp 4 bounces", scenes.cornell_box(), scenes.cornell_camera(512, 512), 512, 512,
        capi.Settings(technique=capi.BRUTE_FORCE, light_bounces=4, sky_color=(0, 0, 0)))
    run("2 banana (reference mesh + texture, data fixture) 1080p cosine 4spp 2 bounces" if scenes.BANANA_FIXTURE.exists() else "2 banana-standin 1080p cosine 4spp 2 bounces", scenes.banana_scene(), scenes.banana_camera(1920, 1080), 1920, 1080,
        capi.Settings(technique=capi.COSINE_WEIGHTED_SAMPLING, sample_count=4, light_bounces=2, sky_color=(0.3, 0.4, 0.5)))
    run("3 hall 1M 1080p NEE+MIS 1spp 2 bounces", hall, scenes.hall_camera(1920, 1080), 1920, 1080,
        capi.Settings(technique=capi.NEE, sample_count=1, light_bounces=2, sky_color=(0, 0, 0)))
    run("4 hall 1M 1080p ReSTIR DI 1spp", hall, scenes.hall_camera(1920, 1080), 1920, 1080,
        capi.Settings(technique=capi.RESTIR_DI, sky_color=(0, 0, 0), use_temporal_reuse=1, use_spatial_reuse=1), frames=50, warm=10)
    run("5 hall 1M 4K ReSTIR GI 1spp 2 bounces (single GPU)", hall, scenes.hall_camera(3840, 2160), 3840, 2160,
        capi.Settings(technique=capi.RESTIR_GI, light_bounces=2, sky_color=(0, 0, 0), use_temporal_reuse=1, use_spatial_reuse=1), frames=10, warm=3)


if __name__ == "__main__":
    main()
