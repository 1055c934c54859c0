#!/usr/bin/env python3
"""One band of an N-way row split of the bench frame, rendered alone with asynchronous frames (what a rank of bench.py --gpus N does):
   usage: python tools/band_probe.py <n> <rank> [frames] [mode: recompute|exchange] [key=value ...]   — for a kernel-trace timeline (tools/timeline_run.sh)."""
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from fypraytracer_amd import capi, multigpu, scenes  # noqa: E402

n, rank = int(sys.argv[1]), int(sys.argv[2])
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 60
mode = sys.argv[4] if len(sys.argv) > 4 else "recompute"
knobs = [kv.split("=") for kv in sys.argv[5:]]
W, H = 1920, 1080
sc, cam = scenes.hall_scene(), scenes.hall_camera(W, H)
ctx = capi.Context(0)
ctx.resize(W, H); ctx.upload_scene(sc); ctx.set_camera(cam)
st = capi.Settings(technique=7, light_bounces=1, sky_color=(0, 0, 0), use_temporal_reuse=1, use_spatial_reuse=1)
for f in range(2):
    st.rand_seed = f + 1; ctx.render_async(st)
halo = multigpu.halo_rows(st, 7, n)
y0, y1 = multigpu.band_rows(H, n, rank)
ctx.set_tuning(13, 1 if mode == "exchange" else 0)
for k, v in knobs:
    ctx.set_tuning(int(k), int(v))
ctx.set_rows(y0, y1, halo)
for f in range(8):
    st.rand_seed = f + 3; ctx.render_async(st)
ctx.synchronize()
t0 = time.perf_counter()
for f in range(frames):
    st.rand_seed = 11 + f; ctx.render_async(st)
ctx.synchronize()
print("band", (y0, y1), "halo", halo, mode, knobs, "wall ms/frame", round((time.perf_counter() - t0) / frames * 1e3, 4), "parts", [round(x, 4) for x in ctx.frame_timings(0)[0]])
