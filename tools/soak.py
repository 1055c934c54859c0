#!/usr/bin/env python3
"""Soak test of the pipelined path: many asynchronous ReSTIR DI frames (two streams, two task queues) must leave exactly the
bits that the same frames rendered one by one (blocking, not pipelined) leave — accumulation, image, reservoirs, history."""
import argparse
import sys
import zlib
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from fypraytracer_amd import capi, scenes  # noqa: E402


def run(sc, cam, W, H, frames, pipelined, rows=None, halo=0):
    ctx = capi.Context(0)
    ctx.resize(W, H)
    if rows:
        ctx.set_rows(rows[0], rows[1], halo)
    ctx.upload_scene(sc)
    ctx.set_camera(cam)
    ctx.set_tuning(11, 1 if pipelined else 0)
    st = capi.Settings(technique=capi.RESTIR_DI, sky_color=(0.1, 0.2, 0.3), use_temporal_reuse=1, use_spatial_reuse=1)
    for f in range(frames):
        st.rand_seed = f + 1
        if pipelined:
            ctx.render_async(st)
        else:
            ctx.render(st)
    ctx.synchronize()
    img, acc = ctx.readback()
    y0, y1 = rows if rows else (0, H)                      # only the rows this context owns are defined
    crc = zlib.crc32(acc[y0:y1].tobytes(), zlib.crc32(img[y0:y1].tobytes()))
    for b in (capi.BUF_DI, capi.BUF_DI_PREV, capi.BUF_DEPTH, capi.BUF_PAYLOAD):
        crc = zlib.crc32(np.ascontiguousarray(ctx.read_buffer(b).reshape(H, W, -1)[y0:y1]).tobytes(), crc)
    ctx.close()
    return crc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=300)
    a = ap.parse_args()
    for name, sc, cam, W, H, n, rows in (("hall_small 320x180", scenes.hall_scene_small(), scenes.hall_camera(320, 180), 320, 180, a.frames, None),
                                   ("hall_small 320x180 top band of 4 (30-row halo, extra row H-1)", scenes.hall_scene_small(), scenes.hall_camera(320, 180), 320, 180, a.frames, (0, 45)),
                                   ("hall_small 320x180 interior band of 4", scenes.hall_scene_small(), scenes.hall_camera(320, 180), 320, 180, a.frames, (45, 90)),
                                   ("hall 1920x1080", scenes.hall_scene(), scenes.hall_camera(1920, 1080), 1920, 1080, max(20, a.frames // 3), None)):
        ref = run(sc, cam, W, H, n, False, rows, 30 if rows else 0)
        for rep in range(3):
            got = run(sc, cam, W, H, n, True, rows, 30 if rows else 0)
            print(name, n, "frames", "blocking", hex(ref), "pipelined", hex(got), "OK" if got == ref else "MISMATCH", flush=True)
            if got != ref:
                sys.exit(1)


if __name__ == "__main__":
    main()
