#!/usr/bin/env python3
"""Random sequences of API calls — techniques, asynchronous / blocking frames, frame-index resets, transform edits through the
SceneManager + device refit, camera moves — on one context, against the same sequence on a context that renders every frame
blocking and not pipelined (and re-uploads the scene instead of refitting it is NOT done: the refitted tree is part of the
state; both contexts refit).  The bits of everything a context owns must agree at the end of every sequence."""
import argparse
import sys
import zlib
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from fypraytracer_amd import capi, scenes  # noqa: E402

TECHS = [capi.RESTIR_DI] * 6 + [capi.RESTIR_GI, capi.NEE, capi.COSINE_WEIGHTED_SAMPLING, capi.BRUTE_FORCE, capi.LIGHT_SOURCE_SAMPLING]


def play(ops, W, H, fast):
    sc = scenes.hall_scene_small()
    cam = scenes.hall_camera(W, H)
    mgr = sc.manager(); mgr.perform_all_scene_updates(sc)
    ctx = capi.Context(0)
    ctx.resize(W, H)
    ctx.upload_scene(sc)
    ctx.set_camera(cam)
    ctx.set_tuning(11, 1 if fast else 0)
    seed = 0
    for op in ops:
        if op[0] == "frame":
            seed += 1
            st = capi.Settings(technique=op[1], sample_count=1, light_bounces=2 if op[1] != capi.RESTIR_DI else 1, sky_color=(0.1, 0.2, 0.3),
                               use_temporal_reuse=1, use_spatial_reuse=1, rand_seed=seed)
            if fast and op[2]:
                ctx.render_async(st)
            else:
                ctx.render(st)
        elif op[0] == "reset":
            ctx.reset_frame_index()
        elif op[0] == "move":
            mgr.set_mesh_transform(sc, op[1], pos=op[2], rotation=op[3])
            mgr.perform_all_scene_updates(sc)
            ctx.update_vertices(sc)
        elif op[0] == "camera":
            cam.set_position(op[1])
            ctx.set_camera(cam)
    ctx.synchronize()
    img, acc = ctx.readback()
    crc = zlib.crc32(acc.tobytes(), zlib.crc32(img.tobytes()))
    for b in (capi.BUF_DI, capi.BUF_DI_PREV, capi.BUF_GI, capi.BUF_GI_PREV, capi.BUF_DEPTH, capi.BUF_PAYLOAD):
        crc = zlib.crc32(ctx.read_buffer(b).tobytes(), crc)
    ctx.close()
    return crc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sequences", type=int, default=12)
    ap.add_argument("--length", type=int, default=40)
    a = ap.parse_args()
    W, H = 224, 128
    n_meshes = len(scenes.hall_scene_small().meshes)
    for s in range(a.sequences):
        rng = np.random.default_rng(1000 + s)
        ops = []
        for _ in range(a.length):
            r = rng.random()
            if r < 0.78:
                ops.append(("frame", int(rng.choice(TECHS)), bool(rng.random() < 0.8)))
            elif r < 0.84:
                ops.append(("reset",))
            elif r < 0.93:
                ops.append(("move", int(rng.integers(0, n_meshes)), tuple(rng.uniform(-0.5, 0.5, 3).tolist()), (0.0, float(rng.uniform(-30, 30)), 0.0)))
            else:
                ops.append(("camera", tuple((np.array([-18.5, 5.5, 6.5]) + rng.uniform(-0.5, 0.5, 3)).tolist())))
        ref, got = play(ops, W, H, False), play(ops, W, H, True)
        print(f"sequence {s}: {sum(o[0] == 'frame' for o in ops)} frames, {sum(o[0] == 'move' for o in ops)} refits  blocking {ref:#010x}  async/pipelined {got:#010x}  {'OK' if ref == got else 'MISMATCH'}", flush=True)
        if ref != got:
            sys.exit(1)


if __name__ == "__main__":
    main()
