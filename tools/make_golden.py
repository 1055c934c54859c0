#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the CPU oracle (reference-order traversal, deterministic
transcendentals).  The reference ships no golden vectors, so these fixtures are the pin:
Cornell box 64x64 (BASELINE config 1 geometry, camera nudged off-axis to avoid exact-t ties), every technique, frames 1-3 accumulated.
Each file: inputs (settings as a dict) + expected outputs (RGBA8 per frame, float4 accumulation after
frame 3, and for the ReSTIR techniques the hit payloads and reservoirs after frame 3)."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

from common import settings_for  # noqa: E402
from fypraytracer_amd import capi, scenes  # noqa: E402
from oraclelib import Oracle  # noqa: E402

W = H = 64
FRAMES = 3


def main():
    out_dir = ROOT / "tests" / "golden"
    out_dir.mkdir(parents=True, exist_ok=True)
    sc = scenes.cornell_box()
    cam = scenes.cornell_camera(W, H, off_axis=True)
    for tech in range(9):
        o = Oracle(sc, W, H)
        o.set_camera(cam)
        st = settings_for(tech, light_bounces=4, sample_count=2, sky_color=(0.0, 0.0, 0.0))
        images = []
        for f in range(FRAMES):
            st.rand_seed = f + 1
            o.render(st)
            images.append(o.image().copy())
        data = {"images": np.stack(images), "accum": o.accum(), "width": W, "height": H, "frames": FRAMES, "technique": tech,
                "light_bounces": 4, "sample_count": 2, "sky_color": np.zeros(3, np.float32)}
        if tech in (capi.RESTIR_DI, capi.RESTIR_GI):
            data["payload"] = o.read_buffer(capi.BUF_PAYLOAD)
            data["reservoir"] = o.read_buffer(capi.BUF_DI if tech == capi.RESTIR_DI else capi.BUF_GI)
            data["prev_reservoir"] = o.read_buffer(capi.BUF_DI_PREV if tech == capi.RESTIR_DI else capi.BUF_GI_PREV)
        np.savez_compressed(out_dir / f"cornell64_{capi.TECHNIQUE_NAMES[tech].lower()}.npz", **data)
        print("wrote", capi.TECHNIQUE_NAMES[tech], "mean radiance", float(np.nanmean(data["accum"][..., :3])) / FRAMES)


if __name__ == "__main__":
    main()
