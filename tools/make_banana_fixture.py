#!/usr/bin/env python3
"""tests/golden/banana_asset.npz — BASELINE config 2's real mesh and texture as a DATA fixture.
The reference ships the assets its application loads at start-up (WalnutApp.cpp:82, :131): `Assets/3D Models/Test/banana.obj`
(Wavefront OBJ, 1590 quads) and `bananaDiffuse.png` (2048 x 2048 RGBA).  /root/reference does not exist on the GPU box, so this
script — run once in the build container — ingests them with THIS repo's own readers (scenes.load_obj: the Assimp flags the
reference asks for are triangulate + generated smooth normals, Scene.cpp:94-146, behaviour unpinned; texture.load_png: Texture.cu:8-40)
and stores the resulting arrays: positions / normals / uvs / triangle indices as the reference's Mesh gets them, and the ABGR8 texture
words.  Arrays only — no reference file is copied."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from fypraytracer_amd import scenes, texture  # noqa: E402

SRC = Path("/root/reference/FYPRayTracer/Assets/3D Models/Test")


def main():
    p, n, uv, idx = scenes.load_obj(str(SRC / "banana.obj"))
    tex = texture.load_png(str(SRC / "bananaDiffuse.png"))
    out = ROOT / "tests" / "golden" / "banana_asset.npz"
    np.savez_compressed(out, positions=p.astype(np.float32), normals=n.astype(np.float32), uvs=uv.astype(np.float32), indices=idx.astype(np.uint32), texture=tex.astype(np.uint32))
    print(f"{out}: {len(idx)} triangles, {len(p)} vertices, texture {tex.shape[1]}x{tex.shape[0]}, {out.stat().st_size / 1e6:.1f} MB")


if __name__ == "__main__":
    main()
