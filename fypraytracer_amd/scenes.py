"""Deterministic synthetic scenes for the BASELINE.json configurations.

These are producers of *input data* (flat vertex / triangle / material arrays built through
the host `Scene` API); both the GPU path and the test-side checker consume the same arrays.
"""
from __future__ import annotations

import math

from pathlib import Path

import numpy as np

from .scene import Camera, Material, Scene

F = np.float32


def _quad(p0, p1, p2, p3, normal):
    """Two triangles (p0,p1,p2), (p0,p2,p3) with a constant vertex normal."""
    pos = np.array([p0, p1, p2, p3], dtype=F)
    nrm = np.tile(np.asarray(normal, dtype=F), (4, 1))
    uv = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], dtype=F)
    idx = np.array([[0, 1, 2], [0, 2, 3]], dtype=np.uint32)
    return pos, nrm, uv, idx


def _box(cx, cy, cz, sx, sy, sz, rot_y_deg, skip_bottom=True):
    """Axis box rotated about y: 5 faces (no bottom) x 2 triangles, face normals."""
    c, s = math.cos(math.radians(rot_y_deg)), math.sin(math.radians(rot_y_deg))
    def tr(p):
        x, y, z = p
        return (cx + c * x + s * z, cy + y, cz - s * x + c * z)
    def trn(n):
        x, y, z = n
        return (c * x + s * z, y, -s * x + c * z)
    hx, hy, hz = sx / 2, sy / 2, sz / 2
    faces = [
        ([(-hx, hy, -hz), (-hx, hy, hz), (hx, hy, hz), (hx, hy, -hz)], (0, 1, 0)),       # top
        ([(-hx, -hy, hz), (hx, -hy, hz), (hx, hy, hz), (-hx, hy, hz)], (0, 0, 1)),       # front
        ([(hx, -hy, -hz), (-hx, -hy, -hz), (-hx, hy, -hz), (hx, hy, -hz)], (0, 0, -1)),  # back
        ([(-hx, -hy, -hz), (-hx, -hy, hz), (-hx, hy, hz), (-hx, hy, -hz)], (-1, 0, 0)),  # left
        ([(hx, -hy, hz), (hx, -hy, -hz), (hx, hy, -hz), (hx, hy, hz)], (1, 0, 0)),       # right
    ]
    P, N, U, I = [], [], [], []
    for k, (pts, n) in enumerate(faces):
        p, nn, uv, idx = _quad(*[tr(q) for q in pts], trn(n))
        P.append(p); N.append(nn); U.append(uv); I.append(idx + 4 * k)
    return np.concatenate(P), np.concatenate(N), np.concatenate(U), np.concatenate(I)


def cornell_box(light_power=40.0):
    """BASELINE config 1: 32 triangles in 8 meshes — 5 wall quads, two
    five-faced boxes, one emissive 0.5x0.5 ceiling quad; materials as WalnutApp.cpp:56-74."""
    sc = Scene()
    sc.materials = [
        Material(albedo=(1, 1, 1), roughness=1.0, metallic=0.0),                                       # 0 white
        Material(albedo=(1, 0, 0), roughness=1.0, metallic=0.0),                                       # 1 red
        Material(albedo=(0, 1, 0), roughness=1.0, metallic=0.0),                                       # 2 green
        Material(albedo=(1, 1, 1), emission_color=(1, 1, 1), emission_power=light_power),              # 3 light
        Material(albedo=(0.2, 0.3, 1.0), roughness=0.75, metallic=0.2),                                # 4 blue-ish (WalnutApp.cpp:51-54)
    ]
    add = sc.add_new_mesh_to_scene
    add(*_quad((-1, -1, 1), (1, -1, 1), (1, -1, -1), (-1, -1, -1), (0, 1, 0)), material_index=0)      # floor
    add(*_quad((-1, 1, -1), (1, 1, -1), (1, 1, 1), (-1, 1, 1), (0, -1, 0)), material_index=0)         # ceiling
    add(*_quad((-1, -1, -1), (1, -1, -1), (1, 1, -1), (-1, 1, -1), (0, 0, 1)), material_index=0)      # back
    add(*_quad((-1, -1, 1), (-1, -1, -1), (-1, 1, -1), (-1, 1, 1), (1, 0, 0)), material_index=1)      # left red
    add(*_quad((1, -1, -1), (1, -1, 1), (1, 1, 1), (1, 1, -1), (-1, 0, 0)), material_index=2)         # right green
    add(*_box(-0.35, -0.4, -0.3, 0.6, 1.2, 0.6, 18.0), material_index=0)                              # tall box
    add(*_box(0.4, -0.7, 0.3, 0.6, 0.6, 0.6, -17.0), material_index=4)                                # short box
    add(*_quad((-0.25, 0.999, -0.25), (0.25, 0.999, -0.25), (0.25, 0.999, 0.25), (-0.25, 0.999, 0.25), (0, -1, 0)), material_index=3)
    sc.init_scene_emissive_triangles()
    return sc


def cornell_camera(width, height, off_axis=False):
    """BASELINE config 1 camera: fov 45, (0,0,3.4) looking down -z.  `off_axis=True` nudges the pose so that rays
    through pixel corners no longer hit quad diagonals / wall seams exactly (equal-t ties, whose winner is
    traversal-order dependent in the reference itself) — used by the golden fixtures."""
    cam = Camera(45.0, 0.1, 100.0)                     # WalnutApp.cpp:44
    cam.on_resize(width, height)
    if off_axis:
        d = np.array([0.0171, -0.0093, -1.0], dtype=np.float64)
        cam.forward = (d / np.linalg.norm(d)).astype(F)
        cam.set_position((0.0137, 0.0071, 3.4))
    else:
        cam.forward = np.array([0, 0, -1], dtype=F)
        cam.set_position((0.0, 0.0, 3.4))
    return cam


# --------------------------------------------------------------------------- config 3/4/5: the hall
def _xorshift32(state):
    state ^= (state << 13) & 0xFFFFFFFF
    state ^= state >> 17
    state ^= (state << 5) & 0xFFFFFFFF
    return state & 0xFFFFFFFF


class _Rng:
    def __init__(self, seed=0xC0FFEE):
        self.s = seed & 0xFFFFFFFF or 1

    def u(self):
        self.s = _xorshift32(self.s)
        return self.s / 4294967296.0

    def rng(self, lo, hi):
        return lo + (hi - lo) * self.u()


def _column(cx, cz, radius, y0, y1, segments, rings, flute):
    """Fluted cylinder: rings x segments quads, smooth radial normals."""
    th = (np.arange(segments, dtype=np.float64) / segments) * 2 * np.pi
    ys = np.linspace(y0, y1, rings + 1)
    r = radius * (1.0 + flute * np.cos(th * 12.0))
    x = cx + r[None, :] * np.cos(th)[None, :] * (1.0 + 0.08 * np.cos(ys * 1.3)[:, None])
    z = cz + r[None, :] * np.sin(th)[None, :] * (1.0 + 0.08 * np.cos(ys * 1.3)[:, None])
    y = np.repeat(ys[:, None], segments, axis=1)
    pos = np.stack([x, y, z], -1).reshape(-1, 3).astype(F)
    nrm = np.stack([np.cos(th)[None, :].repeat(rings + 1, 0), np.zeros_like(x), np.sin(th)[None, :].repeat(rings + 1, 0)], -1).reshape(-1, 3).astype(F)
    uv = np.stack([(th / (2 * np.pi))[None, :].repeat(rings + 1, 0), ((ys - y0) / (y1 - y0))[:, None].repeat(segments, 1)], -1).reshape(-1, 2).astype(F)
    i, j = np.meshgrid(np.arange(rings), np.arange(segments), indexing="ij")
    a = i * segments + j
    b = i * segments + (j + 1) % segments
    c = (i + 1) * segments + (j + 1) % segments
    d = (i + 1) * segments + j
    idx = np.concatenate([np.stack([a, b, c], -1).reshape(-1, 3), np.stack([a, c, d], -1).reshape(-1, 3)]).astype(np.uint32)
    return pos, nrm, uv, idx


def _drape(x0, x1, y0, y1, z, n, phase, amp, along_x=True):
    """Displaced (n+1)x(n+1) grid hanging in a vertical plane; normals from the analytic gradient."""
    s = np.linspace(0.0, 1.0, n + 1)
    u, v = np.meshgrid(s, s, indexing="xy")
    w = amp * (np.sin(u * 9.0 * np.pi + phase) * (0.3 + 0.7 * (1.0 - v)) + 0.35 * np.sin(v * 23.0 + u * 31.0 + phase * 2.0))
    dwdu = amp * (9.0 * np.pi * np.cos(u * 9.0 * np.pi + phase) * (0.3 + 0.7 * (1.0 - v)) + 0.35 * 31.0 * np.cos(v * 23.0 + u * 31.0 + phase * 2.0))
    dwdv = amp * (-0.7 * np.sin(u * 9.0 * np.pi + phase) + 0.35 * 23.0 * np.cos(v * 23.0 + u * 31.0 + phase * 2.0))
    a = x0 + (x1 - x0) * u
    y = y0 + (y1 - y0) * v
    if along_x:
        pos = np.stack([a, y, z + w], -1)
        nrm = np.stack([-dwdu / (x1 - x0), -dwdv / (y1 - y0), np.ones_like(w)], -1)
    else:
        pos = np.stack([z + w, y, a], -1)
        nrm = np.stack([np.ones_like(w), -dwdv / (y1 - y0), -dwdu / (x1 - x0)], -1)
    nrm /= np.linalg.norm(nrm, axis=-1, keepdims=True)
    pos = pos.reshape(-1, 3).astype(F)
    nrm = nrm.reshape(-1, 3).astype(F)
    uv = np.stack([u, v], -1).reshape(-1, 2).astype(F)
    i, j = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    p = i * (n + 1) + j
    idx = np.concatenate([np.stack([p, p + 1, p + n + 2], -1).reshape(-1, 3), np.stack([p, p + n + 2, p + n + 1], -1).reshape(-1, 3)]).astype(np.uint32)
    return pos, nrm, uv, idx


def hall_scene(columns=64, column_segments=96, column_rings=40, drapes=16, drape_n=126, light_quads=128,
               lights_per_mesh=8, seed=0xC0FFEE):
    """BASELINE configs 3-5: deterministic "Sponza-scale" hall, 40 x 15 x 20, xorshift seed 0xC0FFEE.
    Defaults give 999 832 triangles in 108 meshes with 128 emissive quads (256 emissive triangles):
    12 wall quads + 64 fluted columns (7 680 tris each) + 16 displaced drapes (31 752 tris each)
    + 16 light meshes of 8 quads.  Smaller parameters give the parity-test scenes."""
    rng = _Rng(seed)
    sc = Scene()
    LX, LY, LZ = 40.0, 15.0, 20.0
    hx, hz = LX / 2, LZ / 2
    # materials: 0..11 surface palette, then one per light mesh
    palette = []
    for k in range(12):
        metallic = (0.0, 0.5, 1.0)[k % 3]
        palette.append(Material(albedo=(0.35 + 0.6 * rng.u(), 0.35 + 0.6 * rng.u(), 0.35 + 0.6 * rng.u()),
                                roughness=rng.rng(0.2, 1.0), metallic=metallic))
    sc.materials = list(palette)
    add = sc.add_new_mesh_to_scene
    # 12 large quads: floor, ceiling and four walls, each split in two
    def split(p0, p1, p2, p3, n, mat):
        p0, p1, p2, p3 = (np.asarray(p, dtype=np.float64) for p in (p0, p1, p2, p3))
        m01, m32 = (p0 + p1) / 2, (p3 + p2) / 2
        add(*_quad(p0, m01, m32, p3, n), material_index=mat)
        add(*_quad(m01, p1, p2, m32, n), material_index=mat)
    split((-hx, 0, hz), (hx, 0, hz), (hx, 0, -hz), (-hx, 0, -hz), (0, 1, 0), 0)            # floor (diffuse)
    split((-hx, LY, -hz), (hx, LY, -hz), (hx, LY, hz), (-hx, LY, hz), (0, -1, 0), 3)       # ceiling
    split((-hx, 0, -hz), (hx, 0, -hz), (hx, LY, -hz), (-hx, LY, -hz), (0, 0, 1), 6)        # back  z=-hz
    split((hx, 0, hz), (-hx, 0, hz), (-hx, LY, hz), (hx, LY, hz), (0, 0, -1), 9)           # front z=+hz
    split((-hx, 0, hz), (-hx, 0, -hz), (-hx, LY, -hz), (-hx, LY, hz), (1, 0, 0), 0)        # left
    split((hx, 0, -hz), (hx, 0, hz), (hx, LY, hz), (hx, LY, -hz), (-1, 0, 0), 3)           # right
    # columns on a 16 x 4 grid
    ncx = max(1, int(round(math.sqrt(columns * 4))))
    ncz = max(1, (columns + ncx - 1) // ncx)
    k = 0
    for iz in range(ncz):
        for ix in range(ncx):
            if k >= columns:
                break
            cx = -hx + (ix + 0.5) * LX / ncx
            cz = -hz + (iz + 0.5) * LZ / ncz
            add(*_column(cx, cz, rng.rng(0.35, 0.6), 0.0, LY - 1.5, column_segments, column_rings, 0.06), material_index=1 + (k % 11))
            k += 1
    # drapes hanging between column rows
    for d in range(drapes):
        along_x = (d % 2 == 0)
        if along_x:
            x0 = -hx + 1.0 + (d // 2 % 4) * (LX - 2.0) / 4
            zpos = -hz + ((d // 8) + 1) * LZ / 3 + rng.rng(-0.6, 0.6)
            add(*_drape(x0, x0 + (LX - 2.0) / 4 - 0.5, 6.0, LY - 0.5, zpos, drape_n, rng.rng(0, 6.28), 0.25, True), material_index=1 + (d % 11))
        else:
            z0 = -hz + 0.7 + (d // 2 % 2) * (LZ - 1.4) / 2
            xpos = -hx + ((d // 4) + 1) * LX / 5 + rng.rng(-0.6, 0.6)
            add(*_drape(z0, z0 + (LZ - 1.4) / 2 - 0.5, 6.5, LY - 0.5, xpos, drape_n, rng.rng(0, 6.28), 0.25, False), material_index=1 + (d % 11))
    # emissive quads on a ceiling grid, grouped `lights_per_mesh` to a mesh, power U[10,40] per mesh
    gx = max(1, int(round(math.sqrt(light_quads * 2))))
    gz = max(1, (light_quads + gx - 1) // gx)
    quads = []
    for q in range(light_quads):
        ix, iz = q % gx, q // gx
        cx = -hx + (ix + 0.5) * LX / gx
        cz = -hz + (iz + 0.5) * LZ / gz
        quads.append((cx, cz))
    for m0 in range(0, light_quads, lights_per_mesh):
        power = rng.rng(10.0, 40.0)
        tint = (0.8 + 0.2 * rng.u(), 0.8 + 0.2 * rng.u(), 0.8 + 0.2 * rng.u())
        sc.materials.append(Material(albedo=(1, 1, 1), emission_color=tint, emission_power=power))
        mi = len(sc.materials) - 1
        P, N, U, I = [], [], [], []
        for j, (cx, cz) in enumerate(quads[m0:m0 + lights_per_mesh]):
            y = LY - 0.02
            p, n, uv, idx = _quad((cx - 0.4, y, cz - 0.4), (cx + 0.4, y, cz - 0.4), (cx + 0.4, y, cz + 0.4), (cx - 0.4, y, cz + 0.4), (0, -1, 0))
            P.append(p); N.append(n); U.append(uv); I.append(idx + 4 * j)
        add(np.concatenate(P), np.concatenate(N), np.concatenate(U), np.concatenate(I), material_index=mi)
    sc.init_scene_emissive_triangles()
    return sc


def hall_scene_small():
    """~13k-triangle version of the hall for parity tests (same generator, same structure)."""
    return hall_scene(columns=8, column_segments=24, column_rings=10, drapes=4, drape_n=30, light_quads=16, lights_per_mesh=4)


def hall_camera(width, height):
    cam = Camera(45.0, 0.1, 100.0)
    cam.on_resize(width, height)
    d = np.array([1.0, -0.12, -0.18], dtype=np.float64)
    cam.forward = (d / np.linalg.norm(d)).astype(F)
    cam.set_position((-18.5, 5.5, 6.5))
    return cam


def load_obj(path):
    """Minimal Wavefront OBJ reader (positions / normals / uvs, polygons fan-triangulated), standing in
    for Mesh::GenerateMesh (Mesh.cpp:271-314; Assimp is not available — parity there is unpinned).
    Vertices are de-duplicated per (v, vt, vn) triple; missing normals are area-weighted smooth normals."""
    P, T, N = [], [], []
    verts, index, tris = [], {}, []
    with open(path, "r", errors="ignore") as f:
        for line in f:
            s = line.split()
            if not s:
                continue
            if s[0] == "v":
                P.append([float(x) for x in s[1:4]])
            elif s[0] == "vt":
                T.append([float(x) for x in s[1:3]])
            elif s[0] == "vn":
                N.append([float(x) for x in s[1:4]])
            elif s[0] == "f":
                ids = []
                for tok in s[1:]:
                    parts = (tok.split("/") + ["", ""])[:3]
                    key = tuple(int(p) if p else 0 for p in parts)
                    key = tuple((k + (len(P), len(T), len(N))[i] + 1) if k < 0 else k for i, k in enumerate(key))
                    if key not in index:
                        index[key] = len(verts)
                        verts.append(key)
                    ids.append(index[key])
                for k in range(1, len(ids) - 1):
                    tris.append((ids[0], ids[k], ids[k + 1]))
    P = np.asarray(P, dtype=np.float64)
    pos = np.array([P[v[0] - 1] for v in verts], dtype=np.float64)
    uv = np.array([(T[v[1] - 1] if v[1] else (0.0, 0.0)) for v in verts], dtype=np.float64)
    tris = np.asarray(tris, dtype=np.uint32)
    if N and all(v[2] for v in verts):
        nrm = np.array([N[v[2] - 1] for v in verts], dtype=np.float64)
    else:
        fn = np.cross(pos[tris[:, 1]] - pos[tris[:, 0]], pos[tris[:, 2]] - pos[tris[:, 0]])
        nrm = np.zeros_like(pos)
        for k in range(3):
            np.add.at(nrm, tris[:, k], fn)
    ln = np.linalg.norm(nrm, axis=1, keepdims=True)
    nrm = nrm / np.where(ln > 0, ln, 1.0)
    return pos.astype(F), nrm.astype(F), uv.astype(F), tris


# --------------------------------------------------------------------------- config 2: one ~3k-triangle textured mesh
def _bent_tube(n_around=30, n_along=53):
    """Procedural stand-in for Assets/3D Models/Test/banana.obj (1 590 quads -> 3 180 triangles, one BLAS):
    a tapered tube bent along an arc, 30 x 53 quads = 3 180 triangles, smooth normals, cylindrical uvs."""
    s = np.linspace(0.0, 1.0, n_along + 1)
    th = (np.arange(n_around, dtype=np.float64) / n_around) * 2 * np.pi
    ang = (s - 0.5) * 2.2                                           # bend angle along the arc
    R = 2.0
    cx, cy = R * np.sin(ang), R * (1.0 - np.cos(ang))
    tx, ty = np.cos(ang), np.sin(ang)                               # tangent of the arc (xy plane)
    nx, ny = -ty, tx                                                # in-plane normal
    rad = 0.38 * np.sin(np.pi * np.clip(s, 0.02, 0.98)) ** 0.6 + 0.03
    px = cx[:, None] + rad[:, None] * np.cos(th)[None, :] * nx[:, None]
    py = cy[:, None] + rad[:, None] * np.cos(th)[None, :] * ny[:, None]
    pz = rad[:, None] * np.sin(th)[None, :] * np.ones_like(cx)[:, None]
    pos = np.stack([px, py, pz], -1)
    nrm = np.stack([np.cos(th)[None, :] * nx[:, None], np.cos(th)[None, :] * ny[:, None], np.sin(th)[None, :] * np.ones_like(cx)[:, None]], -1)
    uv = np.stack([np.repeat((th / (2 * np.pi))[None, :], n_along + 1, 0), np.repeat(s[:, None], n_around, 1)], -1)
    i, j = np.meshgrid(np.arange(n_along), np.arange(n_around), indexing="ij")
    a = i * n_around + j
    b = i * n_around + (j + 1) % n_around
    c = (i + 1) * n_around + (j + 1) % n_around
    d = (i + 1) * n_around + j
    idx = np.concatenate([np.stack([a, b, c], -1).reshape(-1, 3), np.stack([a, c, d], -1).reshape(-1, 3)]).astype(np.uint32)
    return pos.reshape(-1, 3).astype(F), nrm.reshape(-1, 3).astype(F), uv.reshape(-1, 2).astype(F), idx


def _stripe_texture(n=256):
    """ABGR8 (A<<24|B<<16|G<<8|R) procedural albedo map: yellow with brown speckles and green ends."""
    v, u = np.meshgrid(np.linspace(0, 1, n), np.linspace(0, 1, n), indexing="ij")
    r = 225 - 60 * (np.sin(u * 40) * np.sin(v * 55) > 0.92)
    g = 200 - 70 * (np.sin(u * 40) * np.sin(v * 55) > 0.92) + 30 * (np.abs(v - 0.5) > 0.44)
    b = 40 + 20 * (np.abs(v - 0.5) > 0.44)
    return ((255 << 24) | (b.astype(np.uint32) << 16) | (g.astype(np.uint32) << 8) | r.astype(np.uint32)).astype(np.uint32)


BANANA_FIXTURE = Path(__file__).resolve().parent.parent / "tests" / "golden" / "banana_asset.npz"


def banana_scene(obj_path=None, png_path=None, fixture="auto"):
    """BASELINE config 2: one ~3.2k-triangle mesh as a single BLAS, textured diffuse material, transform of
    WalnutApp.cpp:135-137 (pos (0,-3,0), rot (90,0,0)), a floor quad and one emissive quad above.
    Geometry + texture come from, in this order: `obj_path` / `png_path` (the reference's banana.obj / bananaDiffuse.png read in
    place — build container only); the data fixture tests/golden/banana_asset.npz (the same assets ingested once by
    tools/make_banana_fixture.py: what the GPU box uses); `fixture=None` or no fixture: a procedural stand-in (bent tube, stripe texture)."""
    sc = Scene()
    asset = None
    if obj_path is None and png_path is None and fixture is not None and BANANA_FIXTURE.exists():
        asset = np.load(BANANA_FIXTURE)
    if asset is not None:
        sc.textures = [asset["texture"]]
    elif png_path is not None:
        from . import texture
        sc.textures = [texture.load_png(png_path)]                     # Texture::Texture (Texture.cu:8-40)
    else:
        sc.textures = [_stripe_texture()]
    sc.materials = [
        Material(albedo=(0.9, 0.8, 0.2), roughness=1.0, metallic=0.0, is_use_albedo_map=True, albedo_map_index=0),   # matBanana, WalnutApp.cpp:76-80
        Material(albedo=(1, 1, 1), roughness=1.0, metallic=0.0),
        Material(albedo=(1, 1, 1), emission_color=(1, 1, 1), emission_power=40.0),
    ]
    if asset is not None:
        p, n, uv, idx = asset["positions"], asset["normals"], asset["uvs"], asset["indices"]
    elif obj_path is not None:
        p, n, uv, idx = load_obj(obj_path)
    else:
        p, n, uv, idx = _bent_tube()
    sc.add_new_mesh_to_scene(p, n, uv, idx, pos=(0, -3, 0), rotation=(90, 0, 0), scale_=(1, 1, 1), material_index=0)
    sc.add_new_mesh_to_scene(*_quad((-8, -4.2, 8), (8, -4.2, 8), (8, -4.2, -8), (-8, -4.2, -8), (0, 1, 0)), material_index=1)
    sc.add_new_mesh_to_scene(*_quad((-1.5, 1.5, -1.5), (1.5, 1.5, -1.5), (1.5, 1.5, 1.5), (-1.5, 1.5, 1.5), (0, -1, 0)), material_index=2)
    sc.init_scene_emissive_triangles()
    return sc


def banana_camera(width, height):
    cam = Camera(45.0, 0.1, 100.0)
    cam.on_resize(width, height)
    cam.forward = np.array([-0.6, -0.451, 0.661], dtype=F)          # WalnutApp.cpp:519-520
    cam.set_position((1.752, -0.845, -2.812))
    return cam
