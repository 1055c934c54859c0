"""Edge-case scenes through the real kernels, each against the oracle running the same traversal (bit-exact):
a scene that is a single leaf (no inner node at all), degenerate (zero-area) and needle triangles, coordinates far from the
origin (coarse fp32 grid under the quantised boxes), a camera inside geometry, and the state errors for empty input."""
import numpy as np
import pytest

from common import bits_equal, settings_for
from fypraytracer_amd import capi, scenes
from fypraytracer_amd.scene import Material, Scene

pytestmark = pytest.mark.gpu


def _scene(tris, emissive_quad=True, offset=(0.0, 0.0, 0.0)):
    """tris: (n, 3, 3) float32 positions -> one mesh of n triangles (+ a small emissive quad above, its own mesh)."""
    sc = Scene()
    sc.materials = [Material(albedo=(0.8, 0.7, 0.6), roughness=0.7), Material(albedo=(1, 1, 1), emission_color=(1, 1, 1), emission_power=20.0)]
    off = np.asarray(offset, np.float32)
    t = np.asarray(tris, np.float32) + off
    n = len(t)
    pos = t.reshape(-1, 3)
    e1, e2 = t[:, 1] - t[:, 0], t[:, 2] - t[:, 0]
    nrm = np.cross(e1, e2)
    ln = np.linalg.norm(nrm, axis=1, keepdims=True)
    nrm = np.where(ln > 0, nrm / np.maximum(ln, 1e-30), np.array([0, 1, 0], np.float32)).astype(np.float32)
    sc.add_new_mesh_to_scene(pos, np.repeat(nrm, 3, axis=0), np.zeros((n * 3, 2), np.float32), np.arange(n * 3, dtype=np.uint32).reshape(-1, 3), material_index=0)
    if emissive_quad:
        q = np.array([[-0.3, 1.5, -0.3], [0.3, 1.5, -0.3], [0.3, 1.5, 0.3], [-0.3, 1.5, 0.3]], np.float32) + off
        sc.add_new_mesh_to_scene(q, np.tile(np.array([0, -1, 0], np.float32), (4, 1)), np.zeros((4, 2), np.float32),
                                 np.array([[0, 1, 2], [0, 2, 3]], np.uint32), material_index=1)
    sc.init_scene_emissive_triangles()
    return sc


def _camera(W, H, pos, offset=(0.0, 0.0, 0.0)):
    cam = scenes.cornell_camera(W, H)
    cam.set_position(tuple(np.asarray(pos, np.float64) + np.asarray(offset, np.float64)))
    return cam


def _check(sc, cam, W, H, techs, frames=2):
    from oraclelib import Oracle
    for tech in techs:
        ctx = capi.Context(0)
        ctx.resize(W, H)
        ctx.upload_scene(sc)
        ctx.set_camera(cam)
        orc = Oracle(sc, W, H)
        orc.set_camera(cam)
        orc.use_product_bvh(ctx.export_bvh())
        st = settings_for(tech)
        for f in range(frames):
            st.rand_seed = f + 1
            ctx.render(st)
            orc.render(st)
        img, acc = ctx.readback()
        assert bits_equal(acc, orc.accum()).all(), tech
        assert np.array_equal(img, orc.image()), tech
        ctx.close()


FLOOR = [[[-2, 0, -2], [2, 0, -2], [2, 0, 2]], [[-2, 0, -2], [2, 0, 2], [-2, 0, 2]]]
TECHS = [capi.BRUTE_FORCE, capi.NEE, capi.RESTIR_DI, capi.RESTIR_GI]


def test_single_leaf_scene(oracle_built):
    """Two triangles in one mesh, no light mesh: the root reference is a leaf code, there is no node to fetch."""
    sc = _scene(FLOOR, emissive_quad=False)
    ctx = capi.Context(-1)
    ctx.upload_scene(sc)
    b = ctx.export_bvh()
    assert b["root"] < 0 and len(b["nodes"]) == 0
    ctx.close()
    _check(sc, _camera(48, 32, (0, 1.0, 3.0)), 48, 32, [capi.BRUTE_FORCE, capi.COSINE_WEIGHTED_SAMPLING])


def test_degenerate_and_needle_triangles(oracle_built):
    rng = np.random.default_rng(3)
    tris = list(FLOOR)
    for _ in range(40):
        p = rng.uniform(-1, 1, 3); p[1] = abs(p[1]) * 0.8 + 0.05
        tris.append([p, p, p])                                           # a point
        q = p + rng.normal(size=3) * 0.3
        tris.append([p, q, p + (q - p) * 0.5])                           # collinear: zero area
        tris.append([p, q, q + rng.normal(size=3) * 1e-6])               # needle
    _check(_scene(np.array(tris, np.float32)), _camera(64, 48, (0, 0.8, 3.0)), 64, 48, TECHS)


def test_far_from_the_origin(oracle_built):
    """The whole scene translated by (4096, -2048, 8192): fp32 spacing there is 2^-11..2^-10, far coarser than the
    quantisation grid of small nodes — the builder must still produce conservative boxes (it verifies them in fp32)."""
    rng = np.random.default_rng(5)
    tris = list(FLOOR)
    for _ in range(300):
        c = rng.uniform(-1.5, 1.5, 3); c[1] = abs(c[1]) * 0.6 + 0.02
        tris.append(c + rng.normal(size=(3, 3)) * 0.05)
    off = (4096.0, -2048.0, 8192.0)
    _check(_scene(np.array(tris, np.float32), offset=off), _camera(64, 48, (0, 0.8, 3.0), off), 64, 48, TECHS)


def test_camera_inside_a_closed_box(oracle_built):
    sc = scenes.cornell_box()
    cam = scenes.cornell_camera(64, 64)
    cam.set_position((0.0, 0.0, 0.5))                                    # between the two inner boxes
    _check(sc, cam, 64, 64, TECHS)


def test_empty_scene_renders_the_sky():
    """No triangles at all (triCount == 0: every ray is a miss): the white-sky known answer 0xFF7F7F7F everywhere for the
    techniques that need no emitter; the light-based ones refuse with FYPRT_ENOLIGHT (deterministic reading R6)."""
    sc = Scene()
    sc.materials = [Material(albedo=(1, 1, 1))]
    ctx = capi.Context(0)
    ctx.resize(32, 16)
    ctx.upload_scene(sc)
    ctx.set_camera(scenes.cornell_camera(32, 16))
    for tech in range(9):
        ctx.reset_frame_index()
        try:
            ctx.render(settings_for(tech, sky_color=(1.0, 1.0, 1.0)))
        except capi.FyprtError as e:
            assert tech in (capi.LIGHT_SOURCE_SAMPLING, capi.NEE, capi.RESTIR_DI, capi.RESTIR_GI) and "emissive" in str(e).lower() or "light" in str(e).lower(), (tech, str(e))
            continue
        img, _ = ctx.readback()
        assert (img == 0xFF7F7F7F).all(), (tech, hex(int(img[0, 0])))
    ctx.close()
