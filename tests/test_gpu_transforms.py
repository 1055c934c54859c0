"""SURVEY §8 f-1 / f-3 on the device:
  * fyprt_update_transforms — a transform edit applied ON THE DEVICE (64 bytes per mesh cross the bus; world vertices, records, tree
    boxes, light records recomputed there; only the moved emissive meshes' light trees rebuilt on the host) must leave the context in
    exactly the state fyprt_update_vertices reaches from the host-computed world vertices (Scene.cpp:42-51 via scene.py) — exported
    tree, light trees, frames — which tests/test_gpu_refit.py compares with the oracle;
  * fyprt_compare_image — MisUtils::ComputeMSE / ComputePSNR (MisUtils.cpp:118-157) as a device reduction equals misutils.py exactly."""
import numpy as np
import pytest

from common import SCENES, bits_equal, settings_for, struct_equal
from fypraytracer_amd import capi, misutils

pytestmark = pytest.mark.gpu


def _ctx(sc, cam, W, H):
    ctx = capi.Context(0)
    ctx.resize(W, H)
    ctx.upload_scene(sc)
    ctx.set_camera(cam)
    return ctx


@pytest.mark.parametrize("name,moves", [("hall_small", [(3, dict(pos=(0.7, 0.0, -0.4), rotation=(0, 25, 0))), (-1, dict(pos=(0.0, -0.3, 0.2), rotation=(10, 0, 5)))]),
                                        ("cornell", [(5, dict(pos=(0.2, 0.0, 0.1), rotation=(0, 30, 0))), (7, dict(pos=(0.15, 0.0, -0.1), scale_=(1.1, 0.8, 1.0)))])])
def test_device_transform_update_equals_host_vertex_update(name, moves):
    mk_scene, mk_cam = SCENES[name]
    W, H = 128, 80
    cam = mk_cam(W, H)
    sc = mk_scene()
    mgr = sc.manager()
    mgr.perform_all_scene_updates(sc)
    a, b = _ctx(sc, cam, W, H), _ctx(sc, cam, W, H)
    b.set_object_vertices(sc)
    st = settings_for(capi.RESTIR_DI)
    for c in (a, b):
        c.render(st)                                         # a frame of the old geometry: history the edit must survive identically
    moved = []
    for mesh, tr in moves:
        m = mesh % len(sc.meshes)
        mgr.set_mesh_transform(sc, m, **tr)
        moved.append(m)
    assert mgr.perform_all_scene_updates(sc) is True         # host path: world vertices recomputed by scene.py (numpy float32)
    a.update_vertices(sc)                                    # uploads every vertex
    b.update_transforms(sc, moved)                           # uploads 64 bytes per moved mesh
    ba, bb = a.export_bvh(), b.export_bvh()
    assert ba["nodes"].tobytes() == bb["nodes"].tobytes() and ba["tris"].tobytes() == bb["tris"].tobytes()
    la, lb = a.export_lighttrees(len(sc.meshes)), b.export_lighttrees(len(sc.meshes))
    for k in ("tlas", "blas"):
        assert struct_equal(la[k], lb[k]).all(), k
    assert la["tlas_root"] == lb["tlas_root"] and np.array_equal(la["blas_root"], lb["blas_root"])
    for tech in (capi.RESTIR_DI, capi.NEE, capi.BRDF_SAMPLING):
        s2 = settings_for(tech)
        for f in range(2):
            s2.rand_seed = f + 2
            a.render(s2)
            b.render(s2)
        (ia, aa), (ib, ab) = a.readback(), b.readback()
        assert np.array_equal(ia, ib) and bits_equal(aa, ab).all(), tech
    a.close()
    b.close()


def test_update_transforms_argument_checks():
    mk_scene, mk_cam = SCENES["cornell"]
    sc = mk_scene()
    ctx = _ctx(sc, mk_cam(64, 64), 64, 64)
    with pytest.raises(capi.FyprtError):
        ctx.update_transforms(sc, [0])                       # object vertices not set
    ctx.set_object_vertices(sc)
    with pytest.raises(capi.FyprtError):
        ctx._check(ctx.lib.fyprt_update_transforms(ctx.h, (__import__("ctypes").c_uint32 * 1)(len(sc.meshes)), (__import__("ctypes").c_float * 16)(), 1))   # mesh index out of range
    ctx.close()


def test_device_mse_psnr_equals_misutils():
    mk_scene, mk_cam = SCENES["cornell"]
    W, H = 160, 96
    sc, cam = mk_scene(), mk_cam(W, H)
    ctx = _ctx(sc, cam, W, H)
    st = settings_for(capi.BRUTE_FORCE)
    for f in range(64):
        st.rand_seed = f + 1
        ctx.render(st)
    ref, _ = ctx.readback(want_accum=False)
    ctx.reset_frame_index()
    st2 = settings_for(capi.COSINE_WEIGHTED_SAMPLING)
    ctx.render(st2)
    img, _ = ctx.readback(want_accum=False)
    for flip in (False, True):
        mse, psnr = ctx.compare_image(ref, flip_reference_rows=flip)
        want = misutils.compute_mse(ref if flip else ref[::-1], img)      # compute_mse reads its original flipped (MisUtils.cpp:118-147)
        assert mse == want and mse > 0
        assert psnr == misutils.compute_psnr(want)
    mse0, psnr0 = ctx.compare_image(img)
    assert mse0 == 0.0 and psnr0 == float("inf")
    ctx.set_rows(16, 48, 0)                                  # a band compares its own rows only
    mse_b, _ = ctx.compare_image(ref)
    assert mse_b == misutils.compute_mse(ref[16:48][::-1], img[16:48])
    ctx.close()
