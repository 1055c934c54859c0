"""SURVEY §8 f-2 (dynamic scenes): fyprt_update_vertices refits the acceleration structure on the device after a transform edit
(the reference rebuilds BLAS + TLAS + light trees on the host, SceneManager.cpp:83-129).  The refitted tree keeps its shape, so
parity is checked the usual way — the oracle traverses the EXPORTED (refitted) tree over the moved scene, bit-exact — plus the
structural invariants of the exported tree and the agreement with a context that uploaded the moved scene from scratch."""
import numpy as np
import pytest

from common import SCENES, bits_equal, check_bvh_invariants, settings_for, struct_equal
from fypraytracer_amd import capi

pytestmark = pytest.mark.gpu


def _moved_scene(name):
    mk_scene, mk_cam = SCENES[name]
    sc = mk_scene()
    m = sc.manager()
    m.perform_all_scene_updates(sc)                         # the reference's first call (queues start non-empty)
    return sc, m, mk_cam


@pytest.mark.parametrize("name,moves", [("hall_small", [(3, dict(pos=(0.7, 0.0, -0.4), rotation=(0, 25, 0))), (-1, dict(pos=(0.0, -0.3, 0.2)))]),
                                        ("cornell", [(5, dict(pos=(0.2, 0.0, 0.1), rotation=(0, 30, 0))), (7, dict(pos=(0.15, 0.0, -0.1)))]),
                                        ("banana", [(0, dict(pos=(0.3, -3.0, 0.2), rotation=(90, 40, 10), scale_=(1.2, 0.9, 1.0)))])])
def test_refit_equals_oracle_on_the_moved_scene(oracle_built, name, moves):
    from oraclelib import Oracle
    sc, mgr, mk_cam = _moved_scene(name)
    W, H = 128, 80
    cam = mk_cam(W, H)
    ctx = capi.Context(0)
    ctx.resize(W, H)
    ctx.upload_scene(sc)
    ctx.set_camera(cam)
    ctx.render(settings_for(capi.COSINE_WEIGHTED_SAMPLING))   # a frame of the old geometry first (no ReSTIR history: the oracle below starts fresh)
    for mesh, tr in moves:
        mgr.set_mesh_transform(sc, mesh % len(sc.meshes), **tr)
    assert mgr.perform_all_scene_updates(sc) is True
    ctx.update_vertices(sc)
    bvh = ctx.export_bvh()
    check_bvh_invariants(bvh, sc, require_wide=False)
    for tech in (capi.BRUTE_FORCE, capi.NEE, capi.RESTIR_DI, capi.RESTIR_GI):
        ctx.reset_frame_index()
        orc = Oracle(sc, W, H)
        orc.set_camera(cam)
        orc.use_product_bvh(bvh)
        if tech == capi.NEE:
            lt_p, lt_o = ctx.export_lighttrees(len(sc.meshes)), orc.export_lighttrees()
            for k in ("tlas", "blas"):
                assert struct_equal(lt_p[k], lt_o[k]).all()
        st = settings_for(tech)
        for f in range(2):
            st.rand_seed = f + 1
            ctx.render(st)
            orc.render(st)
        img, acc = ctx.readback()
        assert bits_equal(acc, orc.accum()).all() and np.array_equal(img, orc.image()), tech
    # a context that uploads the moved scene from scratch builds a different tree: same picture except exact-t ties
    fresh = capi.Context(0)
    fresh.resize(W, H)
    fresh.upload_scene(sc)
    fresh.set_camera(cam)
    ctx.reset_frame_index()
    st = settings_for(capi.COSINE_WEIGHTED_SAMPLING)
    ctx.render(st); fresh.render(st)
    a, b = ctx.readback()[1], fresh.readback()[1]
    assert bits_equal(a, b).all(axis=-1).mean() > 0.97
    ctx.close(); fresh.close()


def test_refit_errors():
    mk_scene, _ = SCENES["cornell"]
    sc = mk_scene()
    ctx = capi.Context(0)
    with pytest.raises(capi.FyprtError):
        ctx.update_vertices(sc)                                             # no scene yet
    ctx.upload_scene(sc)
    short = mk_scene()
    short.world_vertices = short.world_vertices[:-1]
    with pytest.raises(capi.FyprtError, match="vertex count"):
        ctx.update_vertices(short)
    ctx.close()
    host = capi.Context(-1)
    host.upload_scene(sc)
    with pytest.raises(capi.FyprtError, match="device"):
        host.update_vertices(sc)
    host.close()
