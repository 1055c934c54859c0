"""the device builder (tuning key 12: Morton sort + Karras radix tree + wide collapse + refit, rt_lbvh.h).
Same bars as for the host builder: structural invariants of the exported tree, bit-exact parity against the oracle traversing
that exported tree, and the same picture as the host-built tree except for exact-t ties."""
import numpy as np
import pytest

from common import SCENES, bits_equal, check_bvh_invariants, settings_for
from fypraytracer_amd import capi

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["hall_small", "cornell", "banana"])
def test_device_built_tree(oracle_built, name):
    from oraclelib import Oracle
    mk_scene, mk_cam = SCENES[name]
    sc, W, H = mk_scene(), 128, 80
    cam = mk_cam(W, H)
    ctx = capi.Context(0)
    ctx.resize(W, H)
    ctx.set_tuning(12, 1)
    ctx.upload_scene(sc)
    ctx.set_camera(cam)
    bvh = ctx.export_bvh()
    check_bvh_invariants(bvh, sc, require_wide=False)
    assert len(bvh["nodes"]) > 0 and (bvh["nodes"]["meta"] & 7).mean() > 2.5
    for tech in (capi.BRUTE_FORCE, capi.NEE, capi.RESTIR_DI, capi.RESTIR_GI):
        ctx.reset_frame_index()
        orc = Oracle(sc, W, H)
        orc.set_camera(cam)
        orc.use_product_bvh(bvh)
        st = settings_for(tech)
        for f in range(2):
            st.rand_seed = f + 1
            ctx.render(st)
            orc.render(st)
        img, acc = ctx.readback()
        assert bits_equal(acc, orc.accum()).all() and np.array_equal(img, orc.image()), tech
    # moving a mesh afterwards refits the device-built tree as well
    mgr = sc.manager(); mgr.perform_all_scene_updates(sc)
    mgr.set_mesh_transform(sc, 0, pos=(0.1, -0.05, 0.2))
    mgr.perform_all_scene_updates(sc)
    ctx.update_vertices(sc)
    check_bvh_invariants(ctx.export_bvh(), sc, require_wide=False)
    # host-built tree: same picture except exact-t ties
    host = capi.Context(0)
    host.resize(W, H)
    host.upload_scene(sc)
    host.set_camera(cam)
    ctx.reset_frame_index()
    st = settings_for(capi.COSINE_WEIGHTED_SAMPLING)
    ctx.render(st); host.render(st)
    assert bits_equal(ctx.readback()[1], host.readback()[1]).all(axis=-1).mean() > 0.97
    ctx.close(); host.close()
