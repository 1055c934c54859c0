"""The device-side instrumentation (rays / node visits / box tests / triangle tests / hits, fyprt_set_ray_counting)
must equal the oracle's instrumented restatement of the same traversal — these counts define the
algorithmic bytes of the roofline, so they are checked exactly."""
import pytest

from common import SCENES, settings_for
from fypraytracer_amd import capi

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tech", [capi.BRUTE_FORCE, capi.NEE, capi.RESTIR_DI, capi.RESTIR_GI])
def test_counters_match_oracle(oracle_built, tech):
    from oraclelib import Oracle
    mk_scene, mk_cam = SCENES["hall_small"]
    sc, W, H = mk_scene(), 128, 72
    cam = mk_cam(W, H)
    ctx = capi.Context(0)
    ctx.resize(W, H)
    ctx.upload_scene(sc)
    ctx.set_camera(cam)
    ctx.set_ray_counting(True)
    orc = Oracle(sc, W, H)
    orc.set_camera(cam)
    orc.use_product_bvh(ctx.export_bvh())
    st = settings_for(tech)
    for f in range(2):
        st.rand_seed = f + 1
        g = ctx.render(st)
        o = orc.render(st)
        assert (g.rays, g.box_tests, g.tri_tests, g.hits, g.node_visits) == (o["rays"], o["box_tests"], o["tri_tests"], o["hits"], o["node_visits"])
        assert sum(g.part_node_visits) == g.node_visits and 0 < g.node_visits < g.box_tests
    if tech == capi.RESTIR_DI:
        assert g.part_rays[0] == W * H                 # one primary ray per pixel in Part 1
        assert sum(g.part_rays) == g.rays              # Part 2's rays are counted in its trace launch
    ctx.close()
