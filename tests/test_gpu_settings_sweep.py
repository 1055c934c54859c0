"""The wavefront stages are state machines over (sample, bounce / neighbour) counters: sweep the settings that shape them — zero and
one bounce, one and several samples, more than eight ray steps per frame (the early-exit read-back path), NEE with a single bounce
(one ray per step instead of two), ReSTIR without reuse / with no neighbours / a single candidate — against the oracle, bit-exact."""
import numpy as np
import pytest

from common import SCENES, bits_equal, settings_for
from fypraytracer_amd import capi

pytestmark = pytest.mark.gpu

SWEEP = [
    (capi.BRUTE_FORCE, dict(light_bounces=0)), (capi.BRUTE_FORCE, dict(light_bounces=1)), (capi.BRUTE_FORCE, dict(light_bounces=6)),
    (capi.UNIFORM_SAMPLING, dict(light_bounces=0, sample_count=3)), (capi.COSINE_WEIGHTED_SAMPLING, dict(light_bounces=4, sample_count=5)),   # 20 steps
    (capi.GGX_SAMPLING, dict(light_bounces=2, sample_count=1)), (capi.BRDF_SAMPLING, dict(light_bounces=1, sample_count=4)),
    (capi.BRDF_SAMPLING, dict(light_bounces=3, sample_count=0)),                                                                          # 0 / 0 -> NaN -> black
    (capi.LIGHT_SOURCE_SAMPLING, dict(sample_count=1)), (capi.LIGHT_SOURCE_SAMPLING, dict(sample_count=4)),
    (capi.NEE, dict(light_bounces=1, sample_count=3)), (capi.NEE, dict(light_bounces=0, sample_count=2)), (capi.NEE, dict(light_bounces=4, sample_count=3)),   # 12 steps
    (capi.RESTIR_DI, dict(use_temporal_reuse=0, use_spatial_reuse=0)), (capi.RESTIR_DI, dict(spatial_neighbor_num=0)), (capi.RESTIR_DI, dict(light_candidate_count=1, spatial_neighbor_num=2, spatial_neighbor_radius=7)),
    (capi.RESTIR_DI, dict(spatial_neighbor_num=9)),                                                                                      # more neighbours than the speculative gather handles
    (capi.RESTIR_GI, dict(light_bounces=0)), (capi.RESTIR_GI, dict(light_bounces=1, use_spatial_reuse=0)), (capi.RESTIR_GI, dict(light_bounces=4, spatial_neighbor_num=1)),
    (capi.RESTIR_GI, dict(use_temporal_reuse=0, spatial_neighbor_num=8, spatial_neighbor_radius=3)),
]


@pytest.mark.parametrize("scene_name", ["cornell", "hall_small"])
def test_settings_sweep_bit_exact(oracle_built, scene_name):
    from oraclelib import Oracle
    mk_scene, mk_cam = SCENES[scene_name]
    W, H = 80, 56
    sc, cam = mk_scene(), mk_cam(W, H)
    ctx = capi.Context(0)
    ctx.resize(W, H)
    ctx.upload_scene(sc)
    ctx.set_camera(cam)
    bvh = ctx.export_bvh()
    for tech, kw in SWEEP:
        for key14 in ((0, 1) if tech == capi.RESTIR_DI else (0,)):
            ctx.set_tuning(14, key14)
            ctx.resize(W, H)                                 # zero-filled ReSTIR history, as the fresh oracle below has
            ctx.set_camera(cam)
            orc = Oracle(sc, W, H)
            orc.set_camera(cam)
            orc.use_product_bvh(bvh)
            st = settings_for(tech, **kw)
            for f in range(2):
                st.rand_seed = f + 1
                ctx.render(st)
                orc.render(st)
            img, acc = ctx.readback()
            eq = bits_equal(acc, orc.accum()).all(axis=-1)
            assert eq.all(), (capi.TECHNIQUE_NAMES[tech], kw, key14, int((~eq).sum()))
            assert np.array_equal(img, orc.image()), (capi.TECHNIQUE_NAMES[tech], kw)
            orc.close()
    ctx.close()
