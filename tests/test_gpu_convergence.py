"""The convergence pin of tests/test_oracle_convergence.py (the reference's own accumulate-and-compare workflow,
WalnutApp.cpp:826-876) through the HIP kernels: long accumulations of every unbiased technique converge to the brute-force
image on the Cornell box.  Same cases and thresholds as the oracle test; 128x128, more frames (they are ~30 us each here)."""
import numpy as np
import pytest

from common import mse_psnr, settings_for
from fypraytracer_amd import capi, scenes
from test_oracle_convergence import CASES

pytestmark = pytest.mark.gpu
W = H = 128


def _accumulate(sc, cam, tech, frames, **kw):
    ctx = capi.Context(0)
    ctx.resize(W, H)
    ctx.upload_scene(sc)
    ctx.set_camera(cam)
    st = settings_for(tech, sky_color=(0.0, 0.0, 0.0), **kw)
    for f in range(frames):
        st.rand_seed = f + 1
        ctx.render_async(st)
    ctx.synchronize()
    img, acc = ctx.readback()
    ctx.close()
    return img, float(acc[..., :3].mean()) / frames


@pytest.mark.parametrize("bounces", [1, 3])
def test_unbiased_techniques_converge_to_brute_force_on_gpu(bounces):
    sc, cam = scenes.cornell_box(), scenes.cornell_camera(W, H)
    ref_img, ref_mean = _accumulate(sc, cam, capi.BRUTE_FORCE, 6000, light_bounces=bounces, sample_count=1)
    for tech, kw, max_rel, min_psnr in CASES[bounces]:
        img, mean = _accumulate(sc, cam, tech, 1000, light_bounces=bounces, sample_count=2, **kw)
        _, psnr = mse_psnr(img, ref_img)
        assert abs(mean - ref_mean) / ref_mean < max_rel, (capi.TECHNIQUE_NAMES[tech], kw, mean, ref_mean)
        assert psnr >= min_psnr, (capi.TECHNIQUE_NAMES[tech], kw, psnr)
