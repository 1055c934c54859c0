"""Shared helpers of the parity tests."""
import numpy as np

from fypraytracer_amd import capi, scenes


def bits_equal(a, b):
    """Bit-level equality of two float arrays, treating any-NaN == any-NaN."""
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    return (a == b) | (np.isnan(a) & np.isnan(b))


def struct_equal(a, b):
    """Per-record equality of two structured arrays (float fields NaN-aware)."""
    ok = np.ones(a.shape, dtype=bool)
    for name in a.dtype.names:
        fa, fb = a[name], b[name]
        if fa.dtype.kind == "f":
            e = bits_equal(fa, fb)
        else:
            e = fa == fb
        if e.ndim > 1:
            e = e.all(axis=tuple(range(1, e.ndim)))
        ok &= e
    return ok


SCENES = {
    "cornell": (lambda: scenes.cornell_box(), scenes.cornell_camera),
    "hall_small": (lambda: scenes.hall_scene_small(), scenes.hall_camera),
    "banana": (lambda: scenes.banana_scene(), scenes.banana_camera),          # textured single-BLAS mesh (config 2 stand-in)
}


def settings_for(tech, **kw):
    base = dict(technique=tech, light_bounces=3, sample_count=2, sky_color=(0.1, 0.2, 0.3), light_candidate_count=4,
                use_temporal_reuse=1, use_spatial_reuse=1, temporal_history_limit=2, spatial_neighbor_num=5,
                spatial_neighbor_radius=30, rand_seed=1)
    base.update(kw)
    return capi.Settings(**base)


def mse_psnr(img_a, img_b):
    """MisUtils::ComputeMSE / ComputePSNR (MisUtils.cpp:118-157): RGB channels of the 8-bit image."""
    a = np.stack([(img_a >> s) & 0xFF for s in (0, 8, 16)], -1).astype(np.float64)
    b = np.stack([(img_b >> s) & 0xFF for s in (0, 8, 16)], -1).astype(np.float64)
    mse = float(np.mean((a - b) ** 2))
    psnr = float("inf") if mse == 0 else 10.0 * np.log10(255.0 * 255.0 / mse)
    return mse, psnr
