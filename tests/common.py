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
    "banana": (lambda: scenes.banana_scene(), scenes.banana_camera),          # config 2: the reference's banana mesh + 2048x2048 texture (data fixture tests/golden/banana_asset.npz)
    "banana_standin": (lambda: scenes.banana_scene(fixture=None), scenes.banana_camera),   # procedural stand-in of the same size
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


def check_bvh_invariants(b, sc, require_wide=True):
    """Structural checks of an exported acceleration structure (fyprt_export_bvh) against the scene it was built / refitted for:
    every triangle in exactly one leaf record, leaf records = (v0, v1 - v0, v2 - v0) bit for bit, every quantised child box
    contains everything below it, grid anchored at the node's own box, unused slots inverted, level counts as recorded, every
    node reachable exactly once."""
    import sys
    nodes, tris = b["nodes"], b["tris"]
    assert sorted(tris["tri"].tolist()) == list(range(len(sc.triangles)))   # every triangle exactly once
    assert b["max_stack"] <= 31                                              # wide levels: what node_step's stack rule needs (<= 31)
    pos = sc.world_vertices["position"]
    t = sc.triangles
    i = tris["tri"]
    assert np.array_equal(tris["v0"], pos[t["v0"][i]])
    assert np.array_equal(tris["e1"], pos[t["v1"][i]] - pos[t["v0"][i]])
    assert np.array_equal(tris["e2"], pos[t["v2"][i]] - pos[t["v0"][i]])
    tri_lo = np.minimum(np.minimum(pos[t["v0"]], pos[t["v1"]]), pos[t["v2"]])
    tri_hi = np.maximum(np.maximum(pos[t["v0"]], pos[t["v1"]]), pos[t["v2"]])
    seen = np.zeros(len(nodes), dtype=int)
    assert (((nodes["meta"] & 7) >= 2) & ((nodes["meta"] & 7) <= 4)).all() and len(nodes) < (1 << 26)

    def bounds(ref):
        if ref >= 0:
            seen[ref] += 1
            n = nodes[ref]
            k = int(n["meta"]) & 7
            step = np.ldexp(np.float32(1.0), n["ex"].astype(np.int32) - 127).astype(np.float32)
            lo_all, hi_all, need = [], [], 0
            for c in range(k):
                l, h, cn = bounds(int(n["child"][c]))
                qlo = (n["origin"] + n["qlo"][:, c].astype(np.float32) * step).astype(np.float32)   # exact: q * 2^e, then one rounding
                qhi = (n["origin"] + n["qhi"][:, c].astype(np.float32) * step).astype(np.float32)
                assert (qlo <= l).all() and (qhi >= h).all()
                lo_all.append(l); hi_all.append(h); need = max(need, cn)
            assert (n["qlo"][:, k:] == 255).all() and (n["qhi"][:, k:] == 0).all()                   # unused slots: inverted box
            assert (n["origin"] == np.min(lo_all, axis=0)).all()                                      # grid anchored at the node's own box
            assert int(n["meta"]) >> 3 == 1 + need                                                    # recorded level count
            return np.min(lo_all, axis=0), np.max(hi_all, axis=0), 1 + need
        code = ~ref
        first, cnt = code >> 2, (code & 3) + 1
        ids = tris["tri"][first:first + cnt]
        return tri_lo[ids].min(0), tri_hi[ids].max(0), 0

    old = sys.getrecursionlimit()
    sys.setrecursionlimit(10000)
    try:
        _, _, need = bounds(b["root"])
    finally:
        sys.setrecursionlimit(old)
    assert (seen == 1).all()
    assert need == b["max_stack"] <= 31
    if require_wide and len(nodes):
        assert (nodes["meta"] & 7).mean() > 3.0                              # the collapse really is wide
