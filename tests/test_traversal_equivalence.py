"""CPU: the product's acceleration structure + traversal order (restated in oracle_product_trace.h over
the arrays exported through the C ABI) must find the same closest hit as the reference's TLAS/BLAS
traversal (Renderer.cu:460-561) — same Möller–Trumbore arithmetic, so t/u/v/normal are bit-identical
whenever the same triangle wins; only exact-t ties may pick the other triangle."""
import numpy as np
import pytest

from common import struct_equal
from fypraytracer_amd import capi, scenes
from oraclelib import Oracle


@pytest.mark.parametrize("tight_stack", [False, True])
@pytest.mark.parametrize("name,min_same", [("hall_small", 0.9995), ("cornell", 0.99)])
def test_random_rays_same_closest_hit(oracle_built, name, min_same, tight_stack):
    """tight_stack: the stack rule's budget is lowered to the tree's level count, so nearly every multi-hit visit goes
    through a resume entry (one pending entry per level) instead of pushing the siblings one by one."""
    sc = scenes.cornell_box() if name == "cornell" else scenes.hall_scene_small()
    ctx = capi.Context(-1)
    ctx.upload_scene(sc)
    bvh = ctx.export_bvh()
    o = Oracle(sc, 8, 8)
    rng = np.random.default_rng(42)
    lo = sc.world_vertices["position"].min(0)
    hi = sc.world_vertices["position"].max(0)
    n = 4000
    origins = rng.uniform(lo + 0.05 * (hi - lo), hi - 0.05 * (hi - lo), (n, 3)).astype(np.float32)
    dirs = rng.normal(size=(n, 3))
    dirs = (dirs / np.linalg.norm(dirs, axis=1, keepdims=True)).astype(np.float32)
    dirs[:50, 0] = 0.0                                                # axis-parallel components (d == 0 branch of the slab test)
    dirs[50:100, 1] = 0.0
    ref = np.zeros(n, dtype=capi.PAYLOAD_DTYPE)
    for i in range(n):
        ref[i], _ = o.trace(origins[i], dirs[i])
    o.use_product_bvh(bvh)
    if tight_stack:
        o.set_product_stack_budget(bvh["max_stack"])
    got = np.zeros(n, dtype=capi.PAYLOAD_DTYPE)
    box = tri = 0
    for i in range(n):
        got[i], c = o.trace(origins[i], dirs[i])
        box += c["box_tests"]; tri += c["tri_tests"]
    same = struct_equal(ref, got)
    assert same.mean() >= min_same, f"{(~same).sum()} of {n} rays differ"
    # the differing rays are ties: same distance to within an ulp, both hit
    d = ~same
    assert np.allclose(ref["hitDistance"][d], got["hitDistance"][d], rtol=1e-6)
    assert (ref["hitDistance"] > 0).mean() > 0.8                      # (the Cornell box is open at the front)
    assert box / n < (400 if tight_stack else 200) and tri / n < 20   # ordered + culled traversal stays cheap
    assert o.product_max_stack() <= (bvh["max_stack"] if tight_stack else 31)   # pending entries never exceed the budget
    ctx.close()


def test_full_frame_reference_vs_product_order(oracle_built):
    sc, W, H = scenes.hall_scene_small(), 96, 54
    cam = scenes.hall_camera(W, H)
    ctx = capi.Context(-1)
    ctx.upload_scene(sc)
    from common import settings_for
    outs = []
    for product in (False, True):
        o = Oracle(sc, W, H)
        o.set_camera(cam)
        if product:
            o.use_product_bvh(ctx.export_bvh())
        for f in range(2):
            o.render(settings_for(capi.RESTIR_DI, rand_seed=f + 1))
        outs.append(o.accum())
    same = ((outs[0] == outs[1]) | (np.isnan(outs[0]) & np.isnan(outs[1]))).all(-1)
    assert same.mean() >= 0.999
    ctx.close()


def _brute_force_closest(sc, origins, dirs):
    """Closest hit by testing EVERY triangle — numpy float32, the reference's Möller–Trumbore operation order (Renderer.cu:513-537):
    no acceleration structure, no code shared with either tracer.  Returns the closest t per ray (-1: miss)."""
    F = np.float32
    pos = sc.world_vertices["position"].astype(F)
    t = sc.triangles
    v0, v1, v2 = pos[t["v0"]], pos[t["v1"]], pos[t["v2"]]
    e1, e2 = (v1 - v0).astype(F), (v2 - v0).astype(F)

    def cross(a, b):
        return np.stack([a[..., 1] * b[..., 2] - b[..., 1] * a[..., 2], a[..., 2] * b[..., 0] - b[..., 2] * a[..., 0], a[..., 0] * b[..., 1] - b[..., 0] * a[..., 1]], -1).astype(F)

    def dot(a, b):
        return ((a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1]) + a[..., 2] * b[..., 2]).astype(F)
    out = np.full(len(origins), -1.0, dtype=F)
    with np.errstate(all="ignore"):
        for i, (o, d) in enumerate(zip(origins, dirs)):
            h = cross(np.broadcast_to(d, e2.shape), e2)
            f = (F(1.0) / dot(e1, h)).astype(F)
            s = (o[None, :] - v0).astype(F)
            u = (f * dot(s, h)).astype(F)
            q = cross(s, e1)
            v = (f * dot(np.broadcast_to(d, q.shape), q)).astype(F)
            tt = (f * dot(e2, q)).astype(F)
            ok = ~((u < 0) | (u > 1)) & ~((v < 0) | ((u + v).astype(F) > 1)) & (tt > F(0.0001))
            if ok.any():
                out[i] = tt[ok].min()
    return out


@pytest.mark.parametrize("name", ["hall_small", "cornell"])
def test_both_tracers_find_the_brute_force_closest_hit(oracle_built, name):
    """An independent pin of the oracle's two traversals (and, through the bit-exact GPU tests, of the HIP traversal): neither the
    reference-order TLAS/BLAS walk nor the product's quantised 4-wide walk may miss the triangle an exhaustive test finds — the hit
    DISTANCE must agree bit for bit (which of two triangles at exactly that distance is reported is the tie freedom of DESIGN.md §5)."""
    sc = scenes.cornell_box() if name == "cornell" else scenes.hall_scene_small()
    ctx = capi.Context(-1)
    ctx.upload_scene(sc)
    o = Oracle(sc, 8, 8)
    rng = np.random.default_rng(7)
    lo, hi = sc.world_vertices["position"].min(0), sc.world_vertices["position"].max(0)
    n = 600
    origins = rng.uniform(lo + 0.05 * (hi - lo), hi - 0.05 * (hi - lo), (n, 3)).astype(np.float32)
    dirs = rng.normal(size=(n, 3))
    dirs = (dirs / np.linalg.norm(dirs, axis=1, keepdims=True)).astype(np.float32)
    want = _brute_force_closest(sc, origins, dirs)
    for product in (False, True):
        if product:
            o.use_product_bvh(ctx.export_bvh())
        got = np.array([o.trace(origins[i], dirs[i])[0]["hitDistance"] for i in range(n)], dtype=np.float32)
        assert np.array_equal(got, want), (product, int((got != want).sum()))
    assert (want > 0).mean() > 0.7
    ctx.close()
