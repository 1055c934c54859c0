"""Every result-neutral tuning knob (fyprt_set_tuning) must leave every output bit unchanged: tile order, fused vs
wavefront Part 2, persistent grid size, light-sorted task order, chunk size, static chunks, refill threshold, node-loop quorum."""
import numpy as np
import pytest

from common import SCENES, settings_for
from fypraytracer_amd import capi

pytestmark = pytest.mark.gpu

VARIANTS = [
    {},                                             # defaults
    {0: 0}, {0: 1},                                 # tile order: linear, one eighth per XCD
    {1: 0},                                         # Part 2 fused in one kernel
    {1: 0, 6: 0},                                   # ... with the classic while-while loop
    {2: 2}, {4: 32, 5: 8}, {4: 256, 5: 48},         # persistent grid / chunk / refill
    {2: 1, 4: 16, 5: 8}, {2: 1, 4: 16, 9: 3},       # small grid + small chunks: the queue is longer than the static chunks -> dynamic stealing runs
    {9: 4}, {2: 1, 4: 64, 10: 16}, {11: 0},          # static chunks per wave; guided claims down to 16 tasks; no two-stream pipelining
    {3: 1},                                         # shadow tasks sorted by light bin
    {6: 0}, {6: 40}, {7: 16},                       # node-loop quorum (shadow rays / every other kernel)
    {14: 1}, {14: 2},                               # Part-2 setup: speculative neighbour gathers / neighbourhood hot fields in LDS
    {15: 1}, {15: 2},                               # wavefront stages' ray kernel: persistent / one thread per ray
    {18: 0}, {18: 0, 1: 0},                         # every ReSTIR DI shadow ray traced, also those whose pixel is black in every outcome (the default skips them)
]


@pytest.mark.parametrize("tech", [capi.RESTIR_DI, capi.NEE])
def test_knobs_do_not_change_results(tech):
    mk_scene, mk_cam = SCENES["hall_small"]
    sc, W, H = mk_scene(), 176, 104                  # not a multiple of 16: partial tiles
    cam = mk_cam(W, H)
    ref = None
    for knobs in VARIANTS:
        ctx = capi.Context(0)
        ctx.resize(W, H)
        ctx.upload_scene(sc)
        ctx.set_camera(cam)
        for k, v in knobs.items():
            ctx.set_tuning(k, v)
        st = settings_for(tech)
        for f in range(3):
            st.rand_seed = f + 1
            ctx.render(st)
        img, acc = ctx.readback()
        ctx.close()
        if ref is None:
            ref = (img, acc)
        else:
            assert np.array_equal(img, ref[0]), knobs
            assert np.array_equal(acc, ref[1], equal_nan=True), knobs


@pytest.mark.parametrize("scene_name", ["cornell", "banana"])
@pytest.mark.parametrize("tech", [capi.BRUTE_FORCE, capi.UNIFORM_SAMPLING, capi.COSINE_WEIGHTED_SAMPLING, capi.GGX_SAMPLING, capi.BRDF_SAMPLING, capi.LIGHT_SOURCE_SAMPLING])
def test_fused_small_scene_frame_equals_the_stage_frame(scene_name, tech):
    """Techniques 0-5 on a small tree run as ONE launch (k_path_fused, tuning key 17 = 2 / auto) — the same step functions as the
    wavefront stages (key 17 = 1) with the traversal in between: every pixel, and every instrumentation count, must be the stages'."""
    mk_scene, mk_cam = SCENES[scene_name]
    sc, W, H = mk_scene(), 120, 88                   # not a multiple of 16: partial tiles
    cam = mk_cam(W, H)
    outs = []
    for mode in (1, 2, 0):                           # stages, fused, auto (= fused: both scenes are far below 64 k triangles)
        ctx = capi.Context(0)
        ctx.resize(W, H)
        ctx.upload_scene(sc)
        ctx.set_camera(cam)
        ctx.set_tuning(17, mode)
        ctx.set_ray_counting(True)
        st = settings_for(tech, light_bounces=3, sample_count=2)
        counts = []
        for f in range(3):
            st.rand_seed = f + 1
            s = ctx.render(st)
            counts.append((s.rays, s.box_tests, s.tri_tests, s.hits, s.node_visits))
            assert s.launches == 1
        outs.append((ctx.readback(), counts))
        ctx.close()
    for (img, acc), counts in outs[1:]:
        assert np.array_equal(img, outs[0][0][0]) and np.array_equal(acc, outs[0][0][1], equal_nan=True)
        assert counts == outs[0][1]


@pytest.mark.parametrize("scene_name", ["cornell", "hall_small"])
@pytest.mark.parametrize("skip_dead", [1, 0])
def test_gi_part2_in_one_launch_equals_the_stages(scene_name, skip_dead):
    """Tuning key 19: ReSTIR GI Part 2 as ONE launch (1 = k_gi2_fused: a thread keeps its pixel for the whole neighbour loop, the visibility
    ray traced in place; 2 = k_gi2_persistent: a lane owns a pixel, lanes without a ray in flight are serviced together) against the 2 x neighbours + 1 stage launches: every buffer a later frame reads (image, accumulation, depth, the
    reservoirs that become the history) and every instrumentation count must be the stages', frame after frame (temporal reuse feeds on it)."""
    mk_scene, mk_cam = SCENES[scene_name]
    sc, W, H = mk_scene(), 136, 88                   # not a multiple of 16: partial tiles
    cam = mk_cam(W, H)
    outs = []
    for mode in (0, 1, 2):
        ctx = capi.Context(0)
        ctx.resize(W, H)
        ctx.upload_scene(sc)
        ctx.set_camera(cam)
        ctx.set_tuning(19, mode)
        ctx.set_tuning(18, skip_dead)
        ctx.set_ray_counting(True)
        st = settings_for(capi.RESTIR_GI, light_bounces=2)
        counts = []
        for f in range(4):
            st.rand_seed = f + 1
            s = ctx.render(st)
            counts.append((s.rays, s.box_tests, s.tri_tests, s.hits, s.node_visits, tuple(s.part_rays)))
        bufs = [ctx.read_buffer(b) for b in (capi.BUF_DEPTH, capi.BUF_GI, capi.BUF_GI_PREV)]
        outs.append((ctx.readback(), counts, bufs))
        ctx.close()
    (img0, acc0), counts0, bufs0 = outs[0]
    for (img1, acc1), counts1, bufs1 in outs[1:]:
        assert np.array_equal(img0, img1) and np.array_equal(acc0, acc1, equal_nan=True)
        assert counts0 == counts1
        for a, b in zip(bufs0, bufs1):
            assert a.tobytes() == b.tobytes()


def test_dead_shadow_rays_are_not_traced_and_nothing_changes():
    """Tuning key 18 (default on): a ReSTIR DI shadow ray whose pixel is black whatever the ray finds — reservoir weight zero, light facing
    away, black sky — is answered without a ray.  Same pixels, fewer rays; with a sky that is not black the miss outcome is not zero and
    every ray is traced again."""
    mk_scene, mk_cam = SCENES["hall_small"]
    sc, W, H = mk_scene(), 160, 96
    cam = mk_cam(W, H)
    out = {}
    for sky in ((0.0, 0.0, 0.0), (0.3, 0.4, 0.5)):
        for skip in (1, 0):
            ctx = capi.Context(0)
            ctx.resize(W, H); ctx.upload_scene(sc); ctx.set_camera(cam)
            ctx.set_tuning(18, skip)
            ctx.set_ray_counting(True)
            st = settings_for(capi.RESTIR_DI, sky_color=sky)
            rays = 0
            for f in range(3):
                st.rand_seed = f + 1
                rays += ctx.render(st).part_rays[2]
            out[(sky, skip)] = (ctx.readback(), rays)
            ctx.close()
        (a, ra), (b, rb) = out[(sky, 1)], out[(sky, 0)]
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1], equal_nan=True)
        if sky == (0.0, 0.0, 0.0):
            assert ra < rb                                       # some rays were dead
        else:
            assert ra <= rb


def test_async_frames_equal_blocking_frames():
    mk_scene, mk_cam = SCENES["cornell"]
    sc, W, H = mk_scene(), 96, 96
    cam = mk_cam(W, H)
    outs = []
    for use_async in (False, True):
        ctx = capi.Context(0)
        ctx.resize(W, H)
        ctx.upload_scene(sc)
        ctx.set_camera(cam)
        st = settings_for(capi.RESTIR_DI)
        for f in range(4):
            st.rand_seed = f + 1
            if use_async:
                ctx.render_async(st)
            else:
                ctx.render(st)
        ctx.synchronize()
        if use_async:
            ms, n = ctx.frame_timings(0)
            assert n == 3 and all(m > 0 for m in ms[:3])
        outs.append(ctx.readback())
        ctx.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1], equal_nan=True)


@pytest.mark.parametrize("pipelined", [1, 0])
def test_pipelined_frames_with_technique_switches(pipelined):
    """ReSTIR DI frames are pipelined over two streams (frame N+1's Part 1 + setup beside frame N's trace kernel, tuning key
    11); frames of other techniques, instrumented frames and blocking frames in between must still see everything before them.
    A long asynchronous sequence with technique switches and a frame-index reset equals the same sequence rendered blocking."""
    mk_scene, mk_cam = SCENES["hall_small"]
    sc, W, H = mk_scene(), 208, 120
    cam = mk_cam(W, H)
    seq = [capi.RESTIR_DI] * 5 + [capi.RESTIR_GI, capi.RESTIR_DI, capi.RESTIR_DI, capi.COSINE_WEIGHTED_SAMPLING] + [capi.RESTIR_DI] * 4
    outs = []
    for use_async in (False, True):
        ctx = capi.Context(0)
        ctx.resize(W, H)
        ctx.upload_scene(sc)
        ctx.set_camera(cam)
        ctx.set_tuning(11, pipelined)
        for f, tech in enumerate(seq):
            st = settings_for(tech)
            st.rand_seed = f + 1
            if f == 10:
                ctx.reset_frame_index()                    # the accumulator restarts while earlier frames may still be in flight
            if use_async and f != 7:
                ctx.render_async(st)
            else:
                ctx.render(st)
        ctx.synchronize()
        img, acc = ctx.readback()
        bufs = [ctx.read_buffer(b) for b in (capi.BUF_DI, capi.BUF_DI_PREV, capi.BUF_DEPTH)]
        outs.append((img, acc, bufs))
        ctx.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1], equal_nan=True)
    for a, b in zip(outs[0][2], outs[1][2]):
        assert a.tobytes() == b.tobytes()


def test_light_sorted_tasks_with_pipelined_async_frames():
    """Key 3 (counting sort of the shadow tasks) together with key 11 (two-stream pipelining): frame N+1's setup / scan / scatter
    run beside frame N's trace kernel, which still reads its sorted index — the sort scratch alternates with the frame parity
    like the task queues.  A long asynchronous run on a frame large enough for the kernels to overlap equals blocking frames."""
    mk_scene, mk_cam = SCENES["hall_small"]
    sc, W, H = mk_scene(), 640, 360
    cam = mk_cam(W, H)
    outs = []
    for use_async in (False, True):
        ctx = capi.Context(0)
        ctx.resize(W, H)
        ctx.upload_scene(sc)
        ctx.set_camera(cam)
        ctx.set_tuning(3, 1)
        ctx.set_tuning(11, 1)
        st = settings_for(capi.RESTIR_DI)
        for f in range(24):
            st.rand_seed = f + 1
            if use_async:
                ctx.render_async(st)
            else:
                ctx.render(st)
        ctx.synchronize()
        outs.append(ctx.readback())
        ctx.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1], equal_nan=True)


def test_tuning_values_are_range_checked():
    ctx = capi.Context(0)
    for key, bad in ((5, 65), (0, 3), (4, -1), (8, 32), (11, 2), (13, 2), (14, 3), (15, 3), (16, 1025), (17, 3), (18, 2), (19, 3), (20, 65), (24, 0), (-1, 0)):
        with pytest.raises(capi.FyprtError):
            ctx.set_tuning(key, bad)
    ctx.set_tuning(5, 64)
    ctx.set_tuning(5, 24)
    ctx.close()
