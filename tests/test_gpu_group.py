"""Multi-GPU layer of the C ABI on real kernels (include/fyprt.h "multi-GPU"): several contexts of one process render one frame in
row bands (fyprt_group_*).  The box has one GPU, so the contexts share it — the code path is the one an 8-GPU host runs, with
hipMemcpyPeerAsync degenerating to a same-device copy.
  * halo mode 1 (exchange of the Part-1 records + temporal history of the halo rows): a static-camera ReSTIR sequence is
    bit-identical to the single-context sequence on EVERY frame, for uneven bands too;
  * halo mode 0 (recompute): frame 1 identical, later frames differ only near the borders (tests/test_gpu_multiband.py states the bound);
  * fyprt_group_gather assembles the frame on the root context; fyprt_balance_rows + fyprt_group_set_rows move the borders;
  * fyprt_comm_* (RCCL) with a communicator of ONE rank: init / render / gather / destroy run through librccl (the multi-rank
    exchange itself cannot run on a one-GPU box: RCCL refuses two ranks on one device)."""
import ctypes as C

import numpy as np
import pytest

from common import SCENES, bits_equal, settings_for
from fypraytracer_amd import capi

pytestmark = pytest.mark.gpu


def _contexts(n, sc, cam, W, H):
    out = []
    for _ in range(n):
        c = capi.Context(0)
        c.resize(W, H)
        c.upload_scene(sc)
        c.set_camera(cam)
        out.append(c)
    return out


def _single(sc, cam, W, H, tech, frames):
    ctx = _contexts(1, sc, cam, W, H)[0]
    st = settings_for(tech)
    outs = []
    for f in range(frames):
        st.rand_seed = f + 1
        ctx.render(st)
        outs.append(ctx.readback())
    ctx.close()
    return outs


def _stitched(ctxs, bounds, H, W):
    img = np.zeros((H, W), np.uint32)
    acc = np.zeros((H, W, 4), np.float32)
    for c, (b, e) in zip(ctxs, zip(bounds, bounds[1:])):
        c.set_rows(b, e, 0)                                  # readback covers the band's rows
        i, a = c.readback()
        img[b:e], acc[b:e] = i[b:e], a[b:e]
    return img, acc


@pytest.mark.parametrize("tech", [capi.RESTIR_DI, capi.RESTIR_GI])
@pytest.mark.parametrize("bounds", [[0, 96, 192], [0, 40, 70, 150, 192]])
def test_halo_exchange_equals_single_gpu_on_every_frame(tech, bounds):
    mk_scene, mk_cam = SCENES["hall_small"]
    sc, W, H = mk_scene(), 160, 192
    cam = mk_cam(W, H)
    frames = 4
    ref = _single(sc, cam, W, H, tech, frames)
    ctxs = _contexts(len(bounds) - 1, sc, cam, W, H)
    grp = capi.Group(ctxs, bounds, halo_mode=1)
    st = settings_for(tech)
    for f in range(frames):
        st.rand_seed = f + 1
        grp.render(st)
        grp.synchronize()
        img, acc = _stitched(ctxs, bounds, H, W)
        eq = bits_equal(acc, ref[f][1]).all(axis=-1)
        assert eq.all(), f"frame {f + 1}: {(~eq).sum()} pixels differ (rows {sorted(set(np.argwhere(~eq)[:, 0].tolist()))[:8]})"
        assert np.array_equal(img, ref[f][0])
    grp.gather(0)
    grp.synchronize()
    ctxs[0].set_rows(0, H, 0)
    full, _ = ctxs[0].readback(want_accum=False)
    assert np.array_equal(full, ref[-1][0])                  # the gathered frame on the root context
    grp.close()
    for c in ctxs:
        c.close()


def test_recompute_mode_and_other_techniques_through_the_group():
    mk_scene, mk_cam = SCENES["hall_small"]
    sc, W, H = mk_scene(), 160, 192
    cam = mk_cam(W, H)
    bounds = [0, 64, 128, 192]
    for tech, frames in ((capi.RESTIR_DI, 1), (capi.NEE, 2), (capi.COSINE_WEIGHTED_SAMPLING, 2)):
        ref = _single(sc, cam, W, H, tech, frames)
        ctxs = _contexts(3, sc, cam, W, H)
        grp = capi.Group(ctxs, bounds, halo_mode=0)
        st = settings_for(tech)
        for f in range(frames):
            st.rand_seed = f + 1
            grp.render(st)
        grp.synchronize()
        img, acc = _stitched(ctxs, bounds, H, W)
        assert np.array_equal(img, ref[-1][0]) and bits_equal(acc, ref[-1][1]).all(), tech
        grp.close()
        for c in ctxs:
            c.close()


def test_balanced_rows_move_the_borders_and_the_sequence_stays_exact():
    mk_scene, mk_cam = SCENES["hall_small"]
    sc, W, H = mk_scene(), 160, 192
    cam = mk_cam(W, H)
    bounds = [0, 96, 192]
    ctxs = _contexts(2, sc, cam, W, H)
    grp = capi.Group(ctxs, bounds, halo_mode=1)
    st = settings_for(capi.RESTIR_DI)
    for f in range(6):
        st.rand_seed = f + 1
        grp.render(st)
        grp.synchronize()
        if f == 2:
            new = capi.balance_rows(bounds, [2.0, 1.0], min_rows=16, max_shift=24)      # band 0 took twice as long: it shrinks
            assert new[0] == 0 and new[2] == H and 72 <= new[1] < 96
            grp.set_rows(new)
            bounds = new
    img, acc = _stitched(ctxs, bounds, H, W)
    ref = _single(sc, cam, W, H, capi.RESTIR_DI, 6)[-1]
    # the rows that changed owner took their accumulation and history along: still the single-GPU sequence, bit for bit
    assert np.array_equal(img, ref[0]) and bits_equal(acc, ref[1]).all()
    grp.close()
    for c in ctxs:
        c.close()


def test_rccl_communicator_of_one_rank():
    mk_scene, mk_cam = SCENES["cornell"]
    sc, W, H = mk_scene(), 96, 80
    cam = mk_cam(W, H)
    ctx = _contexts(1, sc, cam, W, H)[0]
    lib = ctx.lib
    uid = (C.c_char * 128)()
    assert lib.fyprt_comm_unique_id(uid) == 0, lib.fyprt_last_error(None)
    bounds = (C.c_uint32 * 2)(0, H)
    ctx._check(lib.fyprt_comm_init_rank(ctx.h, 1, 0, uid, bounds))
    ctx._check(lib.fyprt_comm_set_halo_mode(ctx.h, 1))
    st = settings_for(capi.RESTIR_DI)
    ref = _single(sc, cam, W, H, capi.RESTIR_DI, 2)[-1]
    for f in range(2):
        st.rand_seed = f + 1
        ctx._check(lib.fyprt_comm_render(ctx.h, C.byref(st)))
        ctx._check(lib.fyprt_comm_gather(ctx.h, -1))
    ctx.synchronize()
    img, acc = ctx.readback()
    assert np.array_equal(img, ref[0]) and bits_equal(acc, ref[1]).all()
    lib.fyprt_comm_destroy(ctx.h)
    ctx.close()


def test_comm_set_rows_between_pipelined_frames():
    """fyprt_comm_set_rows between ASYNCHRONOUS (pipelined) ReSTIR DI frames: the front stream may still run Part 1 + setup of a frame when
    the call arrives; it must drain the context before it touches the history rows and return with them in place.  With
    one rank no row moves, so the sequence must stay the single-context sequence bit for bit."""
    mk_scene, mk_cam = SCENES["cornell"]
    sc, W, H = mk_scene(), 96, 80
    cam = mk_cam(W, H)
    ctx = _contexts(1, sc, cam, W, H)[0]
    lib = ctx.lib
    uid = (C.c_char * 128)()
    assert lib.fyprt_comm_unique_id(uid) == 0, lib.fyprt_last_error(None)
    bounds = (C.c_uint32 * 2)(0, H)
    ctx._check(lib.fyprt_comm_init_rank(ctx.h, 1, 0, uid, bounds))
    st = settings_for(capi.RESTIR_DI)
    frames = 6
    ref = _single(sc, cam, W, H, capi.RESTIR_DI, frames)[-1]
    for f in range(frames):
        st.rand_seed = f + 1
        ctx._check(lib.fyprt_comm_render(ctx.h, C.byref(st)))          # recompute mode, one rank: a plain pipelined frame
        if f in (1, 2, 4):
            ctx._check(lib.fyprt_comm_set_rows(ctx.h, bounds))           # no synchronize in between
    ctx.synchronize()
    img, acc = ctx.readback()
    assert np.array_equal(img, ref[0]) and bits_equal(acc, ref[1]).all()
    lib.fyprt_comm_destroy(ctx.h)
    ctx.close()


def test_asynchronous_frames_with_a_gather_per_frame():
    """No synchronisation between frames: the next frame's epilogues must not overwrite a band the root's gather is still copying, the
    halo pulls of frame N+1 must not overtake frame N.  The frame gathered at the end is the single-context frame."""
    mk_scene, mk_cam = SCENES["hall_small"]
    sc, W, H = mk_scene(), 320, 192
    cam = mk_cam(W, H)
    frames, bounds = 12, [0, 50, 120, 192]
    ref = _single(sc, cam, W, H, capi.RESTIR_DI, frames)[-1]
    ctxs = _contexts(3, sc, cam, W, H)
    grp = capi.Group(ctxs, bounds, halo_mode=1)
    st = settings_for(capi.RESTIR_DI)
    for f in range(frames):
        st.rand_seed = f + 1
        grp.render(st)
        grp.gather(1)                                        # the middle band's context presents
    grp.synchronize()
    ctxs[1].set_rows(0, H, 0)
    full, _ = ctxs[1].readback(want_accum=False)
    assert np.array_equal(full, ref[0])
    grp.close()
    for c in ctxs:
        c.close()


# ---------------------------------------------------------------------------------------------- interleaved stripes
def _owned_rows(H, stripe, parts, part):
    return np.array([y for y in range(H) if (y // stripe) % parts == part], np.int64)


@pytest.mark.parametrize("tech", [capi.BRUTE_FORCE, capi.COSINE_WEIGHTED_SAMPLING, capi.LIGHT_SOURCE_SAMPLING, capi.NEE])
@pytest.mark.parametrize("stripe,parts", [(16, 3), (8, 4), (5, 2)])
def test_striped_context_renders_exactly_its_stripes(tech, stripe, parts):
    """fyprt_set_row_stripes: part k of n renders the stripes k, k + n, ... — bit for bit the single-context pixels there (the
    techniques' pixels are independent), and touches nothing else.  H = 100 leaves a cut last stripe; stripe 5 is not a multiple of
    the 8-row wave tile (placement is free, results are not)."""
    mk_scene, mk_cam = SCENES["hall_small"]
    sc, W, H = mk_scene(), 96, 100
    cam = mk_cam(W, H)
    ref = _single(sc, cam, W, H, tech, 2)
    ctx = _contexts(1, sc, cam, W, H)[0]
    st = settings_for(tech)
    seen = np.zeros(H, bool)
    for part in range(parts):
        ctx.resize(W, H)                                      # zero image / accumulation
        ctx.set_row_stripes(stripe, parts, part)
        for f in range(2):
            st.rand_seed = f + 1
            ctx.render(st)
        img, acc = ctx.readback()
        rows = _owned_rows(H, stripe, parts, part)
        other = np.setdiff1d(np.arange(H), rows)
        assert bits_equal(acc[rows], ref[1][1][rows]).all() and np.array_equal(img[rows], ref[1][0][rows])
        assert not acc[other].any() and not img[other].any()
        seen[rows] = True
    assert seen.all()
    ctx.set_row_stripes(0)                                    # back to the whole frame
    ctx.resize(W, H)
    st.rand_seed = 1
    ctx.render(st)
    assert np.array_equal(ctx.readback(want_accum=False)[0], ref[0][0])
    ctx.close()


def test_restir_refuses_stripes_and_bad_arguments():
    mk_scene, mk_cam = SCENES["hall_small"]
    sc, W, H = mk_scene(), 64, 64
    ctx = _contexts(1, sc, mk_cam(W, H), W, H)[0]
    with pytest.raises(capi.FyprtError):
        ctx.set_row_stripes(16, 2, 2)                         # part >= parts
    with pytest.raises(capi.FyprtError):
        ctx.set_row_stripes(16, 8, 7)                         # 4 stripes only: part 7 owns nothing
    ctx.set_row_stripes(16, 2, 1)
    for tech in (capi.RESTIR_DI, capi.RESTIR_GI):
        with pytest.raises(capi.FyprtError, match="contiguous"):
            ctx.render(settings_for(tech))
    ctx.render(settings_for(capi.NEE))                        # still usable
    ctx.close()


@pytest.mark.parametrize("n,stripe", [(2, 16), (3, 8), (5, 16)])
def test_group_interleave_gathers_the_single_gpu_frame(n, stripe):
    """fyprt_group_set_interleave: the per-pixel techniques run striped and the gather moves stripes; a ReSTIR frame of the same group
    keeps the row bands (with the halo exchange) — both equal the single-context frames bit for bit."""
    mk_scene, mk_cam = SCENES["hall_small"]
    sc, W, H = mk_scene(), 128, 168
    cam = mk_cam(W, H)
    bounds = [round(H * k / n) for k in range(n + 1)]
    ctxs = _contexts(n, sc, cam, W, H)
    grp = capi.Group(ctxs, bounds, halo_mode=1)
    grp.set_interleave(stripe)
    for tech in (capi.NEE, capi.RESTIR_DI, capi.GGX_SAMPLING):
        ref = _single(sc, cam, W, H, tech, 3)
        for c in ctxs:
            c.resize(W, H)                                    # new technique: accumulation restarts (Renderer::ResetFrameIndex)
        st = settings_for(tech)
        for f in range(3):
            st.rand_seed = f + 1
            grp.render(st)
            grp.gather(0)
            grp.synchronize()
            ctxs[0].set_row_stripes(0)
            ctxs[0].set_rows(0, H, 0)
            full, _ = ctxs[0].readback(want_accum=False)
            assert np.array_equal(full, ref[f][0]), f"technique {tech}, frame {f + 1}"
    with pytest.raises(capi.FyprtError):
        grp.set_interleave(H)                                 # fewer stripes than contexts
    grp.close()
    for c in ctxs:
        c.close()
