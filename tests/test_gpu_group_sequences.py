"""Random sequences on a GROUP of contexts (one process, one context per band; include/fyprt.h "multi-GPU") against ONE context:
with the halo exchange on and a static camera the group's frames must be the single-GPU frames bit for bit — through changes of
technique and settings (incl. ReSTIR DI <-> GI switches: the shared "previous normals"), moved band borders (fyprt_group_set_rows
migrates accumulation and history), restarted accumulations, and the interleaved split for the per-pixel techniques.  The box has
one GPU: the contexts share it, the code path is the multi-GPU one (peer copies degenerate to same-device copies)."""
import numpy as np
import pytest

from common import SCENES, bits_equal
from fypraytracer_amd import capi

pytestmark = pytest.mark.gpu


def _ctx(sc, cam, W, H):
    c = capi.Context(0)
    c.resize(W, H)
    c.upload_scene(sc)
    c.set_camera(cam)
    return c


def _random_settings(rng):
    return capi.Settings(technique=int(rng.integers(0, 9)), light_bounces=int(rng.integers(1, 3)), sample_count=int(rng.integers(1, 3)),
                         sky_color=tuple(float(x) for x in rng.uniform(0.0, 0.4, 3)), light_candidate_count=int(rng.integers(1, 6)),
                         use_temporal_reuse=int(rng.integers(0, 2)), use_spatial_reuse=int(rng.integers(0, 2)),
                         temporal_history_limit=int(rng.integers(1, 5)), spatial_neighbor_num=int(rng.integers(0, 5)),
                         spatial_neighbor_radius=int(rng.integers(1, 25)), rand_seed=int(rng.integers(0, 1 << 30)))


def _random_bounds(rng, n, H, min_rows=12):
    while True:
        cuts = sorted(int(x) for x in rng.integers(min_rows, H - min_rows, n - 1))
        b = [0] + cuts + [H]
        if all(b[k + 1] - b[k] >= min_rows for k in range(n)):
            return b


def _seeds():
    import os
    if "FYPRT_SEQ_FIRST" in os.environ:
        return list(range(int(os.environ["FYPRT_SEQ_FIRST"]), int(os.environ.get("FYPRT_SEQ_LAST", os.environ["FYPRT_SEQ_FIRST"])) + 1))
    return list(range(1, 9))


@pytest.mark.parametrize("seed", _seeds())
def test_group_sequences_equal_the_single_context(seed):
    rng = np.random.default_rng(4000 + seed)
    mk_scene, mk_cam = SCENES["hall_small" if seed % 2 else "cornell"]
    sc, W, H = mk_scene(), 96, 112
    cam = mk_cam(W, H)
    n = int(rng.integers(2, 5))
    bounds = _random_bounds(rng, n, H)
    one = _ctx(sc, cam, W, H)
    ctxs = [_ctx(sc, cam, W, H) for _ in range(n)]
    grp = capi.Group(ctxs, bounds, halo_mode=1)
    striped, last_split, log = False, "bands", []
    try:
        for step in range(14):
            op = rng.choice(["frame", "frame", "frame", "rows", "reset", "interleave"])
            if op == "frame":
                st = _random_settings(rng)
                split = "stripes" if (striped and st.technique < capi.RESTIR_DI) else "bands"
                if split != last_split:                      # with the interleave on, ReSTIR frames use bands and the others stripes: the rows a context
                    one.reset_frame_index()                  # accumulates change with the kind of technique, so the accumulation restarts — as the
                    for c in ctxs:                           # reference's host restarts it on any change of settings (Renderer::ResetFrameIndex)
                        c.reset_frame_index()
                    last_split = split
                log.append(f"frame tech {st.technique} T{st.use_temporal_reuse} S{st.use_spatial_reuse}")
                one.render(st)
                grp.render(st)
                grp.gather(0)
                grp.synchronize()
                ctxs[0].set_row_stripes(0)
                ctxs[0].set_rows(0, H, 0)
                full, _ = ctxs[0].readback(want_accum=False)
                ref_img, ref_acc = one.readback()
                assert np.array_equal(full, ref_img), f"gathered image differs in {(full != ref_img).sum()} pixels"
                if not (striped and st.technique < capi.RESTIR_DI):          # bands: every context's accumulation rows are the single GPU's
                    for c, (b, e) in zip(ctxs, zip(bounds, bounds[1:])):
                        c.set_row_stripes(0)
                        c.set_rows(b, e, 0)
                        acc = c.readback()[1]
                        assert bits_equal(acc[b:e], ref_acc[b:e]).all(), f"accumulation of band {b}:{e} differs"
            elif op == "rows":
                bounds = _random_bounds(rng, n, H)
                grp.set_rows(bounds)
                log.append(f"rows {bounds}")
            elif op == "reset":
                one.reset_frame_index()
                for c in ctxs:
                    c.reset_frame_index()
                log.append("reset")
            else:
                striped = not striped
                grp.set_interleave(int(rng.choice([4, 8])) if striped else 0)
                one.reset_frame_index()                      # a host restarts the accumulation when it changes the split
                for c in ctxs:
                    c.reset_frame_index()
                log.append(f"interleave {striped}")
    except AssertionError as e:
        raise AssertionError(f"n={n} after {log}: {e}") from None
    finally:
        grp.close()
        one.close()
        for c in ctxs:
            c.close()
