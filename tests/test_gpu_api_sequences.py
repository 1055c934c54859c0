"""State-machine parity: seeded random SEQUENCES of boundary calls — frames of any technique with random settings, camera moves that
keep the previous matrices (Camera::OnUpdate) and ones that reset them (SetPosition), frame-index resets, resizes, scene
replacement (with any of the three tree builders), mesh moves refitted on the device, row bands with a halo, interleaved stripes, blocking and asynchronous frames, the instrumented kernel
variants, every tuning key that must not change a result — mirrored on
the oracle (walking the product's exported tree), compared bit for bit after every frame.  (A 320-sequence soak of this test found
two things the single-feature tests could not: frame 1 has to clear the WHOLE accumulation buffer, and the "previous normals" of a
ReSTIR frame are the last ReSTIR frame's whichever of the two techniques rendered it.)  The single-feature tests start each case
from a fresh context; this one catches state that leaks from one call into the next (history of another technique or scene, stale
rows, a queue parity, a half-applied tuning change).  Mirrors what Renderer::Render sees from the application's main loop
(WalnutApp.cpp:878-910) over a session."""
import numpy as np
import pytest

from common import SCENES, bits_equal, struct_equal
from fypraytracer_amd import capi

pytestmark = pytest.mark.gpu

SIZES = [(64, 48), (80, 40), (48, 64)]
NEUTRAL_KEYS = {0: [0, 1, 2], 1: [0, 1], 2: [0, 1, 3], 3: [0, 1], 4: [0, 16, 128], 5: [0, 8, 48], 6: [0, 16], 7: [0, 16], 9: [0, 1, 4], 10: [0, 8], 11: [0, 1], 14: [0, 1, 2],
                15: [0, 1, 2], 17: [0, 1, 2], 18: [0, 1],       # r03: fused small-scene frame on / off, dead rays skipped / traced
                19: [0, 1, 2], 20: [0, 8, 48]}                  # r03: ReSTIR GI Part 2 as stages, one launch, one persistent launch


def _random_settings(rng):
    tech = int(rng.integers(0, 9))
    edge = rng.integers(0, 12) == 0                                       # now and then the values the kernels' uint8 casts and empty loops see
    return capi.Settings(technique=tech, light_bounces=int(rng.choice([0, 257])) if edge else int(rng.integers(1, 4)),
                         sample_count=int(rng.choice([0, 258])) if (edge and rng.integers(0, 2)) else int(rng.integers(1, 3)),
                         to_accumulate=int(rng.integers(0, 8) != 0),
                         sky_color=tuple(float(x) for x in rng.uniform(0.0, 0.4, 3)), light_candidate_count=int(rng.integers(1, 9)),
                         use_temporal_reuse=int(rng.integers(0, 2)), use_spatial_reuse=int(rng.integers(0, 2)),
                         temporal_history_limit=int(rng.integers(1, 6)), spatial_neighbor_num=int(rng.integers(0, 6)),
                         spatial_neighbor_radius=int(rng.integers(1, 31)), rand_seed=int(rng.integers(0, 1 << 30)))


def _seeds():
    """16 sequences by default; FYPRT_SEQ_FIRST / FYPRT_SEQ_LAST select a range for soak runs (what the soaks found is scripted below)."""
    import os
    if "FYPRT_SEQ_FIRST" in os.environ:
        return list(range(int(os.environ["FYPRT_SEQ_FIRST"]), int(os.environ.get("FYPRT_SEQ_LAST", os.environ["FYPRT_SEQ_FIRST"])) + 1))
    return list(range(1, 17))


class Mirror:
    """the product context and its oracle twin, kept in the same state"""

    def __init__(self, rng):
        from oraclelib import Oracle
        self.Oracle, self.rng = Oracle, rng
        self.ctx = capi.Context(0)
        self.W, self.H = SIZES[0]
        self.ctx.resize(self.W, self.H)
        self.orc = None
        self.rows, self.halo, self.stripes = None, 0, None
        self.load("hall_small")

    def load(self, name):
        mk_scene, mk_cam = SCENES[name]
        self.name, self.sc, self.mk_cam = name, mk_scene(), mk_cam
        self.cam = mk_cam(self.W, self.H)
        # A scene replaced on a live renderer keeps every per-pixel buffer (the reference's SceneToGPU touches the scene only): the
        # product's upload leaves them alone, the oracle's new twin adopts the old one's frame state.  History of another scene is
        # then reused as far as it is valid (a DI history light index beyond the new emissive list reads as none).
        self.ctx.set_tuning(12, int(self.rng.integers(0, 3)))      # host SAH / device radix tree / device PLOC: the oracle walks whatever was built
        self.ctx.upload_scene(self.sc)
        self.ctx.set_camera(self.cam)
        self.bvh = self.ctx.export_bvh()
        self.mgr = self.sc.manager()
        self.mgr.perform_all_scene_updates(self.sc)
        self.ctx.set_object_vertices(self.sc)
        self._new_oracle(adopt=True)

    def move_mesh(self):
        """SceneManager transform edit -> device refit (by matrix or by re-uploaded vertices); the oracle's twin of the edited scene walks
        the REFITTED tree and adopts the frame state: reservoirs and history survive the edit on both sides (as in the reference)."""
        m = int(self.rng.integers(0, len(self.sc.meshes)))
        self.mgr.set_mesh_transform(self.sc, m, pos=tuple(float(x) for x in self.rng.uniform(-0.3, 0.3, 3)), rotation=(0.0, float(self.rng.uniform(-40, 40)), 0.0))
        self.mgr.perform_all_scene_updates(self.sc)
        by_matrix = bool(self.rng.integers(0, 2))
        if by_matrix:
            self.ctx.update_transforms(self.sc, [m])
        else:
            self.ctx.update_vertices(self.sc)
        self.bvh = self.ctx.export_bvh()
        self._new_oracle(adopt=True)
        return m, by_matrix

    def _new_oracle(self, adopt=False):
        old = self.orc
        self.orc = self.Oracle(self.sc, self.W, self.H)
        self.orc.set_camera(self.cam)
        self.orc.use_product_bvh(self.bvh)
        if old is not None:
            if adopt:
                self.orc.adopt_frame(old)
            old.close()

    def resize(self, W, H):
        self.W, self.H = W, H
        self.ctx.resize(W, H)
        self.ctx.set_rows(0, H, 0)                         # (a band that still fits would survive the resize)
        self.cam = self.mk_cam(W, H)
        self.ctx.set_camera(self.cam)
        self.rows, self.halo, self.stripes = None, 0, None
        self._new_oracle()

    def set_stripes(self, stripes):
        """interleaved part (or None = off); accumulation restarts on both sides, as a host does when it changes the split"""
        self.stripes = stripes
        if stripes:
            self.ctx.set_row_stripes(*stripes)
            self.rows, self.halo = None, 0
        else:
            self.ctx.set_row_stripes(0)
        self.ctx.reset_frame_index(); self.orc.reset_frame_index()

    def frame(self, st, asynchronous):
        if self.stripes and st.technique >= capi.RESTIR_DI:
            self.set_stripes(None)                                           # ReSTIR frames need contiguous rows
        if asynchronous and st.technique >= capi.RESTIR_DI and self.rng.integers(0, 2):
            self.ctx.render_part(st, 1)                                      # the frame in two calls (a host that moves the halo rows itself)
            self.ctx.render_part(st, 2)
            self.ctx.synchronize()
        elif asynchronous:
            self.ctx.render_async(st)
            self.ctx.synchronize()
        else:
            self.ctx.render(st)
        self.orc.render(st, rows=self.rows, halo=self.halo)
        y0, y1 = self.rows if self.rows else (0, self.H)
        img, acc = self.ctx.readback()
        if self.stripes and st.technique < capi.RESTIR_DI:                  # an interleaved part renders its stripes only (and its accumulation of the
            stripe, parts, part = self.stripes                              # other rows is not the frame's): compare the frame's pixels there
            own = np.array([(y // stripe) % parts == part for y in range(self.H)])
            ok = bits_equal(acc[own], self.orc.accum()[own]).all(axis=-1)
            assert ok.all(), f"{(~ok).sum()} accumulation pixels of the stripes differ"
            assert np.array_equal(img[own], self.orc.image()[own])
            return
        ok = bits_equal(acc[y0:y1], self.orc.accum()[y0:y1]).all(axis=-1)
        assert ok.all(), f"{(~ok).sum()} accumulation pixels differ"
        assert np.array_equal(img[y0:y1], self.orc.image()[y0:y1])
        sl = slice(y0 * self.W, y1 * self.W)
        # (the per-pixel techniques own no buffer besides accumulation and image: Renderer.cu:87-166)
        bufs = [] if st.technique < capi.RESTIR_DI else [capi.BUF_PAYLOAD, capi.BUF_DEPTH, capi.BUF_DI_PREV if st.technique == capi.RESTIR_DI else capi.BUF_GI_PREV]
        for b in bufs:
            g, o = self.ctx.read_buffer(b)[sl], self.orc.read_buffer(b)[sl]
            e = struct_equal(g, o) if g.dtype.names else bits_equal(g, o)
            assert e.all(), f"buffer {b}: {(~e).sum()} records differ"

    def close(self):
        self.orc.close()
        self.ctx.close()


@pytest.mark.parametrize("seed", _seeds())
def test_random_call_sequences_against_the_oracle(oracle_built, seed):
    rng = np.random.default_rng(1000 + seed)
    m = Mirror(rng)
    log = []
    try:
        for step in range(26):
            op = rng.choice(["frame", "frame", "frame", "frames", "pose", "teleport", "reset", "resize", "scene", "band", "tuning", "counting", "move", "stripes"])
            if op in ("frame", "frames"):
                st = _random_settings(rng)
                if m.rows and m.halo == 0 and st.technique >= capi.RESTIR_DI:
                    st.use_spatial_reuse = 0                 # a band without halo rows has no neighbours beyond its border to reuse (include/fyprt.h: fyprt_set_rows)
                n = 1 if op == "frame" else 3
                log.append(f"{op} tech {st.technique} T{st.use_temporal_reuse} S{st.use_spatial_reuse} r{st.spatial_neighbor_radius}")
                for k in range(n):
                    st.rand_seed = int(rng.integers(0, 1 << 30))
                    if k:
                        m.cam.on_update(0.05, "W" if k == 1 else "DQ", (float(rng.uniform(-80, 80)), float(rng.uniform(-40, 40))))
                        m.ctx.set_camera(m.cam); m.orc.set_camera(m.cam)
                    m.frame(st, asynchronous=bool(rng.integers(0, 2)))
            elif op == "pose":
                keys = "".join(rng.choice(list("WASDQE"), size=int(rng.integers(0, 3))))
                m.cam.on_update(0.05, keys, (float(rng.uniform(-150, 150)), float(rng.uniform(-90, 90))))
                m.ctx.set_camera(m.cam); m.orc.set_camera(m.cam)
                log.append(f"pose {keys}")
            elif op == "teleport":
                p = np.asarray(m.cam.position, np.float64) + rng.uniform(-0.3, 0.3, 3)
                m.cam.set_position(tuple(float(x) for x in p))
                m.ctx.set_camera(m.cam); m.orc.set_camera(m.cam)
                log.append("teleport")
            elif op == "reset":
                m.ctx.reset_frame_index(); m.orc.reset_frame_index()
                log.append("reset")
            elif op == "resize":
                W, H = SIZES[int(rng.integers(0, len(SIZES)))]
                m.resize(W, H)
                log.append(f"resize {W}x{H}")
            elif op == "scene":
                m.load("cornell" if m.name != "cornell" else "hall_small")
                log.append(f"scene {m.name}")
            elif op == "stripes":
                if m.stripes or rng.integers(0, 3) == 0:
                    m.set_stripes(None)
                else:
                    parts = int(rng.integers(2, 4))
                    m.set_stripes((int(rng.choice([4, 8, 16])), parts, int(rng.integers(0, parts))))
                log.append(f"stripes {m.stripes}")
            elif op == "band":
                if m.stripes:
                    m.set_stripes(None)
                if rng.integers(0, 3) == 0:
                    m.rows, m.halo = None, 0
                    m.ctx.set_rows(0, m.H, 0)
                else:
                    y0 = int(rng.integers(0, m.H - 8)); y1 = int(rng.integers(y0 + 4, m.H + 1)); halo = int(rng.choice([0, 30]))
                    m.rows, m.halo = (y0, y1), halo
                    m.ctx.set_rows(y0, y1, halo)
                log.append(f"band {m.rows} halo {m.halo}")
            elif op == "move":
                mesh, by_matrix = m.move_mesh()
                log.append(f"move mesh {mesh} {'by matrix' if by_matrix else 'by vertices'}")
            elif op == "counting":
                on = bool(rng.integers(0, 2))
                m.ctx.set_ray_counting(on)                   # the instrumented kernel variants compute the same pixels
                log.append(f"counting {on}")
            else:
                key = int(rng.choice(list(NEUTRAL_KEYS)))
                val = int(rng.choice(NEUTRAL_KEYS[key]))
                m.ctx.set_tuning(key, val)
                log.append(f"tuning {key}={val}")
    except AssertionError as e:
        raise AssertionError(f"after {log}: {e}") from None
    finally:
        m.close()


def _settings(tech, **kw):
    base = dict(technique=tech, light_bounces=2, sample_count=1, sky_color=(0.1, 0.2, 0.3), light_candidate_count=4, use_temporal_reuse=1, use_spatial_reuse=1,
                temporal_history_limit=3, spatial_neighbor_num=4, spatial_neighbor_radius=12, rand_seed=1)
    base.update(kw)
    return capi.Settings(**base)


def test_restir_technique_switch_under_a_moving_camera(oracle_built):
    """The reference keeps ONE pair of normal buffers for both ReSTIRs: the "previous normals" of a ReSTIR frame are the last ReSTIR
    frame's, whichever technique rendered it, also across frames of other techniques in between (found by the soak: seeds 348, 400, 418)."""
    m = Mirror(np.random.default_rng(7))
    try:
        script = [capi.RESTIR_GI, capi.RESTIR_GI, capi.RESTIR_DI, capi.RESTIR_DI, capi.NEE, capi.RESTIR_GI, capi.BRUTE_FORCE, capi.RESTIR_DI, capi.RESTIR_GI]
        for k, tech in enumerate(script):
            m.cam.on_update(0.05, "WD"[k % 2], (40.0 - 25.0 * k, 15.0 * (k % 3) - 10.0))
            m.ctx.set_camera(m.cam); m.orc.set_camera(m.cam)
            m.frame(_settings(tech, rand_seed=k + 1), asynchronous=bool(k & 1))
    finally:
        m.close()


def test_band_moved_between_accumulated_frames(oracle_built):
    """Frame 1 clears the whole accumulation buffer, not just the context's rows: a band moved with fyprt_set_rows must not find the sums
    of an earlier accumulation in its new rows (found by the soak: seed 7 of the first batch)."""
    m = Mirror(np.random.default_rng(8))
    try:
        for k in range(2):
            m.frame(_settings(capi.BRUTE_FORCE, rand_seed=k + 1), asynchronous=False)
        m.ctx.reset_frame_index(); m.orc.reset_frame_index()
        m.rows, m.halo = (16, 40), 30
        m.ctx.set_rows(16, 40, 30)
        m.frame(_settings(capi.RESTIR_DI, rand_seed=3), asynchronous=False)
        m.rows, m.halo = (14, 36), 30
        m.ctx.set_rows(14, 36, 30)
        m.frame(_settings(capi.BRDF_SAMPLING, rand_seed=4), asynchronous=False)
    finally:
        m.close()
