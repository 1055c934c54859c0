#!/usr/bin/env python3
"""Scheduling study of the persistent trace kernel WITHOUT a GPU (test infrastructure: uses oracle/).

The trace kernels are bound by vector-instruction issue (profiles/r03/microbench.jsonl: most of a node visit's instructions cost 4-5
cycles per wave, not 2), at a lane utilisation of about one half: the issue cost of a node round or a triangle round is paid for the
whole wave however few lanes take part.  Which lanes take part is pure scheduling — it cannot change a result — so policies can be compared
offline: the oracle's twin of the product traversal logs every ray's sequence of node visits and leaf triangle tests for one small ReSTIR DI
frame of the 1 M-triangle hall (single-threaded render), and this script replays the shadow rays through a 64-lane wave model with the
kernel's refill rule.  Cost model (cycles of SIMD issue per wave, from the instruction mix of the ISA and the measured issue costs):
node round 600, triangle round 230 (+40 per leaf phase), refill + epilogues 600.

  usage: python tools/wave_sim.py [width height]      -> one JSON line per policy (total issue cycles, lane utilisation)"""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import ctypes as C  # noqa: E402

from fypraytracer_amd import capi, scenes  # noqa: E402
from oraclelib import Oracle  # noqa: E402

C_NODE, C_TRI, C_LEAF_PHASE, C_REFILL = 600, 230, 40, 600


def record(W, H):
    sc = scenes.hall_scene()
    cam = scenes.hall_camera(W, H)
    ctx = capi.Context(-1)
    ctx.upload_scene(sc)
    orc = Oracle(sc, W, H)
    orc.set_camera(cam)
    orc.use_product_bvh(ctx.export_bvh())
    orc.lib.orc_set_threads(1)
    orc.lib.orc_record_events.argtypes = [C.c_void_p, C.c_int]
    orc.lib.orc_take_events.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    orc.lib.orc_take_events.restype = C.c_size_t
    orc.lib.orc_record_events(orc.h, 1)
    st = capi.Settings(technique=capi.RESTIR_DI, light_bounces=1, sky_color=(0, 0, 0), use_temporal_reuse=1, use_spatial_reuse=1)
    t = time.time()
    orc.render(st)
    n = orc.lib.orc_take_events(orc.h, None, 0)
    buf = np.empty(n, dtype=np.uint8)
    orc.lib.orc_take_events(orc.h, buf.ctypes.data, n)
    orc.lib.orc_record_events(orc.h, 0)
    print(f"# recorded {n} event bytes in {time.time() - t:.1f} s", file=sys.stderr)
    return buf


def split_rays(buf):
    """-> list of (kind, events) with events = list of ints: 0 = node visit, k > 0 = leaf with k triangle tests (k may be 0: light-only leaf)"""
    rays = []
    starts = np.flatnonzero((buf >= 0xF0) & (buf != 0xFF))
    ends = np.flatnonzero(buf == 0xFF)
    # nested fall-back rays (a shadow ray that misses its light runs a closest-hit ray) do not occur inside another ray's record: the
    # fall-back is decided before the shadow record starts
    assert len(starts) == len(ends)
    for s, e in zip(starts, ends):
        kind = int(buf[s]) & 0xF
        ev = buf[s + 1:e]
        seq = np.where(ev == 1, 0, (ev & 0xF).astype(np.int16) + np.where(ev >= 0x10, 100, 0)).astype(np.int16)   # leaf with k tests -> 100 + k
        rays.append((kind, seq))
    return rays


class Wave:
    """64 lanes, each with an optional ray = (event sequence, position, triangles left in the current leaf)"""

    def __init__(self, tasks, slots=1):
        self.tasks = tasks                  # list of event arrays, consumed from the front
        self.next = 0
        self.slots = slots
        self.seq = [[None] * slots for _ in range(64)]
        self.pos = np.zeros((64, slots), dtype=np.int64)
        self.tri = np.zeros((64, slots), dtype=np.int64)       # > 0: in a leaf with this many triangle rounds left
        self.active = np.zeros((64, slots), dtype=bool)
        self.cycles = 0
        self.lane_cycles = 0                # cycles x lanes that took part

    def state(self, l, s):
        """'n' node visit wanted, 't' triangle round wanted, None finished"""
        if not self.active[l, s]:
            return None
        if self.tri[l, s] > 0:
            return 't'
        q = self.seq[l][s]
        if self.pos[l, s] >= len(q):
            return None
        return 'n' if q[self.pos[l, s]] == 0 else 't'

    def want(self):
        n = np.zeros((64, self.slots), dtype=bool); t = np.zeros((64, self.slots), dtype=bool)
        for l in range(64):
            for s in range(self.slots):
                st = self.state(l, s)
                if st == 'n':
                    n[l, s] = True
                elif st == 't':
                    t[l, s] = True
        return n, t

    def retire(self):
        for l in range(64):
            for s in range(self.slots):
                if self.active[l, s] and self.tri[l, s] == 0 and self.pos[l, s] >= len(self.seq[l][s]):
                    self.active[l, s] = False

    def node_round(self, n):
        """one node visit for one ray of every lane that has a ray in node state (slot 0 first)"""
        took = 0
        for l in range(64):
            for s in range(self.slots):
                if n[l, s]:
                    self.pos[l, s] += 1
                    took += 1
                    break
        self.cycles += C_NODE; self.lane_cycles += C_NODE * took

    def enter_leaves(self):
        for l in range(64):
            for s in range(self.slots):
                if self.active[l, s] and self.tri[l, s] == 0 and self.pos[l, s] < len(self.seq[l][s]):
                    e = self.seq[l][s][self.pos[l, s]]
                    if e >= 100:
                        self.tri[l, s] = e - 100
                        self.pos[l, s] += 1

    def tri_round(self, t):
        took = 0
        for l in range(64):
            for s in range(self.slots):
                if t[l, s] and self.tri[l, s] > 0:
                    self.tri[l, s] -= 1
                    took += 1
                    break
        self.cycles += C_TRI; self.lane_cycles += C_TRI * took

    def refill(self):
        got = 0
        for l in range(64):
            for s in range(self.slots):
                if not self.active[l, s] and self.next < len(self.tasks):
                    self.seq[l][s] = self.tasks[self.next]; self.next += 1
                    self.pos[l, s] = 0; self.tri[l, s] = 0; self.active[l, s] = True
                    got += 1
        self.cycles += C_REFILL; self.lane_cycles += C_REFILL * min(64, got)


def run_kernel_policy(tasks, quorum=24, refill_lanes=24, slots=1, greedy=False, tri_weight=None):
    """the persistent kernel's loop (rt_wavefront.h) on one wave; greedy: every round is the kind with the larger (lanes / cost)"""
    w = Wave(tasks, slots)
    more = True
    while True:
        idle = 64 * slots - int(w.active.sum())
        more = w.next < len(w.tasks)
        if (more and idle >= refill_lanes * slots) or (not w.active.any() and more):
            w.refill()
        if not w.active.any():
            if w.next >= len(w.tasks):
                break
            continue
        while True:
            w.enter_leaves()
            n, t = w.want()
            ln, lt = int(n.any(axis=1).sum()), int(t.any(axis=1).sum())
            if greedy:
                if ln == 0 and lt == 0:
                    pass
                elif lt * C_NODE * (tri_weight or 1.0) > ln * C_TRI:
                    w.tri_round(t)
                else:
                    w.node_round(n)
            else:
                while ln >= 1:                                   # node loop: until fewer than `quorum` lanes are still in it
                    w.node_round(n)
                    w.enter_leaves()
                    n, t = w.want()
                    ln = int(n.any(axis=1).sum())
                    if ln < quorum:
                        break
                n, t = w.want()
                if t.any():
                    w.cycles += C_LEAF_PHASE
                    while True:                                  # leaf phase: every lane at a leaf tests all its triangles
                        n2, t = w.want()
                        if not t.any():
                            break
                        w.tri_round(t)
            w.retire()
            act = int(w.active.any(axis=1).sum())
            if act == 0:
                break
            more = w.next < len(w.tasks)
            if more and (64 * slots - int(w.active.sum())) >= refill_lanes * slots:
                break
    return w.cycles, w.lane_cycles


def main():
    W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (256, 144)
    cache = Path(f"/tmp/wave_sim_events_{W}x{H}.npy")
    if cache.exists():
        buf = np.load(cache)
    else:
        buf = record(W, H)
        np.save(cache, buf)
    rays = split_rays(buf)
    shadow = [q for k, q in rays if k == 1]
    primary = [q for k, q in rays if k == 0]
    nv = np.mean([int((q == 0).sum()) for q in shadow]); tt = np.mean([int((q[q >= 100] - 100).sum()) for q in shadow])
    print(json.dumps({"rays_primary": len(primary), "rays_shadow": len(shadow), "shadow_node_visits_per_ray": round(float(nv), 2), "shadow_tri_tests_per_ray": round(float(tt), 2),
                      "ideal_cycles_per_ray_at_full_utilisation": round((nv * C_NODE + tt * C_TRI) / 64, 1)}))
    # queue order ~ pixel order of the oracle (row-major): re-tile into 8x8 pixel waves as the setup kernel's compaction does
    # (only live pixels carry a shadow ray; their order inside the queue follows the tile order)
    chunk = 128
    policies = [("kernel q24 r24", dict(quorum=24, refill_lanes=24)), ("kernel q16 r24", dict(quorum=16, refill_lanes=24)), ("kernel q32 r24", dict(quorum=32, refill_lanes=24)),
                ("kernel q24 r16", dict(quorum=24, refill_lanes=16)), ("kernel q24 r8", dict(quorum=24, refill_lanes=8)),
                ("greedy r24", dict(greedy=True, refill_lanes=24)), ("greedy r16", dict(greedy=True, refill_lanes=16)), ("greedy r8", dict(greedy=True, refill_lanes=8)),
                ("greedy r24 tri x1.5", dict(greedy=True, refill_lanes=24, tri_weight=1.5)),
                ("two rays per lane, kernel q24 r24", dict(quorum=24, refill_lanes=24, slots=2)), ("two rays per lane, greedy r16", dict(greedy=True, refill_lanes=16, slots=2))]
    nwaves = 24
    per = (len(shadow) // nwaves // chunk) * chunk
    for name, kw in policies:
        t0 = time.time()
        cyc = lane = 0
        for wv in range(nwaves):
            c, l = run_kernel_policy(shadow[wv * per:(wv + 1) * per], **kw)
            cyc += c; lane += l
        nr = nwaves * per
        print(json.dumps({"policy": name, "issue_cycles_per_ray": round(cyc / nr, 1), "lane_utilisation": round(lane / (cyc * 64), 3), "rays": nr, "sim_s": round(time.time() - t0, 1)}), flush=True)


if __name__ == "__main__":
    main()
