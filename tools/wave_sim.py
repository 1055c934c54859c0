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
    # which pixels carry a shadow ray: the primary ray hit a surface that does not emit (the k-th shadow ray of the log is the k-th of them, row-major)
    pay = orc.read_buffer(capi.BUF_PAYLOAD)
    emits = np.array([m.emission_power * max(m.emission_color) > 0 for m in sc.materials])
    obj = pay["objectIndex"]
    live = (pay["hitDistance"] >= 0) & ~emits[sc.triangles["materialIndex"][np.maximum(obj, 0)]]
    return buf, np.flatnonzero(live).astype(np.int64)


def split_rays(buf):
    """-> list of (kind, events) with events = list of ints: 0 = node visit, k > 0 = leaf with k triangle tests (k may be 0: light-only leaf)"""
    rays = []
    k, n = 0, len(buf)
    while k < n:                                             # sequential parse: a shadow ray's header carries 3 bytes of light-triangle id
        assert 0xF0 <= buf[k] < 0xFF, (k, buf[k])
        kind = int(buf[k]) & 0xF
        k += 1
        light = -1
        if kind == 1:
            light = int(buf[k]) | (int(buf[k + 1]) << 8) | (int(buf[k + 2]) << 16)
            k += 3
        e = k
        while buf[e] != 0xFF:
            e += 1
        ev = buf[k:e]
        seq = np.where(ev == 1, 0, (ev & 0xF).astype(np.int16) + np.where(ev >= 0x10, 100, 0)).astype(np.int16)   # leaf with k tests -> 100 + k
        rays.append((kind, seq, light))
        k = e + 1
    return rays


class Wave:
    """64 lanes x `slots` rays per lane; a ray = a row of the padded event matrix E (0 node visit, 100 + k leaf with k triangle tests, -1 end)"""

    def __init__(self, E, first, count, slots=1, postpone=0):
        self.E, self.next, self.end, self.slots = E, first, first + count, slots
        self.postpone = postpone            # leaves a lane may carry along while it keeps walking inner nodes ("speculative" traversal)
        self.held = np.zeros((64, slots), dtype=np.int64)
        self.task = np.full((64, slots), -1, dtype=np.int64)
        self.pos = np.zeros((64, slots), dtype=np.int64)
        self.tri = np.zeros((64, slots), dtype=np.int64)
        self.cycles = 0
        self.lane_cycles = 0                # cycles x lanes that took part

    def cur(self):
        e = self.E[np.maximum(self.task, 0), self.pos]
        return np.where(self.task >= 0, e, -1)

    def settle(self):
        """enter leaves (a leaf's triangle count becomes pending triangle rounds), retire finished rays"""
        while True:
            e = self.cur()
            at = (self.task >= 0) & (self.held <= self.postpone) & (e >= 100)
            if not at.any():
                break
            self.tri = np.where(at, self.tri + e - 100, self.tri)
            self.held = np.where(at, self.held + 1, self.held)
            self.held = np.where(self.tri == 0, 0, self.held)          # leaves without a triangle test are nothing to carry
            self.pos = np.where(at, self.pos + 1, self.pos)
        e = self.cur()
        done = (self.task >= 0) & (self.tri == 0) & (e < 0)
        self.task = np.where(done, -1, self.task)

    def want(self):
        e = self.cur()
        act = self.task >= 0
        return act & (self.held <= self.postpone) & ((self.tri == 0) | (self.postpone > 0)) & (e == 0), act & (self.tri > 0)

    @staticmethod
    def first_slot(m):
        """per lane only the first wanting slot takes part in a round"""
        out = np.zeros_like(m)
        taken = np.zeros(m.shape[0], dtype=bool)
        for s in range(m.shape[1]):
            out[:, s] = m[:, s] & ~taken
            taken |= m[:, s]
        return out

    def node_round(self, n):
        n = self.first_slot(n)
        self.pos = np.where(n, self.pos + 1, self.pos)
        self.cycles += C_NODE; self.lane_cycles += C_NODE * int(n.sum())

    def tri_round(self, t):
        t = self.first_slot(t)
        self.tri = np.where(t, self.tri - 1, self.tri)
        self.held = np.where(self.tri == 0, 0, self.held)
        self.cycles += C_TRI; self.lane_cycles += C_TRI * int(t.sum())

    def idle_slots(self):
        return int((self.task < 0).sum())

    def refill(self):
        free = np.argwhere(self.task < 0)
        got = min(len(free), self.end - self.next)
        for k in range(got):
            l, sl = free[k]
            self.task[l, sl] = self.next + k; self.pos[l, sl] = 0; self.tri[l, sl] = 0; self.held[l, sl] = 0
        self.next += got
        self.cycles += C_REFILL; self.lane_cycles += C_REFILL * min(64, got)


def run_kernel_policy(E, first, count, quorum=24, refill_lanes=24, slots=1, greedy=False, tri_weight=1.0, select_cost=0, select_cost_tri=None, postpone=0, leaf_quorum=0):
    """the persistent kernel's loop (rt_wavefront.h) on one wave; greedy: every round is the kind with the larger (lanes / cost)"""
    w = Wave(E, first, count, slots, postpone)
    while True:
        more = w.next < w.end
        if more and (w.idle_slots() >= refill_lanes * slots or not (w.task >= 0).any()):
            w.refill()
        w.settle()
        if not (w.task >= 0).any():
            if w.next >= w.end:
                break
            continue
        while True:
            w.settle()
            n, t = w.want()
            ln, lt = int(n.any(axis=1).sum()), int(t.any(axis=1).sum())
            if greedy:
                if lt * C_NODE * tri_weight > ln * C_TRI:
                    w.tri_round(t); w.cycles += select_cost if select_cost_tri is None else select_cost_tri
                elif ln:
                    w.node_round(n); w.cycles += select_cost
            else:
                while ln >= 1:                                   # node loop: until fewer than `quorum` lanes are still in it
                    w.node_round(n)
                    w.settle()
                    n, t = w.want()
                    ln = int(n.any(axis=1).sum())
                    if ln < quorum:
                        break
                if t.any():
                    w.cycles += C_LEAF_PHASE
                    while t.any():                               # leaf phase: every lane at a leaf tests all its triangles
                        w.tri_round(t)
                        n, t = w.want()
                        if leaf_quorum and int(t.any(axis=1).sum()) < leaf_quorum:
                            break                                # leaf quorum: the few lanes with triangles left keep them for the next leaf phase
            w.settle()
            if not (w.task >= 0).any():
                break
            if w.next < w.end and w.idle_slots() >= refill_lanes * slots:
                break
    return w.cycles, w.lane_cycles


def run_workgroup_repack(E, first, count, repack_cost=0, refill_rays=96, lanes=256):
    """Upper bound for RE-PACKING rays between the four waves of a workgroup every round: the workgroup's 256 rays are dealt to its waves by
    the kind of work they want, so a node round runs in ceil(n / 64) full waves and a triangle round in ceil(t / 64); `repack_cost` = issue
    cycles every wave pays per round for moving ray state through LDS.  Returns (issue cycles summed over the waves, lane cycles)."""
    task = np.full(lanes, -1, dtype=np.int64); pos = np.zeros(lanes, dtype=np.int64); tri = np.zeros(lanes, dtype=np.int64)
    nxt, end = first, first + count
    cyc = lane_cyc = 0
    while True:
        idle = np.flatnonzero(task < 0)
        if nxt < end and (len(idle) >= refill_rays or len(idle) == lanes):
            got = min(len(idle), end - nxt)
            task[idle[:got]] = np.arange(nxt, nxt + got); pos[idle[:got]] = 0; tri[idle[:got]] = 0
            nxt += got
            w = -(-got // 64)
            cyc += C_REFILL * w; lane_cyc += C_REFILL * got
        act = task >= 0
        if not act.any():
            if nxt >= end:
                break
            continue
        # settle: enter leaves, retire finished rays
        while True:
            e = np.where(act, E[np.maximum(task, 0), pos], -1)
            at = act & (tri == 0) & (e >= 100)
            if not at.any():
                break
            tri = np.where(at, e - 100, tri); pos = np.where(at, pos + 1, pos)
        e = np.where(act, E[np.maximum(task, 0), pos], -1)
        done = act & (tri == 0) & (e < 0)
        task = np.where(done, -1, task); act = task >= 0
        n = act & (tri == 0) & (e == 0); t = act & (tri > 0)
        nn, nt = int(n.sum()), int(t.sum())
        if nn == 0 and nt == 0:
            continue
        wn, wt = -(-nn // 64), -(-nt // 64)
        cyc += wn * (C_NODE + repack_cost) + wt * (C_TRI + repack_cost); lane_cyc += nn * C_NODE + nt * C_TRI
        pos = np.where(n, pos + 1, pos); tri = np.where(t, tri - 1, tri)
    return cyc, lane_cyc


def main():
    W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (256, 144)
    cache = Path(f"/tmp/wave_sim_events2_{W}x{H}.npz")
    if cache.exists():
        z = np.load(cache); buf, live_px = z["buf"], z["live"]
    else:
        buf, live_px = record(W, H)
        np.savez(cache, buf=buf, live=live_px)
    rays = split_rays(buf)
    shadow = [q for k, q, l in rays if k == 1]
    lights = np.array([l for k, q, l in rays if k == 1])
    primary = [q for k, q, l in rays if k == 0]
    # a shadow ray whose light test failed runs as a closest-hit ray (kind 0) instead: drop those pixels from the map (rare)
    if len(live_px) != len(shadow):
        print(f"# {len(live_px)} live pixels, {len(shadow)} shadow rays: pixel map unavailable", file=sys.stderr)
        live_px = None
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
                ("two rays per lane, kernel q24 r24", dict(quorum=24, refill_lanes=24, slots=2)), ("two rays per lane, greedy r16", dict(greedy=True, refill_lanes=16, slots=2)),
                ("two rays per lane, greedy r8", dict(greedy=True, refill_lanes=8, slots=2)), ("two rays per lane, greedy r24", dict(greedy=True, refill_lanes=24, slots=2)),
                ("two rays per lane, greedy r16, +60 cycles per round for the slot select", dict(greedy=True, refill_lanes=16, slots=2, select_cost=60)),
                ("two rays per lane, greedy r16, slot select +57 per node round, +100 per triangle round", dict(greedy=True, refill_lanes=16, slots=2, select_cost=57, select_cost_tri=100)),
                ("two rays per lane, greedy r16, +57 / +100, triangle rounds weighted x0.7", dict(greedy=True, refill_lanes=16, slots=2, select_cost=57, select_cost_tri=100, tri_weight=0.7)),
                ("three rays per lane, greedy r16", dict(greedy=True, refill_lanes=16, slots=3)), ("four rays per lane, greedy r16", dict(greedy=True, refill_lanes=16, slots=4))]
    nwaves = 24
    per = (len(shadow) // nwaves // chunk) * chunk
    L = max(len(q) for q in shadow) + 1

    def matrix(order):
        E = np.full((len(shadow), L), -1, dtype=np.int16)
        for k, src in enumerate(order):
            q = shadow[src]
            E[k, :len(q)] = q
        return E
    orders = {"row-major": np.arange(len(shadow))}
    if live_px is not None:
        x, y = live_px % W, live_px // W
        tile16 = (y // 16) * ((W + 15) // 16) + (x // 16)
        wave8 = ((y % 16) // 8) * 2 + ((x % 16) // 8)
        lane = (y % 8) * 8 + (x % 8)
        orders["tiles (the setup kernel's order: 16x16 workgroups, 8x8 waves)"] = np.lexsort((lane, wave8, tile16))
        orders["tiles, tasks of a 16x16 tile sorted by light"] = np.lexsort((lane, lights, tile16))
        orders["tiles, sorted by light inside 32x32"] = np.lexsort((lane, lights, (y // 32) * ((W + 31) // 32) + (x // 32)))
    for oname, order in orders.items():
        E = matrix(order)
        c = l = 0
        for wv in range(nwaves):
            cc, ll = run_kernel_policy(E, wv * per, per, quorum=24, refill_lanes=24)
            c += cc; l += ll
        print(json.dumps({"queue_order": oname, "policy": "kernel q24 r24", "issue_cycles_per_ray": round(c / (nwaves * per), 1), "lane_utilisation": round(l / (c * 64), 3)}), flush=True)
    E = matrix(orders.get("tiles (the setup kernel's order: 16x16 workgroups, 8x8 waves)", orders["row-major"]))
    # "Speculative" traversal: a lane that reaches a leaf carries it along and keeps walking inner nodes until its next leaf (or the end), so more
    # lanes share a node round.  A shadow ray that the carried leaf would have ended (occluded) walks on in vain until the leaf phase: priced by
    # appending the ray's mean node run between leaves to every ray that ends on a leaf (pessimistic: not every such ray is occluded).
    gaps = [int((q == 0).sum()) / max(1, int((q >= 100).sum())) for q in shadow]
    Es = np.full((E.shape[0], E.shape[1] + 16), -1, dtype=np.int16); Es[:, :E.shape[1]] = E
    order = orders.get("tiles (the setup kernel's order: 16x16 workgroups, 8x8 waves)", orders["row-major"])
    for k, src in enumerate(order):
        q = shadow[src]
        if len(q) and q[-1] >= 100:
            g = min(16, int(round(gaps[src])))
            Es[k, len(q):len(q) + g] = 0
    for name, kw in [("kernel q24 r24, one leaf carried along", dict(quorum=24, refill_lanes=24, postpone=1)), ("kernel q16 r24, one leaf carried", dict(quorum=16, refill_lanes=24, postpone=1)),
                     ("kernel q32 r24, one leaf carried", dict(quorum=32, refill_lanes=24, postpone=1)), ("kernel q24 r24, two leaves carried", dict(quorum=24, refill_lanes=24, postpone=2)),
                     ("greedy r16, one leaf carried", dict(greedy=True, refill_lanes=16, postpone=1))]:
        cyc = lane = opt = 0
        for wv in range(nwaves):
            c, l = run_kernel_policy(Es, wv * per, per, **kw)
            c2, _ = run_kernel_policy(E, wv * per, per, **kw)
            cyc += c; lane += l; opt += c2
        nr = nwaves * per
        print(json.dumps({"policy": name, "issue_cycles_per_ray_if_no_ray_walked_in_vain": round(opt / nr, 1), "issue_cycles_per_ray": round(cyc / nr, 1), "lane_utilisation": round(lane / (cyc * 64), 3), "rays": nr}), flush=True)
    for lq in (4, 8, 12, 16, 24):
        cyc = lane = 0
        for wv in range(nwaves):
            c, l = run_kernel_policy(E, wv * per, per, quorum=24, refill_lanes=24, leaf_quorum=lq)
            cyc += c; lane += l
        nr = nwaves * per
        print(json.dumps({"policy": f"kernel q24 r24 + leaf quorum {lq} (the leaf loop ends when fewer lanes than this still have triangles; they keep them for the next leaf phase)",
                          "issue_cycles_per_ray": round(cyc / nr, 1), "lane_utilisation": round(lane / (cyc * 64), 3)}), flush=True)
    for cost in (0, 100, 200, 300):
        cyc = lane = 0
        nwg = nwaves // 4
        perwg = (len(shadow) // nwg // chunk) * chunk
        for g in range(nwg):
            c, l = run_workgroup_repack(E, g * perwg, perwg, repack_cost=cost)
            cyc += c; lane += l
        nr = nwg * perwg
        print(json.dumps({"policy": f"rays re-packed between the 4 waves of a workgroup every round (upper bound), + {cost} issue cycles per wave and round for the move",
                          "issue_cycles_per_ray": round(cyc / nr, 1), "lane_utilisation": round(lane / (cyc * 64), 3), "rays": nr}), flush=True)
    for name, kw in policies[:1]:
        t0 = time.time()
        cyc = lane = 0
        for wv in range(nwaves):
            c, l = run_kernel_policy(E, wv * per, per, **kw)
            cyc += c; lane += l
        nr = nwaves * per
        print(json.dumps({"policy": name, "issue_cycles_per_ray": round(cyc / nr, 1), "lane_utilisation": round(lane / (cyc * 64), 3), "rays": nr, "sim_s": round(time.time() - t0, 1)}), flush=True)


if __name__ == "__main__":
    main()
