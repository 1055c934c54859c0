"""Host logic of the multi-GPU layer (no GPU): the halo-exchange transfer plan and the cost-balanced band boundaries, through the
C ABI of the built library (fyprt_halo_plan, fyprt_balance_rows)."""
import numpy as np
import pytest

from fypraytracer_amd import capi


def _coverage(plan, n):
    got = [set() for _ in range(n)]
    for recv, owner, r0, r1 in plan:
        assert recv != owner and r0 < r1
        got[recv] |= set(range(r0, r1))
    return got


@pytest.mark.parametrize("bounds,H,halo", [([0, 135, 270, 405, 540, 675, 810, 945, 1080], 1080, 30), ([0, 20, 45, 60, 200], 200, 30), ([0, 100, 200], 200, 30), ([0, 200], 200, 30)])
def test_halo_plan_covers_exactly_the_rows_part2_reads(bounds, H, halo):
    n = len(bounds) - 1
    plan = capi.halo_plan(bounds, halo, H, wrap_row=True)
    got = _coverage(plan, n)
    for r in range(n):
        b, e = bounds[r], bounds[r + 1]
        need = set(range(max(0, b - halo), b)) | set(range(e, min(H, e + halo)))
        if b < halo and min(H, e + halo) < H:
            need.add(H - 1)                                   # the reference's unsigned neighbour wrap (R.cu:1916-1917)
        assert got[r] == need, r
    for recv, owner, r0, r1 in plan:                          # every row comes from the band that owns it
        assert bounds[owner] <= r0 and r1 <= bounds[owner + 1]
    hist = _coverage(capi.halo_plan(bounds, halo, H, wrap_row=False), n)
    for r in range(n):
        assert hist[r] == set(range(max(0, bounds[r] - halo), bounds[r])) | set(range(bounds[r + 1], min(H, bounds[r + 1] + halo)))


def test_balance_rows():
    b = [0, 270, 540, 810, 1080]
    assert capi.balance_rows(b, [1, 1, 1, 1]) == b                                  # balanced already
    nb = capi.balance_rows(b, [2.0, 1.0, 1.0, 1.0])
    assert nb[0] == 0 and nb[-1] == 1080 and nb[1] < 270 and all(x < y for x, y in zip(nb, nb[1:]))
    # equal cost per band after the move, under the piecewise-constant density the routine assumes
    dens = np.repeat(np.array([2.0, 1.0, 1.0, 1.0]) / 270.0, 270)
    costs = [dens[x:y].sum() for x, y in zip(nb, nb[1:])]
    assert max(costs) - min(costs) < 2 * dens.max()
    capped = capi.balance_rows(b, [2.0, 1.0, 1.0, 1.0], max_shift=8)
    assert all(abs(x - y) <= 8 for x, y in zip(capped, b))
    tight = capi.balance_rows([0, 50, 100], [100.0, 1.0], min_rows=30)
    assert tight == [0, 30, 100]
    with pytest.raises(capi.FyprtError):
        capi.balance_rows([0, 50, 100], [1.0, 0.0])


# ---- the RCCL transport's point-to-point sequences: more than one rank per device cannot run on the one-GPU box, so what
# RCCL's matching needs is checked here for every pair of ranks: rank a's sends to b and rank b's receives from a list the same byte
# counts (and buffers, and rows) in the same order, inside one group section.
def _pairing(ops_of_rank):
    n = len(ops_of_rank)
    total = 0
    for a in range(n):
        for b in range(n):
            if a == b:
                continue
            sends = [(buf, off, nbytes) for (recv, peer, buf, off, nbytes) in ops_of_rank[a] if not recv and peer == b]
            recvs = [(buf, off, nbytes) for (recv, peer, buf, off, nbytes) in ops_of_rank[b] if recv and peer == a]
            assert sends == recvs, (a, b)                     # same count, same order, same bytes; the same rows of the same buffer
            total += len(sends)
    for r, ops in enumerate(ops_of_rank):
        assert all(peer != r and nbytes > 0 for (_, peer, _, _, nbytes) in ops)
    return total


DI_EXCHANGE = [32]                 # DIRec
GI_EXCHANGE = [80, 64, 8]          # reservoir, hot record, normal
SET_ROWS = [16, 32, 80, 8]         # accumulation, DI history, GI history, normals


@pytest.mark.parametrize("bounds,H,halo", [([0, 135, 270, 405, 540, 675, 810, 945, 1080], 1080, 30), ([0, 20, 45, 60, 200], 200, 30), ([0, 100, 200], 200, 30),
                                           ([0, 10, 25, 200], 200, 255), ([0, 200], 200, 30)])
@pytest.mark.parametrize("bpp", [DI_EXCHANGE, GI_EXCHANGE])
@pytest.mark.parametrize("wrap", [True, False])
def test_comm_exchange_sends_and_receives_pair_up(bounds, H, halo, bpp, wrap):
    n = len(bounds) - 1
    W = 64
    ops = [capi.comm_ops(0, bounds, r, W, bpp, halo=halo, height=H, wrap_row=wrap) for r in range(n)]
    matched = _pairing(ops)
    plan = capi.halo_plan(bounds, halo, H, wrap_row=wrap)
    assert matched == len(plan) * len(bpp)                    # every planned transfer of every buffer is issued exactly once at each end
    for r in range(n):                                        # and what a rank receives covers exactly its plan entries
        got = sorted((peer, off // (W * bpp[buf]), (off + nbytes) // (W * bpp[buf])) for (recv, peer, buf, off, nbytes) in ops[r] if recv and buf == 0)
        assert got == sorted((owner, r0, r1) for (recv, owner, r0, r1) in plan if recv == r)


@pytest.mark.parametrize("old,new", [([0, 270, 540, 810, 1080], [0, 200, 560, 800, 1080]), ([0, 100, 200], [0, 150, 200]), ([0, 50, 100, 150, 200], [0, 16, 32, 48, 200]),
                                     ([0, 100, 200], [0, 100, 200])])
def test_comm_set_rows_sends_and_receives_pair_up(old, new):
    n = len(old) - 1
    W = 48
    ops = [capi.comm_ops(1, old, r, W, SET_ROWS, new_bounds=new) for r in range(n)]
    _pairing(ops)
    for k in range(n):                                        # a rank receives exactly the rows it gains, once per buffer
        gained = set(range(new[k], new[k + 1])) - set(range(old[k], old[k + 1]))
        for buf, bpp in enumerate(SET_ROWS):
            rows = set()
            for (recv, peer, b, off, nbytes) in ops[k]:
                if recv and b == buf:
                    r0, r1 = off // (W * bpp), (off + nbytes) // (W * bpp)
                    assert old[peer] <= r0 and r1 <= old[peer + 1]          # from the rank that owned them
                    assert not (rows & set(range(r0, r1)))
                    rows |= set(range(r0, r1))
            assert rows == gained
