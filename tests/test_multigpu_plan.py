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
