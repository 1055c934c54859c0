"""The arithmetic contract on the device: rt_math.h's short sequences for sqrt(x), 1/x and 1/sqrt(x) must be the
correctly rounded IEEE results — checked exhaustively, all 2^32 arguments each, against the compiler's own sequences
(fyprt_selftest_math), and on a sample against numpy's (the host's IEEE arithmetic, which the oracle uses)."""
import numpy as np
import pytest

from fypraytracer_amd import capi

pytestmark = pytest.mark.gpu


def test_lean_sqrt_rcp_rsqrt_are_exact_on_every_argument():
    ctx = capi.Context(0)
    bad, first = ctx.selftest_math()
    ctx.close()
    assert bad == [0, 0, 0], f"mismatches (sqrt, rcp, rsqrt) {bad}, first offending argument bits {[hex(x) for x in first]}"
