"""The oracle's (and the kernels') fixed-algorithm transcendentals against numpy float64 references
rounded to float32 and against the host libm (via liboracle_libm.so): <= 1 ulp everywhere and
correctly rounded on all but a vanishing fraction of inputs."""
import numpy as np

from oraclelib import lib


def _ulp_err(got, want64):
    want = want64.astype(np.float32)
    ulp = np.spacing(np.abs(want)).astype(np.float64)
    return np.abs(got.astype(np.float64) - want64) / np.maximum(ulp, 1e-45), (got == want)


def _vec(fn, xs):
    return np.array([fn(float(x)) for x in xs], dtype=np.float32)


def test_sin_cos(oracle_built):
    L, Lm = lib(), lib(libm=True)
    xs = np.concatenate([np.linspace(0, 2 * np.pi, 20001), np.random.default_rng(1).uniform(0, 6.2831855, 20000)]).astype(np.float32)
    for name, ref in (("orc_sin", np.sin), ("orc_cos", np.cos)):
        got = _vec(getattr(L, name), xs)
        err, exact = _ulp_err(got, ref(xs.astype(np.float64)))
        assert err.max() <= 1.0, (name, err.max())
        assert exact.mean() > 0.9999
        gm = _vec(getattr(Lm, name), xs)
        assert (np.abs(got.view(np.int32).astype(np.int64) - gm.view(np.int32).astype(np.int64)) <= 1).all()


def test_acos(oracle_built):
    L = lib()
    xs = np.concatenate([np.linspace(-1, 1, 40001), [-1.0, -0.5, 0.0, 0.5, 1.0]]).astype(np.float32)
    got = _vec(L.orc_acos, xs)
    err, exact = _ulp_err(got, np.arccos(xs.astype(np.float64)))
    assert err.max() <= 1.0, err.max()
    assert exact.mean() > 0.9999
    assert np.isnan(L.orc_acos(1.0000001)) and np.isnan(L.orc_acos(float("nan")))
    assert L.orc_acos(1.0) == 0.0


def test_acos_monotone_exhaustive(oracle_built):
    """Every adjacent float pair in [-1, 1] (2^31 of them): acos never increases.  This is what lets the product take
    max_k acos(x_k) of ConeThetaToBox as acos(min_k x_k) and still match the oracle (which keeps the max) bit for bit."""
    assert lib().orc_acos_monotone_violations() == 0
    assert lib(libm=True).orc_acos_monotone_violations() == 0


def test_pow5(oracle_built):
    L = lib()
    xs = np.linspace(0, 1, 30001).astype(np.float32)
    got = _vec(L.orc_pow5, xs)
    err, exact = _ulp_err(got, xs.astype(np.float64) ** 5)
    assert err.max() <= 0.5000001 and exact.all()
