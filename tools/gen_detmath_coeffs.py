"""Generate the polynomial used by the deterministic acos (asin kernel R(z)).

R(z) = (asin(sqrt(z))/sqrt(z) - 1)/z on z in [0, 0.25], Chebyshev-interpolated
then converted to the monomial basis.  The printed hex-float doubles are pasted
into fypraytracer_amd/csrc/rt_detmath.h and (independently) oracle/oracle_math.h.
"""
import numpy as np
from numpy.polynomial import chebyshev as C, polynomial as P
from fractions import Fraction


def series_coeffs(n):
    # asin(s)/s = sum_k a_k z^k, a_k = (2k)! / (4^k (k!)^2 (2k+1)); R = sum_{k>=1} a_k z^(k-1)
    out = []
    from math import factorial
    for k in range(1, n + 1):
        a = Fraction(factorial(2 * k), (4 ** k) * factorial(k) ** 2 * (2 * k + 1))
        out.append(float(a))
    return np.array(out)


SER = series_coeffs(60)


def R(z):
    z = np.asarray(z, dtype=np.float64)
    out = np.empty_like(z)
    small = z < 0.05
    out[small] = P.polyval(z[small], SER)
    zz = z[~small]
    ss = np.sqrt(zz)
    out[~small] = (np.arcsin(ss) / ss - 1) / zz
    return out


if __name__ == "__main__":
    for deg in (10, 12, 14, 16):
        n = deg + 1
        k = np.arange(n)
        x = np.cos(np.pi * (k + 0.5) / n)
        z = (x + 1) * 0.125
        c = C.chebfit(x, R(z), deg)
        p = C.cheb2poly(c)
        pz = np.zeros(1)
        base = np.array([-1.0, 8.0])
        acc = np.array([1.0])
        for coef in p:
            pz = P.polyadd(pz, coef * acc)
            acc = P.polymul(acc, base)
        zt = np.linspace(0, 0.25, 400001)
        err = np.max(np.abs(P.polyval(zt, pz) - R(zt)))
        print("degree", deg, "max abs err", err)
        if deg == 10:
            print(",\n".join(float(v).hex() for v in pz))
