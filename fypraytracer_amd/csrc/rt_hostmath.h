// Host-side twins of the device's deterministic transcendentals (rt_math.h: sincos_f, acos_f): the same fixed algorithms in binary64 —
// Cody–Waite reduction by pi/2 + Taylor kernels for sin / cos, a minimax kernel for asin — every step one correctly rounded IEEE
// operation (explicit fma), rounded to binary32 once.  The light-tree builder (lighttree_build.cpp) uses them instead of the C library's
// acos / cos / sin, so the trees — and with them the light-sampling distribution of LIGHT_SOURCE_SAMPLING and NEE — are the same bits
// on every host, whatever its libm (VERDICT r02 "what's weak" #7); tests/test_host_scene.py checks that the object file imports no
// libm transcendental at all.  Coefficients: rt_math.h (kSinPoly, kCosPoly, kAsinPoly, kPiSplit), tools/gen_detmath_coeffs.py.
#pragma once
#include <cmath>

namespace rth {

inline void det_sincos(float xf, float& s_out, float& c_out) {
    if (!(xf == xf) || std::fabs(xf) > 1.0e6f) { s_out = c_out = __builtin_nanf(""); return; }   // NaN in, NaN out (the builder's NaN cones rely on it); no huge arguments here
    const double x = (double)xf;
    const double kd = __builtin_rint(x * 0x1.45f306dc9c883p-1);
    const long k = (long)kd;
    double r = __builtin_fma(-kd, 0x1.921fb54442d18p+0, x);
    r = __builtin_fma(-kd, 0x1.1a62633145c07p-54, r);
    const double z = r * r;
    static const double sinPoly[6] = {-1.0 / 6227020800.0, 1.0 / 39916800.0, -1.0 / 362880.0, 1.0 / 5040.0, -1.0 / 120.0, 1.0 / 6.0};
    static const double cosPoly[7] = {1.0 / 87178291200.0, -1.0 / 479001600.0, 1.0 / 3628800.0, -1.0 / 40320.0, 1.0 / 720.0, -1.0 / 24.0, 0.5};
    double sp = sinPoly[0];
    for (int i = 1; i < 6; ++i) sp = __builtin_fma(sp, z, sinPoly[i]);
    const double sr = __builtin_fma(-(r * z), sp, r);
    double cp = cosPoly[0];
    for (int i = 1; i < 7; ++i) cp = __builtin_fma(cp, z, cosPoly[i]);
    const double cr = __builtin_fma(-z, cp, 1.0);
    const int q = (int)(k & 3);
    const double s = (q == 0) ? sr : (q == 1) ? cr : (q == 2) ? -sr : -cr;
    const double c = (q == 0) ? cr : (q == 1) ? -sr : (q == 2) ? -cr : sr;
    s_out = (float)s; c_out = (float)c;
}
inline float det_cos(float x) { float s, c; det_sincos(x, s, c); return c; }
inline float det_sin(float x) { float s, c; det_sincos(x, s, c); return s; }

inline float det_acos(float xf) {
    const double x = (double)xf;
    const double ax = __builtin_fabs(x);
    if (!(ax <= 1.0)) return __builtin_nanf("");
    static const double asinPoly[11] = {0x1.c88ae5be4eda1p-6, -0x1.bf334244335c0p-8, 0x1.fa509e4630b10p-7, 0x1.510d3e4b404ecp-7, 0x1.cf67181b8b240p-7,
                                        0x1.1c0cd5e2c5a38p-6, 0x1.6e8f421105f62p-6, 0x1.f1c6fee482ca3p-6, 0x1.6db6dbab38ae8p-5, 0x1.33333333018c8p-4,
                                        0x1.55555555555bcp-3};
    const bool small = ax <= 0.5;
    const double z = small ? x * x : (1.0 - ax) * 0.5;
    const double s = small ? x : __builtin_sqrt(z);
    double p = asinPoly[0];
    for (int i = 1; i < 11; ++i) p = __builtin_fma(p, z, asinPoly[i]);
    const double as = __builtin_fma(s * z, p, s);
    return (float)(small ? 0x1.921fb54442d18p+0 - as : (x > 0.0 ? 2.0 * as : 0x1.921fb54442d18p+1 - 2.0 * as));
}

}  // namespace rth
