// Device math for the gfx950 trace + shade kernels.
//
// Arithmetic contract: compiled with -ffp-contract=off, so every fp32
// + - * / sqrt below is one correctly rounded IEEE-754 operation (hipcc's default
// -fhip-fp32-correctly-rounded-divide-sqrt keeps / and sqrt exact); fused multiply-adds
// appear only where written as __builtin_fma*.  Transcendentals (cos, sin, x^5, acos) are
// evaluated in binary64 with a fixed polynomial and rounded once to binary32, so a frame is
// reproducible bit for bit on any IEEE machine (the CUDA reference is not: it inherits
// nvcc's fmad contraction and CUDA libm).  Formulas follow the reference's
// MathUtils.cuh:47-396 and the glm definitions it relies on (normalize = v * (1/sqrt(dot)),
// dot = (x*x' + y*y') + z*z', reflect, mix, clamp = min(max())).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define RT_DEV __device__ __forceinline__

namespace rt {

struct f2 { float x, y; };
struct f3 { float x, y, z; };
struct f4 { float x, y, z, w; };

RT_DEV float gmax(float a, float b) { return (a < b) ? b : a; }
RT_DEV float gmin(float a, float b) { return (b < a) ? b : a; }
RT_DEV float gclamp(float x, float lo, float hi) { return gmin(gmax(x, lo), hi); }

RT_DEV f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
RT_DEV f3 splat3(float a) { return mk3(a, a, a); }
RT_DEV f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_DEV f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_DEV f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
RT_DEV f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
RT_DEV f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
RT_DEV f3 operator*(float s, f3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
RT_DEV f3 operator/(f3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
RT_DEV f3 operator+(f3 a, float s) { return mk3(a.x + s, a.y + s, a.z + s); }
RT_DEV f3 rsub(float s, f3 a) { return mk3(s - a.x, s - a.y, s - a.z); }          // float - vec3
RT_DEV float dot(f3 a, f3 b) { float tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z; return (tx + ty) + tz; }
RT_DEV f3 cross(f3 a, f3 b) { return mk3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y); }

// ---- correctly rounded sqrt(x), 1/x and 1/sqrt(x) in fewer instructions
// hipcc's correctly rounded sqrtf is 16 VALU instructions (denormal pre-scaling, v_sqrt_f32, a test of both neighbours through
// two compares, class fix-up) and its 1/x is 11 (v_div_scale x 2, v_rcp, five fma, v_div_fmas, v_div_fixup), plus the wait states
// between each compare and its v_cndmask.  For an argument whose magnitude is in [2^-96, 2^96) — no intermediate can leave the normal range —
// the sequences below give THE SAME BITS in 8 and 5 instructions: the hardware estimate (v_rsq_f32 / v_rcp_f32, 1 ulp) refined with
// fused residuals (Markstein).  "The same bits" is not an argument but a measurement: fyprt_selftest_math compares each of the three
// functions with the compiler's sequence on ALL 2^32 arguments (tests/test_gpu_math.py; 0 mismatches on gfx950).  Everything outside
// the range (zero, denormals, huge, inf, NaN, negative sqrt arguments) takes the compiler's sequence.  The light-tree importance
// (nine normalisations per cluster) and every normalize() / length() of the shading code go through these.
#ifdef RT_NO_LEAN_MATH       // A/B builds (tools/build_variant.sh): the compiler's sequences everywhere
RT_DEV bool lean_range(float) { return false; }
#else
RT_DEV bool lean_range(float x) { return (__float_as_uint(x) - 0x0F800000u) < (0x6F800000u - 0x0F800000u); }      // 2^-96 <= x < 2^96 (positive, finite)
#endif
RT_DEV float lean_sqrt(float x) {
    const float y = __builtin_amdgcn_rsqf(x);
    float g = x * y, h = 0.5f * y;
    const float r = __builtin_fmaf(-h, g, 0.5f);
    g = __builtin_fmaf(g, r, g); h = __builtin_fmaf(h, r, h);
    const float d = __builtin_fmaf(-g, g, x);
    return __builtin_fmaf(d, h, g);
}
RT_DEV float lean_rcp(float s) {
    float y = __builtin_amdgcn_rcpf(s);
    float e = __builtin_fmaf(-s, y, 1.0f);
    y = __builtin_fmaf(e, y, y);
    e = __builtin_fmaf(-s, y, 1.0f);
    return __builtin_fmaf(e, y, y);
}
RT_DEV float sqrt_exact(float x) { return lean_range(x) ? lean_sqrt(x) : __builtin_sqrtf(x); }                                 // == sqrtf(x)
RT_DEV float rcp_exact(float x) { return lean_range(__builtin_fabsf(x)) ? lean_rcp(x) : 1.0f / x; }                           // == 1.0f / x
RT_DEV float rsqrt_exact(float x) { return lean_range(x) ? lean_rcp(lean_sqrt(x)) : 1.0f / __builtin_sqrtf(x); }              // == 1.0f / sqrtf(x), two roundings

RT_DEV float length(f3 a) { return sqrt_exact(dot(a, a)); }
RT_DEV f3 normalize(f3 a) { float inv = rsqrt_exact(dot(a, a)); return a * inv; }
RT_DEV f3 reflect(f3 I, f3 N) { return I - N * dot(N, I) * 2.0f; }
RT_DEV f3 mix(f3 a, f3 b, float t) { return a * (1.0f - t) + b * t; }
RT_DEV bool finitef(float x) { return __builtin_fabsf(x) <= 3.402823466e+38f; }   // false for NaN / inf

RT_DEV f4 mk4(float x, float y, float z, float w) { f4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
RT_DEV f4 operator+(f4 a, f4 b) { return mk4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
RT_DEV f4 operator*(f4 a, float s) { return mk4(a.x * s, a.y * s, a.z * s, a.w * s); }

struct m4 { f4 c[4]; };   // column-major (glm::mat4 memory order)
RT_DEV f4 mul(const m4& m, f4 v) { return (m.c[0] * v.x + m.c[1] * v.y) + (m.c[2] * v.z + m.c[3] * v.w); }
RT_DEV m4 mul(const m4& a, const m4& b) {
    m4 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) r.c[j] = ((a.c[0] * b.c[j].x + a.c[1] * b.c[j].y) + a.c[2] * b.c[j].z) + a.c[3] * b.c[j].w;
    return r;
}

constexpr float kPi = 3.1415926535f;   // MathUtils.cuh:17

// ------------------------------------------------------------------ transcendentals (binary64 kernels)
// The polynomial coefficients live in constant memory: the compiler fetches them with scalar loads into SGPR pairs and feeds
// them to v_fma_f64 as the scalar operand.  Written as literals they were hoisted out of the light-tree loops into ~45 VGPRs
// (a 64-bit literal cannot be an inline operand), which cost the shading kernels one to two waves of occupancy.
__constant__ double kSinPoly[6] = {-1.0 / 6227020800.0, 1.0 / 39916800.0, -1.0 / 362880.0, 1.0 / 5040.0, -1.0 / 120.0, 1.0 / 6.0};
__constant__ double kCosPoly[7] = {1.0 / 87178291200.0, -1.0 / 479001600.0, 1.0 / 3628800.0, -1.0 / 40320.0, 1.0 / 720.0, -1.0 / 24.0, 0.5};
__constant__ double kAsinPoly[11] = {0x1.c88ae5be4eda1p-6, -0x1.bf334244335c0p-8, 0x1.fa509e4630b10p-7, 0x1.510d3e4b404ecp-7, 0x1.cf67181b8b240p-7,
                                     0x1.1c0cd5e2c5a38p-6, 0x1.6e8f421105f62p-6, 0x1.f1c6fee482ca3p-6, 0x1.6db6dbab38ae8p-5, 0x1.33333333018c8p-4,
                                     0x1.55555555555bcp-3};
__constant__ double kPiSplit[4] = {0x1.45f306dc9c883p-1, 0x1.921fb54442d18p+0, 0x1.1a62633145c07p-54, 0x1.921fb54442d18p+1};   // 2/pi, pi/2 hi, pi/2 lo, pi
// fma(a, b, c) with a wave-uniform c (a coefficient from constant memory) as the instruction's scalar operand.  Left to itself the
// compiler turns `p = fma(p, z, c)` into v_fmac_f64 and first copies c from its SGPR pair into the accumulator with two v_mov_b32 —
// three VALU issues per Horner step instead of one (acos_f: 77 -> 55 instructions).  Same operation, same bits.
RT_DEV double fma_uniform_c(double a, double b, double c) {
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c));
    return r;
}
RT_DEV void sincos_f(float xf, float& s_out, float& c_out) {
    // valid for xf in [0, 2*pi + eps] (2*pi*u) and small positive angles; Cody–Waite by pi/2
    const double x = (double)xf;
    const double kd = __builtin_rint(x * kPiSplit[0]);
    const int k = (int)kd;
    double r = __builtin_fma(-kd, kPiSplit[1], x);
    r = __builtin_fma(-kd, kPiSplit[2], r);
    const double z = r * r;
    double sp = kSinPoly[0];
#pragma unroll
    for (int i = 1; i < 6; ++i) sp = fma_uniform_c(sp, z, kSinPoly[i]);
    const double sr = __builtin_fma(-(r * z), sp, r);
    double cp = kCosPoly[0];
#pragma unroll
    for (int i = 1; i < 7; ++i) cp = fma_uniform_c(cp, z, kCosPoly[i]);
    const double cr = __builtin_fma(-z, cp, 1.0);
    const int q = k & 3;
    const double s = (q == 0) ? sr : (q == 1) ? cr : (q == 2) ? -sr : -cr;
    const double c = (q == 0) ? cr : (q == 1) ? -sr : (q == 2) ? -cr : sr;
    s_out = (float)s; c_out = (float)c;
}
RT_DEV float cos_f(float x) { float s, c; sincos_f(x, s, c); return c; }
RT_DEV float pow5_f(float x) { const double d = (double)x; const double d2 = d * d; return (float)((d2 * d2) * d); }   // glm::pow(x, 5.0f)
RT_DEV double asin_kernel(double z) {   // (asin(sqrt z)/sqrt z - 1)/z on [0, 0.25]; tools/gen_detmath_coeffs.py
    double p = kAsinPoly[0];
#pragma unroll
    for (int i = 1; i < 11; ++i) p = fma_uniform_c(p, z, kAsinPoly[i]);
    return p;
}
RT_DEV float acos_f(float xf) {
    // |x| <= 0.5: pi/2 - asin(x) with asin(x) = x + x*z*P(z), z = x*x;  else: 2*asin(sqrt z) with z = (1 - |x|)/2, reflected for x < 0.
    // Both ranges share ONE evaluation of the polynomial (a wave whose lanes fall in both ranges would otherwise run it twice); the
    // per-lane operations and their order are those of the two-branch form.
    const double x = (double)xf;
    const double ax = __builtin_fabs(x);
    if (!(ax <= 1.0)) return __builtin_nanf("");
    const bool small = ax <= 0.5;
    const double z = small ? x * x : (1.0 - ax) * 0.5;
    double s = x;
    if (!small) s = __builtin_sqrt(z);
    const double as = __builtin_fma(s * z, asin_kernel(z), s);
    return (float)(small ? kPiSplit[1] - as : (x > 0.0 ? 2.0 * as : kPiSplit[3] - 2.0 * as));
}

// ------------------------------------------------------------------ RNG (MathUtils.cuh:47-59)
RT_DEV uint32_t pcg_hash(uint32_t input) {
    uint32_t state = input * 747796405u + 2891336453u;
    uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}
RT_DEV float rnd(uint32_t& seed) { seed = pcg_hash(seed); return (float)seed * 0x1p-32f; }   // == seed / float(UINT32_MAX) (2^32)

// ------------------------------------------------------------------ samplers + BRDF (MathUtils.cuh:61-317)
RT_DEV void onb(f3 n, f3& t, f3& b) {
    if (n.x * n.x > n.z * n.z) t = normalize(mk3(-n.y, n.x, 0.0f));
    else                       t = normalize(mk3(0.0f, -n.z, n.y));
    b = normalize(cross(n, t));
}
RT_DEV f3 to_world(f3 n, float x, float y, float z) { f3 t, b; onb(n, t, b); return normalize(t * x + b * y + n * z); }
RT_DEV f3 sample_cosine(f3 n, uint32_t& seed) {
    float u1 = rnd(seed), u2 = rnd(seed);
    float r = __builtin_sqrtf(u1), s, c; sincos_f(2.0f * kPi * u2, s, c);
    return to_world(n, r * c, r * s, __builtin_sqrtf(gmax(0.0f, 1.0f - u1)));
}
RT_DEV float pdf_cosine(float cosTheta) { return cosTheta / kPi; }
RT_DEV f3 sample_uniform(f3 n, uint32_t& seed) {
    float u1 = rnd(seed), u2 = rnd(seed);
    float s, c; sincos_f(2.0f * kPi * u1, s, c);
    float sinT = __builtin_sqrtf(1.0f - u2 * u2);
    return to_world(n, sinT * c, sinT * s, u2);
}
RT_DEV float pdf_uniform() { return 1 / (2 * kPi); }
RT_DEV f3 sample_ggx(f3 n, f3 V, float roughness, uint32_t& seed, float& pdf) {
    float u1 = rnd(seed), u2 = rnd(seed);
    float alpha = roughness * roughness;
    float sp, cp; sincos_f(2.0f * kPi * u2, sp, cp);
    float cosT = __builtin_sqrtf((1.0f - u1) / (1.0f + (alpha * alpha - 1.0f) * u1));
    cosT = gclamp(cosT, 0.0f, 1.0f);
    float sinT = __builtin_sqrtf(__builtin_fmaxf(0.0f, 1.0f - cosT * cosT));
    f3 T, B; onb(n, T, B);
    f3 H = normalize((sinT * cp) * T + (sinT * sp) * B + cosT * n);
    f3 L = reflect(-V, H);
    float NdotL = dot(n, L);
    if (NdotL <= 0.0f) { pdf = 0.0f; return splat3(0.0f); }
    float NdotH = dot(n, H), VdotH = dot(V, H);
    if (VdotH <= 0.0f || NdotH <= 0.0f) { pdf = 0.0f; return splat3(0.0f); }
    float a2 = alpha * alpha;
    float denom = (NdotH * NdotH) * (a2 - 1.0f) + 1.0f;
    float D = a2 / (kPi * denom * denom);
    pdf = (D * NdotH) / (4.0f * VdotH);
    return L;
}
RT_DEV float pdf_ggx(f3 N, f3 V, f3 L, float roughness) {
    f3 H = normalize(V + L);
    float NdotH = gmax(dot(N, H), 0.0f), VdotH = gmax(dot(V, H), 0.0f);
    if (NdotH <= 0.0f || VdotH <= 0.0f) return 0.0f;
    float alpha = roughness * roughness, a2 = alpha * alpha;
    float denom = (NdotH * NdotH) * (a2 - 1.0f) + 1.0f;
    float D = a2 / (kPi * denom * denom);
    return D * NdotH / (4.0f * VdotH);
}
RT_DEV float fresnel_weight(f3 n, f3 V, f3 albedo, float metallic, bool third) {
    f3 F0 = mix(splat3(0.04f), albedo, metallic);
    f3 F = F0 + rsub(1.0f, F0) * pow5_f(1.0f - gmax(dot(n, V), 0.0f));
    float sum = (F.x + F.y) + F.z;
    return third ? sum * (1.0f / 3.0f) : sum / 3.0f;      // MathUtils.cuh:265 vs :218
}
RT_DEV f3 sample_brdf(f3 n, f3 V, f3 albedo, float metallic, float roughness, uint32_t& seed, float& pdf) {
    if (metallic == 1.0f) return sample_ggx(n, V, roughness, seed, pdf);
    if (metallic == 0.0f) { f3 L = sample_cosine(n, seed); pdf = pdf_cosine(gmax(dot(n, L), 0.0f)); return L; }
    float wS = fresnel_weight(n, V, albedo, metallic, false);
    float r = rnd(seed), pS = 0.0f, pD = 0.0f; f3 L;
    if (r <= wS) { L = sample_ggx(n, V, roughness, seed, pS); pD = pdf_cosine(gmax(dot(n, L), 0.0f)); }
    else { L = sample_cosine(n, seed); pD = pdf_cosine(gmax(dot(n, L), 0.0f)); pS = pdf_ggx(n, V, L, roughness); }
    pdf = wS * pS + (1.0f - wS) * pD;
    return L;
}
RT_DEV float pdf_brdf(f3 N, f3 V, f3 L, f3 albedo, float metallic, float roughness) {
    if (metallic == 1.0f) return pdf_ggx(N, V, L, roughness);
    if (metallic == 0.0f) return pdf_cosine(gmax(dot(N, L), 0.0f));
    float wS = fresnel_weight(N, V, albedo, metallic, true);
    float pS = pdf_ggx(N, V, L, roughness);
    float pD = pdf_cosine(gmax(dot(N, L), 0.0f));
    return wS * pS + (1.0f - wS) * pD;
}
RT_DEV f3 eval_brdf(f3 N, f3 V, f3 L, f3 albedo, float metallic, float roughness) {   // CalculateBRDF
    const float invPI = 1.0f / kPi;
    float a = roughness * roughness, a2 = a * a;
    f3 H = normalize(V + L);
    float NdotL = gmax(dot(N, L), 0.0f), NdotV = gmax(dot(N, V), 0.0f);
    float NdotH = gmax(dot(N, H), 0.0f), VdotH = gmax(dot(V, H), 0.0f);
    if (NdotL == 0.0f || NdotV == 0.0f) return splat3(0.0f);
    f3 F0 = mix(splat3(0.04f), albedo, metallic);
    f3 F = F0 + rsub(1.0f, F0) * pow5_f(1.0f - VdotH);
    float k = roughness / 2.0f;
    float G = (NdotV / (NdotV * (1.0f - k) + k)) * (NdotL / (NdotL * (1.0f - k) + k));
    f3 diffuse = (rsub(1.0f, F) * albedo) * invPI;
    float den = (NdotH * NdotH) * (a2 - 1.0f) + 1.0f;
    float D = a2 * invPI / gmax(den * den, 1e-12f);
    f3 specular = ((D * G) * F) / gmax(4.0f * NdotV * NdotL, 1e-12f);
    return diffuse + specular;
}

// ------------------------------------------------------------------ octahedral normals (MathUtils.cuh:328-374)
RT_DEV f2 oct_encode(f3 v) {
    float s = (__builtin_fabsf(v.x) + __builtin_fabsf(v.y)) + __builtin_fabsf(v.z);
    v = v / s;
    f2 e; e.x = v.x; e.y = v.y;
    if (v.z < 0.0f) {
        float ex = e.x, ey = e.y;
        e.x = (1.0f - __builtin_fabsf(ey)) * ((ex >= 0.0f) ? 1.0f : -1.0f);
        e.y = (1.0f - __builtin_fabsf(ex)) * ((ey >= 0.0f) ? 1.0f : -1.0f);
    }
    return e;
}
RT_DEV f3 oct_decode(f2 e) {
    f3 v = mk3(e.x, e.y, (1.0f - __builtin_fabsf(e.x)) - __builtin_fabsf(e.y));
    if (v.z < 0.0f) {
        float nx = (1.0f - __builtin_fabsf(e.y)) * ((e.x >= 0.0f) ? 1.0f : -1.0f);
        float ny = (1.0f - __builtin_fabsf(e.x)) * ((e.y >= 0.0f) ? 1.0f : -1.0f);
        v.x = nx; v.y = ny;
    }
    return normalize(v);
}

// ------------------------------------------------------------------ colour pack (ColorUtils.cuh:14-41)
RT_DEV uint32_t to_u8(float c) { float s = c * 255.0f; if (!(s >= 0.0f)) return 0u; if (s >= 255.0f) return 255u; return (uint32_t)(int)s; }
RT_DEV uint32_t pack_abgr(f4 c) { return (to_u8(c.w) << 24) | (to_u8(c.z) << 16) | (to_u8(c.y) << 8) | to_u8(c.x); }
RT_DEV f4 unpack_abgr(uint32_t p) {
    const float k = 1.0f / 255.0f;
    return mk4((float)(p & 0xFF) * k, (float)((p >> 8) & 0xFF) * k, (float)((p >> 16) & 0xFF) * k, (float)((p >> 24) & 0xFF) * k);
}

}  // namespace rt
