// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_vec.h header).
//
// Restatement of FYPRayTracer/src/Utility/MathUtils.cuh (RNG, samplers, BRDF,
// octahedral normals, reprojection) plus the transcendental layer.
//
// Transcendentals.  The reference calls cos/sin/pow/acos through glm -> CUDA libm on the
// device (and the host libm in its __host__ instantiation).  Neither is reproducible
// across toolchains, so the oracle evaluates them with a fixed algorithm in binary64
// (every step an exactly-rounded IEEE operation, explicit fma) and rounds once to
// binary32.  The result equals the correctly rounded value except in ~1e-8 of inputs,
// i.e. it agrees with glibc's sinf/cosf/powf/acosf to <= 1 ulp (checked in
// tests/test_oracle_math.py).  Building with -DORC_USE_LIBM swaps in the host libm
// (the reference's own __host__ behaviour) to show the images agree within tolerance.
#pragma once
#include "oracle_vec.h"

namespace orc {

static constexpr float kPi = 3.1415926535f;              // MathUtils.cuh:17

// ---------------------------------------------------------------- transcendentals
#ifdef ORC_USE_LIBM
static inline float t_sin(float x) { return sinf(x); }
static inline float t_cos(float x) { return cosf(x); }
static inline float t_pow5(float x) { return powf(x, 5.0f); }
static inline float t_acos(float x) { return acosf(x); }
#else
// sin/cos for arguments in [0, ~2*pi + small] (the hot path only ever passes 2*pi*u, a
// clamped cone angle, or a light-tree angle in [0, pi]); Cody–Waite by pi/2 in binary64.
static inline void t_sincos_d(float xf, double& s, double& c) {
    const double x = (double)xf;
    const double two_over_pi = 0x1.45f306dc9c883p-1;
    const double pio2_hi = 0x1.921fb54442d18p+0;
    const double pio2_lo = 0x1.1a62633145c07p-54;
    const double kd = __builtin_rint(x * two_over_pi);
    const int k = (int)kd;
    double r = __builtin_fma(-kd, pio2_hi, x);
    r = __builtin_fma(-kd, pio2_lo, r);
    const double z = r * r;
    // Taylor kernels on |r| <= pi/4 (truncation < 1e-15 relative)
    double sp = -1.0 / 6227020800.0;
    sp = __builtin_fma(sp, z, 1.0 / 39916800.0);
    sp = __builtin_fma(sp, z, -1.0 / 362880.0);
    sp = __builtin_fma(sp, z, 1.0 / 5040.0);
    sp = __builtin_fma(sp, z, -1.0 / 120.0);
    sp = __builtin_fma(sp, z, 1.0 / 6.0);
    const double sr = __builtin_fma(-(r * z), sp, r);          // r - r^3 * P(z)
    double cp = 1.0 / 87178291200.0;
    cp = __builtin_fma(cp, z, -1.0 / 479001600.0);
    cp = __builtin_fma(cp, z, 1.0 / 3628800.0);
    cp = __builtin_fma(cp, z, -1.0 / 40320.0);
    cp = __builtin_fma(cp, z, 1.0 / 720.0);
    cp = __builtin_fma(cp, z, -1.0 / 24.0);
    cp = __builtin_fma(cp, z, 0.5);
    const double cr = __builtin_fma(-z, cp, 1.0);              // 1 - z * Q(z)
    switch (k & 3) {
        case 0: s = sr;  c = cr;  break;
        case 1: s = cr;  c = -sr; break;
        case 2: s = -sr; c = -cr; break;
        default: s = -cr; c = sr; break;
    }
}
static inline float t_sin(float x) { double s, c; t_sincos_d(x, s, c); return (float)s; }
static inline float t_cos(float x) { double s, c; t_sincos_d(x, s, c); return (float)c; }
static inline float t_pow5(float x) {            // glm::pow(x, 5.0f) for x in [0,1]
    const double d = (double)x;
    const double d2 = d * d;
    return (float)((d2 * d2) * d);
}
static inline double t_asin_kernel(double z) {    // (asin(sqrt z)/sqrt z - 1)/z on [0, 0.25]
    double p = 0x1.c88ae5be4eda1p-6;
    p = __builtin_fma(p, z, -0x1.bf334244335c0p-8);
    p = __builtin_fma(p, z, 0x1.fa509e4630b10p-7);
    p = __builtin_fma(p, z, 0x1.510d3e4b404ecp-7);
    p = __builtin_fma(p, z, 0x1.cf67181b8b240p-7);
    p = __builtin_fma(p, z, 0x1.1c0cd5e2c5a38p-6);
    p = __builtin_fma(p, z, 0x1.6e8f421105f62p-6);
    p = __builtin_fma(p, z, 0x1.f1c6fee482ca3p-6);
    p = __builtin_fma(p, z, 0x1.6db6dbab38ae8p-5);
    p = __builtin_fma(p, z, 0x1.33333333018c8p-4);
    p = __builtin_fma(p, z, 0x1.55555555555bcp-3);
    return p;
}
static inline float t_acos(float xf) {           // xf in [-1, 1] (callers clamp); NaN -> NaN
    const double x = (double)xf;
    const double pi_d = 0x1.921fb54442d18p+1, pio2_d = 0x1.921fb54442d18p+0;
    const double ax = __builtin_fabs(x);
    if (!(ax <= 1.0)) return __builtin_nanf("");
    if (ax <= 0.5) {
        const double z = x * x;
        const double as = __builtin_fma(x * z, t_asin_kernel(z), x);
        return (float)(pio2_d - as);
    }
    const double z = (1.0 - ax) * 0.5;
    const double s = __builtin_sqrt(z);
    const double as = __builtin_fma(s * z, t_asin_kernel(z), s);   // asin(sqrt z)
    return (float)(x > 0.0 ? 2.0 * as : pi_d - 2.0 * as);
}
#endif

// ---------------------------------------------------------------- RNG (MathUtils.cuh:47-59)
static inline uint32_t pcg_hash(uint32_t input) {
    uint32_t state = input * 747796405u + 2891336453u;
    uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}
static inline float randomFloat(uint32_t& seed) {
    seed = pcg_hash(seed);
    return (float)seed / 4294967296.0f;          // (float)UINT32_MAX rounds to 2^32; result may be 1.0f
}

// ---------------------------------------------------------------- samplers (MathUtils.cuh:61-274)
static inline void BuildOrthonormalBasis(vec3 n, vec3& t, vec3& b) {       // :61-71
    if (n.x * n.x > n.z * n.z) t = normalize(v3(-n.y, n.x, 0.0f));
    else                       t = normalize(v3(0.0f, -n.z, n.y));
    b = normalize(cross(n, t));
}
static inline vec3 CosineSampleHemisphere(vec3 normal, uint32_t& seed) {  // :73-90
    float u1 = randomFloat(seed), u2 = randomFloat(seed);
    float r = sqrtf(u1);
    float theta = 2.0f * kPi * u2;
    float x = r * t_cos(theta), y = r * t_sin(theta);
    float z = sqrtf(gmax(0.0f, 1.0f - u1));
    vec3 t, b; BuildOrthonormalBasis(normal, t, b);
    return normalize(t * x + b * y + normal * z);
}
static inline float CosineHemispherePDF(float cosTheta) { return cosTheta / kPi; }   // :92-95
static inline vec3 UniformSampleHemisphere(vec3 normal, uint32_t& seed) { // :97-114
    float u1 = randomFloat(seed), u2 = randomFloat(seed);
    float phi = 2.0f * kPi * u1;
    float cosTheta = u2;
    float sinTheta = sqrtf(1.0f - cosTheta * cosTheta);
    float x = sinTheta * t_cos(phi), y = sinTheta * t_sin(phi), z = cosTheta;
    vec3 t, b; BuildOrthonormalBasis(normal, t, b);
    return normalize(t * x + b * y + normal * z);
}
static inline float UniformHemispherePDF() { return 1 / (2 * kPi); }       // :116
static inline vec3 GGXSampleHemisphere(vec3 normal, vec3 V, float roughness, uint32_t& seed, float& outPDF) { // :118-174
    float u1 = randomFloat(seed), u2 = randomFloat(seed);
    float alpha = roughness * roughness;
    float phi = 2.0f * kPi * u2;
    float cosTheta = sqrtf((1.0f - u1) / (1.0f + (alpha * alpha - 1.0f) * u1));
    cosTheta = gclamp(cosTheta, 0.0f, 1.0f);
    float sinTheta = sqrtf(fmaxf(0.0f, 1.0f - cosTheta * cosTheta));
    vec3 Ht = v3(sinTheta * t_cos(phi), sinTheta * t_sin(phi), cosTheta);
    vec3 T, B; BuildOrthonormalBasis(normal, T, B);
    vec3 H = normalize(Ht.x * T + Ht.y * B + Ht.z * normal);
    vec3 L = reflect(-V, H);
    float NdotL = dot(normal, L);
    if (NdotL <= 0.0f) { outPDF = 0.0f; return v3(0.0f); }
    float NdotH = dot(normal, H), VdotH = dot(V, H);
    if (VdotH <= 0.0f || NdotH <= 0.0f) { outPDF = 0.0f; return v3(0.0f); }
    float a2 = alpha * alpha;
    float denom = (NdotH * NdotH) * (a2 - 1.0f) + 1.0f;
    float D = a2 / (kPi * denom * denom);
    float p_H = D * NdotH;
    outPDF = p_H / (4.0f * VdotH);
    return L;
}
static inline float GGXHemispherePDF(vec3 N, vec3 V, vec3 L, float roughness) {       // :176-190
    vec3 H = normalize(V + L);
    float NdotH = gmax(dot(N, H), 0.0f), VdotH = gmax(dot(V, H), 0.0f);
    if (NdotH <= 0.0f || VdotH <= 0.0f) return 0.0f;
    float alpha = roughness * roughness, a2 = alpha * alpha;
    float denom = (NdotH * NdotH) * (a2 - 1.0f) + 1.0f;
    float D = a2 / (kPi * denom * denom);
    return D * NdotH / (4.0f * VdotH);
}
static inline vec3 BRDFSampleHemisphere(vec3 normal, vec3 V, vec3 albedo, float metallic, float roughness,
                                        uint32_t& seed, float& outPDF) {              // :192-244
    vec3 L; float pdfSpecular = 0.0f, pdfDiffuse = 0.0f, wSpecular;
    if (metallic == 1.0f) return GGXSampleHemisphere(normal, V, roughness, seed, outPDF);
    else if (metallic == 0.0f) {
        L = CosineSampleHemisphere(normal, seed);
        outPDF = CosineHemispherePDF(gmax(dot(normal, L), 0.0f));
        return L;
    } else {
        vec3 F0 = mix(v3(0.04f), albedo, metallic);
        vec3 F = F0 + (1.0f - F0) * t_pow5(1.0f - gmax(dot(normal, V), 0.0f));
        wSpecular = (F.x + F.y + F.z) / 3.0f;
    }
    float rnd = randomFloat(seed);
    if (rnd <= wSpecular) {
        L = GGXSampleHemisphere(normal, V, roughness, seed, pdfSpecular);
        pdfDiffuse = CosineHemispherePDF(gmax(dot(normal, L), 0.0f));
    } else {
        L = CosineSampleHemisphere(normal, seed);
        pdfDiffuse = CosineHemispherePDF(gmax(dot(normal, L), 0.0f));
        pdfSpecular = GGXHemispherePDF(normal, V, L, roughness);
    }
    outPDF = wSpecular * pdfSpecular + (1.0f - wSpecular) * pdfDiffuse;
    return L;
}
static inline float BRDFHemispherePDF(vec3 N, vec3 V, vec3 L, vec3 albedo, float metallic, float roughness) { // :246-274
    if (metallic == 1.0f) return GGXHemispherePDF(N, V, L, roughness);
    if (metallic == 0.0f) return CosineHemispherePDF(gmax(dot(N, L), 0.0f));
    vec3 F0 = mix(v3(0.04f), albedo, metallic);
    float NdotV = gmax(dot(N, V), 0.0f);
    vec3 F = F0 + (1.0f - F0) * t_pow5(1.0f - NdotV);
    float wSpec = (F.x + F.y + F.z) * (1.0f / 3.0f);
    float pdfSpec = GGXHemispherePDF(N, V, L, roughness);
    float pdfDiff = CosineHemispherePDF(gmax(dot(N, L), 0.0f));
    return wSpec * pdfSpec + (1.0f - wSpec) * pdfDiff;
}
static inline vec3 CalculateBRDF(vec3 N, vec3 V, vec3 L, vec3 albedo, float metallic, float roughness) {      // :276-317
    const float invPI = 1.0f / kPi;
    float a = roughness * roughness, a2 = a * a;
    vec3 H = normalize(V + L);
    float NdotL = gmax(dot(N, L), 0.0f), NdotV = gmax(dot(N, V), 0.0f);
    float NdotH = gmax(dot(N, H), 0.0f), VdotH = gmax(dot(V, H), 0.0f);
    if (NdotL == 0.0f || NdotV == 0.0f) return v3(0.0f);
    vec3 F0 = mix(v3(0.04f), albedo, metallic);
    vec3 F = F0 + (1.0f - F0) * t_pow5(1.0f - VdotH);
    float k = roughness / 2.0f;
    float G_V = NdotV / (NdotV * (1.0f - k) + k);
    float G_L = NdotL / (NdotL * (1.0f - k) + k);
    float G = G_V * G_L;
    vec3 kD = (1.0f - F);
    vec3 diffuse = kD * albedo * invPI;
    float denominator = (NdotH * NdotH) * (a2 - 1.0f) + 1.0f;
    float D = a2 * invPI / gmax(denominator * denominator, 1e-12f);
    vec3 specular = (D * G * F) / gmax(4.0f * NdotV * NdotL, 1e-12f);
    return diffuse + specular;
}

// ---------------------------------------------------------------- octahedral normals (:328-374)
static inline vec2 EncodeOctahedral(vec3 v) {
    v /= (fabsf(v.x) + fabsf(v.y) + fabsf(v.z));
    vec2 enc{v.x, v.y};
    if (v.z < 0.0f) {
        float ex = enc.x, ey = enc.y;
        float xx = 1.0f - fabsf(ey), yy = 1.0f - fabsf(ex);
        float sx = (ex >= 0.0f) ? 1.0f : -1.0f, sy = (ey >= 0.0f) ? 1.0f : -1.0f;
        enc.x = xx * sx; enc.y = yy * sy;
    }
    return enc;
}
static inline vec3 DecodeOctahedral(vec2 e) {
    float ex = e.x, ey = e.y;
    vec3 v = v3(ex, ey, 1.0f - fabsf(ex) - fabsf(ey));
    if (v.z < 0.0f) {
        float sx = (ex >= 0.0f) ? 1.0f : -1.0f, sy = (ey >= 0.0f) ? 1.0f : -1.0f;
        float nx = (1.0f - fabsf(ey)) * sx, ny = (1.0f - fabsf(ex)) * sy;
        v.x = nx; v.y = ny;
    }
    return normalize(v);
}
// GetUVFromNDC(projection, view, worldPos)  (:376-396)
static inline vec2 GetUVFromNDC(const mat4& projection, const mat4& view, vec3 worldPos) {
    vec4 clip = (projection * view) * v4(worldPos, 1.0f);
    vec2 ndc;
    if (clip.w == 0.0f) ndc = {0.0f, 0.0f};
    else { vec3 n = xyz(clip) / clip.w; ndc = {n.x, n.y}; }
    return ndc * 0.5f + 0.5f;
}

// ---------------------------------------------------------------- colour pack (ColorUtils.cuh:14-41)
// (uint8_t)(c*255.0f): truncation; NaN / out-of-range are UB in C++ and give 0 with CUDA's
// saturating conversions after the [0,1] clamp; the oracle fixes NaN -> 0.
static inline uint32_t f2u8(float c) {
    float s = c * 255.0f;
    if (!(s >= 0.0f)) return 0u;
    if (s >= 255.0f) return 255u;
    return (uint32_t)(int)s;
}
static inline uint32_t ConvertToRGBA(vec4 c) {
    return (f2u8(c.w) << 24) | (f2u8(c.z) << 16) | (f2u8(c.y) << 8) | f2u8(c.x);
}
static inline vec4 UnpackABGR(uint32_t p) {
    const float k = 1.0f / 255.0f;
    return {(float)(p & 0xFF) * k, (float)((p >> 8) & 0xFF) * k, (float)((p >> 16) & 0xFF) * k, (float)((p >> 24) & 0xFF) * k};
}

}  // namespace orc
