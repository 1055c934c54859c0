// ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is linked, imported or
// executed by the product path (fypraytracer_amd/).  Only tests/, __graft_entry__.smoke()
// and bench.py's cpu_baseline leg may use it, and only as the checker.
//
// Minimal fp32 vector / matrix layer restating the *documented* semantics of the glm
// subset the reference's hot path uses (glm is an un-vendored submodule dependency of
// the reference — SURVEY.md §8c — so its arithmetic is restated here; parity at this
// layer is "unpinned" by the reference and pinned by this file).
//
//   normalize(v) = v * (1 / sqrt(dot(v,v)))          length(v) = sqrt(dot(v,v))
//   dot(vec3)    = (x*x' + y*y') + z*z'               dot(vec4) = (x+y) + (z+w) pairing
//   cross, reflect(I,N) = I - N*dot(N,I)*2,  mix(a,b,t) = a*(1-t) + b*t
//   clamp(x,lo,hi) = min(max(x,lo),hi),  max(a,b) = (a<b)?b:a,  min(a,b) = (b<a)?b:a
//   mat4 is column-major; mat4*vec4 = (c0*x + c1*y) + (c2*z + c3*w)
//
// Compiled with -ffp-contract=off: every + - * / is one IEEE-754 binary32 operation.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace orc {

struct vec2 { float x, y; };
struct vec3 { float x, y, z; };
struct vec4 { float x, y, z, w; };

static inline float gmax(float a, float b) { return (a < b) ? b : a; }   // glm::max
static inline float gmin(float a, float b) { return (b < a) ? b : a; }   // glm::min
static inline float gclamp(float x, float lo, float hi) { return gmin(gmax(x, lo), hi); }
static inline int   iclamp(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }

static inline vec3 v3(float a) { return {a, a, a}; }
static inline vec3 v3(float x, float y, float z) { return {x, y, z}; }
static inline vec3 operator+(vec3 a, vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline vec3 operator-(vec3 a, vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline vec3 operator-(vec3 a) { return {-a.x, -a.y, -a.z}; }
static inline vec3 operator*(vec3 a, vec3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
static inline vec3 operator*(vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
static inline vec3 operator*(float s, vec3 a) { return {s * a.x, s * a.y, s * a.z}; }
static inline vec3 operator/(vec3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
static inline vec3 operator/(vec3 a, vec3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
static inline vec3 operator+(vec3 a, float s) { return {a.x + s, a.y + s, a.z + s}; }
static inline vec3 operator-(float s, vec3 a) { return {s - a.x, s - a.y, s - a.z}; }
static inline vec3& operator+=(vec3& a, vec3 b) { a = a + b; return a; }
static inline vec3& operator*=(vec3& a, vec3 b) { a = a * b; return a; }
static inline vec3& operator*=(vec3& a, float s) { a = a * s; return a; }
static inline vec3& operator/=(vec3& a, float s) { a = a / s; return a; }

static inline float dot(vec3 a, vec3 b) { float tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z; return (tx + ty) + tz; }
static inline vec3  cross(vec3 a, vec3 b) {
    return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y};
}
static inline float length(vec3 a) { return sqrtf(dot(a, a)); }
static inline float length2(vec3 a) { return dot(a, a); }
static inline vec3  normalize(vec3 a) { float inv = 1.0f / sqrtf(dot(a, a)); return a * inv; }
static inline float distance(vec3 a, vec3 b) { return length(b - a); }
static inline vec3  reflect(vec3 I, vec3 N) { return I - N * dot(N, I) * 2.0f; }
static inline vec3  mix(vec3 a, vec3 b, float t) { return a * (1.0f - t) + b * t; }
static inline vec3  vmin(vec3 a, vec3 b) { return {gmin(a.x, b.x), gmin(a.y, b.y), gmin(a.z, b.z)}; }
static inline vec3  vmax(vec3 a, vec3 b) { return {gmax(a.x, b.x), gmax(a.y, b.y), gmax(a.z, b.z)}; }
static inline float comp(const vec3& a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
static inline float& comp(vec3& a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

static inline vec2 operator*(vec2 a, float s) { return {a.x * s, a.y * s}; }
static inline vec2 operator+(vec2 a, vec2 b) { return {a.x + b.x, a.y + b.y}; }
static inline vec2 operator+(vec2 a, float s) { return {a.x + s, a.y + s}; }
static inline vec2 operator*(vec2 a, vec2 b) { return {a.x * b.x, a.y * b.y}; }

static inline vec4 v4(vec3 a, float w) { return {a.x, a.y, a.z, w}; }
static inline vec4 operator+(vec4 a, vec4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
static inline vec4 operator*(vec4 a, float s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }
static inline vec4 operator/(vec4 a, float s) { return {a.x / s, a.y / s, a.z / s, a.w / s}; }
static inline vec4 operator/(vec4 a, vec4 b) { return {a.x / b.x, a.y / b.y, a.z / b.z, a.w / b.w}; }
static inline vec3 xyz(vec4 a) { return {a.x, a.y, a.z}; }

struct mat4 { vec4 c[4]; };   // column-major, c[j] is column j
static inline vec4 operator*(const mat4& m, vec4 v) {
    vec4 m0 = m.c[0] * v.x, m1 = m.c[1] * v.y, m2 = m.c[2] * v.z, m3 = m.c[3] * v.w;
    return (m0 + m1) + (m2 + m3);
}
static inline mat4 operator*(const mat4& a, const mat4& b) {
    // glm: column j of the product = a.c0*b[j].x + a.c1*b[j].y + a.c2*b[j].z + a.c3*b[j].w (left to right)
    mat4 r;
    for (int j = 0; j < 4; ++j)
        r.c[j] = ((a.c[0] * b.c[j].x + a.c[1] * b.c[j].y) + a.c[2] * b.c[j].z) + a.c[3] * b.c[j].w;
    return r;
}
static inline mat4 mat4_from(const float* p) { mat4 m; std::memcpy(&m, p, 64); return m; }

static inline bool finite4(vec4 a) { return std::isfinite(a.x) && std::isfinite(a.y) && std::isfinite(a.z) && std::isfinite(a.w); }

}  // namespace orc
