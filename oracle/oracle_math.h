/*
 * oracle_math.h — tiny float3/float4 algebra for the CPU oracle.  TEST INFRASTRUCTURE ONLY (see oracle_pt.cpp).
 */
#ifndef ORACLE_MATH_H
#define ORACLE_MATH_H

#include <cmath>
#include <cstdint>
#include <cstring>

namespace orc {

struct float2
{
  float x = 0, y = 0;
  float2() = default;
  float2(float a, float b) : x(a), y(b) {}
  explicit float2(float a) : x(a), y(a) {}
};
struct float3
{
  float x = 0, y = 0, z = 0;
  float3() = default;
  float3(float a, float b, float c) : x(a), y(b), z(c) {}
  explicit float3(float a) : x(a), y(a), z(a) {}
  explicit float3(const float* p) : x(p[0]), y(p[1]), z(p[2]) {}
  float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
struct float4
{
  float x = 0, y = 0, z = 0, w = 0;
  float4() = default;
  float4(float a, float b, float c, float d) : x(a), y(b), z(c), w(d) {}
  float4(float3 v, float d) : x(v.x), y(v.y), z(v.z), w(d) {}
  explicit float4(float a) : x(a), y(a), z(a), w(a) {}
  explicit float4(const float* p) : x(p[0]), y(p[1]), z(p[2]), w(p[3]) {}
  float3 xyz() const { return float3(x, y, z); }
};

inline float2 operator+(float2 a, float2 b) { return {a.x + b.x, a.y + b.y}; }
inline float2 operator-(float2 a, float2 b) { return {a.x - b.x, a.y - b.y}; }
inline float2 operator*(float2 a, float s) { return {a.x * s, a.y * s}; }
inline float2 operator*(float2 a, float2 b) { return {a.x * b.x, a.y * b.y}; }
inline float2 operator/(float2 a, float2 b) { return {a.x / b.x, a.y / b.y}; }

inline float3 operator+(float3 a, float3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline float3 operator-(float3 a, float3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline float3 operator-(float3 a) { return {-a.x, -a.y, -a.z}; }
inline float3 operator*(float3 a, float3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline float3 operator*(float3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline float3 operator*(float s, float3 a) { return {a.x * s, a.y * s, a.z * s}; }
inline float3 operator/(float3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline float3 operator/(float3 a, float3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
inline float3& operator+=(float3& a, float3 b) { a = a + b; return a; }
inline float3& operator-=(float3& a, float3 b) { a = a - b; return a; }
inline float3& operator*=(float3& a, float3 b) { a = a * b; return a; }
inline float3& operator*=(float3& a, float s) { a = a * s; return a; }
inline float3& operator/=(float3& a, float s) { a = a / s; return a; }

inline float4 operator+(float4 a, float4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
inline float4 operator*(float4 a, float4 b) { return {a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }
inline float4 operator*(float4 a, float s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }
inline float4 operator/(float4 a, float s) { return {a.x / s, a.y / s, a.z / s, a.w / s}; }
inline float4& operator+=(float4& a, float4 b) { a = a + b; return a; }
inline float4& operator*=(float4& a, float4 b) { a = a * b; return a; }
inline float4& operator*=(float4& a, float s) { a = a * s; return a; }

inline float dot(float2 a, float2 b) { return a.x * b.x + a.y * b.y; }
inline float dot(float3 a, float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float3 cross(float3 a, float3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float length(float3 a) { return std::sqrt(dot(a, a)); }
inline float length(float2 a) { return std::sqrt(dot(a, a)); }
// normalize multiplies by ONE reciprocal (what a shader compiler emits for it; the device code does the same, pt_math.h, so that both
// sides round alike)
inline float3 normalize(float3 a)
{
  const float r = 1.0f / length(a);
  return {a.x * r, a.y * r, a.z * r};
}
inline float2 normalize(float2 a) { float l = length(a); return {a.x / l, a.y / l}; }
inline float3 reflect(float3 i, float3 n) { return i - n * (2.0f * dot(n, i)); }
inline float clampf(float v, float lo, float hi) { return std::fmin(std::fmax(v, lo), hi); }
inline float saturate(float v) { return clampf(v, 0.0f, 1.0f); }
inline float lerp(float a, float b, float t) { return a + (b - a) * t; }
inline float3 lerp(float3 a, float3 b, float t) { return a + (b - a) * t; }
inline float3 lerp(float3 a, float3 b, float3 t) { return a + (b - a) * t; }
inline float3 max3(float3 a, float3 b) { return {std::fmax(a.x, b.x), std::fmax(a.y, b.y), std::fmax(a.z, b.z)}; }
inline float3 min3(float3 a, float3 b) { return {std::fmin(a.x, b.x), std::fmin(a.y, b.y), std::fmin(a.z, b.z)}; }
inline float maxComp(float3 a) { return std::fmax(a.x, std::fmax(a.y, a.z)); }
inline float3 exp3(float3 a) { return {std::exp(a.x), std::exp(a.y), std::exp(a.z)}; }
inline float3 log3(float3 a) { return {std::log(a.x), std::log(a.y), std::log(a.z)}; }
inline float3 sqrt3(float3 a) { return {std::sqrt(a.x), std::sqrt(a.y), std::sqrt(a.z)}; }
inline float3 clamp3(float3 a, float lo, float hi) { return {clampf(a.x, lo, hi), clampf(a.y, lo, hi), clampf(a.z, lo, hi)}; }
inline float smoothstep(float e0, float e1, float x)
{
  float t = saturate((x - e0) / (e1 - e0));
  return t * t * (3.0f - 2.0f * t);
}
inline float signf(float v) { return v > 0.0f ? 1.0f : (v < 0.0f ? -1.0f : 0.0f); }
inline float sqr(float v) { return v * v; }

inline uint32_t asuint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
inline int32_t  asint(float f) { int32_t u; memcpy(&u, &f, 4); return u; }
inline float    asfloat(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
inline float    asfloat(int32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* Column-major 4x4 (glm memory): the Slang code of the reference sees the transpose, so
 *   Slang mul(v, M)  == M * v          -> mulPoint / mulVector
 *   Slang mul(M, v)  == transpose(M)*v -> mulTransposed
 *   Slang M[i]       == column i        */
struct mat4
{
  float m[16];
};
inline mat4 loadMat(const float* p) { mat4 r; memcpy(r.m, p, 64); return r; }
/* fmaf chains with a fixed order: the device code uses the same order, so world-space vertices are bit-identical. */
inline float3 mulPoint(const mat4& M, float3 p)
{
  return {std::fmaf(M.m[8], p.z, std::fmaf(M.m[4], p.y, std::fmaf(M.m[0], p.x, M.m[12]))),
          std::fmaf(M.m[9], p.z, std::fmaf(M.m[5], p.y, std::fmaf(M.m[1], p.x, M.m[13]))),
          std::fmaf(M.m[10], p.z, std::fmaf(M.m[6], p.y, std::fmaf(M.m[2], p.x, M.m[14])))};
}
inline float3 mulVector(const mat4& M, float3 v)
{
  return {std::fmaf(M.m[8], v.z, std::fmaf(M.m[4], v.y, M.m[0] * v.x)), std::fmaf(M.m[9], v.z, std::fmaf(M.m[5], v.y, M.m[1] * v.x)),
          std::fmaf(M.m[10], v.z, std::fmaf(M.m[6], v.y, M.m[2] * v.x))};
}
inline float3 mulTransposed(const mat4& M, float3 v) /* upper 3x3 of M^T times v */
{
  return {std::fmaf(M.m[2], v.z, std::fmaf(M.m[1], v.y, M.m[0] * v.x)), std::fmaf(M.m[6], v.z, std::fmaf(M.m[5], v.y, M.m[4] * v.x)),
          std::fmaf(M.m[10], v.z, std::fmaf(M.m[9], v.y, M.m[8] * v.x))};
}
inline float4 mulFull(const mat4& M, float4 v) /* M * v */
{
  float4 r;
  r.x = M.m[0] * v.x + M.m[4] * v.y + M.m[8] * v.z + M.m[12] * v.w;
  r.y = M.m[1] * v.x + M.m[5] * v.y + M.m[9] * v.z + M.m[13] * v.w;
  r.z = M.m[2] * v.x + M.m[6] * v.y + M.m[10] * v.z + M.m[14] * v.w;
  r.w = M.m[3] * v.x + M.m[7] * v.y + M.m[11] * v.z + M.m[15] * v.w;
  return r;
}
inline float det3(const mat4& M)
{
  return M.m[0] * (M.m[5] * M.m[10] - M.m[9] * M.m[6]) - M.m[4] * (M.m[1] * M.m[10] - M.m[9] * M.m[2])
         + M.m[8] * (M.m[1] * M.m[6] - M.m[5] * M.m[2]);
}

}  // namespace orc
#endif
