// Device-side vector algebra, RNG and small numeric helpers for the wavefront path tracer (gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#define PT_DEV __device__ __forceinline__

namespace pt {

struct f2
{
  float x, y;
};
struct f3
{
  float x, y, z;
};
struct f4
{
  float x, y, z, w;
};

PT_DEV f2 mk2(float x, float y) { return f2{x, y}; }
PT_DEV f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
PT_DEV f3 mk3(float s) { return f3{s, s, s}; }
PT_DEV f3 mk3(const float* p) { return f3{p[0], p[1], p[2]}; }
PT_DEV f4 mk4(float x, float y, float z, float w) { return f4{x, y, z, w}; }
PT_DEV f4 mk4(f3 v, float w) { return f4{v.x, v.y, v.z, w}; }
PT_DEV f4 mk4(float s) { return f4{s, s, s, s}; }
PT_DEV f4 mk4(float4 v) { return f4{v.x, v.y, v.z, v.w}; }
PT_DEV f3 xyz(f4 v) { return f3{v.x, v.y, v.z}; }
PT_DEV f3 xyz(float4 v) { return f3{v.x, v.y, v.z}; }

PT_DEV f2 operator+(f2 a, f2 b) { return {a.x + b.x, a.y + b.y}; }
PT_DEV f2 operator-(f2 a, f2 b) { return {a.x - b.x, a.y - b.y}; }
PT_DEV f2 operator*(f2 a, float s) { return {a.x * s, a.y * s}; }

PT_DEV f3 operator+(f3 a, f3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
PT_DEV f3 operator-(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
PT_DEV f3 operator-(f3 a) { return {-a.x, -a.y, -a.z}; }
PT_DEV f3 operator*(f3 a, f3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
PT_DEV f3 operator*(f3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
PT_DEV f3 operator*(float s, f3 a) { return {a.x * s, a.y * s, a.z * s}; }
// (pt_kernels.hip defines PT_FAST_SHADING_MATH: a vector over a scalar is ONE hardware reciprocal (v_rcp_f32, 1 ulp) and three multiplies there -- within the
//  2.5 ulp the translation unit's divisions have anyway, see divExact below -- instead of three 8-instruction divisions; everything that must stay IEEE says so
//  explicitly with divExact / normalizeExact.  The builders, the denoiser and the host compile of these headers keep the plain form.)
#if defined(__HIP_DEVICE_COMPILE__) && defined(PT_FAST_SHADING_MATH)
PT_DEV f3 operator/(f3 a, float s)
{
  const float r = __builtin_amdgcn_rcpf(s);
  return {a.x * r, a.y * r, a.z * r};
}
#else
PT_DEV f3 operator/(f3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
#endif
PT_DEV f3 operator/(f3 a, f3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
PT_DEV f3& operator+=(f3& a, f3 b) { a = a + b; return a; }
PT_DEV f3& operator-=(f3& a, f3 b) { a = a - b; return a; }
PT_DEV f3& operator*=(f3& a, f3 b) { a = a * b; return a; }
PT_DEV f3& operator*=(f3& a, float s) { a = a * s; return a; }
PT_DEV f3& operator/=(f3& a, float s) { a = a / s; return a; }

PT_DEV f4 operator+(f4 a, f4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
PT_DEV f4 operator*(f4 a, f4 b) { return {a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }
PT_DEV f4 operator*(f4 a, float s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }
PT_DEV f4 operator/(f4 a, float s) { return {a.x / s, a.y / s, a.z / s, a.w / s}; }
PT_DEV f4& operator+=(f4& a, f4 b) { a = a + b; return a; }
PT_DEV f4& operator*=(f4& a, f4 b) { a = a * b; return a; }
PT_DEV f4& operator*=(f4& a, float s) { a = a * s; return a; }

// ---- correctly rounded division / square root, whatever the translation unit's floating-point options ----------------------------------
// pt_kernels.hip is compiled with -fno-hip-fp32-correctly-rounded-divide-sqrt (csrc/Makefile): `a / b` there is the 2.5-ulp division and
// sqrtf the 1-ulp square root the REFERENCE's own arithmetic has (Vulkan's SPIR-V precision: OpFDiv 2.5 ULP, sqrt / inversesqrt 2 ULP) --
// a fifth of the shade kernels' vector instructions were the IEEE expansions (profiles/r06_fast_div_ab.txt: atrium +3.3 %, helmet +4 %).
// What must agree with the CPU oracle BIT FOR BIT keeps IEEE results through these helpers: camera rays (coverage, selection ids, depth),
// the ray setup and the ray / triangle test (hit records, tie-breaks on coincident geometry), the running mean of k_finish_sample.
// divExact is the AMDGPU back end's own IEEE expansion (v_div_scale / v_rcp / Newton-Raphson in fma / v_div_fmas / v_div_fixup) written with
// its builtins; sqrtExact is OCML's correctly rounded sqrt.  tools/test_exact_math.hip checks both against `/` and sqrtf of a translation
// unit compiled WITHOUT the option, bit for bit, on the device.
#if defined(__HIP_DEVICE_COMPILE__)
PT_DEV float divExact(float a, float b)
{
  bool        vcc  = false, unused = false;
  const float den  = __builtin_amdgcn_div_scalef(a, b, false, &unused);
  const float num  = __builtin_amdgcn_div_scalef(a, b, true, &vcc);
  const float rcp  = __builtin_amdgcn_rcpf(den);
  const float e0   = __builtin_fmaf(-den, rcp, 1.0f);
  const float r1   = __builtin_fmaf(e0, rcp, rcp);
  const float q0   = num * r1;
  const float e1   = __builtin_fmaf(-den, q0, num);
  const float q1   = __builtin_fmaf(e1, r1, q0);
  const float e2   = __builtin_fmaf(-den, q1, num);
  const float fmas = __builtin_amdgcn_div_fmasf(e2, r1, q1, vcc);
  return __builtin_amdgcn_div_fixupf(fmas, b, a);
}
PT_DEV float sqrtExact(float x) { return __ocml_sqrt_f32(x); }
// libm-grade logarithm / sine / cosine whatever the options (camera jitter and lens: the translation unit may be compiled with -fapprox-func, which turns
// logf / sinf / cosf CALLS into the hardware's v_log / v_sin / v_cos approximations; OCML's own entry points are not calls the option rewrites)
PT_DEV float logExact(float x) { return __ocml_log_f32(x); }
PT_DEV float sinExact(float x) { return __ocml_sin_f32(x); }
PT_DEV float cosExact(float x) { return __ocml_cos_f32(x); }
#else  // hipcc's host pass (never runs) and the device headers compiled for the host (tests/host_shim): IEEE by the language
PT_DEV float divExact(float a, float b) { return a / b; }
PT_DEV float sqrtExact(float x) { return sqrtf(x); }
PT_DEV float logExact(float x) { return logf(x); }
PT_DEV float sinExact(float x) { return sinf(x); }
PT_DEV float cosExact(float x) { return cosf(x); }
#endif

PT_DEV float dot(f2 a, f2 b) { return a.x * b.x + a.y * b.y; }
PT_DEV float dot(f3 a, f3 b);
PT_DEV float normalizeScale(f3 a) { return divExact(1.0f, sqrtExact(dot(a, a))); }
PT_DEV float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
PT_DEV f3 cross(f3 a, f3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
PT_DEV float length(f3 a) { return sqrtf(dot(a, a)); }
PT_DEV float length(f2 a) { return sqrtf(dot(a, a)); }
// normalize: ONE correctly rounded reciprocal and three multiplies instead of three divisions (a division is ten vector
// instructions); the oracle rounds the same way (oracle_math.h).  The products must stay products: a division cannot be contracted
// into a following add, these multiplies could (hipcc: -ffp-contract=fast-honor-pragmas), and the oracle, built with contraction
// off, would then differ by an ulp wherever a unit vector is added to something -- hence the pragma.
PT_DEV f3 normalize(f3 a)
{
#pragma clang fp contract(off)
  // DIRECTIONS stay IEEE in every translation unit (normalizeScale: divExact / sqrtExact above), bit for bit the oracle's normalize: a unit vector decides where the
  // next ray goes, and a 1-2 ulp difference there sends three times as many paths another way as the FMA contractions do -- the full-size parity legs measured
  // street 3.6e-4 -> 7.5e-4 and sliver atrium 0.93e-3 -> 1.21e-3 with v_rsq_f32 here, and exactly the IEEE build's figures without it, while every other fast
  // operation of pt_kernels.hip (weights, pdfs, Fresnel, sky: radiometric) changed no digit (profiles/r06_parity_by_arithmetic.txt).  Cost: 1.4 % of the atrium.
  // -DPT_NORMALIZE_FAST (A/B) = the hardware reciprocal square root.
#if defined(__HIP_DEVICE_COMPILE__) && defined(PT_FAST_SHADING_MATH) && defined(PT_NORMALIZE_FAST)
  const float r = __builtin_amdgcn_rsqf(dot(a, a));
#elif defined(__HIP_DEVICE_COMPILE__) && defined(PT_FAST_SHADING_MATH)
  const float r = normalizeScale(a);
#else
  const float r = 1.0f / length(a);
#endif
  return {a.x * r, a.y * r, a.z * r};
}
PT_DEV f2 normalize(f2 a) { float l = length(a); return {a.x / l, a.y / l}; }
// ... with IEEE square root and reciprocal whatever the compile options (camera rays: divExact above)
PT_DEV f3 normalizeExact(f3 a)
{
#pragma clang fp contract(off)
  const float r = divExact(1.0f, sqrtExact(dot(a, a)));
  return {a.x * r, a.y * r, a.z * r};
}
// sin / cos of an angle given in REVOLUTIONS (angle / 2 pi), |t| <= 256: v_sin_f32 / v_cos_f32 take their argument that way, so
// phi = 2 pi u needs neither the multiply nor a range reduction.  Max abs error 1.3e-7 on [0, 1) against double precision.
PT_DEV float sinTurns(float t) { return __builtin_amdgcn_sinf(t); }
PT_DEV float cosTurns(float t) { return __builtin_amdgcn_cosf(t); }
PT_DEV f3 reflect(f3 i, f3 n) { return i - n * (2.0f * dot(n, i)); }
PT_DEV float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }
PT_DEV float saturatef(float v) { return clampf(v, 0.0f, 1.0f); }
PT_DEV float lerpf(float a, float b, float t) { return a + (b - a) * t; }
PT_DEV f3 lerp3(f3 a, f3 b, float t) { return a + (b - a) * t; }
PT_DEV f3 lerp3(f3 a, f3 b, f3 t) { return a + (b - a) * t; }
PT_DEV f3 max3(f3 a, f3 b) { return {fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z)}; }
PT_DEV f3 min3(f3 a, f3 b) { return {fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z)}; }
PT_DEV float maxComp(f3 a) { return fmaxf(a.x, fmaxf(a.y, a.z)); }
PT_DEV f3 exp3(f3 a) { return {expf(a.x), expf(a.y), expf(a.z)}; }
PT_DEV f3 log3(f3 a) { return {logf(a.x), logf(a.y), logf(a.z)}; }
PT_DEV f3 sqrt3(f3 a) { return {sqrtf(a.x), sqrtf(a.y), sqrtf(a.z)}; }
PT_DEV f3 clamp3(f3 a, float lo, float hi) { return {clampf(a.x, lo, hi), clampf(a.y, lo, hi), clampf(a.z, lo, hi)}; }
PT_DEV float smoothstepf(float e0, float e1, float x)
{
  float t = saturatef((x - e0) / (e1 - e0));
  return t * t * (3.0f - 2.0f * t);
}
PT_DEV float signfz(float v) { return v > 0.0f ? 1.0f : (v < 0.0f ? -1.0f : 0.0f); }
PT_DEV float sqr(float v) { return v * v; }
PT_DEV bool  isFinite3(f3 v) { return isfinite(v.x) && isfinite(v.y) && isfinite(v.z); }

// Column-major 4x4 as 16 floats (glm memory). Slang mul(v,M) == M*v ; Slang mul(M,v) == M^T*v ; Slang M[i] == column i.
// The fmaf orders below are part of the parity contract with the oracle (bit-identical world-space vertices).
PT_DEV f3 mulPoint(const float* M, f3 p)
{
  return {__fmaf_rn(M[8], p.z, __fmaf_rn(M[4], p.y, __fmaf_rn(M[0], p.x, M[12]))),
          __fmaf_rn(M[9], p.z, __fmaf_rn(M[5], p.y, __fmaf_rn(M[1], p.x, M[13]))),
          __fmaf_rn(M[10], p.z, __fmaf_rn(M[6], p.y, __fmaf_rn(M[2], p.x, M[14])))};
}
PT_DEV f3 mulVector(const float* M, f3 v)
{
  return {__fmaf_rn(M[8], v.z, __fmaf_rn(M[4], v.y, M[0] * v.x)), __fmaf_rn(M[9], v.z, __fmaf_rn(M[5], v.y, M[1] * v.x)),
          __fmaf_rn(M[10], v.z, __fmaf_rn(M[6], v.y, M[2] * v.x))};
}
PT_DEV f3 mulTransposed(const float* M, f3 v)
{
  return {__fmaf_rn(M[2], v.z, __fmaf_rn(M[1], v.y, M[0] * v.x)), __fmaf_rn(M[6], v.z, __fmaf_rn(M[5], v.y, M[4] * v.x)),
          __fmaf_rn(M[10], v.z, __fmaf_rn(M[9], v.y, M[8] * v.x))};
}
PT_DEV f4 mulFull(const float* M, f4 v)
{
  f4 r;
  r.x = M[0] * v.x + M[4] * v.y + M[8] * v.z + M[12] * v.w;
  r.y = M[1] * v.x + M[5] * v.y + M[9] * v.z + M[13] * v.w;
  r.z = M[2] * v.x + M[6] * v.y + M[10] * v.z + M[14] * v.w;
  r.w = M[3] * v.x + M[7] * v.y + M[11] * v.z + M[15] * v.w;
  return r;
}
PT_DEV float dotFma(f3 a, f3 b) { return __fmaf_rn(a.z, b.z, __fmaf_rn(a.y, b.y, a.x * b.x)); }
PT_DEV f3    crossFma(f3 a, f3 b)
{
  return {__fmaf_rn(a.y, b.z, -(a.z * b.y)), __fmaf_rn(a.z, b.x, -(a.x * b.z)), __fmaf_rn(a.x, b.y, -(a.y * b.x))};
}

// ---- RNG (nvshaders/random.h.slang is external to the reference tree; restated from xxHash32 / PCG-RXS-M-XS) ----------
PT_DEV uint32_t xxhash32(uint32_t px, uint32_t py, uint32_t pz)
{
  const uint32_t P2 = 2246822519U, P3 = 3266489917U, P4 = 668265263U, P5 = 374761393U;
  uint32_t       h  = pz + P5 + px * P3;
  h                 = P4 * ((h << 17) | (h >> 15));
  h += py * P3;
  h = P4 * ((h << 17) | (h >> 15));
  h = P2 * (h ^ (h >> 15));
  h = P3 * (h ^ (h >> 13));
  return h ^ (h >> 16);
}
PT_DEV uint32_t pcg(uint32_t& state)
{
  uint32_t prev = state * 747796405u + 2891336453u;
  uint32_t word = ((prev >> ((prev >> 28u) + 4u)) ^ prev) * 277803737u;
  state         = prev;
  return (word >> 22u) ^ word;
}
PT_DEV float u32ToUnitFloat(uint32_t r) { return __uint_as_float(0x3f800000u | (r >> 9)) - 1.0f; }
PT_DEV float rnd(uint32_t& seed) { return u32ToUnitFloat(pcg(seed)); }
// Order-independent stochastic-alpha draw for one (ray, triangle) pair — see DESIGN.md "determinism".
PT_DEV float candidateRand(uint32_t seed, int rnode, int prim) { return u32ToUnitFloat(xxhash32(seed, uint32_t(rnode), uint32_t(prim))); }

// binary16 round trip (VolumeMedium is stored as float16_t in the reference: pathtrace_functions.h.slang:118-123)
PT_DEV float roundToHalf(float f) { return __half2float(__float2half_rn(f)); }

PT_DEV bool splitRandom(float& u, float p)
{
  if(u < p)
  {
    u = p > 0.0f ? u / p : 0.0f;
    return true;
  }
  u = (1.0f - p) > 0.0f ? (u - p) / (1.0f - p) : 0.0f;
  return false;
}

constexpr float INFINITE_F  = 1e32f;
constexpr float DIRAC       = -1.0f;
constexpr float K_PI        = 3.14159265358979323846f;
constexpr float K_TWO_PI    = 6.28318530717958647692f;
constexpr float K_1_OVER_PI = 0.31830988618379067154f;

}  // namespace pt
