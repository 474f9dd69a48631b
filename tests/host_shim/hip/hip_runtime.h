// TEST INFRASTRUCTURE ONLY (tests/host_shim): a stand-in for <hip/hip_runtime.h> that lets g++ compile the *device headers* of
// vk_gltf_renderer_amd/csrc/device (pt_math.h, pt_bsdf.h, pt_light.h, pt_shading.h) for the host, so that the CPU-only test
// tier (-m "not gpu") can diff the device restatement of the BSDF / light / sky / material functions against the oracle before
// any GPU time is spent.  Nothing here is part of, linked into or shipped with libmi_pt.so: the product has no CPU path.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>

#define __device__
#define __host__
#define __global__
#define __forceinline__ inline __attribute__((always_inline))
#define __noinline__ __attribute__((noinline))
#define __restrict__ __restrict
#define __launch_bounds__(...)

// v_sin_f32 / v_cos_f32 take revolutions (pt_math.h: sinTurns / cosTurns)
#define __builtin_amdgcn_sinf(t) std::sin(6.283185307179586f * (t))
#define __builtin_amdgcn_cosf(t) std::cos(6.283185307179586f * (t))

using std::isfinite;
using std::max;
using std::min;

struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct uint4 { uint32_t x, y, z, w; };
struct uchar4 { unsigned char x, y, z, w; };
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline uint4  make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }

static inline float    __fmaf_rn(float a, float b, float c) { return std::fmaf(a, b, c); }
static inline uint32_t __float_as_uint(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static inline int      __float_as_int(float f) { int u; std::memcpy(&u, &f, 4); return u; }
static inline float    __uint_as_float(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
static inline float    __int_as_float(int u) { float f; std::memcpy(&f, &u, 4); return f; }
static inline uint32_t __umul24(uint32_t a, uint32_t b) { return (a & 0xffffffu) * (b & 0xffffffu); }
static inline int      __ffs(int v) { return __builtin_ffs(v); }
static inline int      __popc(uint32_t v) { return __builtin_popcount(v); }

// ---- for bvh_reinsert.h (tests/host_shim/reinsert_on_host.cpp): single roundings and the two atomics the phases use
static inline float __fadd_rn(float a, float b) { volatile float r = a + b; return r; }
static inline float __fmul_rn(float a, float b) { volatile float r = a * b; return r; }
static inline float __fsub_rn(float a, float b) { volatile float r = a - b; return r; }
static inline void  __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
static inline unsigned int atomicAdd(unsigned int* p, unsigned int v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
static inline unsigned long long atomicMax(unsigned long long* p, unsigned long long v)
{
  unsigned long long old = __atomic_load_n(p, __ATOMIC_SEQ_CST);
  while(old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
  return old;
}
