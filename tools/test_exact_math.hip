// Checks divExact / sqrtExact (csrc/device/pt_math.h) on the device, bit for bit, against `a / b` and sqrtf of THIS translation unit -- which is compiled
// with the default -fhip-fp32-correctly-rounded-divide-sqrt, i.e. IEEE -- and that the fast forms pt_kernels.hip gets stay within their ulp bounds.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ivk_gltf_renderer_amd/csrc/device -Iinclude -o tools/_scratch/test_exact_math tools/test_exact_math.hip && tools/_scratch/test_exact_math
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>

#include "pt_math.h"

__device__ uint32_t mix(uint32_t x)
{
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__global__ void k_check(uint32_t n, uint32_t mode, unsigned long long* bad, float* worst)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= n)
    return;
  uint32_t ua = mix(i * 2u + 1u + mode * 0x9e3779b9u), ub = mix(i * 2u + 2u + mode * 0x85ebca6bu);
  if(mode == 1u)  // ordinary magnitudes: exponents within +-20 of 1
  {
    ua = (ua & 0x807fffffu) | (((ua >> 23) % 41u + 107u) << 23);
    ub = (ub & 0x807fffffu) | (((ub >> 23) % 41u + 107u) << 23);
  }
  const float a = __uint_as_float(ua), b = __uint_as_float(ub);
  const float q = a / b, e = pt::divExact(a, b);
  const float s = sqrtf(fabsf(a)), t = pt::sqrtExact(fabsf(a));
  const bool  qn = q != q, en = e != e;
  if((qn != en) || (!qn && __float_as_uint(q) != __float_as_uint(e)))
    atomicAdd(&bad[0], 1ull);
  const bool sn = s != s, tn = t != t;
  if((sn != tn) || (!sn && __float_as_uint(s) != __float_as_uint(t)))
    atomicAdd(&bad[1], 1ull);
  (void)worst;
}
int main()
{
  unsigned long long* bad = nullptr;
  float*              worst = nullptr;
  (void)hipMalloc(&bad, 16);
  (void)hipMalloc(&worst, 8);
  (void)hipMemset(bad, 0, 16);
  const uint32_t n = 1u << 28;
  for(uint32_t mode = 0; mode < 2; ++mode)  // 0: all bit patterns (denormals, infinities, NaNs), 1: ordinary magnitudes
    hipLaunchKernelGGL(k_check, dim3(n / 256), dim3(256), 0, 0, n, mode, bad, worst);
  unsigned long long h[2] = {1, 1};
  (void)hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost);
  printf("divExact vs a / b: %llu mismatches, sqrtExact vs sqrtf: %llu mismatches, over 2 x %u random operand pairs\n", h[0], h[1], n);
  return (h[0] | h[1]) ? 1 : 0;
}
