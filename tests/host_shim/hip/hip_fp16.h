// TEST INFRASTRUCTURE ONLY: binary16 round trip for the host compile of the device headers (see hip_runtime.h next to this file).
#pragma once
#include <cstdint>
#include <cstring>
struct __half { uint16_t bits; };
static inline __half __float2half_rn(float f)
{
  uint32_t x; std::memcpy(&x, &f, 4);
  const uint32_t sign = (x >> 16) & 0x8000u;
  x &= 0x7fffffffu;
  uint16_t h;
  if(x >= 0x7f800000u) h = uint16_t(0x7c00u | ((x > 0x7f800000u) ? 0x200u : 0u));           // inf / nan
  else if(x >= 0x477ff000u) h = 0x7c00u;                                                      // rounds to inf
  else if(x >= 0x38800000u)                                                                   // normal
  {
    uint32_t m = x - 0x38000000u;  // rebias
    uint32_t r = m + 0xfffu + ((m >> 13) & 1u);
    h = uint16_t(r >> 13);
  }
  else if(x >= 0x33000000u)                                                                   // subnormal
  {
    const int e = int(x >> 23);
    uint32_t  m = (x & 0x7fffffu) | 0x800000u;
    const int shift = 126 - e;  // 14..24
    uint32_t  r = m >> shift;
    const uint32_t rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
    if(rem > half || (rem == half && (r & 1u))) ++r;
    h = uint16_t(r);
  }
  else h = 0;
  return __half{uint16_t(h | sign)};
}
static inline float __half2float(__half hh)
{
  const uint16_t h = hh.bits;
  const uint32_t sign = uint32_t(h & 0x8000u) << 16;
  uint32_t e = (h >> 10) & 0x1fu, m = h & 0x3ffu, x;
  if(e == 0)
  {
    if(m == 0) x = sign;
    else
    {
      e = 1;
      while(!(m & 0x400u)) { m <<= 1; --e; }
      x = sign | ((e + 112u) << 23) | ((m & 0x3ffu) << 13);
    }
  }
  else if(e == 31) x = sign | 0x7f800000u | (m << 13);
  else x = sign | ((e + 112u) << 23) | (m << 13);
  float f; std::memcpy(&f, &x, 4);
  return f;
}
