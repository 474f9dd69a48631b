// Tonemapper: HDR accumulator (or the denoised image) -> display-referred RGBA8, the `eImgTonemapped` image of the reference.
// Replaces GltfRenderer::tonemap -> nvshaders::Tonemapper::runCompute (reference: src/renderer.cpp:992-1056; method set
// [Filmic:0, Uncharted:1, Clip:2, ACES:3, AgX:4, KhronosPBR:5] and the tm* parameters: src/renderer.cpp:173-179; the default
// `autoExposure = 1`: src/resources.hpp:212).  The shader bodies live in the un-vendored nvpro_core2, so the
// operators are written from their PUBLISHED forms: Hejl / Burgess-Dawson filmic, Hable's Uncharted 2 curve (W = 11.2, exposure
// bias 2), S. Hill's ACES RRT+ODT fit, B. Wrensch's minimal AgX (6th-order sigmoid fit), Khronos PBR Neutral (2024 spec), IEC
// 61966-2-1 sRGB OETF.  tests/test_gpu_tonemap.py holds the same formulas in numpy.
//
// Memory-bound image pass: 16 B read + 4 B written per pixel; one thread per pixel, float4 loads, packed byte stores.
#include <hip/hip_runtime.h>

#include "pt_kernels.h"

namespace pt {

namespace {

constexpr int HIST_BINS = 256;

__device__ __forceinline__ float srgbOetf(float x)
{
  return x > 0.0031308f ? fmaf(powf(x, 1.0f / 2.4f), 1.055f, -0.055f) : x * 12.92f;
}

__device__ __forceinline__ float3 toSrgb(float3 c)
{
  return make_float3(srgbOetf(c.x), srgbOetf(c.y), srgbOetf(c.z));
}

__device__ __forceinline__ float filmic1(float x)
{
  const float t = fmaxf(0.0f, x - 0.004f);
  return (t * (6.2f * t + 0.5f)) / (t * (6.2f * t + 1.7f) + 0.06f);
}

__device__ __forceinline__ float hable1(float x)
{
  const float a = 0.15f, b = 0.50f, c = 0.10f, d = 0.20f, e = 0.02f, f = 0.30f;
  return ((x * (a * x + c * b) + d * e) / (x * (a * x + b) + d * f)) - e / f;
}

__device__ __forceinline__ float3 tonemapUncharted2(float3 c)
{
  const float white = 1.0f / hable1(11.2f);
  return toSrgb(make_float3(hable1(c.x * 2.0f) * white, hable1(c.y * 2.0f) * white, hable1(c.z * 2.0f) * white));
}

__device__ __forceinline__ float acesFit1(float v)
{
  return (v * (v + 0.0245786f) - 0.000090537f) / (v * (0.983729f * v + 0.4329510f) + 0.238081f);
}

__device__ __forceinline__ float3 tonemapAces(float3 c)
{
  float3 v = make_float3(0.59719f * c.x + 0.35458f * c.y + 0.04823f * c.z, 0.07600f * c.x + 0.90834f * c.y + 0.01566f * c.z,
                         0.02840f * c.x + 0.13383f * c.y + 0.83777f * c.z);
  v        = make_float3(acesFit1(v.x), acesFit1(v.y), acesFit1(v.z));
  return toSrgb(make_float3(1.60475f * v.x - 0.53108f * v.y - 0.07367f * v.z, -0.10208f * v.x + 1.10813f * v.y - 0.00605f * v.z,
                            -0.00327f * v.x - 0.07276f * v.y + 1.07602f * v.z));
}

__device__ __forceinline__ float agxCurve1(float x)
{
  const float minEv = -12.47393f, maxEv = 4.026069f;
  x              = (fminf(fmaxf(log2f(x), minEv), maxEv) - minEv) / (maxEv - minEv);
  const float x2 = x * x, x4 = x2 * x2;
  return 15.5f * x4 * x2 - 40.14f * x4 * x + 31.96f * x4 - 6.868f * x2 * x + 0.4298f * x2 + 0.1191f * x - 0.00232f;
}

__device__ __forceinline__ float3 tonemapAgx(float3 c)
{
  // inset matrix (columns as published), curve, outset matrix; the result is display encoded (no further OETF)
  float3 v = make_float3(0.842479062253094f * c.x + 0.0784335999999992f * c.y + 0.0792237451477643f * c.z,
                         0.0423282422610123f * c.x + 0.878468636469772f * c.y + 0.0791661274605434f * c.z,
                         0.0423756549057051f * c.x + 0.0784336f * c.y + 0.879142973793104f * c.z);
  v        = make_float3(agxCurve1(v.x), agxCurve1(v.y), agxCurve1(v.z));
  return make_float3(1.19687900512017f * v.x - 0.0980208811401368f * v.y - 0.0990297440797205f * v.z,
                     -0.0528968517574562f * v.x + 1.15190312990417f * v.y - 0.0989611768448433f * v.z,
                     -0.0529716355144438f * v.x - 0.0980434501171241f * v.y + 1.15107367264116f * v.z);
}

__device__ __forceinline__ float3 tonemapKhronosPbr(float3 c)
{
  const float startCompression = 0.8f - 0.04f, desaturation = 0.15f;
  const float x      = fminf(c.x, fminf(c.y, c.z));
  const float offset = x < 0.08f ? x - 6.25f * x * x : 0.04f;
  c                  = make_float3(c.x - offset, c.y - offset, c.z - offset);
  const float peak   = fmaxf(c.x, fmaxf(c.y, c.z));
  if(peak < startCompression)
    return toSrgb(c);
  const float d       = 1.0f - startCompression;
  const float newPeak = 1.0f - d * d / (peak + d - startCompression);
  const float s       = newPeak / peak;
  c                   = make_float3(c.x * s, c.y * s, c.z * s);
  const float g       = 1.0f - 1.0f / (desaturation * (peak - newPeak) + 1.0f);
  return toSrgb(make_float3(c.x + (newPeak - c.x) * g, c.y + (newPeak - c.y) * g, c.z + (newPeak - c.z) * g));
}

__device__ __forceinline__ float3 applyTonemap(const MiTonemapperData& tm, float exposure, float3 c, float u, float v)
{
  c = make_float3(c.x * exposure, c.y * exposure, c.z * exposure);
  float3 r;
  switch(tm.method)
  {
    default:
    case MI_TONEMAP_FILMIC: r = make_float3(filmic1(c.x), filmic1(c.y), filmic1(c.z)); break;
    case MI_TONEMAP_UNCHARTED: r = tonemapUncharted2(c); break;
    case MI_TONEMAP_CLIP: r = toSrgb(make_float3(fmaxf(c.x, 0.0f), fmaxf(c.y, 0.0f), fmaxf(c.z, 0.0f))); break;
    case MI_TONEMAP_ACES: r = tonemapAces(c); break;
    case MI_TONEMAP_AGX: r = tonemapAgx(c); break;
    case MI_TONEMAP_KHRONOS_PBR: r = tonemapKhronosPbr(c); break;
  }
  // contrast about mid grey, clamp, brightness as a display gamma, saturation about Rec.601 luma, radial vignette
  r = make_float3(fminf(fmaxf(0.5f + (r.x - 0.5f) * tm.contrast, 0.0f), 1.0f), fminf(fmaxf(0.5f + (r.y - 0.5f) * tm.contrast, 0.0f), 1.0f),
                  fminf(fmaxf(0.5f + (r.z - 0.5f) * tm.contrast, 0.0f), 1.0f));
  const float ib = 1.0f / tm.brightness;
  r              = make_float3(powf(r.x, ib), powf(r.y, ib), powf(r.z, ib));
  const float l  = 0.299f * r.x + 0.587f * r.y + 0.114f * r.z;
  r              = make_float3(l + (r.x - l) * tm.saturation, l + (r.y - l) * tm.saturation, l + (r.z - l) * tm.saturation);
  const float cu = (u - 0.5f) * 2.0f, cv = (v - 0.5f) * 2.0f;
  const float vg = 1.0f - (cu * cu + cv * cv) * tm.vignette;
  return make_float3(r.x * vg, r.y * vg, r.z * vg);
}

__device__ __forceinline__ uint32_t unorm8(float x)
{
  return uint32_t(fminf(fmaxf(x, 0.0f), 1.0f) * 255.0f + 0.5f);
}

// Auto exposure, pass 1: histogram of log2(luminance) over [evMin, evMax] (256 bins, centre-weighted when asked), privatised in
// LDS: one LDS atomic per pixel, 256 global atomics per block.
__global__ void __launch_bounds__(256) k_lum_histogram(const float4* __restrict__ in, int W, int H, MiTonemapperData tm, uint32_t* __restrict__ hist)
{
  __shared__ uint32_t bins[HIST_BINS];
  bins[threadIdx.x] = 0;
  __syncthreads();
  const size_t n = size_t(W) * size_t(H);
  for(size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += size_t(gridDim.x) * 256)
  {
    const float4 c   = in[i];
    const float  lum = 0.2126f * c.x + 0.7152f * c.y + 0.0722f * c.z;
    if(!(lum > 0.0f))
      continue;
    uint32_t weight = 1;
    if(tm.enableCenterMetering)
    {
      const int   x = int(i % size_t(W)), y = int(i / size_t(W));
      const float du = (float(x) + 0.5f) / float(W) - 0.5f, dv = (float(y) + 0.5f) / float(H) - 0.5f;
      weight = (du * du + dv * dv) < 0.0625f ? 4u : 1u;  // the central disc of radius 1/4 counts four times
    }
    const float t = (log2f(lum) - tm.evMinValue) / (tm.evMaxValue - tm.evMinValue);
    const int   b = min(HIST_BINS - 1, max(0, int(t * float(HIST_BINS))));
    atomicAdd(&bins[b], weight);
  }
  __syncthreads();
  if(bins[threadIdx.x])
    atomicAdd(&hist[threadIdx.x], bins[threadIdx.x]);
}

// Auto exposure, pass 2 (one wave): mean log2 luminance of the histogram without its darkest bin's clamp-ins (bin 0 collects
// everything below evMin), exposure = key 0.18 / 2^mean, eased towards the target with 1 - exp(-dt * speed).
// state[0] = current exposure multiplier (0 = not initialised), state[1] = target (diagnostic).
__global__ void __launch_bounds__(64) k_auto_exposure(const uint32_t* __restrict__ hist, MiTonemapperData tm, float dtSeconds, float* __restrict__ state)
{
  float sumW = 0.0f, sumEv = 0.0f;
  for(int b = threadIdx.x; b < HIST_BINS; b += 64)
  {
    const float w  = b == 0 ? 0.0f : float(hist[b]);
    const float ev = tm.evMinValue + (float(b) + 0.5f) / float(HIST_BINS) * (tm.evMaxValue - tm.evMinValue);
    sumW += w;
    sumEv += w * ev;
  }
  for(int o = 32; o > 0; o >>= 1)
  {
    sumW += __shfl_xor(sumW, o);
    sumEv += __shfl_xor(sumEv, o);
  }
  if(threadIdx.x == 0)
  {
    const float target = sumW > 0.0f ? 0.18f / exp2f(sumEv / sumW) : 1.0f;
    const float prev   = state[0];
    const float a      = (prev > 0.0f && dtSeconds >= 0.0f) ? 1.0f - expf(-dtSeconds * tm.autoExposureSpeed) : 1.0f;
    state[0]           = prev > 0.0f ? prev + (target - prev) * a : target;
    state[1]           = target;
  }
}

__global__ void __launch_bounds__(256) k_tonemap(const float4* __restrict__ in, uint32_t* __restrict__ out, int W, int H, MiTonemapperData tm,
                                                const float* __restrict__ autoState)
{
  const size_t n = size_t(W) * size_t(H);
  const size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
  if(i >= n)
    return;
  const float4 c = in[i];
  if(!tm.isActive)
  {
    out[i] = unorm8(c.x) | (unorm8(c.y) << 8) | (unorm8(c.z) << 16) | (unorm8(c.w) << 24);
    return;
  }
  const int    x = int(i % size_t(W)), y = int(i / size_t(W));
  const float  exposure = tm.exposure * ((tm.autoExposure && autoState) ? autoState[0] : 1.0f);
  const float3 r = applyTonemap(tm, exposure, make_float3(c.x, c.y, c.z), (float(x) + 0.5f) / float(W), (float(y) + 0.5f) / float(H));
  out[i]         = unorm8(r.x) | (unorm8(r.y) << 8) | (unorm8(r.z) << 16) | (unorm8(c.w) << 24);
}

}  // namespace

void launchTonemap(const float4* in, uint32_t* outRgba8, int width, int height, const MiTonemapperData& tm, uint32_t* histogram, float* autoState,
                   float dtSeconds, hipStream_t s)
{
  const size_t n = size_t(width) * size_t(height);
  if(tm.isActive && tm.autoExposure && histogram && autoState)
  {
    (void)hipMemsetAsync(histogram, 0, HIST_BINS * sizeof(uint32_t), s);
    const unsigned blocks = unsigned(std::min<size_t>((n + 255) / 256, 2048));
    hipLaunchKernelGGL(k_lum_histogram, dim3(blocks), dim3(256), 0, s, in, width, height, tm, histogram);
    hipLaunchKernelGGL(k_auto_exposure, dim3(1), dim3(64), 0, s, histogram, tm, dtSeconds, autoState);
  }
  hipLaunchKernelGGL(k_tonemap, dim3(unsigned((n + 255) / 256)), dim3(256), 0, s, in, outRgba8, width, height, tm,
                     (tm.autoExposure && autoState) ? autoState : nullptr);
}

}  // namespace pt
