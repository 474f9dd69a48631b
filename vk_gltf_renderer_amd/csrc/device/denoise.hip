// Edge-avoiding a-trous wavelet denoiser (Dammertz et al. 2010, "Edge-Avoiding A-Trous Wavelet Transform for fast Global
// Illumination Filtering"), guided by first-hit albedo and shading normal.  It replaces the I/O contract of the
// reference's OptiX adapter (src/optix_denoiser.hpp:128-153: noisy RGBA32F + albedo + normal -> denoised RGBA32F);
// the guides are captured where the reference captures them (shaders/gltf_pathtrace.slang:228-264).
#include <hip/hip_runtime.h>

#include "pt_kernels.h"

namespace pt {

namespace {

__global__ void __launch_bounds__(256) k_atrous(const float4* __restrict__ in, float4* __restrict__ out, const float4* __restrict__ albedo,
                                               const float4* __restrict__ normal, int W, int H, int step, float sigmaColor, float sigmaNormal,
                                               float sigmaAlbedo)
{
  const int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
  if(x >= W || y >= H)
    return;
  const float  kern[3] = {3.0f / 8.0f, 1.0f / 4.0f, 1.0f / 16.0f};  // B3 spline
  const size_t c       = size_t(y) * W + x;
  const float4 cc = in[c], ca = albedo[c], cn = normal[c];
  float        sr = 0, sg = 0, sb = 0, sw = 0;
  const float  invC = 1.0f / fmaxf(sigmaColor * sigmaColor, 1e-8f), invA = 1.0f / fmaxf(sigmaAlbedo * sigmaAlbedo, 1e-8f);
  for(int dy = -2; dy <= 2; ++dy)
    for(int dx = -2; dx <= 2; ++dx)
    {
      const int qx = x + dx * step, qy = y + dy * step;
      if(qx < 0 || qy < 0 || qx >= W || qy >= H)
        continue;
      const size_t q  = size_t(qy) * W + qx;
      const float4 qc = in[q], qa = albedo[q], qn = normal[q];
      // only filter geometry with geometry and background with background (alpha = solid-hit fraction)
      if((qa.w > 0.5f) != (ca.w > 0.5f))
        continue;
      float dcx = qc.x - cc.x, dcy = qc.y - cc.y, dcz = qc.z - cc.z;
      float dax = qa.x - ca.x, day = qa.y - ca.y, daz = qa.z - ca.z;
      float wC  = __expf(-(dcx * dcx + dcy * dcy + dcz * dcz) * invC);
      float wA  = __expf(-(dax * dax + day * day + daz * daz) * invA);
      float nd  = fmaxf(0.0f, qn.x * cn.x + qn.y * cn.y + qn.z * cn.z);
      float wN  = (ca.w > 0.5f) ? __powf(nd, sigmaNormal) : 1.0f;
      float w   = kern[abs(dx)] * kern[abs(dy)] * wC * wA * wN;
      sr += qc.x * w;
      sg += qc.y * w;
      sb += qc.z * w;
      sw += w;
    }
  out[c] = sw > 0.0f ? make_float4(sr / sw, sg / sw, sb / sw, cc.w) : cc;
}


//--------------------------------------------------------------------------------------------------------------------------------
// Variance-guided variant (Schied et al. 2017, "Spatiotemporal Variance-Guided Filtering", without reprojection: the camera of a
// progressive accumulation stands still, so the temporal half of SVGF IS the running mean of gltf_pathtrace.slang:626-629 plus its
// second moment).  Inputs: the accumulator (alpha = solid-hit fraction), the first-hit albedo / normal guides, the second moment of
// the per-frame pixel luminance (normal.w, k_finish_sample), the NDC depth of frame 0 and the number of accumulated frames.
//   1. prepare: demodulate by the albedo guide; variance of the mean = (E[l^2] - E[l]^2) / frames, carried over to the demodulated
//      signal; with fewer than 4 frames the temporal estimate is replaced by a 7x7 spatial one (paper §4.2).
//   2. a-trous, 5x5 B3 spline, step 2^i: weights = normal^sigmaN x exp(-|dz| / (sigmaZ |grad z| |dp|)) x exp(-|dl| / (sigmaL
//      sqrt(gauss3x3(var)))) (paper eq. 3-5); colour filtered with w, variance with w^2 (eq. 2).  Geometry never mixes with
//      background (albedo.w = hit fraction), background is filtered on luminance alone.
//   3. finish: re-modulate, alpha passes through.
// Image-space passes: 25 taps x (16 + 16 + 4 B) per pixel and iteration out of L2; the HBM side is 2 x 16 B per pixel and pass.
//--------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float lum709(float x, float y, float z)
{
  return 0.2126f * x + 0.7152f * y + 0.0722f * z;
}

__device__ __forceinline__ float3 demodulator(const float4 a)
{
  return a.w > 0.5f ? make_float3(fmaxf(a.x, 0.02f), fmaxf(a.y, 0.02f), fmaxf(a.z, 0.02f)) : make_float3(1.0f, 1.0f, 1.0f);
}

// view-depth-like quantity from the stored NDC depth: proportional to the distance for a perspective projection with a far plane
// much further than the scene (1 - z ~ near / distance), monotonic for any other; only ratios of its differences are used
__device__ __forceinline__ float depthKey(float ndc)
{
  return 1.0f / fmaxf(1.0f - ndc, 1e-7f);
}

__global__ void __launch_bounds__(256) k_svgf_prepare(const float4* __restrict__ color, const float4* __restrict__ albedo, const float4* __restrict__ normal,
                                                     float4* __restrict__ illum, int W, int H, float frames)
{
  const int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
  if(x >= W || y >= H)
    return;
  const size_t c  = size_t(y) * W + x;
  const float4 cc = color[c], ca = albedo[c];
  const float3 dm = demodulator(ca);
  const float3 il = make_float3(cc.x / dm.x, cc.y / dm.y, cc.z / dm.z);
  float        var;
  if(frames >= 4.0f)
  {
    const float l  = lum709(cc.x, cc.y, cc.z);
    const float dl = lum709(dm.x, dm.y, dm.z);
    var            = fmaxf(0.0f, normal[c].w - l * l) / frames / (dl * dl);
  }
  else
  {
    // spatial estimate over the 7x7 neighbourhood of the same kind (geometry / background) with a similar normal
    const float4 cn = normal[c];
    float        s1 = 0.0f, s2 = 0.0f, sw = 0.0f;
    for(int dy = -3; dy <= 3; ++dy)
      for(int dx = -3; dx <= 3; ++dx)
      {
        const int qx = x + dx, qy = y + dy;
        if(qx < 0 || qy < 0 || qx >= W || qy >= H)
          continue;
        const size_t q  = size_t(qy) * W + qx;
        const float4 qa = albedo[q];
        if((qa.w > 0.5f) != (ca.w > 0.5f))
          continue;
        const float4 qn = normal[q], qc = color[q];
        const float  w  = ca.w > 0.5f ? (fmaxf(0.0f, qn.x * cn.x + qn.y * cn.y + qn.z * cn.z) > 0.9f ? 1.0f : 0.0f) : 1.0f;
        const float3 qd = demodulator(qa);
        const float  l  = lum709(qc.x / qd.x, qc.y / qd.y, qc.z / qd.z);
        s1 += w * l;
        s2 += w * l * l;
        sw += w;
      }
    const float m = sw > 0.0f ? s1 / sw : 0.0f;
    var           = sw > 0.0f ? fmaxf(0.0f, s2 / sw - m * m) : 0.0f;
  }
  illum[c] = make_float4(il.x, il.y, il.z, var);
}

__global__ void __launch_bounds__(256) k_svgf_atrous(const float4* __restrict__ in, float4* __restrict__ out, const float4* __restrict__ albedo,
                                                    const float4* __restrict__ normal, const float* __restrict__ depth, int W, int H, int step, float sigmaL,
                                                    float sigmaN, float sigmaZ)
{
  const int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
  if(x >= W || y >= H)
    return;
  const float  kern[3]  = {3.0f / 8.0f, 1.0f / 4.0f, 1.0f / 16.0f};  // B3 spline
  const float  gauss[2] = {0.5f, 0.25f};                              // 3x3 binomial: (1 2 1)^T (1 2 1) / 16
  const size_t c        = size_t(y) * W + x;
  const float4 cc = in[c], cn = normal[c];
  const bool   solid = albedo[c].w > 0.5f;
  // variance at the centre, 3x3 prefiltered (paper §4.4); taps outside the image are dropped and the kernel renormalised
  float gv = 0.0f, gw = 0.0f;
  for(int dy = -1; dy <= 1; ++dy)
    for(int dx = -1; dx <= 1; ++dx)
    {
      const int qx = x + dx, qy = y + dy;
      if(qx < 0 || qy < 0 || qx >= W || qy >= H)
        continue;
      const float w = gauss[abs(dx)] * gauss[abs(dy)];
      gv += w * in[size_t(qy) * W + qx].w;
      gw += w;
    }
  const float sdev = sqrtf(fmaxf(gv / gw, 0.0f));
  const float lc   = lum709(cc.x, cc.y, cc.z);
  const float zc   = depthKey(depth[c]);
  // depth gradient per pixel step (forward differences, backward at the border)
  const float zx = depthKey(depth[size_t(y) * W + (x + 1 < W ? x + 1 : x - (W > 1))]);
  const float zy = depthKey(depth[size_t(y + 1 < H ? y + 1 : y - (H > 1)) * W + x]);
  const float gz = fmaxf(fabsf(zx - zc), fabsf(zy - zc));
  float sr = cc.x * kern[0] * kern[0], sg = cc.y * kern[0] * kern[0], sb = cc.z * kern[0] * kern[0], sv = cc.w * kern[0] * kern[0] * kern[0] * kern[0];
  float sw = kern[0] * kern[0];
  for(int dy = -2; dy <= 2; ++dy)
    for(int dx = -2; dx <= 2; ++dx)
    {
      if(dx == 0 && dy == 0)
        continue;
      const int qx = x + dx * step, qy = y + dy * step;
      if(qx < 0 || qy < 0 || qx >= W || qy >= H)
        continue;
      const size_t q = size_t(qy) * W + qx;
      if((albedo[q].w > 0.5f) != solid)
        continue;
      const float4 qc = in[q];
      float        w  = expf(-fabsf(lum709(qc.x, qc.y, qc.z) - lc) / (sigmaL * sdev + 1e-6f));
      if(solid)
      {
        const float4 qn   = normal[q];
        const float  dist = float(step) * sqrtf(float(dx * dx + dy * dy));
        w *= powf(fmaxf(0.0f, qn.x * cn.x + qn.y * cn.y + qn.z * cn.z), sigmaN);
        w *= expf(-fabsf(depthKey(depth[q]) - zc) / (sigmaZ * gz * dist + 1e-6f * zc));
      }
      const float h = kern[abs(dx)] * kern[abs(dy)] * w;
      sr += qc.x * h;
      sg += qc.y * h;
      sb += qc.z * h;
      sv += qc.w * h * h;
      sw += h;
    }
  out[c] = make_float4(sr / sw, sg / sw, sb / sw, sv / (sw * sw));
}

__global__ void __launch_bounds__(256) k_svgf_finish(const float4* __restrict__ illum, const float4* __restrict__ color, const float4* __restrict__ albedo,
                                                    float4* __restrict__ out, size_t n)
{
  const size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
  if(i >= n)
    return;
  const float4 il = illum[i];
  const float3 dm = demodulator(albedo[i]);
  out[i]          = make_float4(il.x * dm.x, il.y * dm.y, il.z * dm.z, color[i].w);
}

}  // namespace

void launchAtrous(const float4* in, float4* out, const float4* albedo, const float4* normal, int width, int height, int step, float sigmaColor,
                  float sigmaNormal, float sigmaAlbedo, hipStream_t s)
{
  dim3 grid((width + 15) / 16, (height + 15) / 16);
  hipLaunchKernelGGL(k_atrous, grid, dim3(256), 0, s, in, out, albedo, normal, width, height, step, sigmaColor, sigmaNormal, sigmaAlbedo);
}

// SVGF pass: returns the buffer (bufA or bufB) that holds the result
const float4* launchSvgf(const float4* color, const float4* albedo, const float4* normal, const float* depth, float4* bufA, float4* bufB, int width, int height,
                         int iterations, float frames, float sigmaLuminance, float sigmaNormal, float sigmaDepth, hipStream_t s)
{
  dim3 grid((width + 15) / 16, (height + 15) / 16);
  hipLaunchKernelGGL(k_svgf_prepare, grid, dim3(256), 0, s, color, albedo, normal, bufA, width, height, frames);
  float4 *in = bufA, *out = bufB;
  for(int i = 0; i < iterations; ++i)
  {
    hipLaunchKernelGGL(k_svgf_atrous, grid, dim3(256), 0, s, in, out, albedo, normal, depth, width, height, 1 << i, sigmaLuminance, sigmaNormal, sigmaDepth);
    float4* t = in;
    in        = out;
    out       = t;
  }
  const size_t n = size_t(width) * size_t(height);
  hipLaunchKernelGGL(k_svgf_finish, dim3(unsigned((n + 255) / 256)), dim3(256), 0, s, in, color, albedo, out, n);
  return out;
}

}  // namespace pt
