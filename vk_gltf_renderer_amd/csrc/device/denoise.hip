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

}  // namespace

void launchAtrous(const float4* in, float4* out, const float4* albedo, const float4* normal, int width, int height, int step, float sigmaColor,
                  float sigmaNormal, float sigmaAlbedo, hipStream_t s)
{
  dim3 grid((width + 15) / 16, (height + 15) / 16);
  hipLaunchKernelGGL(k_atrous, grid, dim3(256), 0, s, in, out, albedo, normal, width, height, step, sigmaColor, sigmaNormal, sigmaAlbedo);
}

}  // namespace pt
