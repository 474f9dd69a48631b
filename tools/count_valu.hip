// Static vector-instruction counts of the two inner operations of the per-lane BVH walk, for the VALU roofline of bench.py
// (LABNOTES.md §4): one 8-wide node visit (bvh8Visit: decode + 8 slab tests + child ordering) and one triangle test
// (intersectTri + closest update).  Build + count: tools/count_valu.sh.  Not part of the product.
#include <hip/hip_runtime.h>
#include "pt_bvh8.h"
using namespace pt;
extern "C" __global__ void probe_node_visit(DevScene sc, const float4* rays, uint4* out, const uint4* ldsDummy)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const RaySetup r = makeRaySetup(xyz(rays[2 * i]), xyz(rays[2 * i + 1]));
  NodeGroup      G;
  uint32_t       tb, tm;
  bvh8Visit(sc, r, rays[2 * i].w, rayOctInv(r.idir), uint32_t(__float_as_uint(rays[2 * i + 1].w)), G, tb, tm, nullptr, 0u);
  out[i] = make_uint4(G.base, G.bits, tb, tm);
}
extern "C" __global__ void probe_baseline(DevScene sc, const float4* rays, uint4* out, const uint4* ldsDummy)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const RaySetup r = makeRaySetup(xyz(rays[2 * i]), xyz(rays[2 * i + 1]));
  out[i] = make_uint4(__float_as_uint(r.idir.x), __float_as_uint(r.idir.y), __float_as_uint(r.idir.z), rayOctInv(r.idir));
}
extern "C" __global__ void probe_tri_test(DevScene sc, const float4* rays, float4* out)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const DevTri   T = sc.tris[i];
  TriHit         h;
  float4         best = out[i];
  if(intersectTri(xyz(T.a), xyz(T.b), xyz(T.c), xyz(rays[2 * i]), xyz(rays[2 * i + 1]), h) && h.t > 0.0f && h.t <= best.x)
  {
    const uint32_t flags = __float_as_uint(T.c.w);
    const bool     front = h.front != ((flags & INST_FLIP_FACING) != 0u);
    if(front || (flags & INST_CULL_DISABLE))
      best = make_float4(h.t, __int_as_float(int(i)), h.u, h.v);
  }
  out[i] = best;
}
extern "C" __global__ void probe_tri_baseline(DevScene sc, const float4* rays, float4* out)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const DevTri   T = sc.tris[i];
  float4         best = out[i];
  best.x += T.a.x + T.b.y + T.c.z + rays[2 * i].x + rays[2 * i + 1].y;
  out[i] = best;
}
