// On-device BVH construction (replaces the BLAS/TLAS builds of the reference: src/gltf_scene_rtx.cpp:173-227, :299-385).
//
// All render-node instances are flattened to world-space triangles (288 GB of HBM make instancing-by-reference
// unnecessary for the target scenes and it removes the per-instance ray transform from the traversal loop), then
//   1. k_tri_setup   : world-space vertices (fixed fmaf order, shared with the oracle), AABB, centroid bounds
//   2. k_morton      : 63-bit Morton code of the AABB centre
//   3. rocPRIM radix sort of (code, triangle)
//   4. k_hierarchy   : Karras 2012, "Maximizing Parallelism in the Construction of BVHs, Octrees, and k-d Trees"
//   5. k_fit         : bottom-up AABB refit, second arrival at a node continues upward (agent-scope fences)
//   6. k_emit        : 64-byte traversal nodes holding both children's boxes; triangles re-ordered in Morton order
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "bvh_reinsert.h"
#include "bvh_split.h"
#include "pt_build.h"
#include "pt_bvh.h"

namespace pt {

namespace {

struct BuildTables
{
  const MiGltfRenderNode* nodes;
  const DevPrim*          prims;
  const uint8_t*          instFlags;      // per render node
  const uint32_t*         nodeTriOffset;  // exclusive scan of triangle counts over *visible* render nodes, size numEntries+1
  const int32_t*          entryNode;      // render-node index of each entry
  int                     numEntries;
};

__device__ __forceinline__ uint32_t floatToOrdered(float f)
{
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float orderedToFloat(uint32_t u)
{
  return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

__global__ void k_tri_setup(BuildTables T, uint32_t numTris, DevTri* tris, float4* boxLo, float4* boxHi, uint32_t* sceneBounds)
{
  uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  float    lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
  if(g < numTris)
  {
    // binary search the owning entry
    int a = 0, b = T.numEntries;
    while(b - a > 1)
    {
      int m = (a + b) >> 1;
      if(T.nodeTriOffset[m] <= g)
        a = m;
      else
        b = m;
    }
    const int               rnode = T.entryNode[a];
    const uint32_t          t     = g - T.nodeTriOffset[a];
    const MiGltfRenderNode& rn    = T.nodes[rnode];
    const DevPrim&          rp    = T.prims[rn.renderPrimID];
    const uint32_t          i0 = rp.indices[3 * t], i1 = rp.indices[3 * t + 1], i2 = rp.indices[3 * t + 2];
    f3 p0 = mulPoint(rn.objectToWorld, mk3(rp.positions + 3 * size_t(i0)));
    f3 p1 = mulPoint(rn.objectToWorld, mk3(rp.positions + 3 * size_t(i1)));
    f3 p2 = mulPoint(rn.objectToWorld, mk3(rp.positions + 3 * size_t(i2)));
    // Vertex buffers and instance matrices are untrusted bytes.  A triangle with a NaN, an infinity or a coordinate whose square
    // overflows would poison the scene bounds, the Morton keys and the surface areas the clustering compares; it becomes a point at
    // the origin instead -- zero area, so no ray hits it -- and the rest of the scene builds and renders as if it were not there.
    {
      const float big = 1.0e18f;
      const bool  ok  = fabsf(p0.x) < big && fabsf(p0.y) < big && fabsf(p0.z) < big && fabsf(p1.x) < big && fabsf(p1.y) < big && fabsf(p1.z) < big
                      && fabsf(p2.x) < big && fabsf(p2.y) < big && fabsf(p2.z) < big;  // false for NaN as well
      if(!ok)
        p0 = p1 = p2 = mk3(0.0f);
    }
    f3 e1 = p1 - p0, e2 = p2 - p0;
    DevTri tri;
    tri.a   = make_float4(p0.x, p0.y, p0.z, __int_as_float(rnode));
    tri.b   = make_float4(e1.x, e1.y, e1.z, __int_as_float(int(t)));
    // (a triangle the load-time classification found opaque -- DevPrim::opaqueTriangles -- counts as FORCE_OPAQUE like an opaque instance)
    tri.c   = make_float4(e2.x, e2.y, e2.z, __uint_as_float(uint32_t(T.instFlags[rnode]) | (t < rp.opaqueTriangles ? uint32_t(INST_FORCE_OPAQUE) : 0u)));
    tris[g] = tri;
    // bounds from the same p0 + e arithmetic the intersector sees
    f3 q1 = p0 + e1, q2 = p0 + e2;
    lo[0] = fminf(p0.x, fminf(q1.x, q2.x)); hi[0] = fmaxf(p0.x, fmaxf(q1.x, q2.x));
    lo[1] = fminf(p0.y, fminf(q1.y, q2.y)); hi[1] = fmaxf(p0.y, fmaxf(q1.y, q2.y));
    lo[2] = fminf(p0.z, fminf(q1.z, q2.z)); hi[2] = fmaxf(p0.z, fmaxf(q1.z, q2.z));
    // include the true vertices too (p0+e may round inward)
    lo[0] = fminf(lo[0], fminf(p1.x, p2.x)); hi[0] = fmaxf(hi[0], fmaxf(p1.x, p2.x));
    lo[1] = fminf(lo[1], fminf(p1.y, p2.y)); hi[1] = fmaxf(hi[1], fmaxf(p1.y, p2.y));
    lo[2] = fminf(lo[2], fminf(p1.z, p2.z)); hi[2] = fmaxf(hi[2], fmaxf(p1.z, p2.z));
    boxLo[g] = make_float4(lo[0], lo[1], lo[2], 0.0f);
    boxHi[g] = make_float4(hi[0], hi[1], hi[2], 0.0f);
  }
  // block reduction of centroid bounds -> 6 atomics per block
  __shared__ float s_lo[3][256], s_hi[3][256];
  for(int c = 0; c < 3; ++c)
  {
    float cen            = (g < numTris) ? 0.5f * (lo[c] + hi[c]) : 0.0f;
    s_lo[c][threadIdx.x] = (g < numTris) ? cen : 3.0e38f;
    s_hi[c][threadIdx.x] = (g < numTris) ? cen : -3.0e38f;
  }
  __syncthreads();
  for(int s = 128; s > 0; s >>= 1)
  {
    if(threadIdx.x < s)
      for(int c = 0; c < 3; ++c)
      {
        s_lo[c][threadIdx.x] = fminf(s_lo[c][threadIdx.x], s_lo[c][threadIdx.x + s]);
        s_hi[c][threadIdx.x] = fmaxf(s_hi[c][threadIdx.x], s_hi[c][threadIdx.x + s]);
      }
    __syncthreads();
  }
  if(threadIdx.x < 3)
  {
    atomicMin(&sceneBounds[threadIdx.x], floatToOrdered(s_lo[threadIdx.x][0]));
    atomicMax(&sceneBounds[3 + threadIdx.x], floatToOrdered(s_hi[threadIdx.x][0]));
  }
}

// ---- triangle pre-splitting (bvh_split.h): box area per triangle -> references per triangle (count pass) -> scan -> references (emit pass) ----
__global__ void k_split_area(uint32_t n, const float4* boxLo, const float4* boxHi, float* area)
{
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if(g >= n)
    return;
  const float4 lo = boxLo[g], hi = boxHi[g];
  const float  ex = hi.x - lo.x, ey = hi.y - lo.y, ez = hi.z - lo.z;
  area[g]         = ex * ey + ey * ez + ez * ex;
}
// area of the triangles whose box exceeds `factor` x the mean, 0 for the others (summed to decide whether the scene is split at all)
__global__ void k_split_large_area(uint32_t n, const float* area, const float* areaSum, float factor, float* large)
{
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if(g >= n)
    return;
  const float a = area[g];
  large[g]      = a > factor * (*areaSum / float(n)) ? a : 0.0f;
}
// offsets == nullptr: count pass (counts[g] = references of triangle g); otherwise emit pass (reference offsets[g] + i = the i-th of triangle g)
__global__ void k_split_refs(uint32_t n, const DevTri* tris, const float4* boxLo, const float4* boxHi, const float* areaSum, float factor, int maxDepth,
                             const uint32_t* offsets, uint32_t* counts, DevTri* outTris, float4* outLo, float4* outHi)
{
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if(g >= n)
    return;
  const DevTri T  = tris[g];
  const float4 lo = boxLo[g], hi = boxHi[g];
  SplitBox     tb;
  tb.lo[0] = lo.x; tb.lo[1] = lo.y; tb.lo[2] = lo.z; tb.hi[0] = hi.x; tb.hi[1] = hi.y; tb.hi[2] = hi.z;
  const float p[3][3] = {{T.a.x, T.a.y, T.a.z}, {T.a.x + T.b.x, T.a.y + T.b.y, T.a.z + T.b.z}, {T.a.x + T.c.x, T.a.y + T.c.y, T.a.z + T.c.z}};
  // triangles of transmissive instances keep ONE reference: the recording shadow walk counts its candidates (bvh_split.h)
  const bool  splittable = (__float_as_uint(T.c.w) & INST_TRANSMISSIVE) == 0u;
  const float threshold  = factor * (*areaSum / float(n));
  if(!offsets)
  {
    counts[g] = uint32_t(splitTriangle(p, tb, splittable, threshold, maxDepth, [](const SplitBox&) {}));
    return;
  }
  uint32_t o = offsets[g];
  splitTriangle(p, tb, splittable, threshold, maxDepth, [&](const SplitBox& b) {
    outTris[o] = T;
    outLo[o]   = make_float4(b.lo[0], b.lo[1], b.lo[2], 0.0f);
    outHi[o]   = make_float4(b.hi[0], b.hi[1], b.hi[2], 0.0f);
    ++o;
  });
}
// centroid bounds of the references (k_tri_setup's are those of the triangles)
__global__ void k_centroid_bounds(uint32_t n, const float4* boxLo, const float4* boxHi, uint32_t* sceneBounds)
{
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  float          lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
  if(g < n)
  {
    const float4 a = boxLo[g], b = boxHi[g];
    lo[0] = hi[0] = 0.5f * (a.x + b.x); lo[1] = hi[1] = 0.5f * (a.y + b.y); lo[2] = hi[2] = 0.5f * (a.z + b.z);
  }
  for(int c = 0; c < 3; ++c)
  {
    for(int o = 32; o >= 1; o >>= 1)
    {
      lo[c] = fminf(lo[c], __shfl_xor(lo[c], o));
      hi[c] = fmaxf(hi[c], __shfl_xor(hi[c], o));
    }
    if((threadIdx.x & 63u) == 0u)
    {
      atomicMin(&sceneBounds[c], floatToOrdered(lo[c]));
      atomicMax(&sceneBounds[3 + c], floatToOrdered(hi[c]));
    }
  }
}

__device__ __forceinline__ uint64_t expandBits21(uint32_t v)
{
  uint64_t x = v & 0x1fffffu;
  x          = (x | x << 32) & 0x1f00000000ffffull;
  x          = (x | x << 16) & 0x1f0000ff0000ffull;
  x          = (x | x << 8) & 0x100f00f00f00f00full;
  x          = (x | x << 4) & 0x10c30c30c30c30c3ull;
  x          = (x | x << 2) & 0x1249249249249249ull;
  return x;
}

__global__ void k_morton(uint32_t numTris, const float4* boxLo, const float4* boxHi, const uint32_t* sceneBounds, uint64_t* keys, uint32_t* vals)
{
  uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if(g >= numTris)
    return;
  float blo[3], bhi[3];
  for(int c = 0; c < 3; ++c)
  {
    blo[c] = orderedToFloat(sceneBounds[c]);
    bhi[c] = orderedToFloat(sceneBounds[3 + c]);
  }
  float4   lo = boxLo[g], hi = boxHi[g];
  float    cen[3] = {0.5f * (lo.x + hi.x), 0.5f * (lo.y + hi.y), 0.5f * (lo.z + hi.z)};
  uint32_t q[3];
  for(int c = 0; c < 3; ++c)
  {
    float ext = bhi[c] - blo[c];
    float n   = ext > 0.0f ? (cen[c] - blo[c]) / ext : 0.0f;
    q[c]      = uint32_t(fminf(fmaxf(n * 2097152.0f, 0.0f), 2097151.0f));
  }
  keys[g] = (expandBits21(q[0]) << 2) | (expandBits21(q[1]) << 1) | expandBits21(q[2]);
  vals[g] = g;
}

// common-prefix length of sorted keys i and j, index-augmented so that duplicate codes still form a balanced tree
__device__ __forceinline__ int delta(const uint64_t* keys, int n, int i, int j)
{
  if(j < 0 || j >= n)
    return -1;
  uint64_t a = keys[i], b = keys[j];
  if(a == b)
    return 64 + __clz(uint32_t(i) ^ uint32_t(j));
  return __clzll((long long)(a ^ b));
}

// internal node i in [0, n-2]; child refs: >= 0 internal, < 0 leaf (~sorted position)
__global__ void k_hierarchy(int n, const uint64_t* keys, int2* children, int* parentInternal, int* parentLeaf)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= n - 1)
    return;
  int d      = (delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
  int dmin   = delta(keys, n, i, i - d);
  int lmax   = 2;
  while(delta(keys, n, i, i + lmax * d) > dmin)
    lmax <<= 1;
  int l = 0;
  for(int t = lmax >> 1; t >= 1; t >>= 1)
    if(delta(keys, n, i, i + (l + t) * d) > dmin)
      l += t;
  int j     = i + l * d;
  int dnode = delta(keys, n, i, j);
  int s     = 0;
  for(int t = (l + 1) >> 1;; t = (t + 1) >> 1)
  {
    if(delta(keys, n, i, i + (s + t) * d) > dnode)
      s += t;
    if(t == 1)
      break;
  }
  int gamma = i + s * d + min(d, 0);
  int left  = (min(i, j) == gamma) ? ~gamma : gamma;
  int right = (max(i, j) == gamma + 1) ? ~(gamma + 1) : (gamma + 1);
  children[i] = make_int2(left, right);
  if(left >= 0)
    parentInternal[left] = i;
  else
    parentLeaf[~left] = i;
  if(right >= 0)
    parentInternal[right] = i;
  else
    parentLeaf[~right] = i;
  if(i == 0)
    parentInternal[0] = -1;
}

__global__ void k_fit(int n, const uint32_t* vals, const float4* boxLo, const float4* boxHi, const int2* children, const int* parentInternal,
                      const int* parentLeaf, float4* nodeLo, float4* nodeHi, unsigned int* arrive, int* nodeCnt)
{
  int leaf = blockIdx.x * blockDim.x + threadIdx.x;
  if(leaf >= n)
    return;
  int cur = parentLeaf[leaf];
  while(cur >= 0)
  {
    __threadfence();  // release: boxes written below this point of the tree are visible before the ticket
    unsigned int ticket = atomicAdd(&arrive[cur], 1u);
    if(ticket == 0)
      return;         // first arrival: the sibling subtree finishes this node
    __threadfence();  // acquire: drop stale L1 lines before reading the sibling's box
    int2   ch = children[cur];
    float4 lo0, hi0, lo1, hi1;
    if(ch.x >= 0)
    {
      lo0 = nodeLo[ch.x];
      hi0 = nodeHi[ch.x];
    }
    else
    {
      uint32_t t = vals[~ch.x];
      lo0 = boxLo[t];
      hi0 = boxHi[t];
    }
    if(ch.y >= 0)
    {
      lo1 = nodeLo[ch.y];
      hi1 = nodeHi[ch.y];
    }
    else
    {
      uint32_t t = vals[~ch.y];
      lo1 = boxLo[t];
      hi1 = boxHi[t];
    }
    nodeLo[cur] = make_float4(fminf(lo0.x, lo1.x), fminf(lo0.y, lo1.y), fminf(lo0.z, lo1.z), 0.0f);
    nodeHi[cur] = make_float4(fmaxf(hi0.x, hi1.x), fmaxf(hi0.y, hi1.y), fmaxf(hi0.z, hi1.z), 0.0f);
    nodeCnt[cur] = (ch.x >= 0 ? nodeCnt[ch.x] : 1) + (ch.y >= 0 ? nodeCnt[ch.y] : 1);  // triangles below (read by the 8-wide collapse)
    cur         = parentInternal[cur];
  }
}

__global__ void k_emit_nodes(int n, const uint32_t* vals, const float4* boxLo, const float4* boxHi, const int2* children, const float4* nodeLo,
                             const float4* nodeHi, const int* nodeCnt, float4* outNodes)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= n - 1)
    return;
  int2   ch = children[i];
  float4 lo0, hi0, lo1, hi1;
  if(ch.x >= 0) { lo0 = nodeLo[ch.x]; hi0 = nodeHi[ch.x]; } else { uint32_t t = vals[~ch.x]; lo0 = boxLo[t]; hi0 = boxHi[t]; }
  if(ch.y >= 0) { lo1 = nodeLo[ch.y]; hi1 = nodeHi[ch.y]; } else { uint32_t t = vals[~ch.y]; lo1 = boxLo[t]; hi1 = boxHi[t]; }
  float4* o = outNodes + size_t(i) * 4;
  o[0]      = make_float4(lo0.x, hi0.x, lo0.y, hi0.y);
  o[1]      = make_float4(lo1.x, hi1.x, lo1.y, hi1.y);
  o[2]      = make_float4(lo0.z, hi0.z, lo1.z, hi1.z);
  o[3]      = make_float4(__int_as_float(ch.x), __int_as_float(ch.y), __int_as_float(nodeCnt[i]), 0.0f);
}

// ---- PLOC: parallel locally-ordered clustering (Meister & Bittner 2018) ----------------------------------------------
// Bottom-up agglomerative build over the Morton-sorted leaves: every cluster looks PLOC_RADIUS positions either way for
// the neighbour whose union with it has the smallest surface area; mutual nearest neighbours merge into a new node; the
// survivors are compacted and the round repeats until one cluster is left.  Tree quality is close to a full SAH sweep
// (the Karras topology above only follows the Morton prefix and is ~1.5-2x worse on architectural scenes), and every
// round is three flat kernels plus a prefix sum.  All decisions are tie-broken by position, node indices come from the
// scan (no atomics), so the tree is a pure function of the input.
#ifndef PLOC_RADIUS
#define PLOC_RADIUS 16
#endif

// (the .w of a cluster's lower corner carries the number of triangles below it, as int bits: the 8-wide collapse reads it from
//  the node record, n3.z)
__global__ void k_ploc_init(int n, const uint32_t* vals, const float4* boxLo, const float4* boxHi, int* cid, float4* clo, float4* chi)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= n)
    return;
  uint32_t t = vals[i];
  cid[i]     = ~i;
  float4 lo  = boxLo[t];
  lo.w       = __int_as_float(1);
  clo[i]     = lo;
  chi[i]     = boxHi[t];
}

__global__ void __launch_bounds__(256) k_ploc_nn(int m, const float4* clo, const float4* chi, int* nn)
{
  __shared__ float4 sLo[256 + 2 * PLOC_RADIUS], sHi[256 + 2 * PLOC_RADIUS];
  const int base = int(blockIdx.x) * 256 - PLOC_RADIUS;
  for(int k = threadIdx.x; k < 256 + 2 * PLOC_RADIUS; k += 256)
  {
    int g = base + k;
    if(g >= 0 && g < m)
    {
      sLo[k] = clo[g];
      sHi[k] = chi[g];
    }
  }
  __syncthreads();
  const int i = int(blockIdx.x) * 256 + int(threadIdx.x);
  if(i >= m)
    return;
  const int    me = int(threadIdx.x) + PLOC_RADIUS;
  const float4 lo = sLo[me], hi = sHi[me];
  float        best = FLT_MAX;
  int          bj   = -1;
  for(int d = -PLOC_RADIUS; d <= PLOC_RADIUS; ++d)
  {
    const int g = i + d;
    if(d == 0 || g < 0 || g >= m)
      continue;
    const float4 l2 = sLo[me + d], h2 = sHi[me + d];
    const float  ex = fmaxf(hi.x, h2.x) - fminf(lo.x, l2.x), ey = fmaxf(hi.y, h2.y) - fminf(lo.y, l2.y), ez = fmaxf(hi.z, h2.z) - fminf(lo.z, l2.z);
    const float  a  = __fadd_rn(__fadd_rn(__fmul_rn(ex, ey), __fmul_rn(ey, ez)), __fmul_rn(ez, ex));  // symmetric in the pair
    if(a < best)  // ascending g: ties go to the smaller position, which makes a mutual pair always exist
    {
      best = a;
      bj   = g;
    }
  }
  nn[i] = bj;
}

// flag = survives-this-round | (starts-a-merge << 32); the exclusive scan of it yields the compacted position (low word)
// and the rank of the merge, i.e. the new node's index offset (high word)
__global__ void k_ploc_flags(int m, const int* nn, unsigned long long* flags)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= m)
    return;
  const int  j      = nn[i];
  const bool mutual = j >= 0 && nn[j] == i;
  flags[i]          = (unsigned long long)((mutual && i > j) ? 0u : 1u) | ((unsigned long long)((mutual && i < j) ? 1u : 0u) << 32);
}

__global__ void k_ploc_emit(int m, const int* nn, const unsigned long long* flags, const unsigned long long* pos, const int* cid, const float4* clo,
                            const float4* chi, int* cid2, float4* clo2, float4* chi2, int nodeBase, float4* outNodes, unsigned long long* totals)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= m)
    return;
  const unsigned long long f = flags[i], ps = pos[i];
  if(i == m - 1)
    *totals = ps + f;
  if(!(f & 1ull))
    return;
  const uint32_t p = uint32_t(ps);
  if(f >> 32)
  {
    const int    j = nn[i], k = nodeBase + int(ps >> 32);
    const float4 lo0 = clo[i], hi0 = chi[i], lo1 = clo[j], hi1 = chi[j];
    float4*      o = outNodes + size_t(k) * 4;
    o[0]           = make_float4(lo0.x, hi0.x, lo0.y, hi0.y);
    o[1]           = make_float4(lo1.x, hi1.x, lo1.y, hi1.y);
    o[2]           = make_float4(lo0.z, hi0.z, lo1.z, hi1.z);
    const int      cnt = __float_as_int(lo0.w) + __float_as_int(lo1.w);
    o[3]           = make_float4(__int_as_float(cid[i]), __int_as_float(cid[j]), __int_as_float(cnt), 0.0f);
    cid2[p]        = k;
    clo2[p]        = make_float4(fminf(lo0.x, lo1.x), fminf(lo0.y, lo1.y), fminf(lo0.z, lo1.z), __int_as_float(cnt));
    chi2[p]        = make_float4(fmaxf(hi0.x, hi1.x), fmaxf(hi0.y, hi1.y), fmaxf(hi0.z, hi1.z), 0.0f);
  }
  else
  {
    cid2[p] = cid[i];
    clo2[p] = clo[i];
    chi2[p] = chi[i];
  }
}

__global__ void k_emit_tris(int n, const uint32_t* vals, const DevTri* tris, DevTri* outTris)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i < n)
    outTris[i] = tris[vals[i]];
}

// ---- reinsertion passes over the finished BVH2 (bvh_reinsert.h holds the work of one thread of each phase) ---------------------------
__global__ void k_re_parents(Bvh2Tree T)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i < T.numInner)
    reinsertParents(T, i);
}
__global__ void k_re_search(Bvh2Tree T, int ids, ReinsertMove* moves)
{
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  if(id < ids)
    moves[id] = reinsertSearch(T, id);
}
__global__ void k_re_lock(Bvh2Tree T, int ids, ReinsertMove* moves, unsigned long long* locks)
{
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  if(id < ids)
    reinsertLock(T, moves, locks, id);
}
__global__ void k_re_apply(Bvh2Tree T, int ids, ReinsertMove* moves, unsigned long long* locks, unsigned int* carriedOut)
{
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  if(id < ids && reinsertApply(T, moves, locks, id))
    atomicAdd(carriedOut, 1u);
}
__global__ void k_re_unlock(int ids, unsigned long long* locks)
{
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  if(id < ids)
    reinsertUnlock(locks, id);
}
__global__ void k_re_refit(Bvh2Tree T, unsigned int* arrive)
{
  const int leaf = blockIdx.x * blockDim.x + threadIdx.x;
  if(leaf <= T.numInner)
    reinsertRefit(T, arrive, leaf);
}

// A failed call records the error and LEAVES THE ENCLOSING LOOP: the macro is a plain block (no do { } while(0) of its own, whose `break` would
// only leave the macro), so its `break` belongs to the loop it is written in -- the do { } while(0) around a builder's body, or a pass / round
// loop inside it, whose conditions test `ok` and which are followed by `if(!ok) break;` where more steps come after them.
#define BUILD_CHECK(x)                                                                                                  \
  {                                                                                                                     \
    hipError_t e_ = (x);                                                                                                \
    if(e_ != hipSuccess)                                                                                                \
    {                                                                                                                   \
      err = std::string(#x) + ": " + hipGetErrorString(e_);                                                             \
      ok  = false;                                                                                                      \
      break;                                                                                                            \
    }                                                                                                                   \
  }

}  // namespace

// `passes` searches, each followed by up to `rounds` lock / move rounds and one refit; stops early once a pass moves fewer than one node
// in two thousand.  Leaves the records, the root index and the triangle counts in the form the collapse reads (bvh8.hip).
static bool reinsertBvh2(float4* nodes, int numInner, int root, int passes, int rounds, hipStream_t stream, std::string& err, uint32_t* movesOut)
{
  if(movesOut)
    *movesOut = 0;
  if(passes <= 0 || numInner < 3)
    return true;
  static const bool   timing = getenv("MI_PT_BUILD_TIMING") != nullptr;
  const auto          t0     = std::chrono::steady_clock::now();
  int                 passesRun = 0;
  uint32_t            movesAll  = 0;
  const int           ids = 2 * numInner + 1, B = 256;
  int *               parent = nullptr, *leafParent = nullptr;
  ReinsertMove*       moves  = nullptr;
  unsigned long long* locks  = nullptr;
  unsigned int *      arrive = nullptr, *carried = nullptr;
  bool                ok     = true;
  do
  {
    BUILD_CHECK(hipMalloc(&parent, sizeof(int) * numInner));
    BUILD_CHECK(hipMalloc(&leafParent, sizeof(int) * (numInner + 1)));
    BUILD_CHECK(hipMalloc(&moves, sizeof(ReinsertMove) * ids));
    BUILD_CHECK(hipMalloc(&locks, sizeof(unsigned long long) * ids));
    BUILD_CHECK(hipMalloc(&arrive, sizeof(unsigned int) * numInner));
    BUILD_CHECK(hipMalloc(&carried, sizeof(unsigned int)));
    const Bvh2Tree T{nodes, parent, leafParent, numInner, root};
    const dim3     gN((numInner + B - 1) / B), gI((ids + B - 1) / B), gL((numInner + 1 + B - 1) / B);
    for(int pass = 0; pass < passes && ok; ++pass)
    {
      hipLaunchKernelGGL(k_re_parents, gN, dim3(B), 0, stream, T);
      hipLaunchKernelGGL(k_re_search, gI, dim3(B), 0, stream, T, ids, moves);
      BUILD_CHECK(hipMemsetAsync(locks, 0, sizeof(unsigned long long) * ids, stream));
      BUILD_CHECK(hipMemsetAsync(carried, 0, sizeof(unsigned int), stream));
      for(int round = 0; round < rounds; ++round)
      {
        hipLaunchKernelGGL(k_re_lock, gI, dim3(B), 0, stream, T, ids, moves, locks);
        hipLaunchKernelGGL(k_re_apply, gI, dim3(B), 0, stream, T, ids, moves, locks, carried);
        hipLaunchKernelGGL(k_re_unlock, gI, dim3(B), 0, stream, ids, locks);
      }
      hipLaunchKernelGGL(k_re_parents, gN, dim3(B), 0, stream, T);
      BUILD_CHECK(hipMemsetAsync(arrive, 0, sizeof(unsigned int) * numInner, stream));
      hipLaunchKernelGGL(k_re_refit, gL, dim3(B), 0, stream, T, arrive);
      BUILD_CHECK(hipGetLastError());
      unsigned int done = 0;
      BUILD_CHECK(hipMemcpyAsync(&done, carried, sizeof(done), hipMemcpyDeviceToHost, stream));
      BUILD_CHECK(hipStreamSynchronize(stream));
      ++passesRun;
      movesAll += done;
      if(done < unsigned(numInner / 2000 + 1))
        break;
    }
  } while(0);
  (void)hipFree(parent); (void)hipFree(leafParent); (void)hipFree(moves); (void)hipFree(locks); (void)hipFree(arrive); (void)hipFree(carried);
  if(movesOut)
    *movesOut = movesAll;
  if(timing)
    fprintf(stderr, "[mi_pt build] reinsertion: %d of %d passes x %d rounds over %d nodes and leaves, %u subtrees moved, %.2f ms\n", passesRun, passes, rounds, ids, movesAll,
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  return ok;
}

bool buildBvh(const BvhBuildInput& in, BvhBuildOutput& out, hipStream_t stream, std::string& err)
{
  out               = BvhBuildOutput();
  uint32_t n        = in.numTris;  // becomes the number of REFERENCES once the large triangles are pre-split (below)
  out.numTris       = n;
  out.sceneTris     = n;
  out.root          = BVH_EMPTY;
  if(n == 0)
    return true;
  DevTri*   trisTmp = nullptr;
  float4 *  boxLo = nullptr, *boxHi = nullptr, *nodeLo = nullptr, *nodeHi = nullptr;
  uint32_t *bounds = nullptr, *valsA = nullptr, *valsB = nullptr;
  uint64_t *keysA = nullptr, *keysB = nullptr;
  int2*     children = nullptr;
  int *     parentInternal = nullptr, *parentLeaf = nullptr;
  unsigned* arrive         = nullptr;
  int*      nodeCnt        = nullptr;
  void*     sortTemp       = nullptr;
  size_t    sortBytes      = 0;
  int *     cidA = nullptr, *cidB = nullptr, *nn = nullptr;
  float4 *  cloA = nullptr, *chiA = nullptr, *cloB = nullptr, *chiB = nullptr;
  unsigned long long *flags = nullptr, *pos = nullptr, *totals = nullptr;
  void*     scanTemp  = nullptr;
  size_t    scanBytes = 0;
  const int B              = 256;
  unsigned  gridT          = (n + B - 1) / B;
  float *   splitArea = nullptr, *splitSum = nullptr;
  uint32_t *splitCounts = nullptr, *splitOffsets = nullptr;
  void*     splitTemp = nullptr;
  BuildTables T{in.nodes, in.prims, in.instFlags, in.nodeTriOffset, in.entryNode, in.numEntries};
  const uint32_t initBounds[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
  bool           ok            = true;
  uint32_t       hb[6]         = {0, 0, 0, 0, 0, 0};

  do
  {
  BUILD_CHECK(hipMalloc(&trisTmp, sizeof(DevTri) * n));
  BUILD_CHECK(hipMalloc(&boxLo, sizeof(float4) * n));
  BUILD_CHECK(hipMalloc(&boxHi, sizeof(float4) * n));
  BUILD_CHECK(hipMalloc(&bounds, sizeof(initBounds)));
  BUILD_CHECK(hipMemcpyAsync(bounds, initBounds, sizeof(initBounds), hipMemcpyHostToDevice, stream));
  hipLaunchKernelGGL(k_tri_setup, dim3(gridT), dim3(B), 0, stream, T, n, trisTmp, boxLo, boxHi, bounds);
  BUILD_CHECK(hipGetLastError());

  // ---- pre-splitting: references instead of triangles from here on (bvh_split.h) ----
  if(in.splitFactor > 0.0f && n >= 2)
  {
    size_t tb1 = 0, tb2 = 0;
    BUILD_CHECK(hipMalloc(&splitArea, sizeof(float) * n));
    BUILD_CHECK(hipMalloc(&splitSum, sizeof(float)));
    BUILD_CHECK(hipMalloc(&splitCounts, sizeof(uint32_t) * n));
    BUILD_CHECK(hipMalloc(&splitOffsets, sizeof(uint32_t) * n));
    BUILD_CHECK(hipcub::DeviceReduce::Sum(nullptr, tb1, splitArea, splitSum, int(n), stream));
    BUILD_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, tb2, splitCounts, splitOffsets, int(n), stream));
    BUILD_CHECK(hipMalloc(&splitTemp, std::max(tb1, tb2)));
    hipLaunchKernelGGL(k_split_area, dim3(gridT), dim3(B), 0, stream, n, boxLo, boxHi, splitArea);
    BUILD_CHECK(hipcub::DeviceReduce::Sum(splitTemp, tb1, splitArea, splitSum, int(n), stream));
    // Splitting engages only where large triangles DOMINATE the scene's box area (the SAH is area weighted): the share of the summed box area held by
    // triangles above 64 x the mean is 0 / 0.05 on the evenly tessellated atrium / street stand-ins -- whose trees splitting only perturbs (-1 %) --
    // and 0.20 / 0.89 on their sliver versions (+6 % / +41 %).  The threshold is splitMinShare (default 0.1; 0 = always split).
    bool engage = true;
    if(in.splitMinShare > 0.0f)
    {
      float* large = reinterpret_cast<float*>(splitCounts);  // (not yet in use)
      float  sums[2] = {0.0f, 0.0f};
      hipLaunchKernelGGL(k_split_large_area, dim3(gridT), dim3(B), 0, stream, n, splitArea, splitSum, 64.0f, large);
      BUILD_CHECK(hipMemcpyAsync(&sums[0], splitSum, sizeof(float), hipMemcpyDeviceToHost, stream));
      BUILD_CHECK(hipcub::DeviceReduce::Sum(splitTemp, tb1, large, reinterpret_cast<float*>(splitOffsets), int(n), stream));
      BUILD_CHECK(hipMemcpyAsync(&sums[1], splitOffsets, sizeof(float), hipMemcpyDeviceToHost, stream));
      BUILD_CHECK(hipStreamSynchronize(stream));
      engage = sums[0] > 0.0f && sums[1] >= in.splitMinShare * sums[0];
    }
    // the references are budgeted: more than half as many again as there are triangles means the rule does not fit this scene (a few
    // giants among dust: the mean says little) -- the factor is doubled until it does, at most four times
    float    factor = in.splitFactor;
    uint32_t total  = n;
    for(int attempt = 0; engage && attempt < 5; ++attempt, factor *= 2.0f)
    {
      hipLaunchKernelGGL(k_split_refs, dim3(gridT), dim3(B), 0, stream, n, trisTmp, boxLo, boxHi, splitSum, factor, in.splitMaxDepth, nullptr, splitCounts, nullptr, nullptr, nullptr);
      BUILD_CHECK(hipGetLastError());
      BUILD_CHECK(hipcub::DeviceScan::ExclusiveSum(splitTemp, tb2, splitCounts, splitOffsets, int(n), stream));
      uint32_t lastOff = 0, lastCnt = 0;
      BUILD_CHECK(hipMemcpyAsync(&lastOff, splitOffsets + (n - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
      BUILD_CHECK(hipMemcpyAsync(&lastCnt, splitCounts + (n - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
      BUILD_CHECK(hipStreamSynchronize(stream));
      total = lastOff + lastCnt;
      if(uint64_t(total) <= uint64_t(n) + uint64_t(n) / 2u + 4096u && total < (1u << 26))
        break;
      total = n;
    }
    if(!ok)
      break;
    if(total > n)
    {
      // (the reference arrays replace the triangle arrays only once all three exist and are filled; a failure on the way frees what was allocated)
      DevTri* refTris = nullptr;
      float4 *refLo = nullptr, *refHi = nullptr;
      hipError_t e = hipMalloc(&refTris, sizeof(DevTri) * total);
      if(e == hipSuccess) e = hipMalloc(&refLo, sizeof(float4) * total);
      if(e == hipSuccess) e = hipMalloc(&refHi, sizeof(float4) * total);
      if(e == hipSuccess)
      {
        hipLaunchKernelGGL(k_split_refs, dim3(gridT), dim3(B), 0, stream, n, trisTmp, boxLo, boxHi, splitSum, factor, in.splitMaxDepth, splitOffsets, splitCounts, refTris, refLo, refHi);
        e = hipGetLastError();
      }
      if(e == hipSuccess) e = hipStreamSynchronize(stream);
      if(e != hipSuccess)
      {
        (void)hipFree(refTris); (void)hipFree(refLo); (void)hipFree(refHi);
        err = std::string("triangle pre-splitting: ") + hipGetErrorString(e);
        ok  = false;
        break;
      }
      (void)hipFree(trisTmp); (void)hipFree(boxLo); (void)hipFree(boxHi);
      trisTmp = refTris; boxLo = refLo; boxHi = refHi;
      n           = total;
      gridT       = (n + B - 1) / B;
      out.numTris = n;
      BUILD_CHECK(hipMemcpyAsync(bounds, initBounds, sizeof(initBounds), hipMemcpyHostToDevice, stream));
      hipLaunchKernelGGL(k_centroid_bounds, dim3(gridT), dim3(B), 0, stream, n, boxLo, boxHi, bounds);
      BUILD_CHECK(hipGetLastError());
    }
  }

  BUILD_CHECK(hipMalloc(&out.tris, sizeof(DevTri) * n));
  if(n == 1)
  {
    BUILD_CHECK(hipMemcpyAsync(out.tris, trisTmp, sizeof(DevTri), hipMemcpyDeviceToDevice, stream));
    BUILD_CHECK(hipStreamSynchronize(stream));
    out.root     = ~0;  // leaf 0
    out.numNodes = 0;
  }
  else
  {
    BUILD_CHECK(hipMalloc(&keysA, sizeof(uint64_t) * n));
    BUILD_CHECK(hipMalloc(&keysB, sizeof(uint64_t) * n));
    BUILD_CHECK(hipMalloc(&valsA, sizeof(uint32_t) * n));
    BUILD_CHECK(hipMalloc(&valsB, sizeof(uint32_t) * n));
    hipLaunchKernelGGL(k_morton, dim3(gridT), dim3(B), 0, stream, n, boxLo, boxHi, bounds, keysA, valsA);
    BUILD_CHECK(hipGetLastError());
    BUILD_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, sortBytes, keysA, keysB, valsA, valsB, int(n), 0, 63, stream));
    BUILD_CHECK(hipMalloc(&sortTemp, sortBytes));
    BUILD_CHECK(hipcub::DeviceRadixSort::SortPairs(sortTemp, sortBytes, keysA, keysB, valsA, valsB, int(n), 0, 63, stream));

    BUILD_CHECK(hipMalloc(&out.nodes, sizeof(float4) * 4 * (n - 1)));
    if(!in.karrasTopology)
    {
      BUILD_CHECK(hipMalloc(&cidA, sizeof(int) * n)); BUILD_CHECK(hipMalloc(&cidB, sizeof(int) * n)); BUILD_CHECK(hipMalloc(&nn, sizeof(int) * n));
      BUILD_CHECK(hipMalloc(&cloA, sizeof(float4) * n)); BUILD_CHECK(hipMalloc(&chiA, sizeof(float4) * n));
      BUILD_CHECK(hipMalloc(&cloB, sizeof(float4) * n)); BUILD_CHECK(hipMalloc(&chiB, sizeof(float4) * n));
      BUILD_CHECK(hipMalloc(&flags, sizeof(unsigned long long) * n)); BUILD_CHECK(hipMalloc(&pos, sizeof(unsigned long long) * n));
      BUILD_CHECK(hipMalloc(&totals, sizeof(unsigned long long)));
      BUILD_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, scanBytes, flags, pos, int(n), stream));
      BUILD_CHECK(hipMalloc(&scanTemp, scanBytes));
      hipLaunchKernelGGL(k_ploc_init, dim3(gridT), dim3(B), 0, stream, int(n), valsB, boxLo, boxHi, cidA, cloA, chiA);
      BUILD_CHECK(hipGetLastError());
      int m = int(n), nodeBase = 0, rootRef = BVH_EMPTY;
      while(m > 1 && ok)
      {
        const unsigned g = unsigned(m + B - 1) / B;
        hipLaunchKernelGGL(k_ploc_nn, dim3(g), dim3(256), 0, stream, m, cloA, chiA, nn);
        hipLaunchKernelGGL(k_ploc_flags, dim3(g), dim3(B), 0, stream, m, nn, flags);
        BUILD_CHECK(hipcub::DeviceScan::ExclusiveSum(scanTemp, scanBytes, flags, pos, m, stream));
        hipLaunchKernelGGL(k_ploc_emit, dim3(g), dim3(B), 0, stream, m, nn, flags, pos, cidA, cloA, chiA, cidB, cloB, chiB, nodeBase, out.nodes, totals);
        BUILD_CHECK(hipGetLastError());
        unsigned long long t = 0;
        BUILD_CHECK(hipMemcpyAsync(&t, totals, sizeof(t), hipMemcpyDeviceToHost, stream));
        BUILD_CHECK(hipStreamSynchronize(stream));
        const int survivors = int(uint32_t(t)), merges = int(t >> 32);
        if(merges < 1 || survivors != m - merges)
        {
          err = "PLOC round made no progress";
          ok  = false;
          break;
        }
        nodeBase += merges;
        m = survivors;
        std::swap(cidA, cidB); std::swap(cloA, cloB); std::swap(chiA, chiB);
      }
      if(!ok)
        break;
      BUILD_CHECK(hipMemcpy(&rootRef, cidA, sizeof(int), hipMemcpyDeviceToHost));
      if(nodeBase != int(n) - 1 || rootRef != int(n) - 2)
      {
        err = "PLOC produced an inconsistent tree";
        ok  = false;
        break;
      }
      hipLaunchKernelGGL(k_emit_tris, dim3(gridT), dim3(B), 0, stream, int(n), valsB, trisTmp, out.tris);
      BUILD_CHECK(hipGetLastError());
      BUILD_CHECK(hipStreamSynchronize(stream));
      out.root     = rootRef;
      out.numNodes = n - 1;
    }
    else
    {
    BUILD_CHECK(hipMalloc(&children, sizeof(int2) * (n - 1)));
    BUILD_CHECK(hipMalloc(&parentInternal, sizeof(int) * (n - 1)));
    BUILD_CHECK(hipMalloc(&parentLeaf, sizeof(int) * n));
    BUILD_CHECK(hipMalloc(&nodeLo, sizeof(float4) * (n - 1)));
    BUILD_CHECK(hipMalloc(&nodeHi, sizeof(float4) * (n - 1)));
    BUILD_CHECK(hipMalloc(&arrive, sizeof(unsigned) * (n - 1)));
    BUILD_CHECK(hipMalloc(&nodeCnt, sizeof(int) * (n - 1)));
    BUILD_CHECK(hipMemsetAsync(arrive, 0, sizeof(unsigned) * (n - 1), stream));
    hipLaunchKernelGGL(k_hierarchy, dim3((n - 1 + B - 1) / B), dim3(B), 0, stream, int(n), keysB, children, parentInternal, parentLeaf);
    BUILD_CHECK(hipGetLastError());
    hipLaunchKernelGGL(k_fit, dim3(gridT), dim3(B), 0, stream, int(n), valsB, boxLo, boxHi, children, parentInternal, parentLeaf, nodeLo, nodeHi, arrive, nodeCnt);
    BUILD_CHECK(hipGetLastError());
    hipLaunchKernelGGL(k_emit_nodes, dim3((n - 1 + B - 1) / B), dim3(B), 0, stream, int(n), valsB, boxLo, boxHi, children, nodeLo, nodeHi, nodeCnt, out.nodes);
    BUILD_CHECK(hipGetLastError());
    hipLaunchKernelGGL(k_emit_tris, dim3(gridT), dim3(B), 0, stream, int(n), valsB, trisTmp, out.tris);
    BUILD_CHECK(hipGetLastError());
    BUILD_CHECK(hipStreamSynchronize(stream));
    out.root     = 0;
    out.numNodes = n - 1;
    }
  }
  if(out.numNodes >= 3 && in.reinsertPasses > 0 && !reinsertBvh2(out.nodes, int(out.numNodes), out.root, in.reinsertPasses, in.reinsertRounds, stream, err, &out.reinsertMoves))
  {
    ok = false;
    break;
  }
  BUILD_CHECK(hipMemcpy(hb, bounds, sizeof(hb), hipMemcpyDeviceToHost));
  } while(0);
  if(ok)
  {
    for(int c = 0; c < 3; ++c)
    {
      auto toF = [](uint32_t u) {
        uint32_t v = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
        float    f;
        memcpy(&f, &v, 4);
        return f;
      };
      out.centroidLo[c] = toF(hb[c]);
      out.centroidHi[c] = toF(hb[3 + c]);
    }
  }
  if(!ok)
  {
    if(out.nodes) (void)hipFree(out.nodes);
    if(out.tris) (void)hipFree(out.tris);
    out.nodes = nullptr;
    out.tris  = nullptr;
  }
  {
    (void)hipFree(trisTmp); (void)hipFree(boxLo); (void)hipFree(boxHi); (void)hipFree(nodeLo); (void)hipFree(nodeHi);
    (void)hipFree(bounds); (void)hipFree(valsA); (void)hipFree(valsB); (void)hipFree(keysA); (void)hipFree(keysB);
    (void)hipFree(children); (void)hipFree(parentInternal); (void)hipFree(parentLeaf); (void)hipFree(arrive); (void)hipFree(nodeCnt); (void)hipFree(sortTemp);
    (void)hipFree(cidA); (void)hipFree(cidB); (void)hipFree(nn); (void)hipFree(cloA); (void)hipFree(chiA); (void)hipFree(cloB); (void)hipFree(chiB);
    (void)hipFree(flags); (void)hipFree(pos); (void)hipFree(totals); (void)hipFree(scanTemp);
    (void)hipFree(splitArea); (void)hipFree(splitSum); (void)hipFree(splitCounts); (void)hipFree(splitOffsets); (void)hipFree(splitTemp);
    return ok;
  }
}

}  // namespace pt
