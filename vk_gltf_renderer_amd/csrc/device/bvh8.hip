// BVH2 -> 8-wide compressed BVH (80-byte nodes) for incoherent traversal on gfx950.
//
// Why: with one ray per lane a BVH2 visit is 4 divergent dwordx4 loads (64 distinct cache lines per wave instruction);
// the profile shows the L1/TA request rate, not ALU, bounds the walk.  An 8-wide node with 8-bit quantised child boxes
// (layout after Ylitie, Karras, Laine 2017, "Efficient Incoherent Ray Traversal on GPUs Through Compressed Wide BVHs")
// needs ~3.5x fewer node visits per ray at 80 B per visit.
//
// Node (5 x uint4):
//   [0] p.x p.y p.z (float bits) | ex | ey<<8 | ez<<16 | imask<<24      quantisation frame: lo corner + per-axis 2^e
//   [1] childBase | triBase | valid16 | 0             valid16: bit 2s = leaf child in slot s has a triangle, bit 2s + 1 = it has two
//       (a leaf child's triangles follow those of the leaf children in lower slots: index = triBase + popcount(valid16 below its bit);
//        an inner child is a bit of imask, an empty slot is neither)
//   [2] qlo.x[0..7] qlo.y[0..7]   [3] qlo.z[0..7] qhi.x[0..7]   [4] qhi.y[0..7] qhi.z[0..7]
// Child boxes decode as fmaf(q, 2^e, p) — the builder checks with the same fmaf that every decoded box CONTAINS the true
// box, so the structure is conservative and the image cannot depend on it (parity contract).
// Internal children of a node are stored contiguously from childBase in slot order; the triangles of its leaf children
// are stored contiguously from triBase in slot order (<= 2 per child, <= 16 per node: the walk turns its 8-bit child hit mask into
// triangle bits with one bit-doubling spread and one AND, pt_bvh8.h).  Children sit in octant-ordered slots so that
// `slot ^ (7 ^ rayOctant)` is a front-to-back priority.
//
// The collapse runs ON THE DEVICE, level by level over the BVH2 the builder left in HBM (no download): one thread per 8-wide node of
// the level picks its (up to eight) children -- by default the SAH-optimal choice of Ylitie et al. 2017, read from tables a bottom-up
// dynamic programme over the BVH2 leaves behind (k_dp_solve); MI_PT_COLLAPSE=greedy opens BVH2 children by surface area instead --
// orders them by octant, quantises their boxes and writes the 80-byte record; a prefix sum over the level hands every node the indices of its inner children (= the next level's
// work list, breadth-first order) and the positions of its leaf triangles.  Two kernels and one scan per level, ~10 levels.  The
// single-threaded host collapse it replaces (0.6 s of a 0.86 s scene build at 2.6 M triangles in round 1) is kept behind
// MI_PT_HOST_COLLAPSE=1 as the A/B reference: both emit the same node array.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#include "pt_build.h"
#include "pt_ticket.h"
#include "pt_bvh.h"

namespace pt {

namespace {

struct Node8  // 80 bytes, see header
{
  float    p[3];
  uint8_t  e[3];
  uint8_t  imask;
  uint32_t childBase;
  uint32_t triBase;
  uint16_t valid;      // two bits per slot: triangles of its leaf child (see header)
  uint16_t reserved16;
  uint32_t reserved32;
  uint8_t  qlo[3][8];
  uint8_t  qhi[3][8];
};
static_assert(sizeof(Node8) == 80, "Node8 must be 80 bytes");

// Largest leaf child: 2 triangles.  Two bits of the node's 16-bit valid mask belong to
// a slot, so a leaf child holds one or two triangles -- which is also what measured best when the mask had room for three and four
// (round 2: 2 beats 1, 3 and 4 on the helmet, atrium and street workloads, +0.4 .. +2.5 % over 3).
// SAH-optimal collapse by default: against the greedy one (MI_PT_COLLAPSE=greedy) 40 % fewer, fuller nodes (atrium 68.8 k -> 44 k),
// node visits per secondary ray 19.74 -> 19.27 (atrium), 28.66 -> 27.55 (street), 8.01 -> 7.91 (helmet), and
// atrium 482 -> 490, street 512 -> 518, helmet 3799 -> 3878, glass 537 -> 551 Msamples/s (round 3)

struct Cand
{
  int   ref;  // >= 0 BVH2 inner node, < 0 leaf (~sorted triangle index)
  float lo[3], hi[3];
};

float areaOf(const Cand& c)
{
  float ex = c.hi[0] - c.lo[0], ey = c.hi[1] - c.lo[1], ez = c.hi[2] - c.lo[2];
  return ex * ey + ey * ez + ez * ex;
}

__global__ void k_reorder_tris(uint32_t n, const uint32_t* perm, const DevTri* in, DevTri* out)
{
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i < n)
    out[i] = in[perm[i]];
}


// ---- device collapse -----------------------------------------------------------------------------------------------------------
struct DCand
{
  int   ref;  // >= 0 BVH2 inner node, < 0 leaf (~triangle index in BVH2 order)
  float lo[3], hi[3];
};
__device__ __forceinline__ int d_childRef(const float4* nodes2, int node, int which)
{
  const float4 n3 = nodes2[size_t(node) * 4 + 3];
  return __float_as_int(which == 0 ? n3.x : n3.y);
}
__device__ __forceinline__ uint32_t d_triCount(const float4* nodes2, int ref) { return ref >= 0 ? uint32_t(__float_as_int(nodes2[size_t(ref) * 4 + 3].z)) : 1u; }
__device__ __forceinline__ void d_childBox(const float4* nodes2, int node, int which, float lo[3], float hi[3])
{
  const float4 n0 = nodes2[size_t(node) * 4 + 0], n1 = nodes2[size_t(node) * 4 + 1], n2 = nodes2[size_t(node) * 4 + 2];
  if(which == 0) { lo[0] = n0.x; hi[0] = n0.y; lo[1] = n0.z; hi[1] = n0.w; lo[2] = n2.x; hi[2] = n2.y; }
  else           { lo[0] = n1.x; hi[0] = n1.y; lo[1] = n1.z; hi[1] = n1.w; lo[2] = n2.z; hi[2] = n2.w; }
}
__device__ __forceinline__ float d_area(const DCand& c)
{
  const float ex = c.hi[0] - c.lo[0], ey = c.hi[1] - c.lo[1], ez = c.hi[2] - c.lo[2];
  return __fadd_rn(__fadd_rn(__fmul_rn(ex, ey), __fmul_rn(ey, ez)), __fmul_rn(ez, ex));  // (fixed association; NB __fmul_rn / __fadd_rn are a plain * and + in this
                                                                                         //  ROCm and hipcc may still fuse them -- bvh_reinsert.h: r2Area, where the bits matter, uses the pragma)
}
// ---- SAH-optimal collapse (Ylitie, Karras, Laine 2017, section 3.2) ------------------------------------------------------------
// cost[n][i], i = 1..7: the cheapest way to represent the BVH2 subtree n as AT MOST i children of some 8-wide node -- each such child
// either a leaf (the subtree has <= maxLeaf triangles; cost = area x C_TRI x triangles) or an inner 8-wide node (cost = area x 1 +
// the cheapest forest of <= 8 children below it).  C_TRI = 56 / 235: a triangle test against a node visit in vector instructions.
// split[n][i]: how many of the i roots go to the left BVH2 child (-1: the subtree stays one child); split[n][1]: 0 = leaf, 1 = inner
// node; split[n][0]: the left share of the inner node's eight.  Filled bottom-up (second arrival at a node solves it, like k_fit),
// read top-down by the level kernels instead of the greedy opening.
#ifndef MI_PT_DP_C_TRI
#define MI_PT_DP_C_TRI (56.0f / 235.0f)  // a triangle test against a node visit, in vector instructions (round 3's counts; see LABNOTES.md section 3 for the round-4 sweep)
#endif
constexpr float DP_C_TRI = MI_PT_DP_C_TRI;
struct DpTables
{
  float*  cost;   // numInner x 8 ([0] unused)
  int8_t* split;  // numInner x 8
};
__device__ __forceinline__ float d_boxArea(const float lo[3], const float hi[3])
{
  const float ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
  return __fadd_rn(__fadd_rn(__fmul_rn(ex, ey), __fmul_rn(ey, ez)), __fmul_rn(ez, ex));
}
__global__ void k_dp_parents(int numInner, const float4* nodes2, int* parent, int* leafParent, int root)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= numInner)
    return;
  for(int k = 0; k < 2; ++k)
  {
    const int c = d_childRef(nodes2, i, k);
    if(c >= 0)
      parent[c] = i;
    else
      leafParent[~c] = i;
  }
  if(i == root)
    parent[i] = -1;
}
__device__ void d_dpSolve(const float4* nodes2, DpTables dp, int nd, uint32_t maxLeaf)
{
  int   ref[2];
  float leafCost[2];  // cost of a child that is a single triangle
  float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for(int k = 0; k < 2; ++k)
  {
    ref[k] = d_childRef(nodes2, nd, k);
    float l[3], h[3];
    d_childBox(nodes2, nd, k, l, h);
    leafCost[k] = d_boxArea(l, h) * DP_C_TRI;
    for(int a = 0; a < 3; ++a)
    {
      lo[a] = fminf(lo[a], l[a]);
      hi[a] = fmaxf(hi[a], h[a]);
    }
  }
  const float area = d_boxArea(lo, hi);
  // The cost tables are what one thread of this kernel writes and another (the solver of the parent, possibly on another XCD) reads: they go through
  // agent-scope relaxed atomic loads and stores (coherent past the per-XCD L2s), ordered against the ticket in k_dp_solve by the stores'
  // acknowledgement -- not through plain stores between two device-scope fences, which cost an L2 write-back + invalidate per thread and level
  // (round 5: the same change as the reinsertion refit, bvh_reinsert.h).  The split tables are read by later kernels only.
  float childCost[2][8];
  for(int k = 0; k < 2; ++k)
    for(int i = 1; i < 8; ++i)
      childCost[k][i] = ref[k] < 0 ? leafCost[k] : __hip_atomic_load(dp.cost + size_t(ref[k]) * 8 + size_t(i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  auto get = [&](int k, int i) { return childCost[k][min(i, 7)]; };
  float   c[8];
  int8_t* s = dp.split + size_t(nd) * 8;
  for(int i = 2; i <= 7; ++i)
  {
    float best = FLT_MAX;
    int   bk   = 1;
    for(int k = 1; k < i; ++k)
    {
      const float v = get(0, k) + get(1, i - k);
      if(v < best)
      {
        best = v;
        bk   = k;
      }
    }
    c[i] = best;
    s[i] = int8_t(bk);
  }
  float inner = FLT_MAX;
  int   bk    = 1;
  for(int k = 1; k < 8; ++k)
  {
    const float v = get(0, k) + get(1, 8 - k);
    if(v < inner)
    {
      inner = v;
      bk    = k;
    }
  }
  inner += area;
  const uint32_t cnt  = d_triCount(nodes2, nd);
  const float    leaf = cnt <= maxLeaf ? area * DP_C_TRI * float(cnt) : FLT_MAX;
  c[0] = 0.0f;
  c[1] = leaf <= inner ? leaf : inner;
  s[1] = leaf <= inner ? 0 : 1;
  s[0] = int8_t(bk);
  for(int i = 2; i <= 7; ++i)
    if(c[1] < c[i])
    {
      c[i] = c[1];
      s[i] = -1;
    }
  for(int i = 0; i < 8; ++i)
    __hip_atomic_store(dp.cost + size_t(nd) * 8 + size_t(i), c[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void k_dp_solve(int numLeaves, const float4* nodes2, const int* parent, const int* leafParent, unsigned* arrive, DpTables dp, uint32_t maxLeaf)
{
  const int leaf = blockIdx.x * blockDim.x + threadIdx.x;
  if(leaf >= numLeaves)
    return;
  int cur = leafParent[leaf];
  while(cur >= 0)
  {
    // (this thread's write-through stores of d_dpSolve are acknowledged before its ticket, the ticket has returned before the tables are read: pt_ticket.h)
    if(pt::ticketArrive(&arrive[cur]) == 0u)
      return;  // first arrival: the sibling subtree finishes this node
    d_dpSolve(nodes2, dp, cur, maxLeaf);
    cur = parent[cur];
  }
}
// is the candidate an INNER child (an 8-wide node of its own) or a leaf child?  greedy: by its triangle count; DP: as the table decided
__device__ __forceinline__ bool d_isInner(const float4* nodes2, const DpTables& dp, int ref, uint32_t maxLeaf)
{
  if(ref < 0)
    return false;
  if(dp.split)
    return dp.split[size_t(ref) * 8 + 1] != 0;
  return d_triCount(nodes2, ref) > maxLeaf;
}
// The children of the 8-wide node rooted at BVH2 node `ref` as the DP tables chose them.
__device__ int d_expandDp(const float4* nodes2, const DpTables& dp, int ref, DCand cands[8])
{
  struct Item
  {
    int ref, parentNode, which, share;
  };
  Item stack[16];
  int  sp = 0, count = 0;
  const int k0 = dp.split[size_t(ref) * 8 + 0];
  stack[sp++] = Item{d_childRef(nodes2, ref, 1), ref, 1, 8 - k0};  // (right first: popped last, so the children come out left to right)
  stack[sp++] = Item{d_childRef(nodes2, ref, 0), ref, 0, k0};
  while(sp > 0)
  {
    const Item it = stack[--sp];
    const int  sh = min(it.share, 7);
    const int  sv = it.ref >= 0 ? int(dp.split[size_t(it.ref) * 8 + size_t(sh)]) : -1;
    if(it.ref < 0 || sh <= 1 || sv < 0)
    {
      if(count < 8)
      {
        cands[count].ref = it.ref;
        d_childBox(nodes2, it.parentNode, it.which, cands[count].lo, cands[count].hi);
        ++count;
      }
      continue;
    }
    if(sp + 2 <= 16)
    {
      stack[sp++] = Item{d_childRef(nodes2, it.ref, 1), it.ref, 1, sh - sv};
      stack[sp++] = Item{d_childRef(nodes2, it.ref, 0), it.ref, 0, sv};
    }
  }
  return count;
}
// The children of the 8-wide node that starts at BVH2 node `ref`: greedy, always open the inner child of largest surface area that
// holds more than maxLeaf triangles, until there are eight (or nothing left to open).  Returns how many, and the leaf size used.
__device__ int d_expand(const float4* nodes2, const DpTables& dp, int ref, uint32_t maxLeafIn, DCand cands[8], uint32_t& maxLeafOut)
{
  if(dp.split)
  {
    maxLeafOut = maxLeafIn;  // (<= 3 in this mode: eight leaf children always fit the node's triangle mask)
    return d_expandDp(nodes2, dp, ref, cands);
  }
  uint32_t maxLeaf = maxLeafIn;
  int      count;
  for(;;)
  {
    count = 2;
    for(int k = 0; k < 2; ++k)
    {
      cands[k].ref = d_childRef(nodes2, ref, k);
      d_childBox(nodes2, ref, k, cands[k].lo, cands[k].hi);
    }
    while(count < 8)
    {
      int   best  = -1;
      float bestA = -1.0f;
      for(int k = 0; k < count; ++k)
        if(cands[k].ref >= 0 && d_triCount(nodes2, cands[k].ref) > maxLeaf)
        {
          const float a = d_area(cands[k]);
          if(a > bestA)
          {
            bestA = a;
            best  = k;
          }
        }
      if(best < 0)
        break;
      const int r = cands[best].ref;
      cands[best].ref = d_childRef(nodes2, r, 0);
      d_childBox(nodes2, r, 0, cands[best].lo, cands[best].hi);
      cands[count].ref = d_childRef(nodes2, r, 1);
      d_childBox(nodes2, r, 1, cands[count].lo, cands[count].hi);
      ++count;
    }
    uint32_t leafTris = 0;
    for(int k = 0; k < count; ++k)
    {
      const uint32_t tc = d_triCount(nodes2, cands[k].ref);
      leafTris += cands[k].ref < 0 ? 1u : (tc <= maxLeaf ? tc : 0u);
    }
    if(leafTris <= 31u || maxLeaf <= 3u)
      break;
    maxLeaf = 3u;  // would not fit the node's triangle mask
  }
  maxLeafOut = maxLeaf;
  return count;
}
// pass 1 of a level: how many inner children (low word) and leaf triangles (high word) each node of the level will have
__global__ void k_collapse_count(int numItems, const int* items, const float4* nodes2, DpTables dp, uint32_t maxLeaf, unsigned long long* counts)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= numItems)
    return;
  DCand    cands[8];
  uint32_t ml;
  const int count = d_expand(nodes2, dp, items[i], maxLeaf, cands, ml);
  uint32_t  inner = 0, tris = 0;
  for(int k = 0; k < count; ++k)
  {
    const uint32_t tc = d_triCount(nodes2, cands[k].ref);
    if(d_isInner(nodes2, dp, cands[k].ref, ml))
      ++inner;
    else
      tris += tc;
  }
  counts[i] = (unsigned long long)inner | ((unsigned long long)tris << 32);
}
// pass 2: the node records, the next level's work list, the triangle permutation
__global__ void k_collapse_emit(int numItems, const int* items, const float4* nodes2, DpTables dp, uint32_t maxLeaf, const unsigned long long* counts,
                                const unsigned long long* offsets, uint32_t levelStart, uint32_t nextLevelStart, uint32_t triLevelBase, Node8* nodes8,
                                int* nextItems, uint32_t* perm, unsigned long long* totals)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= numItems)
    return;
  if(i == numItems - 1)
    *totals = offsets[i] + counts[i];
  DCand    cands[8];
  uint32_t ml;
  const int count = d_expand(nodes2, dp, items[i], maxLeaf, cands, ml);
  // node frame
  float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for(int k = 0; k < count; ++k)
    for(int a = 0; a < 3; ++a)
    {
      lo[a] = fminf(lo[a], cands[k].lo[a]);
      hi[a] = fmaxf(hi[a], cands[k].hi[a]);
    }
  // octant-ordered slot assignment (greedy on dot(centroid - centre, slot diagonal)), ties by (candidate, slot) order
  int  candOfSlot[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
  bool slotUsed[8] = {false, false, false, false, false, false, false, false}, done[8] = {false, false, false, false, false, false, false, false};
  for(int round = 0; round < count; ++round)
  {
    float bestCost = -FLT_MAX;
    int   bc = -1, bs = -1;
    for(int c = 0; c < count; ++c)
    {
      if(done[c])
        continue;
      float d[3];
      for(int a = 0; a < 3; ++a)
        d[a] = __fsub_rn(__fmul_rn(0.5f, __fadd_rn(cands[c].lo[a], cands[c].hi[a])), __fmul_rn(0.5f, __fadd_rn(lo[a], hi[a])));
      for(int sl = 0; sl < 8; ++sl)
      {
        if(slotUsed[sl])
          continue;
        const float cost = __fadd_rn(__fadd_rn((sl & 1) ? d[0] : -d[0], (sl & 2) ? d[1] : -d[1]), (sl & 4) ? d[2] : -d[2]);
        if(cost > bestCost)
        {
          bestCost = cost;
          bc       = c;
          bs       = sl;
        }
      }
    }
    candOfSlot[bs] = bc;
    slotUsed[bs]   = true;
    done[bc]       = true;
  }
  Node8 N;
  memset(&N, 0, sizeof(N));
  int ex[3];
  for(int a = 0; a < 3; ++a)
  {
    N.p[a]          = lo[a];
    const float ext = hi[a] - lo[a];
    ex[a]           = ext > 0.0f ? int(ceil(log2(double(ext) / 255.0))) : -126;
    ex[a]           = max(-126, min(ex[a], 126));
  }
  for(int sl = 0; sl < 8; ++sl)
    for(int a = 0; a < 3; ++a)
    {
      N.qlo[a][sl] = 255;  // empty slot: inverted box (and no valid bit)
      N.qhi[a][sl] = 0;
    }
  for(int a = 0; a < 3; ++a)
  {
    for(;;)
    {
      const float scale = ldexpf(1.0f, ex[a]);
      bool        fits  = true;
      for(int sl = 0; sl < 8 && fits; ++sl)
      {
        if(candOfSlot[sl] < 0)
          continue;
        const DCand& c  = cands[candOfSlot[sl]];
        const double ql = floor((double(c.lo[a]) - double(N.p[a])) / double(scale));
        const double qh = ceil((double(c.hi[a]) - double(N.p[a])) / double(scale));
        int          il = int(fmax(0.0, fmin(255.0, ql))), ih = int(fmax(0.0, fmin(255.0, qh)));
        while(il > 0 && __fmaf_rn(float(il), scale, N.p[a]) > c.lo[a])
          --il;
        while(ih < 255 && __fmaf_rn(float(ih), scale, N.p[a]) < c.hi[a])
          ++ih;
        if(__fmaf_rn(float(il), scale, N.p[a]) > c.lo[a] || __fmaf_rn(float(ih), scale, N.p[a]) < c.hi[a])
          fits = false;
        N.qlo[a][sl] = uint8_t(il);
        N.qhi[a][sl] = uint8_t(ih);
      }
      if(fits || ex[a] >= 126)
        break;
      ++ex[a];
    }
    N.e[a] = uint8_t(ex[a] + 127);
  }
  // children: inner ones get consecutive node indices in slot order (they are the next level's items), leaf ones consecutive triangles
  const unsigned long long off = offsets[i];
  uint32_t                 childAt = nextLevelStart + uint32_t(off), triAt = triLevelBase + uint32_t(off >> 32);
  N.childBase = childAt;
  N.triBase   = triAt;
  for(int sl = 0; sl < 8; ++sl)
  {
    if(candOfSlot[sl] < 0)
      continue;
    const DCand&   c  = cands[candOfSlot[sl]];
    const uint32_t tc = d_triCount(nodes2, c.ref);
    if(d_isInner(nodes2, dp, c.ref, ml))
    {
      N.imask |= uint8_t(1u << sl);
      nextItems[childAt - nextLevelStart] = c.ref;
      ++childAt;
    }
    else
    {
      N.valid |= uint16_t((tc >= 2u ? 3u : 1u) << (2 * sl));  // (tc <= 2: d_isInner opens anything larger)
      // the (at most 4) triangles below c.ref, left to right
      int stack[8], sp = 0;
      stack[sp++] = c.ref;
      while(sp > 0)
      {
        const int r = stack[--sp];
        if(r < 0)
          perm[triAt++] = uint32_t(~r);
        else
        {
          stack[sp++] = d_childRef(nodes2, r, 1);
          stack[sp++] = d_childRef(nodes2, r, 0);
        }
      }
    }
  }
  nodes8[levelStart + uint32_t(i)] = N;
}

}  // namespace

bool buildBvh8(const BvhBuildOutput& b2, Bvh8Output& out, hipStream_t stream, std::string& err, const Bvh8Options& opt)
{
  out                = Bvh8Output();
  const uint32_t n   = b2.numTris;
  if(n == 0)
    return true;
  auto check = [&](hipError_t e, const char* what) {
    if(e != hipSuccess)
    {
      err = std::string(what) + ": " + hipGetErrorString(e);
      return false;
    }
    return true;
  };
  const uint32_t numInner = b2.numNodes;
  const bool hostCollapse = opt.hostCollapse;
  const int  leafTrisOpt  = 2;  // (see the note above Cand)
  if(numInner > 0 && !hostCollapse)
  {
    // ---- device collapse, one level at a time (see the header comment) -------------------------------------------------------
    Node8*              dNodes = nullptr;
    int *               itemsA = nullptr, *itemsB = nullptr;
    uint32_t*           dPerm  = nullptr;
    unsigned long long *counts = nullptr, *offsets = nullptr, *totals = nullptr;
    void*               scanTemp  = nullptr;
    size_t              scanBytes = 0;
    bool                ok        = true;
    uint32_t            numNodes8 = 0, trisPlaced = 0;
    DpTables            dp{nullptr, nullptr};
    int *               dpParent = nullptr, *dpLeafParent = nullptr;
    unsigned*           dpArrive = nullptr;
    do
    {
      // every 8-wide node is rooted at a distinct BVH2 inner node: numInner bounds their number and the length of any level
      if(!(ok = check(hipMalloc(&dNodes, sizeof(Node8) * size_t(numInner)), "alloc BVH8 nodes (worst case)"))) break;
      if(!(ok = check(hipMalloc(&itemsA, sizeof(int) * size_t(numInner)), "alloc level items"))) break;
      if(!(ok = check(hipMalloc(&itemsB, sizeof(int) * size_t(numInner)), "alloc level items"))) break;
      if(!(ok = check(hipMalloc(&dPerm, sizeof(uint32_t) * size_t(n)), "alloc perm"))) break;
      if(!(ok = check(hipMalloc(&counts, sizeof(unsigned long long) * size_t(numInner)), "alloc counts"))) break;
      if(!(ok = check(hipMalloc(&offsets, sizeof(unsigned long long) * size_t(numInner)), "alloc offsets"))) break;
      if(!(ok = check(hipMalloc(&totals, sizeof(unsigned long long)), "alloc totals"))) break;
      if(!(ok = check(hipcub::DeviceScan::ExclusiveSum(nullptr, scanBytes, counts, offsets, int(numInner), stream), "scan size"))) break;
      if(!(ok = check(hipMalloc(&scanTemp, scanBytes), "alloc scan"))) break;
      const int root = b2.root;
      if(!(ok = check(hipMemcpyAsync(itemsA, &root, sizeof(int), hipMemcpyHostToDevice, stream), "seed level 0"))) break;
      uint32_t levelStart = 0, levelCount = 1;
      uint32_t   maxLeaf = uint32_t(leafTrisOpt);
      // which BVH2 subtrees become the children of an 8-wide node (Bvh8Options; the images do not depend on it)
      const bool sahDp   = opt.sahCollapse;
      if(sahDp)
      {
        maxLeaf = std::min(maxLeaf, 3u);
        if(!(ok = check(hipMalloc(&dp.cost, sizeof(float) * 8 * size_t(numInner)), "alloc DP cost"))) break;
        if(!(ok = check(hipMalloc(&dp.split, 8 * size_t(numInner)), "alloc DP split"))) break;
        if(!(ok = check(hipMalloc(&dpParent, sizeof(int) * size_t(numInner)), "alloc DP parents"))) break;
        if(!(ok = check(hipMalloc(&dpLeafParent, sizeof(int) * size_t(n)), "alloc DP leaf parents"))) break;
        if(!(ok = check(hipMalloc(&dpArrive, sizeof(unsigned) * size_t(numInner)), "alloc DP tickets"))) break;
        if(!(ok = check(hipMemsetAsync(dpArrive, 0, sizeof(unsigned) * size_t(numInner), stream), "clear DP tickets"))) break;
        hipLaunchKernelGGL(k_dp_parents, dim3((numInner + 255u) / 256u), dim3(256), 0, stream, int(numInner), b2.nodes, dpParent, dpLeafParent, root);
        hipLaunchKernelGGL(k_dp_solve, dim3((n + 255u) / 256u), dim3(256), 0, stream, int(n), b2.nodes, dpParent, dpLeafParent, dpArrive, dp, maxLeaf);
        if(!(ok = check(hipGetLastError(), "DP kernels"))) break;
      }
      while(levelCount > 0)
      {
        const unsigned g = (levelCount + 127u) / 128u;
        hipLaunchKernelGGL(k_collapse_count, dim3(g), dim3(128), 0, stream, int(levelCount), itemsA, b2.nodes, dp, maxLeaf, counts);
        if(!(ok = check(hipcub::DeviceScan::ExclusiveSum(scanTemp, scanBytes, counts, offsets, int(levelCount), stream), "scan"))) break;
        hipLaunchKernelGGL(k_collapse_emit, dim3(g), dim3(128), 0, stream, int(levelCount), itemsA, b2.nodes, dp, maxLeaf, counts, offsets, levelStart,
                           levelStart + levelCount, trisPlaced, dNodes, itemsB, dPerm, totals);
        unsigned long long t = 0;
        if(!(ok = check(hipGetLastError(), "collapse kernels"))) break;
        if(!(ok = check(hipMemcpyAsync(&t, totals, sizeof(t), hipMemcpyDeviceToHost, stream), "read level totals"))) break;
        if(!(ok = check(hipStreamSynchronize(stream), "sync level"))) break;
        levelStart += levelCount;
        levelCount = uint32_t(t);
        trisPlaced += uint32_t(t >> 32);
        if(levelStart + levelCount > numInner || trisPlaced > n)
        {
          err = "BVH8 collapse overran its bounds";
          ok  = false;
          break;
        }
        std::swap(itemsA, itemsB);
      }
      if(!ok)
        break;
      numNodes8 = levelStart;
      if(trisPlaced != n)
      {
        err = "BVH8 collapse lost triangles";
        ok  = false;
        break;
      }
      if(!(ok = check(hipMalloc(reinterpret_cast<void**>(&out.nodes), sizeof(Node8) * size_t(numNodes8)), "alloc BVH8 nodes"))) break;
      if(!(ok = check(hipMemcpyAsync(out.nodes, dNodes, sizeof(Node8) * size_t(numNodes8), hipMemcpyDeviceToDevice, stream), "copy BVH8 nodes"))) break;
      if(!(ok = check(hipMalloc(&out.tris, sizeof(DevTri) * size_t(n)), "alloc BVH8 triangles"))) break;
      hipLaunchKernelGGL(k_reorder_tris, dim3((n + 255) / 256), dim3(256), 0, stream, n, dPerm, b2.tris, out.tris);
      ok = check(hipGetLastError(), "k_reorder_tris") && check(hipStreamSynchronize(stream), "sync");
    } while(0);
    (void)hipFree(dNodes); (void)hipFree(itemsA); (void)hipFree(itemsB); (void)hipFree(dPerm); (void)hipFree(counts); (void)hipFree(offsets);
    (void)hipFree(totals); (void)hipFree(scanTemp);
    (void)hipFree(dp.cost); (void)hipFree(dp.split); (void)hipFree(dpParent); (void)hipFree(dpLeafParent); (void)hipFree(dpArrive);
    if(!ok)
    {
      if(out.nodes) (void)hipFree(out.nodes);
      if(out.tris) (void)hipFree(out.tris);
      out = Bvh8Output();
      return false;
    }
    out.numNodes = numNodes8;
    out.numTris  = n;
    return true;
  }
  // ---- host collapse (A/B reference, and the one-triangle scene): download the BVH2 -------------------------------------------

  std::vector<float4> nodes2(size_t(numInner) * 4);
  if(numInner && !check(hipMemcpy(nodes2.data(), b2.nodes, nodes2.size() * sizeof(float4), hipMemcpyDeviceToHost), "download BVH2"))
    return false;
  std::vector<DevTri> tris2;
  if(numInner == 0)  // single triangle: need its bounds
  {
    tris2.resize(n);
    if(!check(hipMemcpy(tris2.data(), b2.tris, sizeof(DevTri) * n, hipMemcpyDeviceToHost), "download triangles"))
      return false;
  }
  auto childRef = [&](int node, int which) {
    float f = which == 0 ? nodes2[size_t(node) * 4 + 3].x : nodes2[size_t(node) * 4 + 3].y;
    int   r;
    memcpy(&r, &f, 4);
    return r;
  };
  auto childBox = [&](int node, int which, float lo[3], float hi[3]) {
    const float4 &n0 = nodes2[size_t(node) * 4 + 0], &n1 = nodes2[size_t(node) * 4 + 1], &n2 = nodes2[size_t(node) * 4 + 2];
    if(which == 0)
    {
      lo[0] = n0.x; hi[0] = n0.y; lo[1] = n0.z; hi[1] = n0.w; lo[2] = n2.x; hi[2] = n2.y;
    }
    else
    {
      lo[0] = n1.x; hi[0] = n1.y; lo[1] = n1.z; hi[1] = n1.w; lo[2] = n2.z; hi[2] = n2.w;
    }
  };
  // ---- subtree triangle counts ---------------------------------------------------------------------------------------------
  std::vector<uint32_t> cnt(numInner, 0);
  if(numInner)
  {
    std::vector<int> order;  // pre-order; children have larger positions than parents
    order.reserve(numInner);
    std::vector<int> stack{b2.root};
    while(!stack.empty())
    {
      int v = stack.back();
      stack.pop_back();
      order.push_back(v);
      for(int w = 0; w < 2; ++w)
      {
        int c = childRef(v, w);
        if(c >= 0)
          stack.push_back(c);
      }
    }
    for(size_t k = order.size(); k-- > 0;)
    {
      int      v = order[k];
      uint32_t c = 0;
      for(int w = 0; w < 2; ++w)
      {
        int r = childRef(v, w);
        c += r >= 0 ? cnt[size_t(r)] : 1u;
      }
      cnt[size_t(v)] = c;
    }
  }
  auto triCount = [&](int ref) { return ref >= 0 ? cnt[size_t(ref)] : 1u; };
  // triangles below a (small) subtree, left to right
  std::function<void(int, std::vector<uint32_t>&)> collectTris = [&](int ref, std::vector<uint32_t>& dst) {
    if(ref < 0)
    {
      dst.push_back(uint32_t(~ref));
      return;
    }
    collectTris(childRef(ref, 0), dst);
    collectTris(childRef(ref, 1), dst);
  };

  // ---- breadth-first collapse ----------------------------------------------------------------------------------------
  std::vector<Node8>    nodes8;
  std::vector<uint32_t> perm;  // new triangle index -> sorted (BVH2) triangle index
  perm.reserve(n);
  struct Work
  {
    uint32_t          node8;
    std::vector<Cand> cands;  // the (<= 2 at start) children to expand
  };
  std::vector<Work> queue;
  {
    Work w;
    w.node8 = 0;
    if(numInner == 0)
    {
      Cand c;
      c.ref = ~0;
      const DevTri& T = tris2[0];
      float v[3][3] = {{T.a.x, T.a.y, T.a.z}, {T.a.x + T.b.x, T.a.y + T.b.y, T.a.z + T.b.z}, {T.a.x + T.c.x, T.a.y + T.c.y, T.a.z + T.c.z}};
      for(int a = 0; a < 3; ++a)
      {
        c.lo[a] = std::min(v[0][a], std::min(v[1][a], v[2][a]));
        c.hi[a] = std::max(v[0][a], std::max(v[1][a], v[2][a]));
        // one ulp of slack each way: the BVH2 path bounds the true vertices too (k_tri_setup)
        c.lo[a] = std::nextafter(c.lo[a], -FLT_MAX);
        c.hi[a] = std::nextafter(c.hi[a], FLT_MAX);
      }
      w.cands.push_back(c);
    }
    else
      for(int k = 0; k < 2; ++k)
      {
        Cand c;
        c.ref = childRef(b2.root, k);
        childBox(b2.root, k, c.lo, c.hi);
        w.cands.push_back(c);
      }
    queue.push_back(std::move(w));
    nodes8.emplace_back();
  }
  for(size_t qi = 0; qi < queue.size(); ++qi)
  {
    const std::vector<Cand> cands0 = std::move(queue[qi].cands);
    std::vector<Cand>       cands;
    const uint32_t          self = queue[qi].node8;
    uint32_t                MAX_LEAF_TRIS = uint32_t(leafTrisOpt);
    for(;;)
    {
      cands = cands0;
      // greedy: open the largest inner child that is too big to be a leaf until 8 children are reached
      while(cands.size() < 8)
      {
        int   best = -1;
        float bestA = -1.0f;
        for(size_t k = 0; k < cands.size(); ++k)
          if(cands[k].ref >= 0 && triCount(cands[k].ref) > MAX_LEAF_TRIS && areaOf(cands[k]) > bestA)
          {
            bestA = areaOf(cands[k]);
            best  = int(k);
          }
        if(best < 0)
          break;
        int  ref = cands[size_t(best)].ref;
        Cand a, b;
        a.ref = childRef(ref, 0);
        b.ref = childRef(ref, 1);
        childBox(ref, 0, a.lo, a.hi);
        childBox(ref, 1, b.lo, b.hi);
        cands[size_t(best)] = a;
        cands.push_back(b);
      }
      uint32_t leafTris = 0;
      for(const Cand& c : cands)
        leafTris += c.ref < 0 ? 1u : (triCount(c.ref) <= MAX_LEAF_TRIS ? triCount(c.ref) : 0u);
      if(leafTris <= 31u || MAX_LEAF_TRIS <= 3u)
        break;
      MAX_LEAF_TRIS = 3u;  // would not fit the node's triangle mask
    }
    // node frame
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for(const Cand& c : cands)
      for(int a = 0; a < 3; ++a)
      {
        lo[a] = std::min(lo[a], c.lo[a]);
        hi[a] = std::max(hi[a], c.hi[a]);
      }
    // octant-ordered slot assignment (greedy on dot(centroid - centre, slot diagonal))
    int  slotOf[8];
    bool slotUsed[8] = {false, false, false, false, false, false, false, false}, done[8] = {false, false, false, false, false, false, false, false};
    for(size_t round = 0; round < cands.size(); ++round)
    {
      float bestCost = -FLT_MAX;
      int   bc = -1, bs = -1;
      for(size_t c = 0; c < cands.size(); ++c)
      {
        if(done[c])
          continue;
        float d[3];
        for(int a = 0; a < 3; ++a)
          d[a] = 0.5f * (cands[c].lo[a] + cands[c].hi[a]) - 0.5f * (lo[a] + hi[a]);
        for(int s = 0; s < 8; ++s)
        {
          if(slotUsed[s])
            continue;
          float cost = ((s & 1) ? d[0] : -d[0]) + ((s & 2) ? d[1] : -d[1]) + ((s & 4) ? d[2] : -d[2]);
          if(cost > bestCost)
          {
            bestCost = cost;
            bc       = int(c);
            bs       = s;
          }
        }
      }
      slotOf[bc]   = bs;
      slotUsed[bs] = true;
      done[bc]     = true;
    }
    int candOfSlot[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
    for(size_t c = 0; c < cands.size(); ++c)
      candOfSlot[slotOf[c]] = int(c);

    Node8 N;
    memset(&N, 0, sizeof(N));
    // quantisation frame; grow an exponent until every child fits conservatively
    int ex[3];
    for(int a = 0; a < 3; ++a)
    {
      N.p[a]      = lo[a];
      float ext   = hi[a] - lo[a];
      ex[a]       = ext > 0.0f ? int(std::ceil(std::log2(double(ext) / 255.0))) : -126;
      ex[a]       = std::max(-126, std::min(ex[a], 126));
    }
    for(int s = 0; s < 8; ++s)
      for(int a = 0; a < 3; ++a)
      {
        N.qlo[a][s] = 255;  // empty slot: inverted box (and no valid bit)
        N.qhi[a][s] = 0;
      }
    for(int a = 0; a < 3; ++a)
    {
      for(;;)
      {
        const float scale = std::ldexp(1.0f, ex[a]);
        bool        fits  = true;
        for(int s = 0; s < 8 && fits; ++s)
        {
          if(candOfSlot[s] < 0)
            continue;
          const Cand& c  = cands[size_t(candOfSlot[s])];
          double      ql = std::floor((double(c.lo[a]) - double(N.p[a])) / double(scale));
          double      qh = std::ceil((double(c.hi[a]) - double(N.p[a])) / double(scale));
          int         il = int(std::max(0.0, std::min(255.0, ql))), ih = int(std::max(0.0, std::min(255.0, qh)));
          while(il > 0 && std::fmaf(float(il), scale, N.p[a]) > c.lo[a])
            --il;
          while(ih < 255 && std::fmaf(float(ih), scale, N.p[a]) < c.hi[a])
            ++ih;
          if(std::fmaf(float(il), scale, N.p[a]) > c.lo[a] || std::fmaf(float(ih), scale, N.p[a]) < c.hi[a])
            fits = false;
          N.qlo[a][s] = uint8_t(il);
          N.qhi[a][s] = uint8_t(ih);
        }
        if(fits || ex[a] >= 126)
          break;
        ++ex[a];
      }
      N.e[a] = uint8_t(ex[a] + 127);
    }
    // children: inner ones get consecutive node indices in slot order, leaf ones consecutive triangles
    N.childBase = uint32_t(nodes8.size());
    N.triBase   = uint32_t(perm.size());
    for(int s = 0; s < 8; ++s)
    {
      if(candOfSlot[s] < 0)
        continue;
      const Cand& c = cands[size_t(candOfSlot[s])];
      if(c.ref >= 0 && triCount(c.ref) > MAX_LEAF_TRIS)
      {
        N.imask |= uint8_t(1u << s);
        Work w;
        w.node8 = uint32_t(nodes8.size());
        for(int k = 0; k < 2; ++k)
        {
          Cand cc;
          cc.ref = childRef(c.ref, k);
          childBox(c.ref, k, cc.lo, cc.hi);
          w.cands.push_back(cc);
        }
        nodes8.emplace_back();
        queue.push_back(std::move(w));
      }
      else
      {
        const uint32_t count = triCount(c.ref);  // <= MAX_LEAF_TRIS <= 2
        N.valid |= uint16_t((count >= 2u ? 3u : 1u) << (2 * s));
        collectTris(c.ref, perm);
      }
    }
    nodes8[self] = N;
  }
  if(perm.size() != n)
  {
    err = "BVH8 collapse lost triangles";
    return false;
  }
  // ---- upload ----------------------------------------------------------------------------------------------------------
  uint32_t* dPerm = nullptr;
  bool      ok    = check(hipMalloc(&out.nodes, nodes8.size() * sizeof(Node8)), "alloc BVH8 nodes")
            && check(hipMemcpy(out.nodes, nodes8.data(), nodes8.size() * sizeof(Node8), hipMemcpyHostToDevice), "upload BVH8 nodes")
            && check(hipMalloc(&out.tris, sizeof(DevTri) * n), "alloc BVH8 triangles") && check(hipMalloc(&dPerm, sizeof(uint32_t) * n), "alloc perm")
            && check(hipMemcpy(dPerm, perm.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice), "upload perm");
  if(ok)
  {
    hipLaunchKernelGGL(k_reorder_tris, dim3((n + 255) / 256), dim3(256), 0, stream, n, dPerm, b2.tris, out.tris);
    ok = check(hipGetLastError(), "k_reorder_tris") && check(hipStreamSynchronize(stream), "sync");
  }
  if(dPerm)
    (void)hipFree(dPerm);
  if(!ok)
  {
    if(out.nodes) (void)hipFree(out.nodes);
    if(out.tris) (void)hipFree(out.tris);
    out = Bvh8Output();
    return false;
  }
  out.numNodes = uint32_t(nodes8.size());
  out.numTris  = n;
  return true;
}

}  // namespace pt
