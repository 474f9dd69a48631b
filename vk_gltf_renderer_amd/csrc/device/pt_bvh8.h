// Traversal of the 8-wide compressed BVH (layout: bvh8.hip).  One call = one node visit: decode 8 quantised child boxes,
// slab-test them, and return (a) the group of hit INNER children as a front-to-back priority mask and (b) the mask of
// triangles of hit LEAF children.  Traversal state per lane = current node group + an LDS stack of node groups.
#pragma once
#include "pt_bvh.h"

namespace pt {

constexpr int BVH8_STACK_LDS  = 12;  // node groups per lane in LDS (2 dwords each)
constexpr int BVH8_STACK_PRIV = 52;  // overflow (scratch)

struct NodeGroup
{
  uint32_t base;  // index of the first inner child of the visited node
  uint32_t bits;  // hits (8 bits, priority space: bit p = slot ^ octinv) << 8 | imask
};

struct LaneStack2
{
  int*     lds;  // 2 * BVH8_STACK_LDS * stride ints
  int      tid, stride;
  uint32_t privBase[BVH8_STACK_PRIV], privBits[BVH8_STACK_PRIV];
  int      sp;
  PT_DEV void push(NodeGroup g)
  {
    if(sp < BVH8_STACK_LDS)
    {
      lds[(2 * sp) * stride + tid]     = int(g.base);
      lds[(2 * sp + 1) * stride + tid] = int(g.bits);
    }
    else if(sp - BVH8_STACK_LDS < BVH8_STACK_PRIV)
    {
      privBase[sp - BVH8_STACK_LDS] = g.base;
      privBits[sp - BVH8_STACK_LDS] = g.bits;
    }
    ++sp;
  }
  PT_DEV NodeGroup pop()
  {
    --sp;
    NodeGroup g;
    if(sp < BVH8_STACK_LDS)
    {
      g.base = uint32_t(lds[(2 * sp) * stride + tid]);
      g.bits = uint32_t(lds[(2 * sp + 1) * stride + tid]);
    }
    else if(sp - BVH8_STACK_LDS < BVH8_STACK_PRIV)
    {
      g.base = privBase[sp - BVH8_STACK_LDS];
      g.bits = privBits[sp - BVH8_STACK_LDS];
    }
    else
      g.base = g.bits = 0;
    return g;
  }
};

PT_DEV uint32_t rayOctInv(f3 dir)
{
  uint32_t oct = (dir.x < 0.0f ? 1u : 0u) | (dir.y < 0.0f ? 2u : 0u) | (dir.z < 0.0f ? 4u : 0u);
  return 7u ^ oct;
}
// The root is "a group with one inner child in slot 0".
PT_DEV NodeGroup rootGroup(uint32_t octinv)
{
  NodeGroup g;
  g.base = 0;
  g.bits = ((1u << octinv) << 8) | 1u;
  return g;
}
// Removes the nearest pending child from the group and returns its node index.
PT_DEV uint32_t groupPopChild(NodeGroup& g, uint32_t octinv)
{
  const uint32_t hits  = g.bits >> 8;
  const uint32_t p     = 31u - uint32_t(__clz(int(hits)));
  const uint32_t slot  = p ^ octinv;
  const uint32_t imask = g.bits & 0xffu;
  g.bits &= ~(0x100u << p);
  return g.base + uint32_t(__popc(imask & ((1u << slot) - 1u)));
}

PT_DEV float byteF(uint32_t w, int i) { return float((w >> (8 * i)) & 0xffu); }  // v_cvt_f32_ubyteN

// One node visit.  Boxes decode as fmaf(q, 2^e, p) (identical to the builder's containment check) and are slab-tested
// exactly like the BVH2 boxes, widened by a few ulps.
PT_DEV void bvh8Visit(const DevScene& sc, const RaySetup& r, float tmax, uint32_t octinv, uint32_t nodeIndex, NodeGroup& outGroup, uint32_t& triBase,
                      uint32_t& triMask)
{
  const uint4* N  = sc.bvh8Nodes + size_t(nodeIndex) * 5;
  const uint4  n0 = N[0], n1 = N[1], n2 = N[2], n3 = N[3], n4 = N[4];
  const float  px = __uint_as_float(n0.x), py = __uint_as_float(n0.y), pz = __uint_as_float(n0.z);
  const float  sx = __uint_as_float((n0.w & 0xffu) << 23), sy = __uint_as_float(((n0.w >> 8) & 0xffu) << 23), sz = __uint_as_float(((n0.w >> 16) & 0xffu) << 23);
  const uint32_t imask = n0.w >> 24;
  // near / far byte planes per axis, chosen by the ray's direction sign
  const bool nx = r.dir.x < 0.0f, ny = r.dir.y < 0.0f, nz = r.dir.z < 0.0f;
  const uint32_t qlx[2] = {n2.x, n2.y}, qly[2] = {n2.z, n2.w}, qlz[2] = {n3.x, n3.y}, qhx[2] = {n3.z, n3.w}, qhy[2] = {n4.x, n4.y}, qhz[2] = {n4.z, n4.w};
  const uint32_t meta[2] = {n1.z, n1.w};
  uint32_t       hits = 0, tmask = 0;
#pragma unroll
  for(int i = 0; i < 8; ++i)
  {
    const int      w = i >> 2, b = i & 3;
    const uint32_t m = (meta[w] >> (8 * b)) & 0xffu;
    const float lox = __fmaf_rn(byteF(qlx[w], b), sx, px), hix = __fmaf_rn(byteF(qhx[w], b), sx, px);
    const float loy = __fmaf_rn(byteF(qly[w], b), sy, py), hiy = __fmaf_rn(byteF(qhy[w], b), sy, py);
    const float loz = __fmaf_rn(byteF(qlz[w], b), sz, pz), hiz = __fmaf_rn(byteF(qhz[w], b), sz, pz);
    const float tx0 = ((nx ? hix : lox) - r.org.x) * r.idir.x, tx1 = ((nx ? lox : hix) - r.org.x) * r.idir.x;
    const float ty0 = ((ny ? hiy : loy) - r.org.y) * r.idir.y, ty1 = ((ny ? loy : hiy) - r.org.y) * r.idir.y;
    const float tz0 = ((nz ? hiz : loz) - r.org.z) * r.idir.z, tz1 = ((nz ? loz : hiz) - r.org.z) * r.idir.z;
    const float tn  = fmaxf(fmaxf(tx0, ty0), fmaxf(tz0, 0.0f));
    const float tf  = fminf(fminf(tx1, ty1), fminf(tz1, tmax));
    const bool  hit = (m != 0u) && (tn <= tf * 1.0000012f + 1e-30f);
    if(hit)
    {
      if((imask >> i) & 1u)
        hits |= 1u << (uint32_t(i) ^ octinv);
      else
        tmask |= ((1u << (m >> 5)) - 1u) << (m & 31u);
    }
  }
  outGroup.base = n1.x;
  outGroup.bits = (hits << 8) | imask;
  triBase       = n1.y;
  triMask       = tmask;
}

}  // namespace pt
