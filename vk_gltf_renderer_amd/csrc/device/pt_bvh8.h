// Traversal of the 8-wide compressed BVH (layout: bvh8.hip).  One call = one node visit: decode 8 quantised child boxes,
// slab-test them, and return (a) the group of hit INNER children as a front-to-back priority mask and (b) the triangles of hit
// LEAF children as a "leaf word" (below).  Traversal state per lane = current node group + an LDS stack of node groups.
#pragma once
#include "pt_bvh.h"

namespace pt {

constexpr int BVH8_STACK_LDS  = MI_BVH8_STACK_LDS;  // node groups per lane in LDS (2 dwords each): 12
constexpr int BVH8_STACK_PRIV = 64 - MI_BVH8_STACK_LDS;  // overflow (scratch): 52

struct NodeGroup
{
  uint32_t base;  // index of the first inner child of the visited node
  uint32_t bits;  // hits (8 bits, priority space: bit p = slot ^ octinv) << 8 | imask
};

struct LaneStack2
{
  int*     lds;  // 2 * BVH8_STACK_LDS * stride ints
  int      tid, stride;
  uint32_t *privBase, *privBits;  // BVH8_STACK_PRIV words each, in a separate local array of the caller (see LaneStack::priv)
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

PT_DEV uint32_t rayOctInv(f3 idir)  // pass RaySetup::idir (sign well defined for -0.0 components)
{
  uint32_t oct = (idir.x < 0.0f ? 1u : 0u) | (idir.y < 0.0f ? 2u : 0u) | (idir.z < 0.0f ? 4u : 0u);
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

// Child masks live in PRIORITY space on the stack (bit p = slot ^ octinv: the highest set bit is the nearest child): a node visit turns
// its 8-bit hit mask from slot space into that order -- a butterfly on the three index bits, 16 vector instructions -- or, in the
// per-lane walks, looks it up: the 8 x 256 results as bytes in the workgroup's LDS (fillOctLut), one address add and one ds_read_u8.
PT_DEV uint32_t octPermute(uint32_t hits, uint32_t octinv)
{
  hits = (octinv & 1u) ? (((hits & 0x55u) << 1) | ((hits & 0xaau) >> 1)) : hits;
  hits = (octinv & 2u) ? (((hits & 0x33u) << 2) | ((hits & 0xccu) >> 2)) : hits;
  hits = (octinv & 4u) ? (((hits & 0x0fu) << 4) | ((hits & 0xf0u) >> 4)) : hits;
  return hits;
}
constexpr int OCT_LUT_BYTES = 8 * 256;
#ifndef MI_PT_OCT_LUT
#define MI_PT_OCT_LUT 1  // (A/B: 0 = the butterfly in every node visit)
#endif
// whole block; the caller's next barrier (fillNodeCache ends with one) publishes the table
PT_DEV void fillOctLut(uint8_t* lut)
{
  for(uint32_t i = threadIdx.x; i < uint32_t(OCT_LUT_BYTES); i += blockDim.x)
    lut[i] = uint8_t(octPermute(i & 255u, i >> 8));
}

// ---- leaf word: the triangles a node visit found, in ONE register.  A node owns up to 16 triangle bits, two per child slot
// (bvh8.hip: valid16); the low half of the word is the node's valid16, the high half the PENDING triangles (bit 16 + 2s + k = triangle k
// of the leaf child in slot s, hit and not yet handed to a test).  Triangles are stored compactly in slot order, so the index of a
// pending bit is triBase + (valid bits below it).  "Nothing pending" is `word <= 0xffff` -- a comparison with a constant, like the
// `mask != 0` it replaces -- so the walks' bookkeeping costs what it did when the word was a plain 24-bit mask of a node's triangles,
// and the node visit no longer builds that mask child by child (6 of its 29 vector instructions per child, round 4).
PT_DEV bool     leafPending(uint32_t w) { return w > 0xffffu; }
PT_DEV uint32_t leafCount(uint32_t w) { return uint32_t(__popc(w >> 16)); }
// removes the lowest pending triangle of the word and returns its index
PT_DEV uint32_t leafPop(uint32_t& w, uint32_t triBase)
{
  const uint32_t bit = uint32_t(__ffs(int(w & 0xffff0000u))) - 1u;  // 16 .. 31
  w &= ~(1u << bit);
  return triBase + uint32_t(__popc(w & ((1u << (bit - 16u)) - 1u)));  // (bits below bit - 16 lie in the valid half: untouched by the clear)
}

// Per node and axis: a child plane q (0..255) is crossed at t = q * A + B with A = 2^e / dir and B = (p - org) / dir; the walk wants a
// LOWER bound for the planes a ray enters through and an UPPER bound for those it leaves through.  Bn = t0 - e and Bf = t0 + e with
// t0 = fl(P * idir) and e = 2^-21 (255 |A| + |t0|): the roundings of P = p - org, of t0, of t0 -+ e and of the fma that adds q * A
// (q * A itself is exact) sum to less than 4 x 2^-24 (255 |A| + |t0|), half of e -- conservative without a multiplicative fudge,
// including the cancellation case of an origin inside the node.  The same expression whatever the direction's sign (round 4: before,
// the pad went onto P before the multiplication and was added or subtracted by sign: two selects and a multiply more per axis); the
// individual operations are pinned so that the byte and the float-plane form of the test stay bit-identical.
PT_DEV void slabOffsets(float P, float idir, float A, float& Bn, float& Bf)
{
  const float t0 = __fmul_rn(P, idir);
  const float e  = __fmul_rn(4.76837158e-7f, __fmaf_rn(255.0f, fabsf(A), fabsf(t0)));  // 2^-21 (...)
  Bn             = __fsub_rn(t0, e);
  Bf             = __fadd_rn(t0, e);
}

// Slab test of the 8 children of a loaded node: bit i of `hm` = child slot i is hit (inner or leaf), `leafOut` = the leaf word of the
// hit LEAF children (inner children own no valid bit, so their hits fall out in the AND).
PT_DEV void bvh8TestChildren(const uint4& n0, const uint4& n1, const uint4& n2, const uint4& n3, const uint4& n4, const RaySetup& r, float tmax,
                             uint32_t& hmOut, uint32_t& leafOut)
{
  const float  sx = __uint_as_float((n0.w & 0xffu) << 23), sy = __uint_as_float(((n0.w >> 8) & 0xffu) << 23), sz = __uint_as_float(((n0.w >> 16) & 0xffu) << 23);
  const float  Px = __uint_as_float(n0.x) - r.org.x, Py = __uint_as_float(n0.y) - r.org.y, Pz = __uint_as_float(n0.z) - r.org.z;
  // a negative direction enters through the upper plane (the sign is idir's: it is well defined for -0.0 as well)
  const bool   nx = r.idir.x < 0.0f, ny = r.idir.y < 0.0f, nz = r.idir.z < 0.0f;
  const float  Ax = sx * r.idir.x, Ay = sy * r.idir.y, Az = sz * r.idir.z;
  float        Bnx, Bfx, Bny, Bfy, Bnz, Bfz;
  slabOffsets(Px, r.idir.x, Ax, Bnx, Bfx);
  slabOffsets(Py, r.idir.y, Ay, Bny, Bfy);
  slabOffsets(Pz, r.idir.z, Az, Bnz, Bfz);
  // near / far byte planes per axis (4 children per word), chosen once per node by the direction sign
  const uint32_t qnx[2] = {nx ? n3.z : n2.x, nx ? n3.w : n2.y}, qfx[2] = {nx ? n2.x : n3.z, nx ? n2.y : n3.w};
  const uint32_t qny[2] = {ny ? n4.x : n2.z, ny ? n4.y : n2.w}, qfy[2] = {ny ? n2.z : n4.x, ny ? n2.w : n4.y};
  const uint32_t qnz[2] = {nz ? n4.z : n3.x, nz ? n4.w : n3.y}, qfz[2] = {nz ? n3.x : n4.z, nz ? n3.y : n4.w};
  // The loop is VALU-bound (it is the traversal's inner loop), so it is written for instruction count: the near and far
  // plane of an axis go through one packed fma; a child's miss is the sign of tf - tn, shifted into the mask by one alignbit
  // (empty slots hold inverted boxes and miss by themselves).
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const f32x2 A2x = {Ax, Ax}, A2y = {Ay, Ay}, A2z = {Az, Az}, B2x = {Bnx, Bfx}, B2y = {Bny, Bfy}, B2z = {Bnz, Bfz};
  uint32_t    miss = 0;
#pragma unroll
  for(int j = 0; j < 8; ++j)
  {
    const int   i = 7 - j, w = i >> 2, b = i & 3;  // child 7 first: child 0 ends up in bit 0
    const f32x2 tx = __builtin_elementwise_fma(f32x2{byteF(qnx[w], b), byteF(qfx[w], b)}, A2x, B2x);
    const f32x2 ty = __builtin_elementwise_fma(f32x2{byteF(qny[w], b), byteF(qfy[w], b)}, A2y, B2y);
    const f32x2 tz = __builtin_elementwise_fma(f32x2{byteF(qnz[w], b), byteF(qfz[w], b)}, A2z, B2z);
    const float tn = fmaxf(fmaxf(tx.x, ty.x), fmaxf(tz.x, 0.0f));
    const float tf = fminf(fminf(tx.y, ty.y), fminf(tz.y, tmax));
    miss           = __builtin_amdgcn_alignbit(miss, __float_as_uint(tf - tn), 31);  // sign set: tn > tf, a miss
  }
  const uint32_t hm = ~miss & 0xffu;
  // child hit bits -> triangle bits: every bit doubled (slot s -> bits 2s, 2s + 1), then the node's valid16 keeps the triangles that exist
  uint32_t x = hm;
  x = (x | (x << 4)) & 0x0f0fu;
  x = (x | (x << 2)) & 0x3333u;
  x = (x | (x << 1)) & 0x5555u;
  x |= x << 1;
  const uint32_t valid = n1.z & 0xffffu;
  hmOut   = hm;
  leafOut = ((x & valid) << 16) | valid;
}

// The same test for a PACKET whose rays all point into one octant (k_trace_primary): the node is wave-uniform, so the byte -> float
// conversions (48 of the 174 vector instructions of a packet node test) and the choice of the near / far plane words by the
// direction signs (31 more) are no per-lane work at all.  DevScene::bvh8Planes holds every node's 48 plane bytes as floats, six
// blocks of eight children -- block 2 * axis + side, side 0 = lower planes, 1 = upper -- and the caller fetches the near and the
// far block of each axis straight into SGPRs (scalar loads at an offset that depends on the octant only).  What is left per lane
// and child is three packed fmas (two children at a time, the plane pair is an SGPR pair), the two reductions and the mask.
// The products are those of bvh8TestChildren bit for bit -- float(q) is exact, fma(+-1, d, P) is P +- d -- so both forms return
// the same mask.  `sgn*` = +1 for a negative direction component, else -1.
typedef float f32x8s __attribute__((ext_vector_type(8)));
PT_DEV uint32_t bvh8TestChildrenPlanes(const uint4& n0, const f32x8s& pnx, const f32x8s& pny, const f32x8s& pnz, const f32x8s& pfx, const f32x8s& pfy,
                                       const f32x8s& pfz, const RaySetup& r, float tmax, float sgnx, float sgny, float sgnz)
{
  const float sx = __uint_as_float((n0.w & 0xffu) << 23), sy = __uint_as_float(((n0.w >> 8) & 0xffu) << 23), sz = __uint_as_float(((n0.w >> 16) & 0xffu) << 23);
  const float Px = __uint_as_float(n0.x) - r.org.x, Py = __uint_as_float(n0.y) - r.org.y, Pz = __uint_as_float(n0.z) - r.org.z;
  const float Ax = sx * r.idir.x, Ay = sy * r.idir.y, Az = sz * r.idir.z;
  float       Bnx, Bfx, Bny, Bfy, Bnz, Bfz;
  (void)sgnx; (void)sgny; (void)sgnz;  // (the offsets no longer depend on the direction's sign: slabOffsets)
  slabOffsets(Px, r.idir.x, Ax, Bnx, Bfx);
  slabOffsets(Py, r.idir.y, Ay, Bny, Bfy);
  slabOffsets(Pz, r.idir.z, Az, Bnz, Bfz);
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const f32x2 A2x = {Ax, Ax}, A2y = {Ay, Ay}, A2z = {Az, Az};
  const f32x2 Bn2x = {Bnx, Bnx}, Bn2y = {Bny, Bny}, Bn2z = {Bnz, Bnz}, Bf2x = {Bfx, Bfx}, Bf2y = {Bfy, Bfy}, Bf2z = {Bfz, Bfz};
  uint32_t    miss = 0;
#pragma unroll
  for(int j = 0; j < 4; ++j)
  {
    const int   i   = 6 - 2 * j;  // children i + 1, then i: child 0 ends up in bit 0
    const f32x2 tnx = __builtin_elementwise_fma(f32x2{pnx[i], pnx[i + 1]}, A2x, Bn2x), tfx = __builtin_elementwise_fma(f32x2{pfx[i], pfx[i + 1]}, A2x, Bf2x);
    const f32x2 tny = __builtin_elementwise_fma(f32x2{pny[i], pny[i + 1]}, A2y, Bn2y), tfy = __builtin_elementwise_fma(f32x2{pfy[i], pfy[i + 1]}, A2y, Bf2y);
    const f32x2 tnz = __builtin_elementwise_fma(f32x2{pnz[i], pnz[i + 1]}, A2z, Bn2z), tfz = __builtin_elementwise_fma(f32x2{pfz[i], pfz[i + 1]}, A2z, Bf2z);
    const float tn1 = fmaxf(fmaxf(tnx.y, tny.y), fmaxf(tnz.y, 0.0f)), tf1 = fminf(fminf(tfx.y, tfy.y), fminf(tfz.y, tmax));
    miss            = __builtin_amdgcn_alignbit(miss, __float_as_uint(tf1 - tn1), 31);
    const float tn0 = fmaxf(fmaxf(tnx.x, tny.x), fmaxf(tnz.x, 0.0f)), tf0 = fminf(fminf(tfx.x, tfy.x), fminf(tfz.x, tmax));
    miss            = __builtin_amdgcn_alignbit(miss, __float_as_uint(tf0 - tn0), 31);
  }
  return ~miss & 0xffu;
}

// One node visit.  The builder guarantees that the decoded boxes fmaf(q, 2^e, p) contain their triangles.  Here every slab
// plane costs ONE fma: t = q * A + B with A = 2^e / dir and B = (p - org) / dir -+ pad per axis and per node (slabOffsets): the
// near planes pulled towards the ray's origin and the far planes pushed away by what bounds the accumulated rounding.
// Nodes below index `cached` are read from the workgroup's LDS copy (ldsNodes), the rest from global memory.
// `triBase` / `leafWord`: the triangles of the hit leaf children (leaf word above; leafPending(leafWord) says whether there are any).
PT_DEV void bvh8Visit(const DevScene& sc, const RaySetup& r, float tmax, uint32_t octinv, uint32_t nodeIndex, NodeGroup& outGroup, uint32_t& triBase,
                      uint32_t& leafWord, const uint4* ldsNodes, uint32_t cached, const uint8_t* octLut = nullptr)
{
  uint4 n0, n1, n2, n3, n4;
  if(nodeIndex < cached)
  {
    // explicit LDS address space: through a generic pointer the two branches merge into flat loads of a selected address
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(3))) u32x4* LdsNodePtr;
    LdsNodePtr  N = (LdsNodePtr)(ldsNodes) + nodeIndex * 5u;
    const u32x4 a = N[0], b = N[1], c = N[2], d = N[3], e = N[4];
    n0 = make_uint4(a[0], a[1], a[2], a[3]); n1 = make_uint4(b[0], b[1], b[2], b[3]); n2 = make_uint4(c[0], c[1], c[2], c[3]);
    n3 = make_uint4(d[0], d[1], d[2], d[3]); n4 = make_uint4(e[0], e[1], e[2], e[3]);
  }
  else
  {
    const uint4* N = sc.bvh8Nodes + size_t(nodeIndex) * 5;
    n0 = N[0]; n1 = N[1]; n2 = N[2]; n3 = N[3]; n4 = N[4];
  }
  const uint32_t imask = n0.w >> 24;
  uint32_t       hm, leaf;
  bvh8TestChildren(n0, n1, n2, n3, n4, r, tmax, hm, leaf);
  uint32_t hits = hm & imask;
  // slot space -> priority space: bit p = slot ^ octinv
  if(MI_PT_OCT_LUT && octLut)
  {
    typedef const __attribute__((address_space(3))) uint8_t* LdsBytePtr;  // (explicit LDS address space: a generic pointer would be a flat load)
    hits = uint32_t(((LdsBytePtr)octLut)[(octinv << 8) | hits]);
  }
  else
    hits = octPermute(hits, octinv);
  outGroup.base = n1.x;
  outGroup.bits = (hits << 8) | imask;
  triBase       = n1.y;
  leafWord      = leaf;
}

}  // namespace pt
