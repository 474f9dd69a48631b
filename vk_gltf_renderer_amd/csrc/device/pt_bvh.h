// Software BVH traversal for gfx950: replaces VK_KHR_ray_query / the RT pipeline of the reference
// (shaders/raytracer_interface.h.slang:67-188) with a per-lane stack walk whose hot part of the stack lives in LDS.
//
// Node (BVH2, 64 B = 4 x float4, one dwordx4 x4 burst):
//   n0 = (c0.lo.x, c0.hi.x, c0.lo.y, c0.hi.y)   n1 = (c1.lo.x, c1.hi.x, c1.lo.y, c1.hi.y)
//   n2 = (c0.lo.z, c0.hi.z, c1.lo.z, c1.hi.z)   n3 = (child0, child1, -, -) as int bits
//   child >= 0: inner node index; child < 0: leaf, ~child = index of the (single) triangle.
// Triangle (48 B): see DevTri in pt_scene.h.
//
// The walk is exposed as a single-step function so that the trace kernels can run it as a per-lane state machine and
// hand a finished lane a new ray while its neighbours keep walking (wave-level dynamic refill, pt_kernels.hip).
#pragma once
#include "pt_scene.h"

namespace pt {

#ifndef MI_BVH8_STACK_LDS
#define MI_BVH8_STACK_LDS 12
#endif
constexpr int BVH_STACK_LDS  = 2 * MI_BVH8_STACK_LDS;  // entries per lane kept in LDS (24)
constexpr int BVH_STACK_PRIV = 72;  // overflow entries per lane (scratch; touched only by very deep LBVH paths)
constexpr int BVH_EMPTY      = int(0x80000000u);

struct TriHit
{
  float t, u, v;
  bool  front;
};

// Moeller-Trumbore with the evaluation order shared with the oracle (oracle_pt.cpp intersectTri): t, u, v are bit-identical.
PT_DEV bool intersectTri(f3 v0, f3 e1, f3 e2, f3 org, f3 dir, TriHit& h)
{
  f3    pvec = crossFma(dir, e2);
  float det  = dotFma(e1, pvec);
  if(det == 0.0f)
    return false;
  float inv  = divExact(1.0f, det);  // (IEEE whatever the compile options: hit records agree with the oracle bit for bit, pt_math.h)
  f3    tvec = org - v0;
  float u    = dotFma(tvec, pvec) * inv;
  if(u < 0.0f || u > 1.0f)
    return false;
  f3    qvec = crossFma(tvec, e1);
  float v    = dotFma(dir, qvec) * inv;
  if(v < 0.0f || u + v > 1.0f)
    return false;
  h.t     = dotFma(e2, qvec) * inv;
  h.u     = u;
  h.v     = v;
  h.front = det > 0.0f;
  return true;
}

struct RaySetup
{
  f3 org, dir, idir, ood;
};
PT_DEV RaySetup makeRaySetup(f3 org, f3 dir)
{
  RaySetup    r;
  const float eps = 1e-30f;
  r.org           = org;
  r.dir           = dir;
  r.idir.x        = divExact(1.0f, fabsf(dir.x) < eps ? copysignf(eps, dir.x) : dir.x);
  r.idir.y        = divExact(1.0f, fabsf(dir.y) < eps ? copysignf(eps, dir.y) : dir.y);
  r.idir.z        = divExact(1.0f, fabsf(dir.z) < eps ? copysignf(eps, dir.z) : dir.z);
  r.ood           = org * r.idir;
  return r;
}

// LDS stack addressing: [depth][lane] so that a wave's simultaneous push/pop at equal depth is conflict-free.
struct LaneStack
{
  int* lds;      // base of this block's LDS stack
  int  tid;      // thread index in block
  int  stride;   // block size
  int* priv;     // BVH_STACK_PRIV ints of overflow in scratch: a SEPARATE local array of the caller -- as a member array it
                 // would drag the whole struct (sp, tid, stride) into scratch and turn every push/pop into memory round trips
  int  sp;
  PT_DEV void push(int v)
  {
    if(sp < BVH_STACK_LDS)
      lds[sp * stride + tid] = v;
    else if(sp - BVH_STACK_LDS < BVH_STACK_PRIV)
      priv[sp - BVH_STACK_LDS] = v;
    ++sp;
  }
  PT_DEV int pop()
  {
    --sp;
    if(sp < BVH_STACK_LDS)
      return lds[sp * stride + tid];
    if(sp - BVH_STACK_LDS < BVH_STACK_PRIV)
      return priv[sp - BVH_STACK_LDS];
    return BVH_EMPTY;
  }
};

// One inner-node step: tests both children's boxes (conservatively widened by a few ulps so that no triangle accepted by
// intersectTri is ever culled) and returns the next node to visit (BVH_EMPTY when the stack ran dry).
PT_DEV int bvhInnerStep(const DevScene& sc, const RaySetup& r, float tmax, int cur, LaneStack& st)
{
  const float4* n  = sc.bvhNodes + size_t(cur) * 4;
  const float4  n0 = n[0], n1 = n[1], n2 = n[2], n3 = n[3];
  float c0x0 = __fmaf_rn(n0.x, r.idir.x, -r.ood.x), c0x1 = __fmaf_rn(n0.y, r.idir.x, -r.ood.x);
  float c0y0 = __fmaf_rn(n0.z, r.idir.y, -r.ood.y), c0y1 = __fmaf_rn(n0.w, r.idir.y, -r.ood.y);
  float c0z0 = __fmaf_rn(n2.x, r.idir.z, -r.ood.z), c0z1 = __fmaf_rn(n2.y, r.idir.z, -r.ood.z);
  float c1x0 = __fmaf_rn(n1.x, r.idir.x, -r.ood.x), c1x1 = __fmaf_rn(n1.y, r.idir.x, -r.ood.x);
  float c1y0 = __fmaf_rn(n1.z, r.idir.y, -r.ood.y), c1y1 = __fmaf_rn(n1.w, r.idir.y, -r.ood.y);
  float c1z0 = __fmaf_rn(n2.z, r.idir.z, -r.ood.z), c1z1 = __fmaf_rn(n2.w, r.idir.z, -r.ood.z);
  float t0n  = fmaxf(fmaxf(fminf(c0x0, c0x1), fminf(c0y0, c0y1)), fmaxf(fminf(c0z0, c0z1), 0.0f));
  float t0f  = fminf(fminf(fmaxf(c0x0, c0x1), fmaxf(c0y0, c0y1)), fminf(fmaxf(c0z0, c0z1), tmax));
  float t1n  = fmaxf(fmaxf(fminf(c1x0, c1x1), fminf(c1y0, c1y1)), fmaxf(fminf(c1z0, c1z1), 0.0f));
  float t1f  = fminf(fminf(fmaxf(c1x0, c1x1), fmaxf(c1y0, c1y1)), fminf(fmaxf(c1z0, c1z1), tmax));
  bool  hit0 = t0n <= t0f * 1.0000012f + 1e-30f;
  bool  hit1 = t1n <= t1f * 1.0000012f + 1e-30f;
  int   ch0 = __float_as_int(n3.x), ch1 = __float_as_int(n3.y);
  if(hit0 && hit1)
  {
    bool swap = t1n < t0n;
    st.push(swap ? ch0 : ch1);
    return swap ? ch1 : ch0;
  }
  if(hit0)
    return ch0;
  if(hit1)
    return ch1;
  return st.sp > 0 ? st.pop() : BVH_EMPTY;
}
PT_DEV int bvhPop(LaneStack& st) { return st.sp > 0 ? st.pop() : BVH_EMPTY; }

// Whole walk in one call (used by the one-off selection pass). `visit(triIndex, tmax) -> tmax'`; negative terminates.
template <typename Visit>
PT_DEV void bvhWalk(const DevScene& sc, const RaySetup& r, float tmax, LaneStack& st, Visit&& visit)
{
  int cur = sc.bvhRoot;
  st.sp   = 0;
  while(cur != BVH_EMPTY)
  {
    if(cur >= 0)
      cur = bvhInnerStep(sc, r, tmax, cur, st);
    else
    {
      tmax = visit(~cur, tmax);
      if(tmax < 0.0f)
        return;
      cur = bvhPop(st);
    }
  }
}

}  // namespace pt
