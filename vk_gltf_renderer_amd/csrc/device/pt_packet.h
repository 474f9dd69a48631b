// Interval test of a one-octant camera-ray packet against the eight children of a BVH8 node (k_trace_primary, pt_kernels.hip).
//
// The packet walk enters a child when ANY of its 64 rays hits the child's box, and entering one child too many changes no result
// (the triangle tests stay per ray and exact).  With pixel-major path slots a packet is 64 samples of one pixel: rays that differ
// by a sub-pixel jitter -- yet the per-ray form (pt_bvh8.h: bvh8TestChildrenPlanes) spends 174 vector instructions per node
// having every lane test all eight children.  Here the 64 lanes share the work instead: lane = child * 8 + plane tests ONE plane
// of ONE child against the packet's *interval ray* (bounds of the origins and of the inverse directions per axis) and returns a
// lower bound of every ray's entry time through an entry plane, or an upper bound of every ray's exit time through an exit plane;
// two three-step reductions inside the aligned groups of eight lanes give each child's [entry, exit] and a ballot gives the mask.
// About 35 vector instructions per node.  Conservative by construction (the slack below covers the per-ray test's own delta and
// the rounding of this arithmetic), so the mask contains every child the per-ray test would enter for some ray.
//
// Plain float arithmetic only: tests/host_shim compiles this header for the host and tests/test_packet_interval.py checks the
// containment on random nodes and packets without a GPU.
#pragma once
#include "pt_math.h"

namespace pt {

struct PacketBounds  // over the packet's rays (one sign per axis for the inverse directions: the packet is "one-octant")
{
  float omin[3], omax[3];  // ray origins
  float imin[3], imax[3];  // RaySetup::idir
};

// What lane (child * 8 + plane) keeps for the whole walk; plane = 2 * axis + side as in DevScene::bvh8Planes (side 0 = the
// children's lower planes), planes 6 and 7 are idle lanes.
struct PacketLane
{
  float    oSel, oOther;  // the origin bound this plane's time is extreme for, and the other one (for the slack)
  float    iA, iB;        // bounds of the inverse direction on this plane's axis
  float    slack;         // -1 / +1: the slack moves an entry plane towards the rays' origins, an exit plane away from them
  uint32_t planeOffset;   // index of this lane's float inside a node's 48 planes: plane * 8 + child
  uint32_t axis;
  bool     entry, live;
};

// negMask: bit a set = the packet's direction component a is negative (then rays enter through the children's UPPER planes)
PT_DEV PacketLane makePacketLane(uint32_t lane, uint32_t negMask, const PacketBounds& B)
{
  PacketLane     L;
  const uint32_t plane = lane & 7u, child = lane >> 3;
  L.live               = plane < 6u;
  L.axis               = L.live ? (plane >> 1) : 0u;
  const uint32_t side  = plane & 1u;
  const bool     neg   = ((negMask >> L.axis) & 1u) != 0u;
  L.entry              = side == (neg ? 1u : 0u);
  // t = (w - o) * i over o in [omin, omax], i in [imin, imax] (one sign) is bilinear: its extremes lie at the corners.  The entry
  // time is smallest (the exit time largest) at the origin bound chosen here; the two direction bounds are both tried per node.
  const bool useMax = L.entry == !neg;
  L.oSel            = useMax ? B.omax[L.axis] : B.omin[L.axis];
  L.oOther          = useMax ? B.omin[L.axis] : B.omax[L.axis];
  L.iA              = B.imin[L.axis];
  L.iB              = B.imax[L.axis];
  L.slack           = (L.entry != neg) ? -1.0f : 1.0f;
  L.planeOffset     = (L.live ? plane : 0u) * 8u + child;
  return L;
}

// q: this lane's quantised plane (0..255 as a float), P / s: the node's origin and 2^exponent on the lane's axis.  The plane is at
// fmaf(q, s, P) exactly as the builder decodes it.  Slack: the per-ray test moves its planes by 2^-21 (|P - org| + 255 s) to cover
// its own rounding; 2^-19 of the same terms plus |plane| covers that and the three roundings made here.
PT_DEV float packetPlaneTime(const PacketLane& L, float q, float P, float s)
{
  const float w   = __fmaf_rn(q, s, P);
  const float d   = 1.9073486328125e-6f * (fmaxf(fabsf(P - L.oSel), fabsf(P - L.oOther)) + 255.0f * s + fabsf(w));
  const float rel = __fmaf_rn(L.slack, d, w) - L.oSel;
  const float t1 = rel * L.iA, t2 = rel * L.iB;
  return L.entry ? fminf(t1, t2) : fmaxf(t1, t2);
}

// tn = the largest entry bound, tf = the smallest exit bound of a child's three axes; tmax = the largest closest-hit distance any ray
// of the packet still accepts.  Written so that a NaN (inf - inf on a degenerate axis) counts as a hit.
PT_DEV bool packetChildHit(float tn, float tf, float tmax)
{
  const float tol = 9.5367431640625e-7f * (fabsf(tf) + fabsf(tn));  // 2^-20: the products' relative rounding
  return !(tf - tn < -tol) && !(tn > tmax) && !(tf < 0.0f);
}

}  // namespace pt
