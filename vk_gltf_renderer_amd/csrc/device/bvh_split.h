// Triangle pre-splitting for the BVH builder: the work of ONE thread (one triangle), shared by the kernels of bvh_build.hip, the CPU test
// tier (tests/host_shim/split_on_host.cpp) and nothing else.
//
// Why.  The reference builds its BLAS over whatever the asset holds (src/gltf_scene_rtx.cpp:173-227) and the RT hardware copes with a
// hall-sized wall triangle next to millimetre detail; a software BVH over triangle BOXES does not: the box of a long thin or simply huge
// triangle overlaps everything near it, sits high in the tree, and every ray that passes tests it (sliver stand-in of the atrium: 30.5
// triangle tests per secondary ray against 8.0 on the evenly tessellated one, 419 against 695 Msamples/s).  The cure is the classic one
// (Ernst & Greiner 2007 "Early split clipping"; Karras & Aila 2013 section 4): the builder works on REFERENCES -- (triangle, box) pairs --
// and a triangle whose box is large gets several, each with the box of the part of the triangle inside one cell of a recursive
// bisection of its box.  Every reference carries a full copy of the triangle record, so nothing downstream of the builder knows:
// the hit record names a reference, whose shade / alpha records are the triangle's.
//
// What must hold for the image not to move (tests: split_on_host + test_image_independent_of_acceleration_structure):
//  * the reference boxes of a triangle cover it (each is the padded box of triangle ∩ cell, the cells partition the triangle's box);
//  * a ray may meet the same triangle through two references: closest hit -- same (t, u, v), tie broken by (renderNode, primitive), equal,
//    first one stays; any-hit opaque / alpha-tested -- the draw is hash(seed, renderNode, primitive), idempotent.  Only the RECORDING
//    shadow walk counts candidates (transmissive instances, raytracer_interface.h.slang:160-178): triangles of transmissive instances
//    are therefore never split (the caller passes splittable = false).
//
// Rule (the lab's, tools/lab/bvh_lab.cpp split=F): a part whose box area (ex*ey + ey*ez + ez*ex) exceeds `thresholdArea` = F x the mean
// box area of the scene's triangles is cut at the middle of its box's longest axis, recursively, at most `maxDepth` times.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace pt {

#ifndef PT_DEV
#define PT_DEV __device__ __forceinline__
#endif

constexpr int SPLIT_MAX_DEPTH = 10;  // at most 2^10 references per triangle (a 36-m wall against centimetre detail needs ~8)

struct SplitBox
{
  float lo[3], hi[3];
};
PT_DEV float splitHalfArea(const SplitBox& b)
{
  const float ex = b.hi[0] - b.lo[0], ey = b.hi[1] - b.lo[1], ez = b.hi[2] - b.lo[2];
  return ex * ey + ey * ez + ez * ex;
}

// Box of (triangle ∩ cell), Sutherland-Hodgman against the six planes of the cell (boundary points belong to both sides).  Returns false
// when nothing of the triangle lies in the cell.  The result is clamped to the cell.
PT_DEV bool splitClipBox(const float p[3][3], const SplitBox& cell, SplitBox& out)
{
  float a[10][3], b[10][3];
  int   n = 3;
  for(int i = 0; i < 3; ++i)
    for(int c = 0; c < 3; ++c)
      a[i][c] = p[i][c];
  for(int plane = 0; plane < 6 && n > 0; ++plane)
  {
    const int   ax   = plane >> 1;
    const bool  isHi = (plane & 1) != 0;
    const float d    = isHi ? cell.hi[ax] : cell.lo[ax];
    int         m    = 0;
    for(int i = 0; i < n; ++i)
    {
      const float* s = a[i];
      const float* e = a[(i + 1) % n];
      const bool   is = isHi ? s[ax] <= d : s[ax] >= d, ie = isHi ? e[ax] <= d : e[ax] >= d;
      if(is && m < 10)
      {
        b[m][0] = s[0]; b[m][1] = s[1]; b[m][2] = s[2];
        ++m;
      }
      if(is != ie && m < 10)
      {
        const float t = (d - s[ax]) / (e[ax] - s[ax]);
        for(int c = 0; c < 3; ++c)
          b[m][c] = s[c] + (e[c] - s[c]) * t;
        b[m][ax] = d;
        ++m;
      }
    }
    n = m;
    for(int i = 0; i < n; ++i)
      for(int c = 0; c < 3; ++c)
        a[i][c] = b[i][c];
  }
  if(n == 0)
    return false;
  for(int c = 0; c < 3; ++c)
  {
    float lo = a[0][c], hi = a[0][c];
    for(int i = 1; i < n; ++i)
    {
      lo = a[i][c] < lo ? a[i][c] : lo;
      hi = a[i][c] > hi ? a[i][c] : hi;
    }
    out.lo[c] = lo < cell.lo[c] ? cell.lo[c] : lo;
    out.hi[c] = hi > cell.hi[c] ? cell.hi[c] : hi;
  }
  return true;
}

// References of one triangle: calls emit(box) once per reference, in a fixed (depth-first, low half first) order, and returns their number
// (>= 1).  triBox = the triangle's own conservative box (what an unsplit triangle is filed under); every emitted box lies inside it.
// The interpolated clip points carry a rounding error of an ulp or two of the coordinates involved: each reference box is padded by
// 2^-20 of its largest absolute coordinate per axis (and clamped back to triBox), so that the boxes of neighbouring parts overlap
// instead of leaving a crack a ray could pass through.
template <class Emit>
PT_DEV int splitTriangle(const float p[3][3], const SplitBox& triBox, bool splittable, float thresholdArea, int maxDepth, Emit&& emit)
{
  if(!splittable || !(splitHalfArea(triBox) > thresholdArea) || maxDepth <= 0)
  {
    emit(triBox);
    return 1;
  }
  maxDepth = maxDepth > SPLIT_MAX_DEPTH ? SPLIT_MAX_DEPTH : maxDepth;
  SplitBox stackBox[SPLIT_MAX_DEPTH + 1];
  int      stackDepth[SPLIT_MAX_DEPTH + 1];
  int      sp = 0, count = 0;
  stackBox[0]   = triBox;
  stackDepth[0] = 0;
  sp            = 1;
  while(sp > 0)
  {
    --sp;
    const SplitBox cell  = stackBox[sp];
    const int      depth = stackDepth[sp];
    SplitBox       b;
    if(!splitClipBox(p, cell, b))
      continue;
    auto emitPadded = [&](const SplitBox& q) {
      SplitBox r;
      for(int c = 0; c < 3; ++c)
      {
        const float alo = q.lo[c] < 0.0f ? -q.lo[c] : q.lo[c], ahi = q.hi[c] < 0.0f ? -q.hi[c] : q.hi[c];
        const float pad = (alo > ahi ? alo : ahi) * 9.5367431640625e-7f;  // 2^-20
        const float lo = q.lo[c] - pad, hi = q.hi[c] + pad;
        r.lo[c] = lo < triBox.lo[c] ? triBox.lo[c] : lo;
        r.hi[c] = hi > triBox.hi[c] ? triBox.hi[c] : hi;
      }
      emit(r);
      ++count;
    };
    if(depth >= maxDepth || !(splitHalfArea(b) > thresholdArea))
    {
      emitPadded(b);
      continue;
    }
    int   ax  = 0;
    float ext = b.hi[0] - b.lo[0];
    for(int c = 1; c < 3; ++c)
      if(b.hi[c] - b.lo[c] > ext)
      {
        ext = b.hi[c] - b.lo[c];
        ax  = c;
      }
    const float mid = 0.5f * (b.lo[ax] + b.hi[ax]);
    if(!(mid > b.lo[ax] && mid < b.hi[ax]))  // no room left between the two planes in float: this part stays whole
    {
      emitPadded(b);
      continue;
    }
    SplitBox lower = b, upper = b;
    lower.hi[ax] = mid;
    upper.lo[ax] = mid;
    // (upper pushed first: the lower half is taken off the stack first)
    stackBox[sp] = upper; stackDepth[sp] = depth + 1; ++sp;
    stackBox[sp] = lower; stackDepth[sp] = depth + 1; ++sp;
  }
  if(count == 0)  // cannot happen for a finite triangle (the root cell is its own box); never lose a triangle over it
  {
    emit(triBox);
    count = 1;
  }
  return count;
}

}  // namespace pt
