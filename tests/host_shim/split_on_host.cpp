// TEST INFRASTRUCTURE ONLY.  The per-triangle splitter of the BVH builder (csrc/device/bvh_split.h) compiled for the host through the stand-in
// <hip/hip_runtime.h> of this directory, so that the CPU-only test tier (tests/test_bvh_split.py) runs the code the device runs.  Never loaded by the product.
#include "bvh_split.h"

using namespace pt;

extern "C" {
// tri: 9 floats (three vertices); triBox: 6 floats lo xyz, hi xyz; outBoxes: capacity x 6 floats.  Returns the number of references
// (all of them counted, the first `capacity` written).
__attribute__((visibility("default"))) int dev_split_triangle(const float* tri, const float* triBox, int splittable, float thresholdArea, int maxDepth, float* outBoxes,
                                                              int capacity)
{
  const float p[3][3] = {{tri[0], tri[1], tri[2]}, {tri[3], tri[4], tri[5]}, {tri[6], tri[7], tri[8]}};
  SplitBox tb;
  for(int c = 0; c < 3; ++c)
  {
    tb.lo[c] = triBox[c];
    tb.hi[c] = triBox[3 + c];
  }
  int written = 0;
  return splitTriangle(p, tb, splittable != 0, thresholdArea, maxDepth, [&](const SplitBox& b) {
    if(written < capacity)
    {
      for(int c = 0; c < 3; ++c)
      {
        outBoxes[written * 6 + c]     = b.lo[c];
        outBoxes[written * 6 + 3 + c] = b.hi[c];
      }
      ++written;
    }
  });
}
}
