// Host-side interface of the on-device BVH builder (bvh_build.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "pt_scene.h"

namespace pt {

struct BvhBuildInput
{
  const MiGltfRenderNode* nodes;          // device
  const DevPrim*          prims;          // device
  const uint8_t*          instFlags;      // device, per render node
  const uint32_t*         nodeTriOffset;  // device, numEntries + 1
  const int32_t*          entryNode;      // device, numEntries
  int                     numEntries;
  uint32_t                numTris;
  bool                    karrasTopology = false;  // true: plain LBVH (Morton-prefix hierarchy); false: PLOC clustering
  int                     reinsertPasses = 0;      // parallel reinsertion over the finished BVH2 (bvh_reinsert.h): searches ...
  int                     reinsertRounds = 4;      // ... and lock / move rounds per search
  float                   splitFactor    = 0.0f;   // triangle pre-splitting (bvh_split.h): references for parts whose box area exceeds this x the mean; 0 = off
  int                     splitMaxDepth  = 8;      // ... at most 2^this references per triangle
  float                   splitMinShare  = 0.1f;   // ... and only in scenes where triangles above 64 x the mean hold at least this share of the summed box area (0 = always)
};
struct BvhBuildOutput
{
  float4*  nodes    = nullptr;  // device, 4 float4 per node
  DevTri*  tris     = nullptr;  // device, Morton order (leaf reference ~i = triangle i of this array)
  uint32_t numNodes = 0, numTris = 0;   // numTris = REFERENCES: a pre-split triangle appears once per reference (each a full copy of its record)
  uint32_t sceneTris = 0;               // triangles the references were made from
  int      root     = 0;
  uint32_t reinsertMoves = 0;   // subtrees the reinsertion passes moved
  float    centroidLo[3] = {0, 0, 0}, centroidHi[3] = {0, 0, 0};
};
bool buildBvh(const BvhBuildInput& in, BvhBuildOutput& out, hipStream_t stream, std::string& err);

// 8-wide compressed BVH collapsed from the BVH2 above (bvh8.hip)
struct Bvh8Output
{
  uint4*   nodes    = nullptr;  // device, 5 uint4 per node
  DevTri*  tris     = nullptr;  // device, node order (triangles of a node's leaf children are contiguous)
  uint32_t numNodes = 0, numTris = 0;
};
struct Bvh8Options  // (MI_PT_COLLAPSE / MI_PT_HOST_COLLAPSE, read and validated once in mi_pt_create: RunSwitches)
{
  bool sahCollapse  = true;   // SAH-optimal dynamic programme (default) | greedy by surface area
  bool hostCollapse = false;  // the greedy host collapse (A/B reference of the device one)
};
bool buildBvh8(const BvhBuildOutput& b2, Bvh8Output& out, hipStream_t stream, std::string& err, const Bvh8Options& opt = Bvh8Options());

}  // namespace pt
