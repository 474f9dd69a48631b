// Device-resident scene tables and per-path state of the wavefront path tracer.  Layout notes are in DESIGN.md §3.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mi_pt.h"
#include "pt_math.h"

namespace pt {

// ---- static scene -----------------------------------------------------------------------------------------------------
struct DevTexture  // 80 B
{
  uint32_t levelOffset[16];  // texel offset of each mip level inside the shared RGBA8 pool
  uint16_t width, height;
  uint8_t  numLevels;
  uint8_t  srgb;
  uint8_t  magFilter, minFilter, mipmapMode, wrapS, wrapT;
  uint8_t  _pad[5];
};

// GltfRenderPrimitive with device pointers (reference: shaders/gltf_scene_io.h.slang:50-64)
struct DevPrim
{
  const uint32_t* indices;
  const float*    positions;
  const float*    normals;
  const uint32_t* colors;
  const float*    tangents;
  const float*    texCoords0;
  const float*    texCoords1;
};

enum : uint32_t
{
  INST_FORCE_OPAQUE = 1u,  // reference: src/gltf_scene_rtx.cpp:271-295
  INST_CULL_DISABLE = 2u,
  INST_FLIP_FACING  = 4u,  // det(objectToWorld) < 0: world-space winding is the mirror of object-space winding
};

// World-space triangle, 48 B: {v0, rnode} {e1, prim} {e2, instFlags}
struct DevTri
{
  float4 a, b, c;
};

struct DevScene
{
  const MiGltfShadeMaterial* materials;
  const MiGltfTextureInfo*   texInfos;
  const MiGltfRenderNode*    nodes;
  const DevPrim*             prims;
  const MiGltfLight*         lights;
  const DevTexture*          textures;
  const uchar4*              texels;
  const float4*              envPixels;  // rgb + pdf
  const MiEnvAccel*          envAccel;
  const float4*              bvhNodes;  // 4 x float4 per node (see pt_bvh.h)
  const DevTri*              tris;
  const float*               srgbLut;  // 256 floats
  int                        numMaterials, numTextures, numLights, numNodes;
  int                        envWidth, envHeight;
  int                        numTris;
  int                        bvhRoot;  // node index, or ~tri for a single-triangle scene; INT_MIN when empty
};

// ---- per-frame constants (kernel argument, ~600 B) ---------------------------------------------------------------------
struct FrameConsts
{
  MiSceneFrameInfo        frameInfo;
  MiSkyPhysicalParameters sky;
  MiPathtraceParams       pc;
  int                     width, height;
  int                     tileSize, tilesX, tilesY;
  int                     numSlots;  // owned tiles * tileSize^2
  int                     lightWeightValid;
};

// ---- per-path state, structure of arrays indexed by slot -----------------------------------------------------------------
// Per-segment traffic (read + write) is accounted in DESIGN.md §5; keep records 16-byte sized for dwordx4 access.
enum : uint32_t
{
  PF_INSIDE     = 1u << 0,   // pt.isInside
  PF_NOT_SOLID  = 1u << 1,   // !pt.solid
  PF_ALIVE      = 1u << 2,   // still in the bounce loop
  PF_PLANE_HIT  = 1u << 3,
  PF_DEPTH_SHIFT   = 8,      // bits 8..15  surfaceDepth
  PF_SCATTER_SHIFT = 16,     // bits 16..23 scatterBounces (saturating)
};

struct PathSoA
{
  float4*   rayOrg;       // origin.xyz, tmax
  float4*   rayDir;       // direction.xyz, cone.width
  float4*   hit;          // t, triIndex (int bits; -1 = miss), u, v
  float4*   throughput;   // rgb, lastSamplePdf
  float4*   radiance;     // rgb, maxRoughness.x
  float4*   misc;         // maxRoughness.y, flags (uint bits), seed (uint bits), unused
  uint4*    medium;       // VolumeMedium as 7 halves packed
  float4*   firstHit;     // firstHitPos.xyz, unused
  float4*   shadowOrg;    // origin.xyz, dist
  float4*   shadowDir;    // direction.xyz, flags (bit0 = initialInside)
  float4*   shadowContrib;// rgb, unused
  float4*   pixelSum;     // sum over the frame's samples of the clamped radiance rgba
  float4*   guideAlbedo;  // optional (denoiser guides): sum over samples
  float4*   guideNormal;
};

struct Queues
{
  uint32_t* active[2];   // path slots to trace/shade this iteration, ping-pong
  uint32_t* shadow;      // path slots with a pending shadow ray
  uint32_t* sortKeys;    // scratch for the per-bounce sort
  uint32_t* sortTmp;
  // counters live in one small device array: see QC_* indices
  uint32_t* counters;
};
enum : int
{
  QC_ACTIVE0 = 0,
  QC_ACTIVE1 = 1,
  QC_SHADOW  = 2,
  QC_HEAD_TRACE  = 3,  // dynamic-fetch heads (persistent kernels)
  QC_HEAD_SHADE  = 4,
  QC_HEAD_SHADOW = 5,
  QC_COUNT       = 16
};

struct StatCounters  // device mirror of MiPtStats' dynamic part
{
  unsigned long long cameraPaths, segments, shadowRays, nodesClosest, trisClosest, nodesShadow, trisShadow, textureTaps;
};

}  // namespace pt
