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

// GltfTextureInfo + the descriptor of the texture behind it, flattened and indexed like GltfTextureInfo (the texture slots of
// a material): the shade kernel reaches the texels in two dependent loads (this record, the texels) instead of five
// (table pointers, texture info, descriptor, mip offset, texels).  The mip chain is contiguous in the texel pool, so the
// offset of a level is level0 + sum of the sizes of the levels before it.
struct DevTexRef  // 44 B
{
  float    uv[6];          // uvTransform (column-major 3x2)
  uint32_t level0;         // texel offset of mip 0
  uint16_t width, height;  // 0 x 0: no valid texture behind this slot (sampling yields 1, like an out-of-range index)
  uint8_t  numLevels, srgb, magFilter, minFilter, mipmapMode, wrapS, wrapT, texCoord;
  uint32_t _pad;
};
// The five CORE map slots of a material -- base colour, normal, metallic-roughness, emissive, occlusion, the order of MiGltfShadeMaterial's slot words --
// as 16-byte records indexed by MATERIAL: coreTex[5 * materialID + k].  Their address depends on the material index alone, so a shade kernel can fetch a
// slot's record next to the material record instead of behind it (pt_shading.h: coreTexPlan; in use for the base colour).  Slots whose sampler the footprint
// path does not cover (NEAREST filters, MIRRORED_REPEAT) keep the general fetch through `ref`.
struct DevCoreTex  // 16 B
{
  uint32_t level0;  // texel offset of mip 0
  uint32_t wh;      // width | height << 16; 0: no valid texture behind the slot (sampling yields 1)
  uint32_t flags;   // CT_*
  uint32_t ref;     // the slot's texture-info index (DevTexRef): the general fetch, and the uv transform where CT_TRANSFORM says there is one
};
enum : uint32_t
{
  CT_FAST         = 1u,       // LINEAR mag and min filter, no MIRRORED_REPEAT: every shape of the fetch is one or two footprint records
  CT_SRGB         = 1u << 1,
  CT_TEXCOORD1    = 1u << 2,
  CT_TRANSFORM    = 1u << 3,  // KHR_texture_transform other than the identity
  CT_MIP_LINEAR   = 1u << 4,
  CT_WRAPS_SHIFT  = 8,        // 2 bits
  CT_WRAPT_SHIFT  = 10,       // 2 bits
  CT_LEVELS_SHIFT = 16,       // 8 bits
};
struct TexCtx  // what a texture fetch needs, passed BY VALUE (registers) into the non-inlined fetch
{
  const DevTexRef* refs;
  const uchar4*    texels;
  const float*     lut;    // sRGB decode table (global, or the shade kernel's LDS copy)
  const uint4*     quads;  // bilinear footprints, one per texel (DevScene::texQuads), or null
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
  // position + normal + uv0 + tangent of a vertex interleaved in 48 bytes (3 x float4: {p.xyz, n.x} {n.y, n.z, uv0} {tangent}):
  // the hit-attribute fetch of the shade kernel touches 3 cache lines per hit instead of 12 (the separate streams above stay
  // for the builders, the alpha records, uv1 and colours).  Absent attributes are zero; the stream pointers say which exist.
  const float4*   verts;
  // triangles [0, opaqueTriangles) pass their material's alpha test everywhere (MiPtRenderPrimitive::opaqueTriangleCount): the
  // build gives them INST_FORCE_OPAQUE in their triangle record, so the walks never defer an alpha test for them
  uint32_t        opaqueTriangles;
  uint32_t        _pad;
};

enum : uint32_t
{
  INST_FORCE_OPAQUE = 1u,  // reference: src/gltf_scene_rtx.cpp:271-295
  INST_CULL_DISABLE = 2u,
  INST_FLIP_FACING  = 4u,  // det(objectToWorld) < 0: world-space winding is the mirror of object-space winding
  INST_TRANSMISSIVE = 8u,  // material.transmissionFactor > MIN_TRANSMISSION: shadow rays attenuate instead of stopping
  // A non-opaque instance whose material has alphaMode OPAQUE (it is non-opaque because it transmits: getInstanceFlag): getOpacity returns 1
  // (pathtrace_functions.h.slang:196-197), the draw `rand <= 1` always commits -- the candidate needs no alpha test, only the treatment of a
  // non-opaque instance (no FORCE_OPAQUE: shadow rays do not stop at it).  The walks take such a candidate at once instead of deferring a
  // test whose outcome is known (round 5: every sphere of the glass-class workload).
  INST_ALPHA_PASSES = 16u,
};

// World-space triangle, 48 B: {v0, rnode} {e1, prim} {e2, instFlags}
struct DevTri
{
  float4 a, b, c;
};

// Everything the stochastic-alpha test of one triangle needs, in ONE 48-byte record next to the triangle (same index):
// getOpacity (shaders/pathtrace_functions.h.slang:189-234) walks instance -> material -> primitive -> indices -> uvs ->
// texture-info -> texture -> texels, eight dependent loads per candidate; on the software walk that chain, not the
// arithmetic, dominated foliage rays.  Here it is record -> texels.
struct DevAlphaTri
{
  float4 a;  // uv0.x, uv0.y, uv1.x, uv1.y            (of the TEXCOORD set the alpha texture uses)
  float4 b;  // uv2.x, uv2.y, alpha factor, alpha cutoff
  uint4  c;  // level-0 texel offset | width | height << 16 | flags | vertex alpha bytes a0 | a1 << 8 | a2 << 16
};
enum : uint32_t
{
  AT_MODE_MASK   = 3u,       // MiAlphaMode
  AT_HAS_TEXTURE = 1u << 2,
  AT_LINEAR      = 1u << 3,  // magFilter (SampleLevel(uv, 0) is a magnification)
  AT_WRAPS_SHIFT = 4,        // 2 bits
  AT_WRAPT_SHIFT = 6,        // 2 bits
  AT_HAS_COLORS  = 1u << 8,
};

// What the shade kernel needs to start fetching the hit's attributes, per triangle (same index as tris): without it the
// chain is triangle -> render node -> primitive record -> indices -> vertices; with it record -> {node, primitive, material}
// -> vertices.
enum : uint32_t  // DevShadeTri::attrs: which vertex streams the triangle's primitive has
{
  SHADE_HAS_NORMALS  = 1u << 0,
  SHADE_HAS_UV0      = 1u << 1,
  SHADE_HAS_TANGENTS = 1u << 2,
  SHADE_HAS_UV1      = 1u << 3,  // uv1 / colours are not part of the interleaved vertex: a hit that needs them fetches its DevPrim
  SHADE_HAS_COLORS   = 1u << 4,
};
struct DevShadeTri  // 32 B
{
  uint32_t v0, v1, v2;    // the triangle's three interleaved vertices (DevPrim::verts) as float4 indices into DevScene::geomPool: the
                          // shade kernel goes from this record straight to the vertices, without the primitive's pointer table
  uint32_t rnode;         // GltfRenderNode index
  int32_t  renderPrimID;  // GltfRenderNode::renderPrimID
  int32_t  materialID;    // max(0, GltfRenderNode::materialID)
  uint32_t prim;          // triangle index inside the primitive (PrimitiveIndex())
  uint32_t attrs;         // SHADE_HAS_*
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
  // Bilinear footprints: texQuads[i] = the four RGBA8 texels {(x, y), (x+1, y), (x, y+1), (x+1, y+1)} of the footprint whose first
  // texel is texel i of the pool, the neighbours taken under the texture's own wrap modes at upload.  There is no texture unit:
  // a bilinear tap is four gathers and two wrapCoordPair() evaluations, a trilinear one twice that, five maps per hit -- 40
  // gathers.  With the footprint stored per texel it is ONE 16-byte gather and no neighbour arithmetic; 4x the texel pool in HBM,
  // which is what the 288 GB are for.  REPEAT and CLAMP_TO_EDGE only (the neighbour under MIRRORED_REPEAT depends on the period
  // the coordinate is in); null = not built.
  const uint4*               texQuads;
  const float4*              envPixels;  // rgb + pdf
  const MiEnvAccel*          envAccel;
  const float4*              bvhNodes;  // BVH2: 4 x float4 per node (see pt_bvh.h); null when the wide BVH is active
  const uint4*               bvh8Nodes; // BVH8: 5 x uint4 per node (see pt_bvh8.h)
  const float*               bvh8Planes;  // BVH8: the nodes' quantised planes as floats, 48 per node (pt_bvh8.h: bvh8TestChildrenPlanes); may be null
  const DevTri*              tris;      // triangles in the order of the ACTIVE structure (hit records index this array)
  const DevTexRef*           texRefs;   // numTextureInfos entries
  const uint4*               coreTex;   // DevCoreTex, 5 per material (see there)
  const DevShadeTri*         shadeTris; // same indexing as tris
  const float4*              geomPool;  // the geometry pool (every DevPrim stream is a 16-byte aligned piece of it)
  const DevAlphaTri*         alphaTris; // same indexing as tris; valid for triangles of non-FORCE_OPAQUE instances
  const float*               srgbLut;  // 256 floats
  int                        numMaterials, numTextures, numLights, numNodes;
  int                        envWidth, envHeight;
  int                        numTris;
  int                        bvh8NumNodes;
  int                        bvhRoot;  // node index, or ~tri for a single-triangle scene; INT_MIN when empty
  int                        packetInterval;  // k_trace_primary: interval node test for one-pixel packets (pt_packet.h); 0 = per-ray test everywhere
  uint32_t                   shadowOctFlip;   // 7: the any-hit walks of k_trace_shadow (MODE 0 / 1 / 3: order independent) take a node's children far end first; 0: near end first
};

// ---- per-frame constants (kernel argument, ~600 B) ---------------------------------------------------------------------
// What the physical sky model derives from its parameters alone (makeSkyPrecomp, pt_light.h): the same for every ray, so it
// is computed once per parameter change by a one-thread kernel -- with the device's own arithmetic, i.e. bit-identical to
// evaluating it in place -- instead of in every evaluation (an acos, a pow, a tan, two sines and nine exponentials per call).
struct SkyPrecomp
{
  float sunDir[3], cosS;
  float thetaS, tS, T, cosTs;
  float sunE[3], Yz;
  float xz, yz, sunRadius, coneAngle;
  float coneOneMinusCos, omega, denY, denX;
  float denYy, pad0, pad1, pad2;
  float cY[5], cX[5], cYy[5], pad3;  // Perez coefficients of Y, x, y
};
struct FrameConsts
{
  MiSceneFrameInfo        frameInfo;
  MiSkyPhysicalParameters sky;
  const SkyPrecomp*       skyPre;  // device-resident, refreshed by k_sky_precomp when the sky parameters change
  MiPathtraceParams       pc;
  int                     width, height;
  int                     tileSize, tileShift;  // tileSize = 1 << tileShift, >= 16
  int                     numSlots;             // owned tiles * tileSize^2 (a multiple of QCHUNK): pixel slots of ONE frame
  // Frames in flight.  Path slots are MICRO-TILE MAJOR: slot = ((pixelSlot / 64) * numFrames + frame) * 64 + pixelSlot % 64 -- the
  // numFrames samples of an 8x8 pixel block are adjacent, so the waves that run side by side on the device work on the same few
  // pixels: the same BVH nodes, the same triangles, vertices and texels for their camera rays and first hits (round 3; before, the
  // slots were frame major -- frame * numSlots + pixelSlot -- and concurrent waves covered a tenth of a frame).  See pathSlot().
  // With a multiple of 64 frames in flight the layout goes one step further, PIXEL major (slotLayout 1): slot = pixelSlot * numFrames
  // + frame -- a wave is 64 samples of ONE pixel.  Its camera rays are the tightest packet there is, and its first hits share the
  // triangle, the vertices, the material and (within the pixel's footprint) the texels: the bounce-0 gathers of a wave touch a
  // handful of cache lines instead of 64.
  int                     numFrames;
  uint32_t                framesMagic, framesShift;  // w / numFrames == mulhi(w, framesMagic) >> framesShift for w < 2^31 (divideMagic; numFrames >= 2)
  int                     slotLayout;                // 0: micro-tile major, 1: pixel major (numFrames % 64 == 0)
  // 1: misc / throughput / radiance of a living path are records of its queue entry (RayQueue::misc / aux2 / rad); PathSoA::radiance and
  // ::misc are written once, when the path ends.  0: by path slot, as in rounds 1-3 -- frames with the shadow-catcher plane, whose
  // resolve pass ends a path from outside the shade kernel (pathtrace_functions.h.slang:520-534), and MI_PT_STATE_BY_SLOT=1 (A/B).
  int                     stateInQueue;
};

// Exact division of a 31-bit number by d >= 2 as multiply-high + shift: with k = floor(log2 d) and m = ceil(2^(32+k) / d) (< 2^32
// unless d is a power of two, which gets k - 1 and m = 2^31), mulhi(n, m) >> k = floor(n / d) for every n < 2^31: the estimate
// n m / 2^(32+k) exceeds n / d by less than n / 2^(32+k) < 1 / d.
inline void divideMagic(uint32_t d, uint32_t& magic, uint32_t& shift)
{
  uint32_t k = 31;
  while(!(d >> k))
    --k;
  if((d & (d - 1u)) == 0u)
  {
    magic = 0x80000000u;
    shift = k - 1u;
    return;
  }
  const unsigned long long p = 1ull << (32 + k);
  magic                      = uint32_t((p + d - 1ull) / d);
  shift                      = k;
}

#if defined(__HIPCC__)
// A pointer that a kernel reads out of a device-resident struct (DevScene behind `const DevScene*`, DevPrim, TexCtx) is a generic
// ("flat") pointer to the compiler: its loads are flat_load, which count on BOTH memory counters, and since LDS operations return
// out of order every use of such a load drains everything outstanding (s_waitcnt vmcnt(0) lgkmcnt(0)).  All of these tables live in
// global memory; saying so turns them into global_load with counted waits.
// gat(table, i) = table[i], read as global memory (the index is applied in the global address space, which is what keeps the
// compiler's address-space inference from folding the cast away).
template <class T, class I>
__device__ __forceinline__ const T& gat(const T* table, I i)
{
  return *(const T*)((const __attribute__((address_space(1))) T*)table + i);
}
// A descriptor (DevScene, FrameConsts, the sky tables) reaches a non-inlined helper as a reference in VECTOR registers: the helper
// cannot know that every lane holds the same address, nor that nothing writes there, and reads each field with a flat vector load
// (sampleLights: 37 of them, each an address-pipeline slot and most of them a full drain).  uniformConst() states both facts: the
// address is made scalar (readfirstlane) and the memory constant, so the fields arrive through the scalar cache in SGPRs.
template <class T>
__device__ __forceinline__ const T& uniformConst(const T& r)
{
  const unsigned long long p = reinterpret_cast<unsigned long long>(&r);
  // (the builtin returns int: widen through uint32_t, or a low half with bit 31 set sign-extends into the high half)
  const uint32_t           lo = uint32_t(__builtin_amdgcn_readfirstlane(int(uint32_t(p)))), hi = uint32_t(__builtin_amdgcn_readfirstlane(int(uint32_t(p >> 32))));
  const unsigned long long u  = (unsigned long long)lo | ((unsigned long long)hi << 32);
  return *(const T*)(const __attribute__((address_space(4))) T*)u;
}
#else  // the device headers compiled for the host (tests/host_shim)
template <class T, class I>
inline const T& gat(const T* table, I i)
{
  return table[i];
}
template <class T>
inline const T& uniformConst(const T& r)
{
  return r;
}
#endif

// slot <-> (pixel slot, frame) of the micro-tile-major layout (FrameConsts::numFrames)
#if defined(__HIPCC__)
__device__ __forceinline__ uint32_t pathSlot(const FrameConsts& fc, uint32_t pixelSlot, uint32_t frame)
{
  if(fc.slotLayout == 1)
    return pixelSlot * uint32_t(fc.numFrames) + frame;
  return ((pixelSlot >> 6) * uint32_t(fc.numFrames) + frame) * 64u + (pixelSlot & 63u);
}
__device__ __forceinline__ uint32_t pathSlotFrame(const FrameConsts& fc, uint32_t slot)
{
  if(fc.numFrames <= 1)
    return 0u;
  const uint32_t w = fc.slotLayout == 1 ? slot : (slot >> 6), q = __umulhi(w, fc.framesMagic) >> fc.framesShift;
  return w - q * uint32_t(fc.numFrames);
}
// the pixel slot of a path slot (the inverse of pathSlot in its first argument): index of the per-pixel records (PathSoA::firstHit)
__device__ __forceinline__ uint32_t pathSlotPixel(const FrameConsts& fc, uint32_t slot)
{
  if(fc.numFrames <= 1)
    return slot;
  const uint32_t w = fc.slotLayout == 1 ? slot : (slot >> 6), q = __umulhi(w, fc.framesMagic) >> fc.framesShift;
  return fc.slotLayout == 1 ? q : (q * 64u + (slot & 63u));
}
#endif

// ---- per-path state, structure of arrays indexed by slot -----------------------------------------------------------------
// Per-segment traffic (read + write) is accounted in DESIGN.md §4; keep records 16-byte sized for dwordx4 access.
enum : uint32_t
{
  PF_INSIDE     = 1u << 0,   // pt.isInside
  PF_NOT_SOLID  = 1u << 1,   // !pt.solid
  PF_ALIVE      = 1u << 2,   // still in the bounce loop
  PF_PLANE_HIT  = 1u << 3,
  PF_PRIMARY_MISS = 1u << 4, // the camera ray left the scene (k_trace_primary): `radiance` holds its DIRECTION, k_finish_sample evaluates
                             // the environment / backplate for it
  PF_DEPTH_SHIFT   = 8,      // bits 8..15  surfaceDepth
  PF_SCATTER_SHIFT = 16,     // bits 16..23 scatterBounces (saturating)
};
// the two flags k_finish_sample needs, mirrored into PathSoA::radiance.w so that it reads one record per path, not two
constexpr uint32_t RADW_NOT_SOLID    = 0x80000000u;  // sign bit of maxRoughness.x (which is >= 0)
constexpr uint32_t RADW_PRIMARY_MISS = 0xffc00001u;  // a NaN no arithmetic produces (maxRoughness.x of such a path is 0 and never read)

// `radiance` is the one record every batch has by slot (396 B per path slot went to slot-indexed state and queue entries until round 4; 252 now).
// The others exist only for the batches that need them (mi_pt_api.hip: ensureOptionalPathArrays) and are NULL otherwise -- a kernel that writes one
// outside the feature that needs it tests the pointer first.
struct PathSoA
{
  float4*   radiance;     // rgb, maxRoughness.x (>= 0) with the sign bit = !solid (RADW_NOT_SOLID); RADW_PRIMARY_MISS: rgb = the camera
                          // ray's direction (k_trace_primary) -- k_finish_sample reads this record alone.  Written when a path ENDS (its
                          // queue entry carries it while it lives), or at every bounce when the state lives by slot
  float4*   firstHit;     // firstHitPos.xyz, unused.  Per PIXEL slot (pathSlotPixel): only frame 0 of a first-frame batch has one (NDC depth)
  float4*   misc;         // maxRoughness.y, flags (uint bits), seed (uint bits), cone.width.  Multi-sample frames (the next sample starts from the
                          // seed) and state-by-slot frames; else null
  float4*   throughput;   // rgb, lastSamplePdf.  State-by-slot frames (shadow-catcher plane, MI_PT_STATE_BY_SLOT); else null
  uint4*    medium;       // VolumeMedium as 7 halves packed.  Scenes that run the generic shade kernel (volume materials); else null
  float4*   pixelSum;     // sum over the frame's samples of the clamped radiance rgba.  Multi-sample frames; else null
  float4*   guideAlbedo;  // optional (denoiser guides): sum over samples
  float4*   guideNormal;
};

// Work queues.  Every logical queue is split into NSUB sub-queues so that appends hit 16 different counter words instead
// of one (a single device-scope atomic word saturates at ~88 M ops/s on gfx950: MI355X_MICROARCH.md "dequeue" row).
// A consumer sees the concatenation of the sub-queues ("flat" index space, prefix sums of the 16 counts); work is cut
// into chunks of QCHUNK flat indices and chunk c appends its survivors to sub-queue c % NSUB, which bounds every sub-queue
// by ceil(numChunks / NSUB) * QCHUNK entries whatever the scheduling.  An entry of QUEUE_DEAD is skipped by consumers.
constexpr int      NSUB       = 16;
constexpr int      QCHUNK     = 256;
constexpr uint32_t QUEUE_DEAD = 0xffffffffu;
// Rays and hit records live IN the queues (queue-ordered arrays of NSUB x subCap entries), not in the slot-indexed path
// state: the trace kernels then read their rays with unit-stride, single-level loads and never touch the path state.
struct RayQueue
{
  uint32_t* slot;  // path slot of the entry, or QUEUE_DEAD
  float4*   org;   // origin.xyz, (closest: unused | shadow: distance to the light)
  float4*   dir;   // direction.xyz, (closest: unused | shadow: uint bits, bit0 = initialInside)
  float4*   aux;   // closest: hit record written by k_trace_closest (t, triangle index bits or -1, u, v)
                   // shadow : contribution.rgb, seed bits (input of k_trace_shadow)
  float4*   aux2;  // shadow queue: shadow-catcher entries only (dir.w bit1): unshadowed term.rgb, position of the path's
                   // continuation entry in the next active queue (or 0xffffffff).  Active queues: the path's throughput.rgb + lastSamplePdf
  // Active queues only -- the path's state TRAVELS WITH ITS RAY (FrameConsts::stateInQueue, round 4): what PathSoA::misc / ::radiance hold by
  // slot, written by the shade launch that appends the entry and read by the next one as a unit-stride stream.  By the later bounces
  // the surviving paths are sparse in slot space: three 16-byte gathers by slot touched up to three cache lines per path (and three
  // scattered stores), where the entry's records now arrive in the order the wave processes them.
  float4*   misc;  // maxRoughness.y, flags (uint bits), seed (uint bits), cone.width
  float4*   rad;   // radiance so far .rgb, maxRoughness.x | RADW_NOT_SOLID -- k_shadow_resolve adds a shadow ray's term HERE while the path lives
};
// slot field of a SHADOW queue entry when the path's state is in the queue: where k_shadow_resolve adds the ray's term -- bit 31 set:
// entry (slot & 0x7fffffff) of the NEXT active queue's `rad` (the path goes on); clear: PathSoA::radiance[slot] (the path ended with
// this bounce, its radiance already lies where k_finish_sample reads it)
constexpr uint32_t SHADOW_TARGET_QUEUE = 0x80000000u;
struct Queues
{
  RayQueue  active[2];  // paths to trace/shade this iteration, ping-pong
  RayQueue  shadow;     // pending shadow rays
  uint32_t* counters;   // see QC_* indices
  uint32_t  subCap;
  // Shadow rays through transmissive instances (scenes with such instances on the 8-wide BVH, k_trace_shadow MODE 3): the any-hit
  // walk records every transmissive candidate {t, u, v, triangle} in `candPool` -- entries of one ray chained through `candNext`
  // -- and leaves the chain's head in the ray's queue entry; k_shadow_resolve then takes each ray's candidates in increasing
  // (t, renderNode, primitive) order (raytracer_interface.h.slang:160-178).  A ray whose candidates did not fit the pool is listed
  // in `overflow` and re-traced by the ordered-search kernel (MODE 2).  Null when unused.
  float4*   candPool;
  uint32_t* candNext;
  uint32_t* overflow;
  uint32_t  candCap;
};
// Outcome of a shadow ray's walk, left in org.w of its queue entry for k_shadow_resolve: the first recorded candidate, or
constexpr uint32_t CAND_NIL        = 0xffffffffu;  // unoccluded, nothing recorded
constexpr uint32_t SHADOW_OCCLUDED = 0xfffffffeu;
constexpr uint32_t SHADOW_DIR_OVERFLOW = 4u;  // dir.w bit 2: candidates did not fit the pool, the ordered-search kernel re-traces the ray (org.w keeps tmax)
enum : int
{
  // Queue tails.  PAIR q holds, per sub-queue s, two adjacent words: [2s] = entries appended to active queue q, [2s+1] =
  // shadow rays appended by the same shade launch -- so that one 64-bit atomic per block and chunk reserves space in both
  // queues (a device-scope atomic on these lines is served beyond the XCD's L2 and costs microseconds).  Every group lives
  // in its own 128-byte line(s): a kernel may reset one group with plain stores while other waves run atomics on another,
  // and the XCDs' L2s are not coherent with each other.
  QC_PAIR0        = 0,    // 2 * NSUB words
  QC_PAIR1        = 32,   // 2 * NSUB words
  QC_HEADS_TRACE  = 96,   // 8 dynamic-fetch heads (one per XCD by convention) of k_trace_closest
  QC_HEADS_SHADOW = 128,  // 8 heads of k_trace_shadow
  QC_HEADS_OVERFLOW = 160,  // 8 heads of the ordered-search kernel over the overflow list
  QC_CAND_POOL      = 192,  // entries handed out of candPool
  QC_RESOLVE        = 224,  // (unused)
  QC_OVERFLOW       = 256,  // entries of the overflow list
  QC_COUNT          = 288
};

struct StatCounters  // device mirror of MiPtStats' dynamic part
{
  unsigned long long cameraPaths, segments, shadowRays, nodesClosest, trisClosest, nodesShadow, trisShadow, textureTaps, surfaceHits, nodesPrimary, trisPrimary;
};

}  // namespace pt
