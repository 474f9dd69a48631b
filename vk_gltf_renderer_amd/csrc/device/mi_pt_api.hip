// C-ABI of libmi_pt.so (include/mi_pt.h): device resource management and the per-frame wavefront schedule.
// Host counterpart in the reference: PathTracer::{onAttach,onResize,onRender,setupPushConstant,renderRayQuery}
// (src/renderer_pathtracer.cpp:150-260, :500-614, :1404-1431, :1496-1574) plus the uploads of SceneVk / SceneRtx.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "mi_pt.h"
#include "pt_build.h"
#include "pt_bvh.h"
#include "pt_kernels.h"

namespace {

thread_local std::string g_lastError;

int fail(int code, const std::string& msg)
{
  g_lastError = msg;
  return code;
}

#define HIP_TRY(x)                                                                                                       \
  do                                                                                                                    \
  {                                                                                                                     \
    hipError_t e_ = (x);                                                                                                \
    if(e_ != hipSuccess)                                                                                                \
      return fail(MI_PT_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(e_));                                       \
  } while(0)

template <typename T>
struct DevBuf
{
  T*     ptr   = nullptr;
  size_t count = 0;
  hipError_t alloc(size_t n)
  {
    release();
    count = n;
    if(n == 0)
      return hipSuccess;
    return hipMalloc(reinterpret_cast<void**>(&ptr), n * sizeof(T));
  }
  hipError_t upload(const T* src, size_t n)
  {
    hipError_t e = alloc(n);
    if(e != hipSuccess || n == 0)
      return e;
    return hipMemcpy(ptr, src, n * sizeof(T), hipMemcpyHostToDevice);
  }
  void release()
  {
    if(ptr)
      (void)hipFree(ptr);
    ptr   = nullptr;
    count = 0;
  }
  ~DevBuf() { release(); }
};

enum TimedKernel { TK_GENERATE, TK_TRACE, TK_SORT, TK_SHADE, TK_SHADOW, TK_ACCUM, TK_PRIMARY, TK_SHADE_FIRST, TK_COUNT };

}  // namespace

// Run-time switches (A/B and diagnostics; INTEGRATION.md lists them): read from the environment ONCE, in mi_pt_create, so that a
// variable that appears or changes later cannot alter an instance's slot layout, kernels or scene in the middle of an accumulation.
struct RunSwitches
{
  int    packetInterval = 1;       // MI_PT_PACKET_INTERVAL  0: per-ray node test in every camera-ray packet
  int    sortMode       = 2;       // MI_PT_SORT             window sort of the generic shade kernel: 0 | 1 | 2
  bool   noPacket       = false;   // MI_PT_NO_PACKET        k_generate + per-lane walk instead of k_trace_primary
  bool   stateBySlot    = false;   // MI_PT_STATE_BY_SLOT    path state gathered by slot in every launch (rounds 1-3) instead of travelling in the queue entry
  bool   traceSpans     = false;   // MI_PT_TRACE_SPANS      synchronising diagnostics
  int    overlapUpTo    = 32;      // MI_PT_OVERLAP          batches up to this many frames run a bounce's shadow stage on a second stream, next to the
                                   //                        following bounce's closest-hit walk (0 = one stream, as in rounds 1-3); same image
  int    overlapMinTris = 200000;  // MI_PT_OVERLAP_MIN_TRIS ... in scenes of at least this many (flattened) triangles
  bool   noPlanes       = false;   // MI_PT_DIAG_NO_PLANES   no float planes for the packet walk
  bool   noOpaqueTris   = false;   // MI_PT_DIAG_NO_OPAQUE_TRIS  alpha-test the OPAQUE class of the alpha cut too
  bool   noQuads        = false;   // MI_PT_DIAG_NO_QUADS    four texel gathers per bilinear tap instead of one footprint
  bool   shadowFarFirst = true;    // MI_PT_SHADOW_FAR_FIRST=0 any-hit shadow walks (k_trace_shadow MODE 0 / 1 / 3: order independent) take a node's children near end first,
                                   //                        as in rounds 1-4.  Default since round 5: from the ray's FAR end -- a ray that starts on a surface wades through the
                                   //                        boxes around its origin, its occluder is usually the large thing at the other end.  Same image bit for bit
                                   //                        (test_shadow_walk_from_the_far_end_changes_no_bit); atrium 604 -> 656, street 623 -> 661 Msamples/s (profiles/r05_staged_ab.txt)
  int    reinsert       = 16;      // MI_PT_REINSERT         reinsertion passes over the BVH2 before the 8-wide collapse at scene build (bvh_reinsert.h); 0 = the tree as clustered.
                                   //                        Same image bit for bit (test_reinsertion_changes_the_tree_not_the_image), a tenth fewer node visits per ray:
                                   //                        atrium 604 -> 633, street 623 -> 680 Msamples/s; both switches: 682 / 727 (profiles/r05_staged_ab.txt)
  int    reinsertUpdate = 0;       // MI_PT_REINSERT_UPDATE  ... at the rebuilds of mi_pt_update_render_nodes: 0, because a moving instance pays them every time and a pass costs
                                   //                        more than half of what the rest of a rebuild costs (2.8 M triangles: rebuild 15.7 ms, + 10 ms per pass;
                                   //                        profiles/r05_sweep_and_rebuild.txt).  A posed scene that is then accumulated for many frames
                                   //                        can ask for them
  int    reinsertRounds = 4;       // MI_PT_REINSERT_ROUNDS  lock / move rounds per pass
  float  splitFactor    = 4.0f;    // MI_PT_SPLIT            triangle pre-splitting (bvh_split.h): a triangle whose box area exceeds this x the scene's mean gets one
                                   //                        reference per part of a recursive bisection of its box; 0 = one reference per triangle (rounds 1-5).  Same image
                                   //                        bit for bit; sliver stand-in of the atrium 30.5 -> 9.5 triangle tests per secondary ray, 419 -> 594 Msamples/s
                                   //                        (0 / 64 / 16 / 4 / 1: 419 / 533 / 590 / 594 / 614; street sliver 679 / 703 / 719 / 723 / 696: profiles/r06_split_ab.txt)
  int    splitMaxDepth  = 8;       // MI_PT_SPLIT_DEPTH      ... at most 2^this references per triangle
  float  splitMinShare  = 0.1f;    // MI_PT_SPLIT_MIN_SHARE  ... only in scenes where triangles above 64 x the mean hold this share of the summed box area (0 = always):
                                   //                        the evenly tessellated stand-ins (share 0 / 0.05) keep their round-5 trees -- splitting perturbs them by -1 %
  bool   collapseGreedy = false;   // MI_PT_COLLAPSE=sah|greedy  how BVH2 subtrees become children of an 8-wide node (anything else: mi_pt_create fails)
  bool   collapseBad    = false;
  bool   hostCollapse   = false;   // MI_PT_HOST_COLLAPSE    collapse on the host (the greedy reference of the device collapse)
  bool   coreTex        = true;    // MI_PT_CORE_TEX=0: A/B switch -- the per-material slot records (pt_scene.h: DevCoreTex) send every fetch the general way
  int    maxItersDiag   = 0;       // MI_PT_DIAG_MAX_ITERS=N test hook: the bounce loop stops after N iterations whatever is still alive (the truncation a
                                   //                        volume-scatter scene meets at maxDepth * 66 + 512)
  int    failBuildAt    = 0;       // MI_PT_DIAG_FAIL_BUILD=N  test hook: the N-th acceleration REbuild of the instance fails after the old structure is gone
  bool   candPoolSet    = false;   // MI_PT_DIAG_CAND_POOL   entries of the transmissive-candidate pool (tests of the overflow path)
  size_t candPool       = 0;
  void   read()
  {
    auto flag = [](const char* n) { const char* e = getenv(n); return e != nullptr; };
    auto num  = [](const char* n, int d) { const char* e = getenv(n); return e ? atoi(e) : d; };
    packetInterval = num("MI_PT_PACKET_INTERVAL", 1);
    sortMode       = num("MI_PT_SORT", 2);
    noPacket       = flag("MI_PT_NO_PACKET");
    stateBySlot    = flag("MI_PT_STATE_BY_SLOT");
    traceSpans     = flag("MI_PT_TRACE_SPANS");
    overlapUpTo    = num("MI_PT_OVERLAP", 32);
    overlapMinTris = num("MI_PT_OVERLAP_MIN_TRIS", 200000);
    noPlanes       = flag("MI_PT_DIAG_NO_PLANES");
    noOpaqueTris   = flag("MI_PT_DIAG_NO_OPAQUE_TRIS");
    noQuads        = flag("MI_PT_DIAG_NO_QUADS");
    shadowFarFirst = num("MI_PT_SHADOW_FAR_FIRST", 1) != 0;
    reinsert       = std::max(0, num("MI_PT_REINSERT", 16));
    reinsertUpdate = std::max(0, num("MI_PT_REINSERT_UPDATE", 0));
    reinsertRounds = std::max(1, num("MI_PT_REINSERT_ROUNDS", 4));
    if(const char* e = getenv("MI_PT_SPLIT"))
      splitFactor = std::max(0.0f, float(atof(e)));
    if(const char* e = getenv("MI_PT_SPLIT_MIN_SHARE"))
      splitMinShare = std::max(0.0f, float(atof(e)));
    splitMaxDepth  = std::min(std::max(0, num("MI_PT_SPLIT_DEPTH", 8)), 10);  // (bvh_split.h: SPLIT_MAX_DEPTH)
    if(const char* e = getenv("MI_PT_COLLAPSE"))
    {
      collapseGreedy = strcmp(e, "greedy") == 0;
      collapseBad    = !collapseGreedy && strcmp(e, "sah") != 0;  // (a typo must not silently select the other collapse)
    }
    hostCollapse   = flag("MI_PT_HOST_COLLAPSE");
    maxItersDiag   = num("MI_PT_DIAG_MAX_ITERS", 0);
    coreTex        = num("MI_PT_CORE_TEX", 1) != 0;
    failBuildAt    = num("MI_PT_DIAG_FAIL_BUILD", 0);
    if(const char* e = getenv("MI_PT_DIAG_CAND_POOL"))
    {
      candPoolSet = true;
      candPool    = size_t(strtoull(e, nullptr, 10));
    }
  }
};

struct MiPt
{
  int device = 0;
  int numCUs = 256;
  RunSwitches sw;
  int accelBuilds = 0;  // acceleration-structure builds of this instance so far (0 while the first one runs)
  // mi_pt_set_frame_queue: consecutive mi_pt_render_frame calls held back and issued together (flushPending)
  int               frameQueueDepth = 1;
  int               pendingFrames   = 0;
  MiPathtraceParams pendingFirst{};
  void*             pendingStream   = nullptr;
  // scene
  DevBuf<MiGltfShadeMaterial> materials;
  DevBuf<MiGltfTextureInfo>   texInfos;
  DevBuf<pt::DevTexRef>       texRefs;
  DevBuf<pt::DevCoreTex>      coreTex;   // 5 per material (pt_scene.h)
  DevBuf<MiGltfRenderNode>    nodes;
  DevBuf<pt::DevPrim>         prims;
  DevBuf<MiGltfLight>         lights;
  DevBuf<pt::DevTexture>      textures;
  DevBuf<uchar4>              texels;
  DevBuf<uint4>               texQuads;  // bilinear footprints, one per texel of the pool (DevScene::texQuads)
  DevBuf<uint8_t>             geometry;  // all index / attribute streams, 16-byte aligned sub-allocations
  DevBuf<uint8_t>             instFlags;
  DevBuf<float>               srgbLut;
  DevBuf<float4>              envPixels;
  DevBuf<MiEnvAccel>          envAccel;
  DevBuf<pt::DevAlphaTri>     alphaTris;
  DevBuf<pt::DevShadeTri>     shadeTris;
  float4*                     bvhNodes  = nullptr;
  uint4*                      bvh8Nodes = nullptr;
  DevBuf<float>               bvh8Planes;  // the nodes' planes as floats for the packet walk (DevScene::bvh8Planes)
  pt::DevTri*                 bvhTris   = nullptr;
  bool                        wide      = true;
  pt::DevScene                scene{};
  bool                        hasAlpha = false, hasAlphaTest = false, hasVolumeScatter = false, simpleMaterials = true, hasTransmissive = false;
  // device-resident descriptor copies (see k_shade): the scene struct, re-uploaded when the environment changes, and a ring
  // of per-batch frame constants fed from pinned host memory
  static constexpr int        FC_RING = 32;
  DevBuf<pt::DevScene>        sceneDev;
  bool                        sceneDevDirty = true;
  DevBuf<pt::FrameConsts>     fcRing;
  DevBuf<pt::SkyPrecomp>      skyPre;  // constants of the sky model for `skyPreFor` (k_sky_precomp)
  MiSkyPhysicalParameters     skyPreFor{};
  bool                        skyPreValid = false;
  pt::FrameConsts*            fcHost = nullptr;  // pinned, FC_RING entries
  hipEvent_t                  fcDone[FC_RING] = {};
  unsigned                    fcCursor = 0;
  MiPtStats                   staticStats{};
  // host copies of what the acceleration structure is built from, kept so that mi_pt_update_render_nodes can rebuild it
  std::vector<MiGltfRenderNode> hostNodes;
  std::vector<uint8_t>          hostVisible;    // empty = all visible
  std::vector<uint8_t>          matInstFlags;   // per material: INST_FORCE_OPAQUE | INST_CULL_DISABLE | INST_TRANSMISSIVE | INST_ALPHA_PASSES
  std::vector<uint8_t>          matFeatures;    // per material: bit0 volume scatter, bit1 needs the generic shade kernel
  std::vector<uint32_t>         primTriangles;  // per render primitive: triangle count, 0 when it has no usable geometry
  int                           bvhBuilder = 0;
  // frame state
  int                     width = 0, height = 0;
  int                     tileRank = 0, tileWorld = 1, tileSize = 64;
  MiSceneFrameInfo        frameInfo{};
  MiSkyPhysicalParameters sky{};
  bool                    haveFrameInfo = false;
  DevBuf<uint32_t>        ownedTiles;
  int                     numSlots = 0, tilesX = 0, tilesY = 0;
  int                     framesCap = 1;  // frames in flight the path/queue arrays are sized for
  size_t                  queueCap  = 0;  // positions of one queue (NSUB x subCap)
  DevBuf<float4>          pathArrays;  // PathSoA::radiance: the one by-slot record every batch needs
  // ... and the by-slot records only some batches need, allocated the first time one does (ensureOptionalPathArrays): `optMisc` + `optPixelSum`
  // multi-sample frames, `optThroughput` (+ optMisc) frames whose state lives by slot (shadow-catcher plane, MI_PT_STATE_BY_SLOT), `optMedium` scenes with
  // volume materials (the generic shade kernel), `optGuides` batches that capture the denoiser guides, `optShadowAux2` shadow-catcher frames
  DevBuf<float4>          optThroughput, optMisc, optMedium, optPixelSum, optGuides, optShadowAux2;
  DevBuf<float4>          firstHit;    // PathSoA::firstHit: per PIXEL slot (frame 0 of a first-frame batch only)
  pt::PathSoA             paths{};
  DevBuf<uint32_t>        queueMem;
  DevBuf<float4>          queuePayload;
  DevBuf<float4>          candPool;   // recorded transmissive shadow candidates (Queues::candPool)
  DevBuf<uint32_t>        candLists;  // candNext + overflow list
  pt::Queues              queues{};
  DevBuf<float4>          accumOwn, albedo, normal, denoiseA, denoiseB;
  float4*                 albedoBound = nullptr;  // caller-owned guide / depth images (mi_pt_bind_guides), NULL = the internal ones
  float4*                 normalBound = nullptr;
  float*                  depthBound  = nullptr;
  float4*                 albedoImg() { return albedoBound ? albedoBound : albedo.ptr; }
  float4*                 normalImg() { return normalBound ? normalBound : normal.ptr; }
  float*                  depthImg() { return depthBound ? depthBound : depth.ptr; }
  float                   accumFrames = 0.0f;  // frames folded into the accumulator (variance of the mean, SVGF pass)
  float                   momentFrames = 0.0f; // frames folded into the luminance second moment (normal.w): only batches rendered with the guides on
                                               // feed it, so it may lag behind accumFrames -- the SVGF pass then falls back to its spatial variance
  const float4*           denoised = nullptr;  // result of the last mi_pt_denoise (one of denoiseA / denoiseB)
  DevBuf<uint32_t>        tonemapped, tmHistogram;
  DevBuf<float>           tmAutoState;
  float4*                 accum = nullptr;  // accumOwn.ptr or caller-bound memory
  DevBuf<float>           depth;
  DevBuf<uint32_t>        selection;
  DevBuf<pt::StatCounters> stats;
  bool                    collectCounters = false;
  bool                    timingEnabled   = false;
  // deferred timing: events are recorded per launch and only resolved in mi_pt_get_frame_timing (no per-frame sync)
  struct Span
  {
    int        kind;
    hipEvent_t a, b;
  };
  std::vector<Span>       pendingSpans;
  size_t                  evCursor = 0;
  MiPtFrameTiming         accTiming{};
  std::vector<hipEvent_t> eventPool;
  hipStream_t             lastStream = nullptr;
  hipStream_t             sideStream = nullptr;                    // the shadow stage of small batches (RunSwitches::overlapUpTo)
  hipEvent_t              evShaded = nullptr, evShadowed = nullptr;  // main -> side after a shade launch, side -> main after the shadow stage

  ~MiPt()
  {
    if(bvhNodes)
      (void)hipFree(bvhNodes);
    if(bvhTris)
      (void)hipFree(bvhTris);
    if(bvh8Nodes)
      (void)hipFree(bvh8Nodes);
    for(hipEvent_t e : eventPool)
      (void)hipEventDestroy(e);
    for(hipEvent_t e : fcDone)
      if(e)
        (void)hipEventDestroy(e);
    if(fcHost)
      (void)hipHostFree(fcHost);
    if(sideStream)
      (void)hipStreamDestroy(sideStream);
    if(evShaded)
      (void)hipEventDestroy(evShaded);
    if(evShadowed)
      (void)hipEventDestroy(evShadowed);
  }
};

namespace {

size_t align16(size_t v) { return (v + 15u) & ~size_t(15); }

// Path state and ray queues for `frames` frames in flight (path slots are micro-tile major: pt::pathSlot).
int allocPathResources(MiPt* pt, int frames)
{
  if(size_t(pt->numSlots) * size_t(frames) >= 0x7fffffffull)
    return fail(MI_PT_ERR_ARGUMENT, "too many path slots: reduce the frames in flight or the resolution");
  pt->framesCap          = frames;
  const size_t n         = std::max(size_t(pt->numSlots) * size_t(frames), size_t(1));
  // By slot: the radiance record alone (what k_finish_sample folds; a path's other state travels in its queue entry).  The optional
  // records of the previous size are dropped and come back on demand.
  HIP_TRY(pt->pathArrays.alloc(n));
  HIP_TRY(pt->firstHit.alloc(std::max(size_t(pt->numSlots), size_t(1))));
  pt->optThroughput.release(); pt->optMisc.release(); pt->optMedium.release(); pt->optPixelSum.release(); pt->optGuides.release(); pt->optShadowAux2.release();
  pt::PathSoA& P = pt->paths;
  P              = pt::PathSoA{};
  P.radiance     = pt->pathArrays.ptr;
  P.firstHit     = pt->firstHit.ptr;
  // sub-queue capacity: ceil(numChunks / NSUB) chunks (+1 of slack), see pt_scene.h
  const size_t numChunks = (n + pt::QCHUNK - 1) / pt::QCHUNK;
  const size_t subCap    = ((numChunks + pt::NSUB - 1) / pt::NSUB + 1) * pt::QCHUNK;
  const size_t qsize     = subCap * pt::NSUB;
  // path slots and queue positions are 31-bit: bit 31 of a shadow entry's slot field tells the two apart (pt_scene.h: SHADOW_TARGET_QUEUE),
  // 0xffffffff is QUEUE_DEAD, and the frame index of a slot is a 31-bit multiply-high (FrameConsts::framesMagic)
  if(qsize >= (size_t(1) << 31))
    return fail(MI_PT_ERR_ARGUMENT, "mi_pt_render_frames: " + std::to_string(frames) + " frames in flight x " + std::to_string(pt->numSlots)
                                        + " pixel slots exceed 2^31 path slots");
  HIP_TRY(pt->queueMem.alloc(qsize * 3 + pt::QC_COUNT));
  // 14 records of 16 bytes per queue position: the two active queues' ray + state (org, dir, aux2, misc, rad), ONE hit record shared by both (the
  // closest-hit walk writes it, the shade launch of the same bounce reads it, and the next walk -- of the other queue -- starts after that launch:
  // the two queues' hit records are never alive together), the shadow queue's ray + contribution (org, dir, aux).  The shadow queue's aux2 exists
  // on shadow-catcher frames only (optShadowAux2).
  HIP_TRY(pt->queuePayload.alloc(qsize * 14));
  pt->queueCap = qsize;
  pt::RayQueue* qs[3] = {&pt->queues.active[0], &pt->queues.active[1], &pt->queues.shadow};
  for(int i = 0; i < 3; ++i)
    qs[i]->slot = pt->queueMem.ptr + qsize * size_t(i);
  for(int i = 0; i < 2; ++i)  // the living paths' ray and state, in queue order (pt_scene.h: RayQueue, FrameConsts::stateInQueue)
  {
    qs[i]->org  = pt->queuePayload.ptr + qsize * size_t(5 * i + 0);
    qs[i]->dir  = pt->queuePayload.ptr + qsize * size_t(5 * i + 1);
    qs[i]->aux2 = pt->queuePayload.ptr + qsize * size_t(5 * i + 2);
    qs[i]->misc = pt->queuePayload.ptr + qsize * size_t(5 * i + 3);
    qs[i]->rad  = pt->queuePayload.ptr + qsize * size_t(5 * i + 4);
    qs[i]->aux  = pt->queuePayload.ptr + qsize * 10;
  }
  pt->queues.shadow.org  = pt->queuePayload.ptr + qsize * 11;
  pt->queues.shadow.dir  = pt->queuePayload.ptr + qsize * 12;
  pt->queues.shadow.aux  = pt->queuePayload.ptr + qsize * 13;
  pt->queues.shadow.aux2 = nullptr;
  pt->queues.shadow.misc = pt->queues.shadow.rad = nullptr;
  pt->queues.counters = pt->queueMem.ptr + 3 * qsize;
  pt->queues.subCap   = uint32_t(subCap);
  // recorded transmissive shadow candidates (pt_scene.h): two pool entries per shadow-queue entry, and an overflow list as long as
  // the queue.  MI_PT_DIAG_CAND_POOL=<entries> shrinks the pool (tests of the overflow path); 0 disables recording.
  pt->queues.candPool = nullptr; pt->queues.candNext = nullptr; pt->queues.overflow = nullptr;
  pt->queues.candCap  = 0;
  size_t candCap = qsize * 2;
  if(pt->sw.candPoolSet)
    candCap = pt->sw.candPool;
  if(pt->hasTransmissive && pt->wide && candCap > 0)
  {
    candCap = std::min(candCap, size_t(0x7fffffff));
    HIP_TRY(pt->candPool.alloc(candCap));
    HIP_TRY(pt->candLists.alloc(candCap + qsize));
    pt->queues.candPool = pt->candPool.ptr;
    pt->queues.candNext = pt->candLists.ptr;
    pt->queues.overflow = pt->candLists.ptr + candCap;
    pt->queues.candCap  = uint32_t(candCap);
  }
  HIP_TRY(hipMemset(pt->queues.counters, 0, sizeof(uint32_t) * pt::QC_COUNT));
  return MI_PT_OK;
}

// The by-slot / by-position records only some batches need (MiPt::opt*): allocated the first time a batch needs them, at the size of the
// current path resources, and kept until those are reallocated.  An allocation here synchronises the device once.
int ensureOptionalPathArrays(MiPt* pt, bool stateBySlot, bool multiSample, bool guides, bool catcher)
{
  const size_t n = std::max(size_t(pt->numSlots) * size_t(pt->framesCap), size_t(1));
  auto need = [&](DevBuf<float4>& b, size_t count) -> int {
    if(b.ptr)
      return MI_PT_OK;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(b.alloc(count));
    HIP_TRY(hipMemset(b.ptr, 0, count * sizeof(float4)));
    return MI_PT_OK;
  };
  pt::PathSoA& P = pt->paths;
  if(stateBySlot || multiSample)
  {
    if(int rc = need(pt->optMisc, n)) return rc;
    P.misc = pt->optMisc.ptr;
  }
  if(stateBySlot)
  {
    if(int rc = need(pt->optThroughput, n)) return rc;
    P.throughput = pt->optThroughput.ptr;
  }
  if(multiSample)
  {
    if(int rc = need(pt->optPixelSum, n)) return rc;
    P.pixelSum = pt->optPixelSum.ptr;
  }
  if(!pt->simpleMaterials)  // the generic shade kernel keeps the medium a path is inside of
  {
    if(int rc = need(pt->optMedium, n)) return rc;
    P.medium = reinterpret_cast<uint4*>(pt->optMedium.ptr);
  }
  if(guides)
  {
    if(int rc = need(pt->optGuides, 2 * n)) return rc;
    P.guideAlbedo = pt->optGuides.ptr;
    P.guideNormal = pt->optGuides.ptr + n;
  }
  if(catcher)
  {
    if(int rc = need(pt->optShadowAux2, pt->queueCap)) return rc;
    pt->queues.shadow.aux2 = pt->optShadowAux2.ptr;
  }
  return MI_PT_OK;
}

int allocFrameResources(MiPt* pt)
{
  const int W = pt->width, H = pt->height, T = pt->tileSize;
  pt->tilesX = (W + T - 1) / T;
  pt->tilesY = (H + T - 1) / T;
  std::vector<uint32_t> owned;  // pixel origin x0 | y0 << 16 of every tile this rank renders
  for(int t = 0; t < pt->tilesX * pt->tilesY; ++t)
    if(pt->tileWorld <= 1 || (t % pt->tileWorld) == pt->tileRank)
      owned.push_back(uint32_t((t % pt->tilesX) * T) | (uint32_t((t / pt->tilesX) * T) << 16));
  pt->numSlots = int(owned.size()) * T * T;
  HIP_TRY(pt->ownedTiles.upload(owned.data(), owned.size()));
  if(int rc = allocPathResources(pt, pt->framesCap))
    return rc;
  const size_t px = size_t(W) * size_t(H);
  HIP_TRY(pt->accumOwn.alloc(px));
  HIP_TRY(hipMemset(pt->accumOwn.ptr, 0, px * sizeof(float4)));
  pt->accum = pt->accumOwn.ptr;
  HIP_TRY(pt->albedo.alloc(px));
  HIP_TRY(pt->normal.alloc(px));
  HIP_TRY(hipMemset(pt->albedo.ptr, 0, px * sizeof(float4)));
  HIP_TRY(hipMemset(pt->normal.ptr, 0, px * sizeof(float4)));
  HIP_TRY(pt->depth.alloc(px));
  HIP_TRY(pt->selection.alloc(px));
  HIP_TRY(hipMemset(pt->selection.ptr, 0, px * sizeof(uint32_t)));
  {
    std::vector<float> ones(px, 1.0f);
    HIP_TRY(hipMemcpy(pt->depth.ptr, ones.data(), px * sizeof(float), hipMemcpyHostToDevice));
  }
  pt->denoiseA.release();
  pt->denoiseB.release();
  pt->tonemapped.release();
  pt->tmHistogram.release();
  pt->tmAutoState.release();
  return MI_PT_OK;
}

hipEvent_t getEvent(MiPt* pt, size_t& cursor)
{
  if(cursor >= pt->eventPool.size())
  {
    hipEvent_t e;
    (void)hipEventCreate(&e);
    pt->eventPool.push_back(e);
  }
  return pt->eventPool[cursor++];
}


// Flattens the visible render-node instances to world-space triangles and builds the acceleration structure over them on the
// device: Morton sort + PLOC -> BVH2 -> (default) 8-wide collapse, then the per-triangle shading / alpha records.  Everything it
// needs is resident (geometry pool, materials) or kept on the host in MiPt (node matrices, visibility), so it serves the scene
// build (mi_pt_create; reference: SceneRtx BLAS + TLAS build, src/gltf_scene_rtx.cpp:173-385) and the transform / visibility
// updates of animated scenes (mi_pt_update_render_nodes; reference: TLAS update, src/gltf_scene_transform_vk.cpp:534-639) alike.
int buildAccelerationUnguarded(MiPt* pt);
// A failed (re)build must not leave the scene descriptor pointing at freed or half-built arrays: mi_pt_update_render_nodes runs
// this once per animated frame, and the next mi_pt_render_frame would walk them.  On any error the instance falls back to an
// EMPTY structure (frames render the environment only) and the error is returned to the caller.
int buildAcceleration(MiPt* pt)
{
  const int rc = buildAccelerationUnguarded(pt);
  if(rc != MI_PT_OK)
  {
    const std::string why = g_lastError;  // (the frees below must not disturb the message)
    if(pt->bvhNodes) (void)hipFree(pt->bvhNodes);
    if(pt->bvhTris) (void)hipFree(pt->bvhTris);
    if(pt->bvh8Nodes) (void)hipFree(pt->bvh8Nodes);
    pt->bvhNodes = nullptr; pt->bvhTris = nullptr; pt->bvh8Nodes = nullptr;
    pt->bvh8Planes.release(); pt->shadeTris.release(); pt->alphaTris.release();
    pt::DevScene& S = pt->scene;
    S.bvhNodes = nullptr; S.bvh8Nodes = nullptr; S.tris = nullptr; S.bvh8Planes = nullptr; S.shadeTris = nullptr; S.alphaTris = nullptr;
    S.numTris = 0; S.bvh8NumNodes = 0; S.bvhRoot = pt::BVH_EMPTY;
    pt->staticStats.bvhNodeCount = pt->staticStats.bvhTriangleCount = 0;
    pt->sceneDevDirty = true;
    g_lastError = why;
  }
  return rc;
}
int buildAccelerationUnguarded(MiPt* pt)
{
  const int numNodes = int(pt->hostNodes.size()), numMaterials = int(pt->matInstFlags.size()), numPrims = int(pt->primTriangles.size());
  std::vector<uint8_t>  flags(size_t(std::max(numNodes, 1)), 0);
  std::vector<uint32_t> triOffset;
  std::vector<int32_t>  entryNode;
  uint64_t              totalTris = 0;
  triOffset.push_back(0);
  pt->hasAlpha = pt->hasAlphaTest = pt->hasTransmissive = pt->hasVolumeScatter = false;
  pt->simpleMaterials = true;
  for(int n = 0; n < numNodes; ++n)
  {
    const MiGltfRenderNode& rn = pt->hostNodes[size_t(n)];
    const int               m  = std::max(0, std::min(rn.materialID, numMaterials - 1));
    uint32_t                f  = pt->matInstFlags[size_t(m)];
    if(!(f & pt::INST_FORCE_OPAQUE))
      pt->hasAlpha = true;
    if(!(f & (pt::INST_FORCE_OPAQUE | pt::INST_ALPHA_PASSES)))
      pt->hasAlphaTest = true;  // some candidate's alpha test has an open outcome (MASK / BLEND materials)
    if(f & pt::INST_TRANSMISSIVE)
      pt->hasTransmissive = true;
    if(pt->matFeatures[size_t(m)] & 1u)
      pt->hasVolumeScatter = true;
    if(pt->matFeatures[size_t(m)] & 2u)
      pt->simpleMaterials = false;
    const float* M   = rn.objectToWorld;
    float        det = M[0] * (M[5] * M[10] - M[9] * M[6]) - M[4] * (M[1] * M[10] - M[9] * M[2]) + M[8] * (M[1] * M[6] - M[5] * M[2]);
    if(det < 0.0f)
      f |= pt::INST_FLIP_FACING;
    flags[size_t(n)] = uint8_t(f);
    if(!pt->hostVisible.empty() && !pt->hostVisible[size_t(n)])
      continue;  // invisible nodes get no geometry (reference: src/gltf_scene_rtx.cpp:319-323)
    if(rn.renderPrimID < 0 || rn.renderPrimID >= numPrims || pt->primTriangles[size_t(rn.renderPrimID)] == 0)
      continue;
    totalTris += pt->primTriangles[size_t(rn.renderPrimID)];
    if(totalTris > 0x7fffffffull)
      return fail(MI_PT_ERR_ARGUMENT, "more than 2^31 flattened triangles");
    entryNode.push_back(n);
    triOffset.push_back(uint32_t(totalTris));
  }
  HIP_TRY(pt->nodes.upload(pt->hostNodes.data(), pt->hostNodes.size()));
  HIP_TRY(pt->instFlags.upload(flags.data(), flags.size()));
  pt->scene.nodes = pt->nodes.ptr;

  // release the previous structure (an update), then build
  if(pt->bvhNodes) (void)hipFree(pt->bvhNodes);
  if(pt->bvhTris) (void)hipFree(pt->bvhTris);
  if(pt->bvh8Nodes) (void)hipFree(pt->bvh8Nodes);
  pt->bvhNodes = nullptr; pt->bvhTris = nullptr; pt->bvh8Nodes = nullptr;
  if(pt->accelBuilds++ == pt->sw.failBuildAt && pt->sw.failBuildAt > 0)  // test hook (armed at mi_pt_create): this REbuild fails after the old structure is gone
    return fail(MI_PT_ERR_HIP, "BVH build failed: MI_PT_DIAG_FAIL_BUILD");
  {
    DevBuf<uint32_t> dOffset;
    DevBuf<int32_t>  dEntry;
    HIP_TRY(dOffset.upload(triOffset.data(), triOffset.size()));
    HIP_TRY(dEntry.upload(entryNode.data(), entryNode.size()));
    pt::BvhBuildInput in{pt->nodes.ptr, pt->prims.ptr, pt->instFlags.ptr, dOffset.ptr, dEntry.ptr, int(entryNode.size()), uint32_t(totalTris)};
    in.karrasTopology = (pt->bvhBuilder & 2) != 0;
    in.reinsertPasses = pt->accelBuilds == 1 ? pt->sw.reinsert : pt->sw.reinsertUpdate;  // (accelBuilds counts this build already)
    in.reinsertRounds = pt->sw.reinsertRounds;
    in.splitFactor    = pt->sw.splitFactor;
    in.splitMaxDepth  = pt->sw.splitMaxDepth;
    in.splitMinShare  = pt->sw.splitMinShare;
    pt::BvhBuildOutput bo;
    std::string        err;
    if(!pt::buildBvh(in, bo, nullptr, err))
      return fail(MI_PT_ERR_HIP, "BVH build failed: " + err);
    pt->bvhNodes = bo.nodes;
    pt->bvhTris  = bo.tris;
    pt->scene.bvhRoot = bo.root;
    pt->scene.numTris = int(bo.numTris);
    pt->scene.bvh8NumNodes = 0;
    pt->staticStats.bvhNodeCount     = bo.numNodes;
    pt->staticStats.bvhTriangleCount = bo.numTris;
    pt->staticStats.bvhNodeBytes     = 64;
    pt->staticStats.bvhTriangleBytes = sizeof(pt::DevTri);
    pt->wide = (pt->bvhBuilder & 1) == 0;
    // the triangle rounds of the 8-wide walk pack (triangle index | owner lane << 26) into one word (pt_kernels.hip)
    if(pt->wide && bo.numTris >= (1u << 26))
      return fail(MI_PT_ERR_ARGUMENT, "scene has 2^26 or more triangles: beyond what the 8-wide BVH walk indexes (select the BVH2 walk, bvhBuilder bit 0)");
    if(pt->wide && bo.numTris > 0)
    {
      pt::Bvh8Output b8;
      pt::Bvh8Options b8opt;
      b8opt.sahCollapse = !pt->sw.collapseGreedy; b8opt.hostCollapse = pt->sw.hostCollapse;
      if(!pt::buildBvh8(bo, b8, nullptr, err, b8opt))
      {
        if(b8.nodes)
          (void)hipFree(b8.nodes);
        if(b8.tris)
          (void)hipFree(b8.tris);
        return fail(MI_PT_ERR_HIP, "BVH8 collapse failed: " + err);
      }
      // the wide structure owns its own triangle order; the BVH2 arrays are no longer needed
      (void)hipFree(pt->bvhNodes);
      (void)hipFree(pt->bvhTris);
      pt->bvhNodes  = nullptr;
      pt->bvhTris   = b8.tris;
      pt->bvh8Nodes = b8.nodes;
      pt->scene.bvhRoot = 0;
      pt->scene.bvh8NumNodes = int(b8.numNodes);
      pt->staticStats.bvhNodeCount = b8.numNodes;
      pt->staticStats.bvhNodeBytes = 80;
      if(!pt->sw.noPlanes)  // A/B switch of the packet walk's float planes
      {
        HIP_TRY(pt->bvh8Planes.alloc(size_t(b8.numNodes) * 48));
        pt::launchBvh8Planes(b8.nodes, b8.numNodes, pt->bvh8Planes.ptr, nullptr);
        HIP_TRY(hipGetLastError());
      }
    }
  }
  pt::DevScene& S = pt->scene;
  S.bvhNodes = pt->bvhNodes; S.bvh8Nodes = pt->bvh8Nodes; S.tris = pt->bvhTris;
  S.bvh8Planes = pt->bvh8Nodes ? pt->bvh8Planes.ptr : nullptr;
  S.shadeTris = nullptr;
  S.alphaTris = nullptr;
  if(S.numTris > 0)
  {
    HIP_TRY(pt->shadeTris.alloc(size_t(S.numTris)));
    pt::launchBuildShadeRecords(S, uint32_t(S.numTris), pt->shadeTris.ptr, nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    S.shadeTris = pt->shadeTris.ptr;
  }
  if(pt->hasAlpha && S.numTris > 0)
  {
    HIP_TRY(pt->alphaTris.alloc(size_t(S.numTris)));
    pt::launchBuildAlphaRecords(S, uint32_t(S.numTris), pt->alphaTris.ptr, nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    S.alphaTris = pt->alphaTris.ptr;
  }
  pt->sceneDevDirty = true;
  return MI_PT_OK;
}

}  // namespace

extern "C" {

const char* mi_pt_last_error(void)
{
  return g_lastError.c_str();
}
// Issues the frames mi_pt_render_frame has been holding back (mi_pt_set_frame_queue) as ONE mi_pt_render_frames batch.  Every entry point that
// reads or changes what those frames depend on calls this first, so a caller never observes the deferral except through time.
static int flushPending(MiPt* pt)
{
  if(!pt || pt->pendingFrames == 0)
    return MI_PT_OK;
  const int n       = pt->pendingFrames;
  pt->pendingFrames = 0;
  return mi_pt_render_frames(pt, &pt->pendingFirst, n, pt->pendingStream);
}
#define FLUSH_PENDING(pt)              \
  do                                   \
  {                                    \
    if(int rcFlush_ = flushPending(pt)) \
      return rcFlush_;                 \
  } while(0)

int mi_pt_create(const MiPtSceneDesc* sd, const MiPtCreateOptions* options, MiPt** out)
{
  if(!sd || !out)
    return fail(MI_PT_ERR_ARGUMENT, "mi_pt_create: null argument");
  int deviceCount = 0;
  if(hipGetDeviceCount(&deviceCount) != hipSuccess || deviceCount <= 0)
    return fail(MI_PT_ERR_NO_DEVICE, "mi_pt_create: no HIP device visible (this library has no CPU path)");
  const int device = options ? options->device : 0;
  if(device < 0 || device >= deviceCount)
    return fail(MI_PT_ERR_ARGUMENT, "mi_pt_create: bad device ordinal");
  HIP_TRY(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  if(sd->numMaterials <= 0 || sd->numTextureInfos <= 0)
    return fail(MI_PT_ERR_ARGUMENT, "mi_pt_create: scene needs at least one material and the reserved texture-info slot 0");

  // ---- cross-table consistency (include/mi_pt.h: MI_PT_ERR_ARGUMENT for inconsistent tables).  The kernels index these tables
  // without further checks, so a caller other than the in-tree loader gets an error here instead of a device fault.
  if(sd->numRenderNodes < 0 || sd->numRenderPrimitives < 0 || sd->numLights < 0 || sd->numTextures < 0 || sd->numMaterials > 65535 * 4
     || (sd->numRenderNodes > 0 && !sd->renderNodes) || (sd->numRenderPrimitives > 0 && !sd->renderPrimitives) || (sd->numLights > 0 && !sd->lights)
     || (sd->numTextures > 0 && !sd->textures) || !sd->materials || !sd->textureInfos)
    return fail(MI_PT_ERR_ARGUMENT, "mi_pt_create: negative table size or null table");
  for(int n = 0; n < sd->numRenderNodes; ++n)
  {
    const MiGltfRenderNode& rn = sd->renderNodes[n];
    if(rn.materialID >= sd->numMaterials)  // (negative ids mean "default material", reference: max(0, materialID))
      return fail(MI_PT_ERR_ARGUMENT, "mi_pt_create: render node " + std::to_string(n) + " references material " + std::to_string(rn.materialID) + " of "
                                          + std::to_string(sd->numMaterials));
  }
  for(int m = 0; m < sd->numMaterials; ++m)
  {
    const uint16_t* slots = &sd->materials[m].pbrBaseColorTexture;  // the 22 texture-info slots are contiguous (mi_pt_shaderio.h)
    for(int k = 0; k < 22; ++k)
      if(int(slots[k]) >= sd->numTextureInfos)
        return fail(MI_PT_ERR_ARGUMENT, "mi_pt_create: material " + std::to_string(m) + " references texture info " + std::to_string(slots[k]) + " of "
                                            + std::to_string(sd->numTextureInfos));
  }
  for(int i = 0; i < sd->numRenderPrimitives; ++i)
  {
    const MiPtRenderPrimitive& p = sd->renderPrimitives[i];
    if(p.vertexCount > 0 && !p.positions)
      return fail(MI_PT_ERR_ARGUMENT, "mi_pt_create: render primitive " + std::to_string(i) + " has vertices but no positions");
    if(p.triangleCount > 0 && !p.indices)
      return fail(MI_PT_ERR_ARGUMENT, "mi_pt_create: render primitive " + std::to_string(i) + " has triangles but no indices");
    for(size_t k = 0, n = size_t(p.triangleCount) * 3; k < n; ++k)
      if(p.indices[k] >= p.vertexCount)
        return fail(MI_PT_ERR_ARGUMENT, "mi_pt_create: render primitive " + std::to_string(i) + " has a vertex index beyond its vertexCount");
  }

  // MI_PT_BUILD_TIMING=1: wall time of the phases of the scene build on stderr (diagnostics)
  static const bool buildTiming = getenv("MI_PT_BUILD_TIMING") != nullptr;
  auto              tPhase      = std::chrono::steady_clock::now();
  auto              phase       = [&](const char* what) {
    if(!buildTiming)
      return;
    (void)hipDeviceSynchronize();
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[mi_pt build] %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - tPhase).count());
    tPhase = now;
  };

  std::unique_ptr<MiPt> pt(new MiPt());
  pt->sw.read();
  if(pt->sw.collapseBad)
    return fail(MI_PT_ERR_ARGUMENT, std::string("MI_PT_COLLAPSE=") + getenv("MI_PT_COLLAPSE") + " is neither \"sah\" nor \"greedy\"");
  pt->device          = device;
  pt->numCUs          = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  pt->collectCounters = options && options->collectCounters;

  HIP_TRY(pt->materials.upload(sd->materials, size_t(sd->numMaterials)));
  HIP_TRY(pt->texInfos.upload(sd->textureInfos, size_t(sd->numTextureInfos)));
  HIP_TRY(pt->nodes.upload(sd->renderNodes, size_t(sd->numRenderNodes)));
  HIP_TRY(pt->lights.upload(sd->lights, size_t(sd->numLights)));

  // ---- geometry pool ------------------------------------------------------------------------------------------------
  size_t geomBytes = 0;
  for(int i = 0; i < sd->numRenderPrimitives; ++i)
  {
    const MiPtRenderPrimitive& p = sd->renderPrimitives[i];
    geomBytes += align16(size_t(p.triangleCount) * 12);
    geomBytes += align16(size_t(p.vertexCount) * 12);
    if(p.normals) geomBytes += align16(size_t(p.vertexCount) * 12);
    if(p.colors) geomBytes += align16(size_t(p.vertexCount) * 4);
    if(p.tangents) geomBytes += align16(size_t(p.vertexCount) * 16);
    if(p.texCoords0) geomBytes += align16(size_t(p.vertexCount) * 8);
    if(p.texCoords1) geomBytes += align16(size_t(p.vertexCount) * 8);
    geomBytes += align16(size_t(p.vertexCount) * 48);  // interleaved copy (DevPrim::verts)
  }
  if(geomBytes / 16 >= (size_t(1) << 32))  // DevShadeTri addresses vertices as 32-bit float4 indices into this pool
    return fail(MI_PT_ERR_ARGUMENT, "the geometry pool exceeds 64 GiB");
  HIP_TRY(pt->geometry.alloc(std::max<size_t>(geomBytes, 16)));
  std::vector<uint8_t>     staging(std::max<size_t>(geomBytes, 16));
  std::vector<pt::DevPrim> devPrims(size_t(sd->numRenderPrimitives));
  {
    size_t off = 0;
    auto   put = [&](const void* src, size_t bytes) -> const uint8_t* {
      if(!src || bytes == 0)
        return nullptr;
      memcpy(staging.data() + off, src, bytes);
      const uint8_t* d = pt->geometry.ptr + off;
      off += align16(bytes);
      return d;
    };
    for(int i = 0; i < sd->numRenderPrimitives; ++i)
    {
      const MiPtRenderPrimitive& p = sd->renderPrimitives[i];
      pt::DevPrim&               d = devPrims[size_t(i)];
      d.indices    = reinterpret_cast<const uint32_t*>(put(p.indices, size_t(p.triangleCount) * 12));
      d.positions  = reinterpret_cast<const float*>(put(p.positions, size_t(p.vertexCount) * 12));
      d.normals    = reinterpret_cast<const float*>(put(p.normals, size_t(p.vertexCount) * 12));
      d.colors     = reinterpret_cast<const uint32_t*>(put(p.colors, size_t(p.vertexCount) * 4));
      d.tangents   = reinterpret_cast<const float*>(put(p.tangents, size_t(p.vertexCount) * 16));
      d.texCoords0 = reinterpret_cast<const float*>(put(p.texCoords0, size_t(p.vertexCount) * 8));
      d.texCoords1 = reinterpret_cast<const float*>(put(p.texCoords1, size_t(p.vertexCount) * 8));
      d.opaqueTriangles = pt->sw.noOpaqueTris ? 0u : std::min(p.opaqueTriangleCount, p.triangleCount);  // (test switch: alpha-test them all -- same image)
      d._pad            = 0;
      {
        std::vector<float> iv(size_t(p.vertexCount) * 12, 0.0f);
        for(uint32_t v = 0; v < p.vertexCount; ++v)
        {
          float* o = &iv[size_t(v) * 12];
          o[0] = p.positions[3 * size_t(v)]; o[1] = p.positions[3 * size_t(v) + 1]; o[2] = p.positions[3 * size_t(v) + 2];
          if(p.normals) { o[3] = p.normals[3 * size_t(v)]; o[4] = p.normals[3 * size_t(v) + 1]; o[5] = p.normals[3 * size_t(v) + 2]; }
          if(p.texCoords0) { o[6] = p.texCoords0[2 * size_t(v)]; o[7] = p.texCoords0[2 * size_t(v) + 1]; }
          if(p.tangents) memcpy(o + 8, p.tangents + 4 * size_t(v), 16);
        }
        d.verts = reinterpret_cast<const float4*>(put(iv.data(), iv.size() * sizeof(float)));
      }
    }
    HIP_TRY(hipMemcpy(pt->geometry.ptr, staging.data(), staging.size(), hipMemcpyHostToDevice));
  }
  HIP_TRY(pt->prims.upload(devPrims.data(), devPrims.size()));
  phase("tables + geometry upload");

  // ---- what the acceleration structure is built from (see buildAcceleration) ------------------------------------------------
  pt->hostNodes.assign(sd->renderNodes, sd->renderNodes + sd->numRenderNodes);
  if(sd->renderNodeVisible)
    pt->hostVisible.assign(sd->renderNodeVisible, sd->renderNodeVisible + sd->numRenderNodes);
  pt->matInstFlags.resize(size_t(sd->numMaterials));
  pt->matFeatures.resize(size_t(sd->numMaterials));
  for(int m = 0; m < sd->numMaterials; ++m)
  {
    const MiGltfShadeMaterial& mat = sd->materials[m];
    uint32_t                   f   = 0;
    // reference: getInstanceFlag, src/gltf_scene_rtx.cpp:271-295
    if(mat.transmissionFactor == 0.0f && mat.alphaMode == MI_ALPHA_OPAQUE && mat.diffuseTransmissionFactor == 0.0f)
      f |= pt::INST_FORCE_OPAQUE;
    if(mat.doubleSided == 1 || mat.thicknessFactor > 0.0f || mat.transmissionFactor > 0.0f)
      f |= pt::INST_CULL_DISABLE;
    if(mat.transmissionFactor > 0.01f)  // MIN_TRANSMISSION, shaders/pathtrace_functions.h.slang:36,256
      f |= pt::INST_TRANSMISSIVE;
    if(!(f & pt::INST_FORCE_OPAQUE) && mat.alphaMode == MI_ALPHA_OPAQUE)  // getOpacity == 1: the alpha draw always commits (pt_scene.h)
      f |= pt::INST_ALPHA_PASSES;
    pt->matInstFlags[size_t(m)] = uint8_t(f);
    uint32_t g = 0;
    if(mat.multiscatterColorFactor[0] > 0.0f || mat.multiscatterColorFactor[1] > 0.0f || mat.multiscatterColorFactor[2] > 0.0f)
      g |= 1u;
    // the materials the specialised shade kernel cannot serve (see evaluateMaterial<SIMPLE>, pt_shading.h)
    if(mat.transmissionFactor != 0.0f || mat.diffuseTransmissionFactor != 0.0f || mat.clearcoatFactor != 0.0f || mat.iridescenceFactor != 0.0f
       || mat.anisotropyStrength > 0.0f || mat.retroreflectionFactor != 0.0f || mat.sheenColorFactor[0] != 0.0f || mat.sheenColorFactor[1] != 0.0f
       || mat.sheenColorFactor[2] != 0.0f)
      g |= 2u;
    pt->matFeatures[size_t(m)] = uint8_t(g);
  }
  pt->primTriangles.resize(size_t(sd->numRenderPrimitives));
  for(int i = 0; i < sd->numRenderPrimitives; ++i)
  {
    const MiPtRenderPrimitive& rp = sd->renderPrimitives[i];
    pt->primTriangles[size_t(i)]  = (rp.positions && rp.indices) ? rp.triangleCount : 0u;
  }
  pt->bvhBuilder = options ? options->bvhBuilder : 0;

  // ---- textures: one RGBA8 pool with all mip chains -----------------------------------------------------------------------
  {
    std::vector<pt::DevTexture> dt(size_t(sd->numTextures));
    size_t                      totalTexels = 0;
    for(int i = 0; i < sd->numTextures; ++i)
    {
      const MiPtTexture& t = sd->textures[i];
      if(t.numLevels < 1 || t.numLevels > 16 || t.width < 1 || t.height < 1 || t.width > 65535 || t.height > 65535)
        return fail(MI_PT_ERR_ARGUMENT, "mi_pt_create: texture dimensions / level count out of range");
      for(int l = 0; l < t.numLevels; ++l)
        totalTexels += size_t(std::max(1, t.width >> l)) * size_t(std::max(1, t.height >> l));
    }
    if(totalTexels > 0xffffffffull)
      return fail(MI_PT_ERR_ARGUMENT, "mi_pt_create: texture pool exceeds 2^32 texels");
    std::vector<uchar4> pool(std::max<size_t>(totalTexels, 1));
    size_t              off = 0;
    for(int i = 0; i < sd->numTextures; ++i)
    {
      const MiPtTexture& t = sd->textures[i];
      pt::DevTexture&    d = dt[size_t(i)];
      memset(&d, 0, sizeof(d));
      d.width = uint16_t(t.width); d.height = uint16_t(t.height); d.numLevels = uint8_t(t.numLevels); d.srgb = uint8_t(t.srgb ? 1 : 0);
      d.magFilter = uint8_t(t.magFilter); d.minFilter = uint8_t(t.minFilter); d.mipmapMode = uint8_t(t.mipmapMode);
      d.wrapS = uint8_t(t.wrapS); d.wrapT = uint8_t(t.wrapT);
      for(int l = 0; l < t.numLevels; ++l)
      {
        size_t n         = size_t(std::max(1, t.width >> l)) * size_t(std::max(1, t.height >> l));
        d.levelOffset[l] = uint32_t(off);
        memcpy(&pool[off], t.levels[l], n * 4);
        off += n;
      }
    }
    HIP_TRY(pt->textures.upload(dt.data(), dt.size()));
    HIP_TRY(pt->texels.upload(pool.data(), pool.size()));
    // bilinear footprints (pt_scene.h: texQuads), built on the device from the pool just uploaded; MI_PT_DIAG_NO_QUADS=1: A/B switch
    if(totalTexels > 0 && !pt->sw.noQuads)
    {
      HIP_TRY(pt->texQuads.alloc(pool.size()));
      for(const pt::DevTexture& d : dt)
        for(int l = 0; l < d.numLevels; ++l)
          pt::launchTextureQuads(pt->texels.ptr, pt->texQuads.ptr, d.levelOffset[l], std::max(1, int(d.width) >> l), std::max(1, int(d.height) >> l), d.wrapS,
                                 d.wrapT, nullptr);
      HIP_TRY(hipGetLastError());
    }
    // texture info + descriptor, flattened per texture slot (pt_scene.h: DevTexRef)
    std::vector<pt::DevTexRef> refs(size_t(std::max(sd->numTextureInfos, 1)));
    memset(refs.data(), 0, refs.size() * sizeof(pt::DevTexRef));
    for(int i = 0; i < sd->numTextureInfos; ++i)
    {
      const MiGltfTextureInfo& ti = sd->textureInfos[i];
      pt::DevTexRef&           r  = refs[size_t(i)];
      memcpy(r.uv, ti.uvTransform, sizeof(r.uv));
      r.texCoord = uint8_t(ti.texCoord);
      if(ti.index >= 0 && ti.index < sd->numTextures)
      {
        const pt::DevTexture& d = dt[size_t(ti.index)];
        r.level0 = d.levelOffset[0]; r.width = d.width; r.height = d.height; r.numLevels = d.numLevels; r.srgb = d.srgb;
        r.magFilter = d.magFilter; r.minFilter = d.minFilter; r.mipmapMode = d.mipmapMode; r.wrapS = d.wrapS; r.wrapT = d.wrapT;
      }
    }
    HIP_TRY(pt->texRefs.upload(refs.data(), refs.size()));
    // the five core map slots per material (pt_scene.h: DevCoreTex), in the order of the material's slot words
    std::vector<pt::DevCoreTex> core(size_t(std::max(sd->numMaterials, 1)) * 5);
    memset(core.data(), 0, core.size() * sizeof(pt::DevCoreTex));
    static const float identityUv[6] = {1.0f, 0.0f, 0.0f, 1.0f, 0.0f, 0.0f};
    for(int m = 0; m < sd->numMaterials; ++m)
    {
      const MiGltfShadeMaterial& M = sd->materials[m];
      const uint16_t slots[5] = {M.pbrBaseColorTexture, M.normalTexture, M.pbrMetallicRoughnessTexture, M.emissiveTexture, M.occlusionTexture};
      for(int k = 0; k < 5; ++k)
      {
        pt::DevCoreTex& c = core[size_t(m) * 5 + size_t(k)];
        c.ref = slots[k];
        if(slots[k] == 0 || int(slots[k]) >= sd->numTextureInfos)
          continue;
        const pt::DevTexRef& r = refs[slots[k]];
        c.level0 = r.level0;
        c.wh     = uint32_t(r.width) | (uint32_t(r.height) << 16);
        const bool fast = pt->sw.coreTex && r.width > 0 && r.magFilter == MI_FILTER_LINEAR && r.minFilter == MI_FILTER_LINEAR && r.wrapS != MI_WRAP_MIRRORED_REPEAT && r.wrapT != MI_WRAP_MIRRORED_REPEAT;
        c.flags  = (fast ? pt::CT_FAST : 0u) | (r.srgb ? pt::CT_SRGB : 0u) | (r.texCoord ? pt::CT_TEXCOORD1 : 0u) | (memcmp(r.uv, identityUv, sizeof(identityUv)) != 0 ? pt::CT_TRANSFORM : 0u)
                  | (r.mipmapMode == MI_FILTER_LINEAR ? pt::CT_MIP_LINEAR : 0u) | (uint32_t(r.wrapS) << pt::CT_WRAPS_SHIFT) | (uint32_t(r.wrapT) << pt::CT_WRAPT_SHIFT)
                  | (uint32_t(r.numLevels) << pt::CT_LEVELS_SHIFT);
      }
    }
    HIP_TRY(pt->coreTex.upload(core.data(), core.size()));
  }
  {
    float lut[256];
    for(int i = 0; i < 256; ++i)
    {
      float c = float(i) / 255.0f;
      lut[i]  = c <= 0.04045f ? c / 12.92f : std::pow((c + 0.055f) / 1.055f, 2.4f);
    }
    HIP_TRY(pt->srgbLut.upload(lut, 256));
  }
  phase("textures upload");

  pt::DevScene& S = pt->scene;
  S.materials = pt->materials.ptr; S.texInfos = pt->texInfos.ptr; S.nodes = pt->nodes.ptr; S.prims = pt->prims.ptr; S.lights = pt->lights.ptr;
  S.textures = pt->textures.ptr; S.texels = pt->texels.ptr; S.texQuads = pt->texQuads.ptr; S.envPixels = nullptr; S.envAccel = nullptr; S.bvhNodes = nullptr; S.bvh8Nodes = nullptr; S.bvh8Planes = nullptr; S.tris = nullptr;
  S.geomPool = reinterpret_cast<const float4*>(pt->geometry.ptr);
  S.texRefs = pt->texRefs.ptr; S.coreTex = reinterpret_cast<const uint4*>(pt->coreTex.ptr); S.alphaTris = nullptr; S.shadeTris = nullptr; S.srgbLut = pt->srgbLut.ptr; S.numMaterials = sd->numMaterials; S.numTextures = sd->numTextures;
  S.numLights = sd->numLights; S.numNodes = sd->numRenderNodes;
  S.envWidth = 0; S.envHeight = 0;
  S.packetInterval = pt->sw.packetInterval;  // A/B switch (LABNOTES.md section 2): 0 = the per-ray node test in every packet
  S.shadowOctFlip  = pt->sw.shadowFarFirst ? 7u : 0u;
  if(int rc = buildAcceleration(pt.get()))
    return rc;
  phase("shade / alpha records");
  HIP_TRY(pt->stats.alloc(1));
  HIP_TRY(hipMemset(pt->stats.ptr, 0, sizeof(pt::StatCounters)));
  // SkyPhysicalParameters{} defaults, so a caller that never calls mi_pt_set_sky still renders the default sky
  MiSkyPhysicalParameters s{};
  s.rgbUnitConversion[0] = s.rgbUnitConversion[1] = s.rgbUnitConversion[2] = 1.0f / 80000.0f;
  s.multiplier = 0.1f; s.haze = 0.1f; s.redblueshift = 0.1f; s.saturation = 1.0f; s.groundColor[0] = s.groundColor[1] = s.groundColor[2] = 0.4f;
  s.horizonBlur = 0.3f; s.sunDiskIntensity = 1.0f; s.sunDirection[0] = s.sunDirection[1] = s.sunDirection[2] = 0.5773502691896258f;
  s.sunDiskScale = 1.0f; s.sunGlowIntensity = 1.0f; s.yIsUp = 1;
  pt->sky = s;
  *out    = pt.release();
  return MI_PT_OK;
}

int mi_pt_update_render_nodes(MiPt* pt, const MiGltfRenderNode* renderNodes, int numRenderNodes, const uint8_t* renderNodeVisible)
{
  FLUSH_PENDING(pt);
  if(!pt || !renderNodes || numRenderNodes != int(pt->hostNodes.size()))
    return fail(MI_PT_ERR_ARGUMENT, "mi_pt_update_render_nodes: the render-node count must be the one the instance was created with");
  for(int n = 0; n < numRenderNodes; ++n)
    if(renderNodes[n].materialID >= int(pt->matInstFlags.size()))
      return fail(MI_PT_ERR_ARGUMENT, "mi_pt_update_render_nodes: render node references a material beyond the table");
  HIP_TRY(hipSetDevice(pt->device));
  HIP_TRY(hipDeviceSynchronize());  // nothing in flight may still walk the old structure
  pt->hostNodes.assign(renderNodes, renderNodes + numRenderNodes);
  if(renderNodeVisible)
    pt->hostVisible.assign(renderNodeVisible, renderNodeVisible + numRenderNodes);
  else
    pt->hostVisible.clear();
  return buildAcceleration(pt);
}

int mi_pt_update_lights(MiPt* pt, const MiGltfLight* lights, int numLights)
{
  FLUSH_PENDING(pt);
  if(!pt || numLights != pt->scene.numLights || (numLights > 0 && !lights))
    return fail(MI_PT_ERR_ARGUMENT, "mi_pt_update_lights: the light count must be the one the instance was created with");
  if(numLights == 0)
    return MI_PT_OK;
  HIP_TRY(hipSetDevice(pt->device));
  HIP_TRY(hipDeviceSynchronize());  // frames in flight still sample the old table
  HIP_TRY(hipMemcpy(pt->lights.ptr, lights, sizeof(MiGltfLight) * size_t(numLights), hipMemcpyHostToDevice));
  return MI_PT_OK;
}

int mi_pt_destroy(MiPt* pt)
{
  if(pt)
    pt->pendingFrames = 0;  // (frames still held back are dropped with the instance)
  if(!pt)
    return MI_PT_OK;
  (void)hipSetDevice(pt->device);
  (void)hipDeviceSynchronize();
  pt::dumpTraceProfile();
  delete pt;
  return MI_PT_OK;
}

int mi_pt_set_environment(MiPt* pt, const MiPtEnvironment* env)
{
  FLUSH_PENDING(pt);
  if(!pt)
    return fail(MI_PT_ERR_ARGUMENT, "mi_pt_set_environment: null instance");
  HIP_TRY(hipSetDevice(pt->device));
  HIP_TRY(hipDeviceSynchronize());
  if(!env || !env->rgba || !env->accel || env->width <= 0 || env->height <= 0)
  {
    pt->envPixels.release();
    pt->envAccel.release();
    pt->scene.envPixels = nullptr;
    pt->scene.envAccel  = nullptr;
    pt->scene.envWidth = pt->scene.envHeight = 0;
    pt->sceneDevDirty  = true;
    return MI_PT_OK;
  }
  const size_t n = size_t(env->width) * size_t(env->height);
  HIP_TRY(pt->envPixels.upload(reinterpret_cast<const float4*>(env->rgba), n));
  HIP_TRY(pt->envAccel.upload(env->accel, n));
  pt->scene.envPixels = pt->envPixels.ptr;
  pt->scene.envAccel  = pt->envAccel.ptr;
  pt->scene.envWidth  = env->width;
  pt->scene.envHeight = env->height;
  pt->sceneDevDirty   = true;
  return MI_PT_OK;
}

int mi_pt_resize(MiPt* pt, int width, int height)
{
  FLUSH_PENDING(pt);
  if(!pt || width <= 0 || height <= 0 || width > 32768 || height > 32768)
    return fail(MI_PT_ERR_ARGUMENT, "mi_pt_resize: need 0 < width, height <= 32768");
  HIP_TRY(hipSetDevice(pt->device));
  HIP_TRY(hipDeviceSynchronize());
  pt->width       = width;
  pt->height      = height;
  pt->denoised    = nullptr;
  pt->accumFrames = 0.0f;
  pt->momentFrames = 0.0f;
  return allocFrameResources(pt);
}

int mi_pt_set_frame_info(MiPt* pt, const MiSceneFrameInfo* info)
{
  if(!pt || !info)
    return fail(MI_PT_ERR_ARGUMENT, "mi_pt_set_frame_info: null argument");
  // (the reference updates bFrameInfo / bSkyParams in EVERY onRender, changed or not -- src/renderer.cpp:675-708 -- and so does a drop-in caller: the same values again
  //  are no reason to issue the frames mi_pt_set_frame_queue is holding back)
  if(pt->haveFrameInfo && memcmp(&pt->frameInfo, info, sizeof(*info)) == 0)
    return MI_PT_OK;
  FLUSH_PENDING(pt);
  pt->frameInfo     = *info;
  pt->haveFrameInfo = true;
  return MI_PT_OK;
}
int mi_pt_set_sky(MiPt* pt, const MiSkyPhysicalParameters* sky)
{
  if(!pt || !sky)
    return fail(MI_PT_ERR_ARGUMENT, "mi_pt_set_sky: null argument");
  if(memcmp(&pt->sky, sky, sizeof(*sky)) == 0)
    return MI_PT_OK;
  FLUSH_PENDING(pt);
  pt->sky = *sky;
  return MI_PT_OK;
}

int mi_pt_set_tile_partition(MiPt* pt, int rank, int world, int tileSize)
{
  FLUSH_PENDING(pt);
  if(!pt || world < 1 || rank < 0 || rank >= world || tileSize < 16 || tileSize > 4096 || (tileSize & (tileSize - 1)) != 0)
    return fail(MI_PT_ERR_ARGUMENT, "mi_pt_set_tile_partition: need 0 <= rank < world and tileSize a power of two in [16, 4096]");
  pt->tileRank  = rank;
  pt->tileWorld = world;
  pt->tileSize  = tileSize;
  if(pt->width > 0)
  {
    HIP_TRY(hipSetDevice(pt->device));
    HIP_TRY(hipDeviceSynchronize());
    void* bound = (pt->accum != pt->accumOwn.ptr) ? pt->accum : nullptr;
    int   rc    = allocFrameResources(pt);
    if(rc == MI_PT_OK && bound)
      pt->accum = reinterpret_cast<float4*>(bound);
    return rc;
  }
  return MI_PT_OK;
}

int mi_pt_bind_accum(MiPt* pt, void* deviceRGBA32F)
{
  FLUSH_PENDING(pt);
  if(!pt)
    return fail(MI_PT_ERR_ARGUMENT, "mi_pt_bind_accum: null instance");
  pt->accum = deviceRGBA32F ? reinterpret_cast<float4*>(deviceRGBA32F) : pt->accumOwn.ptr;
  return MI_PT_OK;
}

int mi_pt_bind_guides(MiPt* pt, void* albedoRGBA32F, void* normalRGBA32F, void* depthR32F)
{
  FLUSH_PENDING(pt);
  if(!pt)
    return fail(MI_PT_ERR_ARGUMENT, "mi_pt_bind_guides: null instance");
  pt->albedoBound = reinterpret_cast<float4*>(albedoRGBA32F);
  pt->normalBound = reinterpret_cast<float4*>(normalRGBA32F);
  pt->depthBound  = reinterpret_cast<float*>(depthR32F);
  return MI_PT_OK;
}

int mi_pt_set_frame_queue(MiPt* pt, int depth)
{
  if(!pt || depth < 1 || depth > 1024)
    return fail(MI_PT_ERR_ARGUMENT, "mi_pt_set_frame_queue: 1 <= depth <= 1024 required");
  FLUSH_PENDING(pt);
  pt->frameQueueDepth = depth;
  return MI_PT_OK;
}

int mi_pt_render_frame(MiPt* pt, const MiPathtraceParams* params, void* hipStream)
{
  if(pt && params && pt->frameQueueDepth > 1)
  {
    // frame k of a batch is the first frame's parameters with frameCount + k and totalSamples + k * numSamples (mi_pt_render_frames): a call joins
    // the pending batch when it is exactly that, on the same stream, and does not start a new accumulation
    if(pt->pendingFrames > 0)
    {
      MiPathtraceParams expect = pt->pendingFirst;
      expect.frameCount += pt->pendingFrames;
      expect.totalSamples += pt->pendingFrames * expect.numSamples;
      expect.flags &= ~MI_PT_FIRST_FRAME;
      expect.mouseCoord[0] = params->mouseCoord[0];
      expect.mouseCoord[1] = params->mouseCoord[1];
      if(hipStream == pt->pendingStream && memcmp(&expect, params, sizeof(expect)) == 0)
      {
        if(++pt->pendingFrames >= pt->frameQueueDepth)
          return flushPending(pt);
        return MI_PT_OK;
      }
      FLUSH_PENDING(pt);
    }
    // argument errors are reported by THIS call, not by a later flush
    if(params->numSamples < 1 || params->maxDepth < 1 || params->maxDepth > 255)
      return fail(MI_PT_ERR_ARGUMENT, "mi_pt_render_frame: numSamples >= 1 and 1 <= maxDepth <= 255 required");
    if(pt->width <= 0 || !pt->haveFrameInfo)
      return fail(MI_PT_ERR_STATE, "mi_pt_render_frame: call mi_pt_resize and mi_pt_set_frame_info first");
    pt->pendingFirst  = *params;
    pt->pendingStream = hipStream;
    pt->pendingFrames = 1;
    return MI_PT_OK;
  }

  return mi_pt_render_frames(pt, params, 1, hipStream);
}

int mi_pt_render_frames(MiPt* pt, const MiPathtraceParams* params, int numFrames, void* hipStream)
{
  FLUSH_PENDING(pt);
  if(!pt || !params)
    return fail(MI_PT_ERR_ARGUMENT, "mi_pt_render_frame: null argument");
  if(numFrames < 1 || numFrames > 1024)
    return fail(MI_PT_ERR_ARGUMENT, "mi_pt_render_frames: 1 <= numFrames <= 1024 required");
  if(pt->width <= 0 || !pt->haveFrameInfo)
    return fail(MI_PT_ERR_STATE, "mi_pt_render_frame: call mi_pt_resize and mi_pt_set_frame_info first");
  // (maxDepth 0: the reference's loop `for(depth = 0; depth < maxDepth; ...)` never runs and the image is black; here no shade launch would run and
  //  the queued camera hits would keep the radiance and guide records of an earlier batch -- refused instead of special-cased)
  if(params->numSamples < 1 || params->maxDepth < 1 || params->maxDepth > 255)
    return fail(MI_PT_ERR_ARGUMENT, "mi_pt_render_frame: numSamples >= 1 and 1 <= maxDepth <= 255 required");
  if((pt->frameInfo.flags & MI_SCENE_USE_HDR_ENVIRONMENT) && !pt->scene.envPixels)
    return fail(MI_PT_ERR_STATE, "mi_pt_render_frame: HDR environment requested but none was set");
  HIP_TRY(hipSetDevice(pt->device));
  hipStream_t stream = reinterpret_cast<hipStream_t>(hipStream);
  pt->lastStream     = stream;
  if(pt->numSlots == 0)
    return MI_PT_OK;
  if(numFrames > pt->framesCap)  // grow the path/queue arrays once; later batches of this size reuse them
  {
    HIP_TRY(hipDeviceSynchronize());
    if(int rc = allocPathResources(pt, numFrames))
      return rc;
  }

  pt::LaunchCtx c;
  c.scene = pt->scene;
  memset(&c.fc, 0, sizeof(c.fc));
  c.fc.frameInfo = pt->frameInfo;
  c.fc.sky       = pt->sky;
  if(!pt->skyPreValid || memcmp(&pt->skyPreFor, &pt->sky, sizeof(pt->sky)) != 0)
  {
    // stream-ordered behind the batches that still read the previous values
    if(!pt->skyPre.ptr)
      HIP_TRY(pt->skyPre.alloc(1));
    pt::launchSkyPrecomp(pt->sky, pt->skyPre.ptr, stream);
    pt->skyPreFor   = pt->sky;
    pt->skyPreValid = true;
  }
  c.fc.skyPre    = pt->skyPre.ptr;
  c.fc.pc        = *params;
  c.fc.width     = pt->width;
  c.fc.height    = pt->height;
  c.fc.tileSize  = pt->tileSize;
  c.fc.tileShift = 0;
  while((1 << c.fc.tileShift) < pt->tileSize)
    ++c.fc.tileShift;
  c.fc.numSlots  = pt->numSlots;
  c.fc.numFrames = numFrames;
  pt::divideMagic(uint32_t(std::max(numFrames, 2)), c.fc.framesMagic, c.fc.framesShift);
  c.fc.slotLayout = numFrames % 64 == 0 ? 1 : 0;
  c.fc.stateInQueue = (!pt->sw.stateBySlot && !(pt->frameInfo.flags & MI_SCENE_INFINITE_PLANE_SHADOW_CATCHER)) ? 1 : 0;
  const bool guides = (params->flags & MI_PT_USE_OPTIX_DENOISER) != 0;
  if(int rc = ensureOptionalPathArrays(pt, c.fc.stateInQueue == 0, params->numSamples > 1, guides, (pt->frameInfo.flags & MI_SCENE_INFINITE_PLANE_SHADOW_CATCHER) != 0))
    return rc;
  c.paths        = pt->paths;
  {
    // optional records stay allocated (MiPt::opt*) once a batch needed them; THIS batch sees only the ones it needs itself -- a later
    // single-sample batch must not keep paying the by-slot writes of an earlier multi-sample or catcher batch
    const bool stateBySlot = c.fc.stateInQueue == 0, multiSample = params->numSamples > 1;
    if(!(stateBySlot || multiSample))
      c.paths.misc = nullptr;
    if(!stateBySlot)
      c.paths.throughput = nullptr;
    if(!multiSample)
      c.paths.pixelSum = nullptr;
  }
  pt->accumFrames   = float(params->totalSamples) / float(params->numSamples) + float(numFrames);
  // the second moment covers the accumulation only if every batch since its start carried the guides
  if(guides && (params->totalSamples == 0 || pt->momentFrames == pt->accumFrames - float(numFrames)))
    pt->momentFrames = pt->accumFrames;
  else if(!guides || params->totalSamples == 0)
    pt->momentFrames = 0.0f;
  if(!guides)
  {
    c.paths.guideAlbedo = nullptr;
    c.paths.guideNormal = nullptr;
  }
  c.queues           = pt->queues;
  if(!(pt->frameInfo.flags & MI_SCENE_INFINITE_PLANE_SHADOW_CATCHER))
    c.queues.shadow.aux2 = nullptr;
  c.ownedTiles       = pt->ownedTiles.ptr;
  c.stats            = pt->stats.ptr;
  c.stream           = stream;
  c.persistentBlocks = unsigned(pt->numCUs) * 8u;
  c.hasAlpha         = pt->hasAlpha;
  // the closest-hit walks run their alpha machinery only where some alpha test has an open outcome: a scene whose non-opaque instances all have
  // alphaMode OPAQUE (transmissive glass) takes the plain kernels -- every candidate commits (INST_ALPHA_PASSES)
  c.hasAlphaClosest  = c.hasAlpha && pt->hasAlphaTest;
  c.hasTransmissive  = pt->hasTransmissive;
  c.simpleMaterials  = pt->simpleMaterials;
  c.wide             = pt->wide;
  c.collectCounters  = pt->collectCounters;
  {
    // Per-bounce sort of the shade kernel (k_shade): grouping the hits by material pays where the materials take different ways
    // through the BSDF (glass workload: generic shade kernel -11 %); where every material runs the same code (the SIMPLE
    // kernel, which therefore has no sort at all) it only cost (helmet +4 %, atrium +12 %, street +4 % of the shade kernel).
    // MI_PT_SORT = 0 | 1 | 2 is the A/B switch of the generic kernel.
    c.sortMode = pt->sw.sortMode;
  }
  // descriptor copies for the kernels that read them through a pointer
  if(pt->sceneDevDirty)
  {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(pt->sceneDev.upload(&pt->scene, 1));
    pt->sceneDevDirty = false;
  }
  if(!pt->fcHost)
  {
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&pt->fcHost), sizeof(pt::FrameConsts) * MiPt::FC_RING, hipHostMallocDefault));
    HIP_TRY(pt->fcRing.alloc(MiPt::FC_RING));
    for(hipEvent_t& e : pt->fcDone)
      HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  const unsigned fcSlot = pt->fcCursor++ % unsigned(MiPt::FC_RING);
  if(pt->fcCursor > unsigned(MiPt::FC_RING))
    HIP_TRY(hipEventSynchronize(pt->fcDone[fcSlot]));  // the batch that used this slot FC_RING calls ago must have drained
  pt->fcHost[fcSlot] = c.fc;
  HIP_TRY(hipMemcpyAsync(pt->fcRing.ptr + fcSlot, pt->fcHost + fcSlot, sizeof(pt::FrameConsts), hipMemcpyHostToDevice, stream));
  c.sceneDev = pt->sceneDev.ptr;
  c.fcDev    = pt->fcRing.ptr + fcSlot;

  auto timedOn = [&](hipStream_t on, int kind, auto&& launch) {
    if(pt->timingEnabled)
    {
      hipEvent_t a = getEvent(pt, pt->evCursor), b = getEvent(pt, pt->evCursor);
      (void)hipEventRecord(a, on);
      launch();
      (void)hipEventRecord(b, on);
      pt->pendingSpans.push_back({kind, a, b});
    }
    else
      launch();
  };
  auto timed = [&](int kind, auto&& launch) { timedOn(stream, kind, launch); };
  hipEvent_t frameA = nullptr, frameB = nullptr;
  if(pt->timingEnabled)
  {
    frameA = getEvent(pt, pt->evCursor);
    frameB = getEvent(pt, pt->evCursor);
    (void)hipEventRecord(frameA, stream);
  }
  int iterations = 0, traceLaunches = 0, shadeLaunches = 0, shadowLaunches = 0;
  const bool usePacket = !pt->sw.noPacket;
  const bool debugSpans = pt->sw.traceSpans;
  // Small batches: a bounce's shadow stage (any-hit walk + resolve) on a second stream, next to the NEXT bounce's closest-hit walk.  The two
  // are independent -- the walk reads the continuation rays the shade launch wrote, the shadow stage adds into radiance records that only
  // the next SHADE launch reads (which therefore waits for it) -- and with few frames in flight neither fills the 256 CUs: most
  // workgroups of a persistent grid find no work and leave, so the other kernel's find room.  Needs the state in the queue entry (on
  // catcher frames the resolve pass may end a path the walk is about to trace).
  // ... and launches long enough to be worth two cross-stream hand-overs per bounce: measured (round 4, frames in flight 1 / 4 / 8 / 16 / 32 / 64 / 128)
  // atrium-class (406 k triangles, depth 12) +15 / +8 / +6 / +3.5 / +1.7 / +1.0 / +0.3 %, street-class 4K +8 % at 1, +1.6 % at 8, +0.3 % from 32;
  // helmet-class (74 k, 60 % of the camera paths leave at bounce 0) -6.5 / -0.4 / -0.8 / -0.6 %: scenes below 2e5 triangles keep one stream.
  // Default up to 32 frames: the interactive range, where it pays; larger batches fill the device by themselves, and their per-launch times
  // (bench.py's kernel table) stay those of kernels that have the device to themselves.
  // Volume-scatter scenes run it at every batch size and triangle count: their random walks leave ~100 bounce iterations of a few thousand paths per
  // batch, three dependent launches each, and taking the shadow stage off that chain measured +31 % for a single frame, +8.7 % at 32 and +2.2 % at 256
  // frames in flight on the glass-class workload (the host's poll of the queue length every eighth iteration only waits for the main stream).
  bool overlap = pt->sw.overlapUpTo > 0 && ((numFrames <= pt->sw.overlapUpTo && pt->scene.numTris >= pt->sw.overlapMinTris) || pt->hasVolumeScatter)
                 && c.fc.stateInQueue != 0 && !debugSpans;
  if(overlap && !pt->sideStream)
    overlap = hipStreamCreateWithFlags(&pt->sideStream, hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&pt->evShaded, hipEventDisableTiming) == hipSuccess
              && hipEventCreateWithFlags(&pt->evShadowed, hipEventDisableTiming) == hipSuccess;
  pt::LaunchCtx cSide = c;
  cSide.stream        = pt->sideStream;
  // Any early return below (a failed queue poll, a launch error) must not leave shadow-stage work queued on the side stream behind the caller's
  // back: a caller that synchronises its own stream and then resizes or updates would race with kernels still adding into the radiance records.
  // On every exit taken while a shadow stage is outstanding the side stream is drained (error paths only: the ordinary path joins it by event).
  struct SideJoin
  {
    hipStream_t side        = nullptr;
    bool        outstanding = false;
    ~SideJoin()
    {
      if(outstanding && side)
        (void)hipStreamSynchronize(side);
    }
  } sideJoin;
  sideJoin.side = overlap ? pt->sideStream : nullptr;
  for(int s = 0; s < params->numSamples; ++s)
  {
    pt::launchResetCounters(c.queues, stream);
    // 8-wide BVH: ONE kernel generates the camera rays, walks them as packets, finishes the paths that leave the scene and queues
    // the hits (k_trace_primary); BVH2 / MI_PT_NO_PACKET: k_generate writes the rays and the per-lane kernel walks them
    const bool fusedPrimary = c.wide && usePacket && !debugSpans;
    if(fusedPrimary)
    {
      timed(TK_PRIMARY, [&] { pt::launchTracePrimary(c, s); });
      if(pt->timingEnabled)
        ++pt->accTiming.tracePrimaryLaunches;
    }
    else
      timed(TK_GENERATE, [&] { pt::launchGenerate(c, s); });
    int cur = 0;
    // Every iteration either ends a path or consumes one unit of surfaceDepth, except volume scatter events
    // (pathtrace_functions.h.slang:925-931), which are free for VOLUME_FREE_BUDGET bounces and then Russian-rouletted.
    int maxIters = params->maxDepth;
    // a shadow-catcher bounce does not consume surfaceDepth (eEarlyContinue) but is always followed by a miss or a surface hit
    if((pt->frameInfo.flags & MI_SCENE_USE_INFINITE_PLANE) && (pt->frameInfo.flags & MI_SCENE_INFINITE_PLANE_SHADOW_CATCHER))
      maxIters = 2 * params->maxDepth + 2;
    if(pt->hasVolumeScatter)
      maxIters = params->maxDepth * 66 + 512;
    if(pt->sw.maxItersDiag > 0)
      maxIters = std::min(maxIters, pt->sw.maxItersDiag);
    for(int it = 0; it < maxIters; ++it)
    {
      if(pt->hasVolumeScatter && it >= params->maxDepth && (it % 8) == 0)
      {
        uint32_t counts[2 * pt::NSUB];
        HIP_TRY(hipMemcpyAsync(counts, &c.queues.counters[cur ? pt::QC_PAIR1 : pt::QC_PAIR0], sizeof(counts), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        uint32_t remaining = 0;
        for(int q = 0; q < pt::NSUB; ++q)
          remaining += counts[2 * q];
        if(remaining == 0)
          break;
      }
      if(debugSpans)  // MI_PT_TRACE_SPANS=1: per-launch wall time and queue lengths (synchronising; diagnostics only)
      {
        auto count = [&](int base) {  // base: first of 16 tails stored 2 words apart
          uint32_t v[2 * pt::NSUB];
          (void)hipMemcpy(v, &c.queues.counters[base], sizeof(uint32_t) * (2 * pt::NSUB - 1), hipMemcpyDeviceToHost);
          uint32_t t = 0;
          for(int q = 0; q < pt::NSUB; ++q) t += v[2 * q];
          return t;
        };
        auto span = [&](auto&& launch) {
          (void)hipStreamSynchronize(stream);
          auto t0 = std::chrono::steady_clock::now();
          launch();
          (void)hipStreamSynchronize(stream);
          return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        };
        auto segs = [&]() {
          pt::StatCounters h{};
          (void)hipMemcpy(&h, pt->stats.ptr, sizeof(h), hipMemcpyDeviceToHost);
          return h;
        };
        uint32_t nIn = count(cur ? pt::QC_PAIR1 : pt::QC_PAIR0);
        const pt::StatCounters s0 = segs();
        double   tTrace = span([&] { pt::launchTraceClosest(c, cur); });
        const pt::StatCounters s1 = segs();
        double   tShade = span([&] { pt::launchShade(c, cur, it == 0); });
        uint32_t nSh    = count((cur ? pt::QC_PAIR0 : pt::QC_PAIR1) + 1);
        double   tShadow = span([&] { pt::launchTraceShadow(c, cur ^ 1); });
        const pt::StatCounters s2 = segs();
        fprintf(stderr, "[mi_pt span] frame %d it %2d rays %8u (traced %8llu, nodes %9llu tris %9llu) trace %8.3f ms shade %8.3f ms | shadow rays %8u (traced %8llu, nodes %9llu tris %9llu) %8.3f ms\n",
                params->frameCount, it, nIn, (unsigned long long)(s1.segments - s0.segments), (unsigned long long)(s1.nodesClosest - s0.nodesClosest),
                (unsigned long long)(s1.trisClosest - s0.trisClosest), tTrace, tShade, nSh, (unsigned long long)(s2.shadowRays - s1.shadowRays),
                (unsigned long long)(s2.nodesShadow - s1.nodesShadow), (unsigned long long)(s2.trisShadow - s1.trisShadow), tShadow);
      }
      else
      {
        if(!(it == 0 && fusedPrimary))
          timed(TK_TRACE, [&] { pt::launchTraceClosest(c, cur); });
        if(overlap && it > 0)
          (void)hipStreamWaitEvent(stream, pt->evShadowed, 0);  // the previous bounce's shadow terms are in before this shade launch reads them
        timed(it == 0 ? TK_SHADE_FIRST : TK_SHADE, [&] { pt::launchShade(c, cur, it == 0); });
        if(it == 0 && pt->timingEnabled)
          ++pt->accTiming.shadeFirstLaunches;
        if(overlap)
        {
          (void)hipEventRecord(pt->evShaded, stream);
          (void)hipStreamWaitEvent(pt->sideStream, pt->evShaded, 0);
          timedOn(pt->sideStream, TK_SHADOW, [&] { pt::launchTraceShadow(cSide, cur ^ 1); });
          (void)hipEventRecord(pt->evShadowed, pt->sideStream);
          sideJoin.outstanding = true;
        }
        else
          timed(TK_SHADOW, [&] { pt::launchTraceShadow(c, cur ^ 1); });
      }
      ++iterations; ++traceLaunches; ++shadeLaunches; ++shadowLaunches;
      cur ^= 1;
    }
    if(overlap && iterations > 0)
    {
      (void)hipStreamWaitEvent(stream, pt->evShadowed, 0);  // the last bounce's shadow terms, before the sample is folded
      sideJoin.outstanding = false;                         // (joined: the caller's stream now orders everything after the side stream's work)
    }
    // paths the loop left alive (its iteration bound cut them: volume random walks) hand their radiance and seed over by slot
    if(c.fc.stateInQueue && (pt->hasVolumeScatter || pt->sw.maxItersDiag > 0))
      pt::launchFlushSurvivors(c, cur);
    timed(TK_ACCUM, [&] {
      pt::launchFinishSample(c, s, pt->accum, pt->depthImg(), guides ? pt->albedoImg() : nullptr, guides ? pt->normalImg() : nullptr);
    });
  }
  if(params->flags & MI_PT_FIRST_FRAME)
    pt::launchSelection(c, pt->selection.ptr);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(pt->fcDone[fcSlot], stream));
  if(pt->timingEnabled)
  {
    (void)hipEventRecord(frameB, stream);
    pt->pendingSpans.push_back({TK_COUNT, frameA, frameB});
    pt->accTiming.traceClosestLaunches += traceLaunches;
    pt->accTiming.shadeLaunches += shadeLaunches;
    pt->accTiming.traceShadowLaunches += shadowLaunches;
    pt->accTiming.bounceIterations += iterations;
  }
  return MI_PT_OK;
}

int mi_pt_synchronize(MiPt* pt)
{
  FLUSH_PENDING(pt);
  if(!pt)
    return fail(MI_PT_ERR_ARGUMENT, "mi_pt_synchronize: null instance");
  HIP_TRY(hipSetDevice(pt->device));
  HIP_TRY(hipDeviceSynchronize());
  return MI_PT_OK;
}

int mi_pt_read_accum(MiPt* pt, float* host)
{
  FLUSH_PENDING(pt);
  if(!pt || !host || pt->width <= 0)
    return fail(MI_PT_ERR_ARGUMENT, "mi_pt_read_accum: bad arguments");
  HIP_TRY(hipSetDevice(pt->device));
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(host, pt->accum, size_t(pt->width) * size_t(pt->height) * sizeof(float4), hipMemcpyDeviceToHost));
  return MI_PT_OK;
}
int mi_pt_write_accum(MiPt* pt, const float* host)
{
  FLUSH_PENDING(pt);
  if(!pt || !host || pt->width <= 0)
    return fail(MI_PT_ERR_ARGUMENT, "mi_pt_write_accum: bad arguments");
  HIP_TRY(hipSetDevice(pt->device));
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(pt->accum, host, size_t(pt->width) * size_t(pt->height) * sizeof(float4), hipMemcpyHostToDevice));
  pt->accumFrames  = 1.0f;  // a frame from elsewhere: denoisable, with no temporal moment behind it
  pt->momentFrames = 0.0f;
  return MI_PT_OK;
}
int mi_pt_read_guides(MiPt* pt, float* albedo, float* normal)
{
  FLUSH_PENDING(pt);
  if(!pt || pt->width <= 0)
    return fail(MI_PT_ERR_ARGUMENT, "mi_pt_read_guides: bad arguments");
  HIP_TRY(hipSetDevice(pt->device));
  HIP_TRY(hipDeviceSynchronize());
  const size_t bytes = size_t(pt->width) * size_t(pt->height) * sizeof(float4);
  if(albedo)
    HIP_TRY(hipMemcpy(albedo, pt->albedoImg(), bytes, hipMemcpyDeviceToHost));
  if(normal)
    HIP_TRY(hipMemcpy(normal, pt->normalImg(), bytes, hipMemcpyDeviceToHost));
  return MI_PT_OK;
}
int mi_pt_read_selection(MiPt* pt, uint32_t* host)
{
  FLUSH_PENDING(pt);
  if(!pt || !host || pt->width <= 0)
    return fail(MI_PT_ERR_ARGUMENT, "mi_pt_read_selection: bad arguments");
  HIP_TRY(hipSetDevice(pt->device));
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(host, pt->selection.ptr, size_t(pt->width) * size_t(pt->height) * sizeof(uint32_t), hipMemcpyDeviceToHost));
  return MI_PT_OK;
}
int mi_pt_read_depth(MiPt* pt, float* host)
{
  FLUSH_PENDING(pt);
  if(!pt || !host || pt->width <= 0)
    return fail(MI_PT_ERR_ARGUMENT, "mi_pt_read_depth: bad arguments");
  HIP_TRY(hipSetDevice(pt->device));
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(host, pt->depthImg(), size_t(pt->width) * size_t(pt->height) * sizeof(float), hipMemcpyDeviceToHost));
  return MI_PT_OK;
}
void* mi_pt_accum_device_ptr(MiPt* pt)
{
  (void)flushPending(pt);
  return pt ? pt->accum : nullptr;
}

int mi_pt_denoise(MiPt* pt, int iterations, float sigmaColor, float sigmaNormal, float sigmaAlbedo, float* host, void* hipStream)
{
  FLUSH_PENDING(pt);
  if(!pt || pt->width <= 0 || iterations < 1 || iterations > 8)
    return fail(MI_PT_ERR_ARGUMENT, "mi_pt_denoise: bad arguments");
  HIP_TRY(hipSetDevice(pt->device));
  hipStream_t  stream = reinterpret_cast<hipStream_t>(hipStream);
  const size_t px     = size_t(pt->width) * size_t(pt->height);
  if(pt->denoiseA.count != px)
  {
    HIP_TRY(pt->denoiseA.alloc(px));
    HIP_TRY(pt->denoiseB.alloc(px));
  }
  const float4* in  = pt->accum;
  float4*       out = pt->denoiseA.ptr;
  for(int i = 0; i < iterations; ++i)
  {
    pt::launchAtrous(in, out, pt->albedoImg(), pt->normalImg(), pt->width, pt->height, 1 << i, sigmaColor, sigmaNormal, sigmaAlbedo, stream);
    in  = out;
    out = (out == pt->denoiseA.ptr) ? pt->denoiseB.ptr : pt->denoiseA.ptr;
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(stream));
  pt->denoised = in;
  if(host)
    HIP_TRY(hipMemcpy(host, in, px * sizeof(float4), hipMemcpyDeviceToHost));
  return MI_PT_OK;
}

int mi_pt_denoise_svgf(MiPt* pt, int iterations, float sigmaLuminance, float sigmaNormal, float sigmaDepth, float* host, void* hipStream)
{
  FLUSH_PENDING(pt);
  if(!pt || pt->width <= 0 || iterations < 1 || iterations > 8 || !(sigmaLuminance > 0.0f) || !(sigmaDepth > 0.0f) || !(sigmaNormal >= 0.0f))
    return fail(MI_PT_ERR_ARGUMENT, "mi_pt_denoise_svgf: bad arguments");
  if(!(pt->accumFrames >= 1.0f))
    return fail(MI_PT_ERR_STATE, "mi_pt_denoise_svgf: nothing rendered yet");
  HIP_TRY(hipSetDevice(pt->device));
  hipStream_t  stream = reinterpret_cast<hipStream_t>(hipStream);
  const size_t px     = size_t(pt->width) * size_t(pt->height);
  if(pt->denoiseA.count != px)
  {
    HIP_TRY(pt->denoiseA.alloc(px));
    HIP_TRY(pt->denoiseB.alloc(px));
  }
  // (mean and second moment must cover the same frames for E[l^2] - E[l]^2 to mean anything: otherwise -- guides switched on
  //  mid-accumulation, or an accumulator written by mi_pt_write_accum -- the pass estimates the variance spatially, as it does below 4 frames)
  const float varianceFrames = pt->momentFrames == pt->accumFrames ? pt->accumFrames : 1.0f;
  pt->denoised = pt::launchSvgf(pt->accum, pt->albedoImg(), pt->normalImg(), pt->depthImg(), pt->denoiseA.ptr, pt->denoiseB.ptr, pt->width, pt->height, iterations,
                                varianceFrames, sigmaLuminance, sigmaNormal, sigmaDepth, stream);
  HIP_TRY(hipGetLastError());
  if(host)
  {
    HIP_TRY(hipStreamSynchronize(stream));
    HIP_TRY(hipMemcpy(host, pt->denoised, px * sizeof(float4), hipMemcpyDeviceToHost));
  }
  return MI_PT_OK;
}

void mi_pt_default_tonemapper(MiTonemapperData* tm, int autoExposure)
{
  if(!tm)
    return;
  *tm                      = MiTonemapperData{};
  tm->method               = MI_TONEMAP_FILMIC;
  tm->isActive             = 1;
  tm->exposure             = 1.0f;
  tm->brightness           = 1.0f;
  tm->contrast             = 1.0f;
  tm->saturation           = 1.0f;
  tm->vignette             = 0.0f;
  tm->autoExposure         = autoExposure ? 1 : 0;
  tm->autoExposureSpeed    = 1.0f;
  tm->evMinValue           = -24.0f;
  tm->evMaxValue           = 8.0f;
  tm->enableCenterMetering = 0;
}

int mi_pt_tonemap(MiPt* pt, const MiTonemapperData* tm, int source, float dtSeconds, uint8_t* host, void* hipStream)
{
  FLUSH_PENDING(pt);
  if(!pt || !tm || pt->width <= 0 || source < 0 || source > 1)
    return fail(MI_PT_ERR_ARGUMENT, "mi_pt_tonemap: bad arguments");
  if(tm->method < MI_TONEMAP_FILMIC || tm->method > MI_TONEMAP_KHRONOS_PBR || !(tm->brightness > 0.0f) || !(tm->evMaxValue > tm->evMinValue))
    return fail(MI_PT_ERR_ARGUMENT, "mi_pt_tonemap: method outside [0, 5], brightness <= 0 or an empty EV range");
  const size_t px = size_t(pt->width) * size_t(pt->height);
  if(source == 1 && (!pt->denoised || pt->denoiseA.count != px))
    return fail(MI_PT_ERR_STATE, "mi_pt_tonemap: source 1 needs a mi_pt_denoise result of the current size");
  HIP_TRY(hipSetDevice(pt->device));
  hipStream_t stream = reinterpret_cast<hipStream_t>(hipStream);
  if(pt->tonemapped.count != px)
    HIP_TRY(pt->tonemapped.alloc(px));
  if(!pt->tmHistogram.ptr)
  {
    HIP_TRY(pt->tmHistogram.alloc(256));
    HIP_TRY(pt->tmAutoState.alloc(2));
    HIP_TRY(hipMemset(pt->tmAutoState.ptr, 0, 2 * sizeof(float)));
  }
  pt::launchTonemap(source == 1 ? pt->denoised : pt->accum, pt->tonemapped.ptr, pt->width, pt->height, *tm, pt->tmHistogram.ptr, pt->tmAutoState.ptr,
                    dtSeconds, stream);
  HIP_TRY(hipGetLastError());
  if(host)
  {
    HIP_TRY(hipStreamSynchronize(stream));
    HIP_TRY(hipMemcpy(host, pt->tonemapped.ptr, px * sizeof(uint32_t), hipMemcpyDeviceToHost));
  }
  return MI_PT_OK;
}

const void* mi_pt_denoised_device_ptr(MiPt* pt)
{
  return pt ? pt->denoised : nullptr;
}

void* mi_pt_tonemapped_device_ptr(MiPt* pt)
{
  return pt ? pt->tonemapped.ptr : nullptr;
}

int mi_pt_get_memory(MiPt* pt, MiPtMemory* out)
{
  if(!pt || !out)
    return fail(MI_PT_ERR_ARGUMENT, "mi_pt_get_memory: null argument");
  HIP_TRY(hipSetDevice(pt->device));
  auto bytes = [](const auto& b) { return uint64_t(b.count) * sizeof(*b.ptr); };
  const pt::DevScene& sc = pt->scene;
  uint64_t scene = bytes(pt->materials) + bytes(pt->texInfos) + bytes(pt->nodes) + bytes(pt->prims) + bytes(pt->lights) + bytes(pt->textures) + bytes(pt->texels) + bytes(pt->texQuads)
                   + bytes(pt->geometry) + bytes(pt->instFlags) + bytes(pt->srgbLut) + bytes(pt->envPixels) + bytes(pt->envAccel) + bytes(pt->alphaTris)
                   + bytes(pt->shadeTris) + bytes(pt->texRefs) + bytes(pt->coreTex) + bytes(pt->bvh8Planes);
  // the acceleration structure is raw allocations: 64-B BVH2 nodes or 80-B BVH8 nodes + 48-B triangle records
  scene += uint64_t(pt->staticStats.bvhNodeCount) * pt->staticStats.bvhNodeBytes + uint64_t(sc.numTris) * sizeof(pt::DevTri);
  const uint64_t pathState = bytes(pt->pathArrays) + bytes(pt->optThroughput) + bytes(pt->optMisc) + bytes(pt->optMedium) + bytes(pt->optPixelSum) + bytes(pt->optGuides)
                             + bytes(pt->optShadowAux2) + bytes(pt->queueMem) + bytes(pt->queuePayload) + bytes(pt->candPool) + bytes(pt->candLists);
  const uint64_t renderer = pathState + bytes(pt->firstHit) + bytes(pt->accumOwn)
                            + bytes(pt->albedo) + bytes(pt->normal) + bytes(pt->denoiseA) + bytes(pt->denoiseB) + bytes(pt->tonemapped) + bytes(pt->depth)
                            + bytes(pt->selection) + bytes(pt->ownedTiles) + bytes(pt->sceneDev) + bytes(pt->fcRing) + bytes(pt->stats);
  size_t freeB = 0, totalB = 0;
  HIP_TRY(hipMemGetInfo(&freeB, &totalB));
  out->sceneBytes       = scene;
  out->rendererBytes    = renderer;
  out->deviceUsedBytes  = uint64_t(totalB - freeB);
  out->deviceTotalBytes = uint64_t(totalB);
  out->pathStateBytes   = pathState;
  out->pathSlots        = uint64_t(std::max(pt->numSlots, 0)) * uint64_t(std::max(pt->framesCap, 1));
  return MI_PT_OK;
}

int mi_pt_get_stats(MiPt* pt, MiPtStats* out)
{
  FLUSH_PENDING(pt);
  if(!pt || !out)
    return fail(MI_PT_ERR_ARGUMENT, "mi_pt_get_stats: null argument");
  HIP_TRY(hipSetDevice(pt->device));
  HIP_TRY(hipDeviceSynchronize());
  pt::StatCounters h{};
  HIP_TRY(hipMemcpy(&h, pt->stats.ptr, sizeof(h), hipMemcpyDeviceToHost));
  *out              = pt->staticStats;
  out->cameraPaths  = h.cameraPaths;
  out->segments     = h.segments;
  out->shadowRays   = h.shadowRays;
  out->nodesClosest = h.nodesClosest;
  out->trisClosest  = h.trisClosest;
  out->nodesShadow  = h.nodesShadow;
  out->trisShadow   = h.trisShadow;
  out->textureTaps  = h.textureTaps;
  out->surfaceHits  = h.surfaceHits;
  out->nodesPrimary = h.nodesPrimary;
  out->trisPrimary  = h.trisPrimary;
  return MI_PT_OK;
}
int mi_pt_reset_stats(MiPt* pt)
{
  FLUSH_PENDING(pt);
  if(!pt)
    return fail(MI_PT_ERR_ARGUMENT, "mi_pt_reset_stats: null instance");
  HIP_TRY(hipSetDevice(pt->device));
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemset(pt->stats.ptr, 0, sizeof(pt::StatCounters)));
  return MI_PT_OK;
}
int mi_pt_enable_timing(MiPt* pt, int enable)
{
  FLUSH_PENDING(pt);
  if(!pt)
    return fail(MI_PT_ERR_ARGUMENT, "mi_pt_enable_timing: null instance");
  pt->timingEnabled = enable != 0;
  if(enable)
  {
    pt->pendingSpans.clear();
    pt->evCursor  = 0;
    pt->accTiming = MiPtFrameTiming{};
  }
  return MI_PT_OK;
}
int mi_pt_get_frame_timing(MiPt* pt, MiPtFrameTiming* out)
{
  FLUSH_PENDING(pt);
  if(!pt || !out)
    return fail(MI_PT_ERR_ARGUMENT, "mi_pt_get_frame_timing: null argument");
  // Totals since mi_pt_enable_timing(1): resolves the recorded events (one device synchronisation, here, not per frame).
  HIP_TRY(hipSetDevice(pt->device));
  HIP_TRY(hipDeviceSynchronize());
  MiPtFrameTiming& t = pt->accTiming;
  for(const MiPt::Span& sp : pt->pendingSpans)
  {
    float ms = 0.0f;
    (void)hipEventElapsedTime(&ms, sp.a, sp.b);
    switch(sp.kind)
    {
      case TK_GENERATE: t.generateMs += ms; break;
      case TK_TRACE: t.traceClosestMs += ms; break;
      case TK_SORT: t.sortMs += ms; break;
      case TK_SHADE: t.shadeMs += ms; break;
      case TK_PRIMARY: t.traceClosestMs += ms; t.tracePrimaryMs += ms; break;
      case TK_SHADE_FIRST: t.shadeMs += ms; t.shadeFirstMs += ms; break;
      case TK_SHADOW: t.traceShadowMs += ms; break;
      case TK_ACCUM: t.accumulateMs += ms; break;
      default: t.totalMs += ms; break;
    }
  }
  pt->pendingSpans.clear();
  pt->evCursor = 0;
  *out         = t;
  return MI_PT_OK;
}
}
