// Host-visible launch interface of the wavefront kernels (pt_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "pt_scene.h"

namespace pt {

struct LaunchCtx
{
  DevScene        scene;
  FrameConsts     fc;
  const DevScene*    sceneDev;  // device-resident copies of `scene` / `fc` for kernels that hand them to non-inlined helpers
  const FrameConsts* fcDev;
  PathSoA         paths;
  Queues          queues;
  const uint32_t* ownedTiles;
  StatCounters*   stats;
  hipStream_t     stream;
  unsigned        persistentBlocks;
  bool            hasAlpha;
  bool            hasAlphaClosest;  // hasAlpha and some instance needs a real alpha test (not INST_ALPHA_PASSES): the closest-hit walks carry the alpha machinery
  bool            hasTransmissive;  // some instance carries INST_TRANSMISSIVE (ordered shadow transmission needed)
  bool            simpleMaterials;  // no material needs the transmission / clearcoat / sheen / iridescence / anisotropy paths
  bool            wide;  // traverse the 8-wide compressed BVH (scene.bvh8Nodes) instead of the BVH2
  bool            collectCounters;
  int             sortMode;  // per-bounce sort of the generic shade kernel: 0 off, 1 surface hits / others / dead, 2 hits grouped by material too
};

void launchBuildShadeRecords(const DevScene& scene, uint32_t numTris, DevShadeTri* out, hipStream_t s);
void launchBuildAlphaRecords(const DevScene& scene, uint32_t numTris, DevAlphaTri* out, hipStream_t s);
void dumpTraceProfile();  // prints the -DTRACE_PROFILE section timers (no-op in the product build)
void launchBvh8Planes(const uint4* nodes, uint32_t numNodes, float* planes, hipStream_t s);
void launchTextureQuads(const uchar4* texels, uint4* quads, uint32_t offset, int width, int height, int wrapS, int wrapT, hipStream_t s);
void launchResetCounters(const Queues& Q, hipStream_t s);
void launchSkyPrecomp(const MiSkyPhysicalParameters& sky, SkyPrecomp* out, hipStream_t stream);
void launchGenerate(const LaunchCtx& c, int sampleIndex);
void launchTraceClosest(const LaunchCtx& c, int cur);
void launchTracePrimary(const LaunchCtx& c, int sampleIndex);  // bounce 0 of an 8-wide-BVH scene: camera rays generated, packet-walked, misses finished, hits into queue 0
void launchShade(const LaunchCtx& c, int cur, bool first);  // first: bounce 0 (paths still carry k_generate's initial state)
void launchTraceShadow(const LaunchCtx& c, int nxt);  // nxt: active queue the preceding shade launch appended to
void launchFlushSurvivors(const LaunchCtx& c, int cur);  // cur: the active queue the last shade launch appended to (paths alive when the bounce loop stopped)
void launchFinishSample(const LaunchCtx& c, int sampleIndex, float4* accum, float* depth, float4* albedo, float4* normal);
void launchSelection(const LaunchCtx& c, uint32_t* selection);

// a-trous edge-avoiding wavelet filter (denoise.hip)
void launchAtrous(const float4* in, float4* out, const float4* albedo, const float4* normal, int width, int height, int step, float sigmaColor,
                  float sigmaNormal, float sigmaAlbedo, hipStream_t s);

const float4* launchSvgf(const float4* color, const float4* albedo, const float4* normal, const float* depth, float4* bufA, float4* bufB, int width, int height,
                         int iterations, float frames, float sigmaLuminance, float sigmaNormal, float sigmaDepth, hipStream_t s);

// tonemapper (tonemap.hip): optional auto-exposure metering (histogram: 256 u32, autoState: 2 floats) + the curve, RGBA32F -> RGBA8
void launchTonemap(const float4* in, uint32_t* outRgba8, int width, int height, const MiTonemapperData& tm, uint32_t* histogram, float* autoState,
                   float dtSeconds, hipStream_t s);

}  // namespace pt
