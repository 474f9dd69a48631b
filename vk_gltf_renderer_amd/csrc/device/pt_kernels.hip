// Wavefront path-trace kernels for gfx950.  One frame = numSamples x { generate -> [trace -> shade -> shadow]* -> finish }.
// The megakernel of the reference (shaders/gltf_pathtrace.slang:87-671, one thread per pixel running the whole bounce
// loop around hardware ray queries) is split at its two Trace calls so that traversal (memory/latency bound, small
// register footprint, LDS stack) and shading (ALU bound, large live state) run as separate persistent grids over
// compacted queues of path slots.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#ifndef MI_PT_EXACT_FP  // (A/B: tools/build_variant.sh exactfp -DMI_PT_EXACT_FP -fhip-fp32-correctly-rounded-divide-sqrt -fno-reciprocal-math -fno-approx-func = IEEE everywhere)
#define PT_FAST_SHADING_MATH 1  // pt_math.h: hardware reciprocal / reciprocal square root in the vector helpers of THIS translation unit (see csrc/Makefile: PT_KERNELS_FP)
#endif
#include "pt_kernels.h"
#include "pt_shading.h"
#include "pt_bvh.h"
#include "pt_bvh8.h"
#include "pt_packet.h"
#include "pt_feed.h"

namespace pt {

namespace {

// One trace workgroup per CU: 16 waves share one LDS copy of the top of the 8-wide BVH (the first nodes of the BFS-ordered
// node array), next to the per-lane traversal stacks (96 KiB), the waves' triangle-round lists (8 KiB) and, in the kernels
// that defer alpha tests, their alpha lists (16 KiB): 55 or 39 KiB of nodes of the CU's 160 KiB.
constexpr int TRACE_BLOCK = 1024;
#ifndef NODE_CACHE_EXTRA
#define NODE_CACHE_EXTRA 0  // (A/B: LDS given back by a shallower LDS stack, MI_BVH8_STACK_LDS.  Measured in round 3: 10 / 8 node groups per
                            //  lane in LDS with 200 / 400 more cached nodes: within +-0.5 % on all four workloads; 6 + 600: street -2 %)
#endif
constexpr int NODE_CACHE  = 712 + NODE_CACHE_EXTRA;  // BVH8 nodes (80 B each) resident in LDS
constexpr int NODE_CACHE_ALPHA = 504 + NODE_CACHE_EXTRA;
// The any-hit walk also keeps the 2-KiB octant table of pt_bvh8.h there (it has the room: no triangle-round result slots): measured in
// round 4 on both walks -- shadow walk -1.5 ... -3 % on every workload; closest-hit walk -2 % on the helmet but +1.5 ... +2.3 % on atrium,
// glass and street (its node step then waits for one more LDS round trip where the butterfly's 16 instructions were hidden by the
// other waves, and the table took the room of 26 cached nodes), so that kernel keeps the butterfly.  // ... in the kernels that also keep a list of deferred alpha tests there (16 B x 1024)
constexpr int SEL_BLOCK   = 256;
#ifndef TRACE_MIN_WAVES
#define TRACE_MIN_WAVES 1
#endif
constexpr int SHADE_BLOCK = 256;
// Per-bounce sort in the shade kernel (see k_shade): window of queue entries sorted in LDS, and its bins.
constexpr uint32_t SORT_ROUNDS   = 4;
constexpr uint32_t SORT_WINDOW   = SORT_ROUNDS * SHADE_BLOCK;  // 1024 entries
constexpr uint32_t SORT_SEGMENTS = SORT_ROUNDS * (SHADE_BLOCK / 64);
constexpr int      SORT_BINS     = 32;  // one lane of the scan per bin
constexpr uint32_t SORT_BIN_MISS = 30;  // bins 0..29: surface hits, materialID mod 30; 30: no surface hit (environment, infinite plane)
constexpr uint32_t SORT_BIN_DEAD = 31;  // dead queue entries: not laid out
#ifndef SHADE_SIMPLE_WAVES
#define SHADE_SIMPLE_WAVES 3
#endif

// ---- queues ------------------------------------------------------------------------------------------------------------------
PT_DEV uint32_t laneId() { return __lane_id(); }

// Exclusive prefix of the NSUB sub-queue counts into LDS (s_prefix[NSUB] = total).  Call from every thread of the block.
PT_DEV void queuePrefix(const uint32_t* counts, uint32_t* s_prefix)  // counts: 16 tails, 2 words apart (see QC_PAIR*)
{
  if(threadIdx.x < 64)  // first wave: one load per lane, shuffle scan (every block of every launch pays this latency)
  {
    const uint32_t lane = threadIdx.x;
    // (plain cached loads: every block of every launch reads these 16 words; coherent agent-scope loads are served beyond the
    // XCD's L2 one by one and cost ~100 us per launch on short queues)
    uint32_t       v    = lane < NSUB ? counts[2u * lane] : 0u;
#pragma unroll
    for(int d = 1; d < NSUB; d <<= 1)
    {
      const uint32_t t = uint32_t(__shfl_up(int(v), d));
      if(lane >= uint32_t(d))
        v += t;
    }
    if(lane < NSUB)
      s_prefix[lane + 1] = v;
    if(lane == 0)
      s_prefix[0] = 0;
  }
  __syncthreads();
}
// Array position of entry `flat` of the concatenated sub-queues.
PT_DEV uint32_t queuePos(uint32_t subCap, const uint32_t* s_prefix, uint32_t flat)
{
  uint32_t q = 0;
#pragma unroll
  for(uint32_t step = NSUB / 2; step > 0; step >>= 1)
    if(s_prefix[q + step] <= flat)
      q += step;
  return q * subCap + (flat - s_prefix[q]);
}
// Block-aggregated append of the survivors of one chunk to BOTH output queues of a shade launch: one LDS atomic per wave and
// queue, ONE 64-bit global atomic per block for the two tails (`pair` = &counters[QC_PAIRq + 2 * sub]).  Must be called by
// every thread of the block (contains __syncthreads).  s_tmp: 4 words of LDS.  Returns the array positions reserved for this
// thread's entries (meaningless when the predicate is false).
struct PushPos
{
  uint32_t next, shadow;
};
PT_DEV PushPos queuePushBlock2(bool predNext, bool predShadow, uint32_t subCap, uint32_t* pair, uint32_t sub, uint32_t* s_tmp)
{
  if(threadIdx.x == 0)
  {
    s_tmp[0] = 0;
    s_tmp[1] = 0;
  }
  __syncthreads();
  const unsigned long long maskN = __ballot(predNext), maskS = __ballot(predShadow);
  const uint32_t           lane  = laneId();
  uint32_t                 wbaseN = 0, wbaseS = 0;
  if(lane == 0)
  {
    if(maskN != 0ull)
      wbaseN = atomicAdd(&s_tmp[0], uint32_t(__popcll(maskN)));
    if(maskS != 0ull)
      wbaseS = atomicAdd(&s_tmp[1], uint32_t(__popcll(maskS)));
  }
  wbaseN = uint32_t(__shfl(int(wbaseN), 0));
  wbaseS = uint32_t(__shfl(int(wbaseS), 0));
  __syncthreads();
  if(threadIdx.x == 0)
  {
    const uint32_t     nN = s_tmp[0], nS = s_tmp[1];
    unsigned long long base = 0ull;
    if((nN | nS) != 0u)
      base = atomicAdd(reinterpret_cast<unsigned long long*>(pair), (static_cast<unsigned long long>(nS) << 32) | nN);
    s_tmp[2] = uint32_t(base);
    s_tmp[3] = uint32_t(base >> 32);
  }
  __syncthreads();
  const unsigned long long below = (1ull << lane) - 1ull;
  PushPos                  p;
  p.next   = sub * subCap + s_tmp[2] + wbaseN + uint32_t(__popcll(maskN & below));
  p.shadow = sub * subCap + s_tmp[3] + wbaseS + uint32_t(__popcll(maskS & below));
  return p;
}


// (A per-WAVE version of this append -- one 64-bit global atomic per wave, no barrier, so that the waves of a block run their chunks
//  independently -- was measured in round 3: helmet 3793 against 3799 Msamples/s, atrium 478 / 482, street 504 / 512, glass 529 / 537.
//  The three barriers per chunk are not what the shade kernel waits for; four times the device-scope atomics are.)

// ---- slot <-> pixel -----------------------------------------------------------------------------------------------------------
// Slots are tile-major; inside a tile 8x8 micro-tiles, so one 64-lane wave owns one 8x8 pixel block (coherent camera rays).
// ownedTiles[i] = x0 | y0 << 16 (pixel origin of the i-th tile this rank renders).
PT_DEV bool slotToPixel(const FrameConsts& fc, const uint32_t* ownedTiles, uint32_t slot, int& px, int& py)
{
  const uint32_t tile  = ownedTiles[slot >> (2 * fc.tileShift)];
  const uint32_t w     = slot & ((1u << (2 * fc.tileShift)) - 1u);
  const uint32_t micro = w >> 6, lane = w & 63u;
  const uint32_t mshift = uint32_t(fc.tileShift) - 3u;
  px                   = int((tile & 0xffffu) + (micro & ((1u << mshift) - 1u)) * 8u + (lane & 7u));
  py                   = int((tile >> 16) + (micro >> mshift) * 8u + (lane >> 3));
  return px < fc.width && py < fc.height;
}

// ---- camera (pathtrace_functions.h.slang:784-811, gltf_pathtrace.slang:502-529) ------------------------------------------
PT_DEV void getRay(const FrameConsts& fc, f2 samplePos, f2 offset, f3& origin, f3& direction)
{
  const MiSceneFrameInfo& fi = fc.frameInfo;
  // (IEEE division / square root whatever the compile options -- divExact, normalizeExact: camera rays agree with the oracle bit for bit)
  f2 clip = mk2(divExact(samplePos.x + offset.x, float(fc.width)) * 2.0f - 1.0f, divExact(samplePos.y + offset.y, float(fc.height)) * 2.0f - 1.0f);
  f4 view = mulFull(fi.projInv, mk4(clip.x, clip.y, -1.0f, 1.0f));
  view    = mk4(divExact(view.x, view.w), divExact(view.y, view.w), divExact(view.z, view.w), divExact(view.w, view.w));
  if(hasFlag(fi.flags, MI_SCENE_IS_ORTHOGRAPHIC))
  {
    origin    = xyz(mulFull(fi.viewInv, view));
    direction = normalizeExact(xyz(mulFull(fi.viewInv, mk4(0, 0, -1, 0))));
  }
  else
  {
    origin    = mk3(fi.viewInv[12], fi.viewInv[13], fi.viewInv[14]);
    direction = normalizeExact(xyz(mulFull(fi.viewInv, view)) - origin);
  }
}

PT_DEV uint4 packMedium(f3 ext, f3 sc, float g)
{
  __half2 a = __floats2half2_rn(ext.x, ext.y), b = __floats2half2_rn(ext.z, sc.x), c = __floats2half2_rn(sc.y, sc.z), d = __floats2half2_rn(g, 0.0f);
  uint4   r;
  r.x = *reinterpret_cast<uint32_t*>(&a);
  r.y = *reinterpret_cast<uint32_t*>(&b);
  r.z = *reinterpret_cast<uint32_t*>(&c);
  r.w = *reinterpret_cast<uint32_t*>(&d);
  return r;
}
PT_DEV void unpackMedium(uint4 m, f3& ext, f3& sc, float& g)
{
  float2 a = __half22float2(*reinterpret_cast<__half2*>(&m.x)), b = __half22float2(*reinterpret_cast<__half2*>(&m.y));
  float2 c = __half22float2(*reinterpret_cast<__half2*>(&m.z)), d = __half22float2(*reinterpret_cast<__half2*>(&m.w));
  ext = mk3(a.x, a.y, b.x);
  sc  = mk3(b.y, c.x, c.y);
  g   = d.x;
}

//================================================================================================================================
// Camera paths: seed, AA jitter, camera ray, thin-lens DoF  (gltf_pathtrace.slang:546-596, 502-529)
//================================================================================================================================
struct CameraPath
{
  bool     valid;  // the slot maps to a pixel of the image
  uint32_t frame;  // frame of the batch the slot belongs to (wave-uniform)
  uint32_t seed;   // RNG state after the camera draws
  f3       origin, direction;
};
// Start of sample `sampleIndex` of path slot `slot` (slot < batchSlots).  Resets the denoiser guides of the slot on sample 0; the
// caller stores the seed (PathSoA::misc).  PathTracerState{} of gltf_pathtrace.slang:443 is implicit: throughput 1, lastSamplePdf
// DIRAC, radiance 0, maxRoughness 0, not inside, depth 0 -- the bounce-0 shade launch knows it as constants (k_shade<FIRST>), the
// medium is written when a path first enters one, firstHit by the first shade of the path, pixelSum by k_finish_sample.
// Cost note: the bounce-0 kernel is bound by vector-instruction issue (2 500 per wave of 64 camera rays, of which this function
// was 900 in its first form).  What is wave-uniform runs on the scalar unit: a wave's 64 slots are consecutive and 64-aligned
// and the slots are micro-tile major (FrameConsts::numFrames), so the frame index, the tile and the micro-tile are computed once per
// wave (the division by numFrames is a multiply-high by FrameConsts::framesMagic).  With aperture = 0 (wave-uniform) the lens offset is
// (cos, sin) * sqrt(0) = 0 whatever the angle: the two draws still advance the seed, the sine and cosine are skipped.
// The arithmetic that forms the ray stays correctly rounded (IEEE division, square root) -- it is what makes the camera rays,
// and with them coverage, selection ids and the segment counters, agree with the oracle bit for bit; hardware reciprocals and
// v_sin / v_cos here bought 1 % more and moved silhouette samples (tried, dropped).
PT_DEV CameraPath generateCameraPath(const FrameConsts& fc, const PathSoA& P, const uint32_t* ownedTiles, uint32_t slot, int sampleIndex)
{
  CameraPath cp;
  cp.frame     = 0u;
  cp.seed      = 0u;
  cp.origin    = mk3(0.0f);
  cp.direction = mk3(0.0f);
  // ---- slot -> (frame, pixel): see slotToPixel; per wave
  const uint32_t lane      = laneId();
  const uint32_t base      = __builtin_amdgcn_readfirstlane(slot - lane);
  uint32_t frame, pixelBase, inMicro;
  if(fc.slotLayout == 1)
  {
    // pixel major: the wave is 64 consecutive frames of pixel slot base / numFrames
    const uint32_t pslot = __umulhi(base, fc.framesMagic) >> fc.framesShift;
    frame                = base - pslot * uint32_t(fc.numFrames) + lane;
    pixelBase            = pslot & ~63u;
    inMicro              = pslot & 63u;
  }
  else
  {
    // micro-tile major: wave w = base / 64 works on micro-tile w / numFrames of frame w % numFrames (pathSlot)
    const uint32_t waveIdx = base >> 6;
    const uint32_t mtile   = fc.numFrames > 1 ? (__umulhi(waveIdx, fc.framesMagic) >> fc.framesShift) : waveIdx;
    frame                  = waveIdx - mtile * uint32_t(fc.numFrames);
    pixelBase              = mtile * 64u;
    inMicro                = lane;
  }
  cp.frame                 = frame;
  const uint32_t tile      = ownedTiles[pixelBase >> (2 * fc.tileShift)];
  const uint32_t micro     = (pixelBase & ((1u << (2 * fc.tileShift)) - 1u)) >> 6;
  const uint32_t mshift    = uint32_t(fc.tileShift) - 3u;
  const int      px        = int((tile & 0xffffu) + (micro & ((1u << mshift) - 1u)) * 8u + (inMicro & 7u));
  const int      py        = int((tile >> 16) + (micro >> mshift) * 8u + (inMicro >> 3));
  cp.valid                 = px < fc.width && py < fc.height;
  if(!cp.valid)
    return cp;
  uint32_t seed;
  f2       jitter;
  if(sampleIndex == 0)
  {
    seed     = xxhash32(uint32_t(px), uint32_t(py), uint32_t(fc.pc.frameCount) + frame);
    float u1 = rnd(seed), u2 = rnd(seed);
    // sampleGaussian (Box-Muller), pathtrace_functions.h.slang:784-789
    float r     = sqrtExact(-2.0f * logExact(fmaxf(1e-38f, u1)));
    float theta = 2.0f * K_PI * u2;
    jitter      = mk2(0.5f + ANTIALIASING_STANDARD_DEVIATION * (r * cosExact(theta)), 0.5f + ANTIALIASING_STANDARD_DEVIATION * (r * sinExact(theta)));
    // (single-sample frames: the guide records are WRITTEN by the path's first shade -- or zeroed where a path has none -- instead of zeroed here and
    //  read, added to and written back there: 64 B per path less)
    if(P.guideAlbedo && fc.pc.numSamples > 1)
    {
      P.guideAlbedo[slot] = make_float4(0, 0, 0, 0);
      P.guideNormal[slot] = make_float4(0, 0, 0, 0);
    }
  }
  else
  {
    seed     = __float_as_uint(P.misc[slot].z);
    float u1 = rnd(seed), u2 = rnd(seed);
    jitter   = mk2(u1, u2);
  }
  // getRay (pathtrace_functions.h.slang:791-811)
  const MiSceneFrameInfo& fi = fc.frameInfo;
  const f2 clip = mk2(divExact(float(px) + jitter.x, float(fc.width)) * 2.0f - 1.0f, divExact(float(py) + jitter.y, float(fc.height)) * 2.0f - 1.0f);
  f4       view = mulFull(fi.projInv, mk4(clip.x, clip.y, -1.0f, 1.0f));
  view          = mk4(divExact(view.x, view.w), divExact(view.y, view.w), divExact(view.z, view.w), divExact(view.w, view.w));
  f3 origin, direction;
  if(hasFlag(fi.flags, MI_SCENE_IS_ORTHOGRAPHIC))
  {
    origin    = xyz(mulFull(fi.viewInv, view));
    direction = normalizeExact(xyz(mulFull(fi.viewInv, mk4(0, 0, -1, 0))));
  }
  else
  {
    origin    = mk3(fi.viewInv[12], fi.viewInv[13], fi.viewInv[14]);
    direction = normalizeExact(xyz(mulFull(fi.viewInv, view)) - origin);
    // thin lens (gltf_pathtrace.slang:502-529)
    const float* V          = fi.viewInv;
    f3           focalPoint = direction * fc.pc.focalDistance;
    float        cam_r1     = rnd(seed) * K_TWO_PI;
    float        cam_r2     = rnd(seed) * fc.pc.aperture;
    f3           aperturePos = mk3(0.0f);
    if(fc.pc.aperture != 0.0f)
    {
      f3 cam_right = mk3(V[0], V[4], V[8]);  // Slang mul(viewMatrixI, float4(1,0,0,0)) = M^T e0
      f3 cam_up    = mk3(V[1], V[5], V[9]);
      aperturePos  = (cam_right * cosExact(cam_r1) + cam_up * sinExact(cam_r1)) * sqrtExact(cam_r2);
    }
    direction = normalizeExact(focalPoint - aperturePos);
    origin += aperturePos;
  }
  cp.seed      = seed;
  cp.origin    = origin;
  cp.direction = normalizeExact(direction);  // pathTrace loop head, gltf_pathtrace.slang:447
  return cp;
}

// tryPrimaryMissBackplate (pathtrace_functions.h.slang:944-971): what a camera ray that leaves the scene shows instead of the
// environment, if anything.
// (not inlined, like missEnvironment below: the bounce-0 kernel and the shade kernel must get the same bits out of it, and two
//  inlined copies may be contracted into fused multiply-adds differently)
__device__ __noinline__ f4 primaryMissBackplateCall(const DevScene& scIn, const FrameConsts& fcIn, f3 rayDir)  // .w != 0: there is a backplate, .xyz
{
  const DevScene&    sc = uniformConst(scIn);
  const FrameConsts& fc = uniformConst(fcIn);
  if(hasFlag(fc.frameInfo.flags, MI_SCENE_USE_SOLID_BACKGROUND))
  {
    const f3 c = mk3(fc.frameInfo.backgroundColor);
    return mk4(c.x, c.y, c.z, 1.0f);
  }
  if(hasFlag(fc.frameInfo.flags, MI_SCENE_USE_HDR_ENVIRONMENT) && fc.frameInfo.envBlur > 0.0f)
  {
    f3       dir = rotateAxis(rayDir, mk3(0, 1, 0), -fc.frameInfo.envRotation);
    const f3 c   = smoothHDRBlur(sc, getSphericalUv(dir), fc.frameInfo.envBlur) * fc.frameInfo.envIntensity;
    return mk4(c.x, c.y, c.z, 1.0f);
  }
  return mk4(0.0f);
}
PT_DEV bool primaryMissBackplate(const DevScene& sc, const FrameConsts& fc, f3 rayDir, f3& radiance)
{
  const f4 r = primaryMissBackplateCall(sc, fc, rayDir);
  if(r.w == 0.0f)
    return false;
  radiance = xyz(r);
  return true;
}
// Environment seen by a ray that leaves the scene and its MIS weight against next-event estimation (gltf_pathtrace.slang:139-156).
__device__ __noinline__ f4 missEnvironmentCall(const DevScene& scIn, const FrameConsts& fcIn, f3 rayDir, float lastSamplePdf)
{
  const DevScene&    sc = uniformConst(scIn);
  const FrameConsts& fc = uniformConst(fcIn);
  float envPdf;
  f3    envColor;
  sampleEnvironment(sc, fc, rayDir, envColor, envPdf);
  return mk4(envColor.x, envColor.y, envColor.z, computeEnvHitMisWeight(sc, fc, lastSamplePdf, envPdf));
}
PT_DEV void missEnvironment(const DevScene& sc, const FrameConsts& fc, f3 rayDir, float lastSamplePdf, f3& envColor, float& mis)
{
  const f4 r = missEnvironmentCall(sc, fc, rayDir, lastSamplePdf);  // by value: registers, not the caller's scratch
  envColor   = xyz(r);
  mis        = r.w;
}
// checkInfinitePlaneIntersection (pathtrace_functions.h.slang:556-585): distance along the ray, or a negative number.
PT_DEV float infinitePlaneT(const FrameConsts& fc, f3 rayOrigin, f3 rayDir, float hitT)
{
  if(!hasFlag(fc.frameInfo.flags, MI_SCENE_USE_INFINITE_PLANE))
    return -1.0f;
  const float planeHeight = fc.frameInfo.infinitePlaneDistance, Dn = rayDir.y;
  if(rayOrigin.y > planeHeight && fabsf(Dn) > 1e-6f)
  {
    const float t = divExact(-rayOrigin.y + planeHeight, Dn);
    if(t > 0.0f && t < hitT)
      return t;
  }
  return -1.0f;
}

//================================================================================================================================
// k_generate: camera paths into queue 0 for the per-lane trace kernel (BVH2 scenes and the A/B switch MI_PT_NO_PACKET; 8-wide
// BVH scenes generate their camera rays inside k_trace_primary)
//================================================================================================================================
__global__ void __launch_bounds__(256) k_generate(DevScene sc, FrameConsts fc, PathSoA P, Queues Q, const uint32_t* ownedTiles, int sampleIndex,
                                                   StatCounters* stats)
{
  const uint32_t batchSlots = uint32_t(fc.numSlots) * uint32_t(fc.numFrames);
  uint32_t       slot       = blockIdx.x * blockDim.x + threadIdx.x;
  CameraPath     cp{};
  if(slot < batchSlots)
    cp = generateCameraPath(fc, P, ownedTiles, slot, sampleIndex);
  if(cp.valid)
  {
    if(P.misc)
      P.misc[slot] = make_float4(0.0f, __uint_as_float(PF_ALIVE), __uint_as_float(cp.seed), 0.0f);  // cone.width = 0
    if(stats)
      atomicAdd(&stats->cameraPaths, 1ull);
  }
  // Queue placement is a pure function of the slot (chunk = slot / QCHUNK goes to sub-queue chunk % NSUB): no atomics.
  if(slot < batchSlots)
  {
    const uint32_t chunk = slot / QCHUNK;
    const uint32_t pos   = (chunk % NSUB) * Q.subCap + (chunk / NSUB) * QCHUNK + (slot % QCHUNK);
    Q.active[0].slot[pos] = cp.valid ? slot : QUEUE_DEAD;
    if(cp.valid)
    {
      Q.active[0].org[pos] = make_float4(cp.origin.x, cp.origin.y, cp.origin.z, 0.0f);
      Q.active[0].dir[pos] = make_float4(cp.direction.x, cp.direction.y, cp.direction.z, __uint_as_float(cp.seed));  // .w: the path's seed (alpha draws of the walk)
      if(fc.stateInQueue)
        Q.active[0].misc[pos] = make_float4(0.0f, __uint_as_float(PF_ALIVE), __uint_as_float(cp.seed), 0.0f);  // the state travels with the ray (pt_scene.h)
    }
  }
  if(blockIdx.x == 0 && threadIdx.x < NSUB)
  {
    const uint32_t numChunks = batchSlots / QCHUNK;
    Q.counters[QC_PAIR0 + 2 * threadIdx.x] = QCHUNK * (numChunks / NSUB + (threadIdx.x < numChunks % NSUB ? 1u : 0u));
  }
  if(blockIdx.x == 0 && threadIdx.x < 8)
    Q.counters[QC_HEADS_TRACE + threadIdx.x] = 0;
  (void)sc;
}

//================================================================================================================================
// k_trace_closest: RayQueryRaytracer::Trace (raytracer_interface.h.slang:69-122) on the software BVH
//================================================================================================================================
#ifdef TRACE_PROFILE
// Diagnostics build only (-DTRACE_PROFILE): shader-clock ticks per wave spent in the sections of k_trace_closest.
// [8..15]: lane occupancy -- node steps / lanes visiting / triangle rounds / lanes testing / refills / lanes idle at refill /
// triangle phases / lanes blocked (both park records full) per node step
__device__ unsigned long long g_traceProf[20];  // [18] alpha rounds, [19] triangle-round publish
__device__ unsigned long long g_shadowProf[16];  // the same for k_trace_shadow (8-wide BVH, deferring modes)
#define PROF_T() __builtin_amdgcn_s_memtime()
#define PROF_ADD(i, t0) profAcc[i] += PROF_T() - (t0)
#define PROF_CNT(i, n) profCnt[i] += (unsigned long long)(n)
#else
#define PROF_T() 0ull
#define PROF_ADD(i, t0) (void)(t0)
#define PROF_CNT(i, n) (void)0
#endif

#ifdef SHADE_PROFILE
// Diagnostics build only (-DSHADE_PROFILE, tools/build_variant.sh): shader-clock ticks per wave and lane-ticks (ticks x lanes of the wave inside the region) of the regions of the
// LATER-BOUNCE k_shade<., ., FIRST = false>: [2r] ticks, [2r + 1] lane-ticks; r = 0 entry + hit state, 1 material + textures, 2 sampleLights, 3 bsdfEvaluate, 4 bsdfSample,
// 5 rest of the bounce (ratchet, emissive, shadow origin, roulette, state), 6 miss (inline or deferred list), 7 append + stores, 8 finishMisses, 9 whole round
__device__ unsigned long long g_shadeProf[20];
#define SPROF_BEGIN() const unsigned long long sprofT0_ = __builtin_amdgcn_s_memtime(); const int sprofLanes_ = __popcll(__ballot(true))
#define SPROF_END(r)                                                                                                             \
  do                                                                                                                             \
  {                                                                                                                              \
    if(!FIRST)                                                                                                                   \
    {                                                                                                                            \
      const unsigned long long dt_ = __builtin_amdgcn_s_memtime() - sprofT0_;                                                   \
      if(laneId() == uint32_t(__ffsll((long long)__ballot(true)) - 1))                                                          \
      {                                                                                                                          \
        atomicAdd(&g_shadeProf[2 * (r)], dt_);                                                                                   \
        atomicAdd(&g_shadeProf[2 * (r) + 1], dt_ * (unsigned long long)sprofLanes_);                                             \
      }                                                                                                                          \
    }                                                                                                                            \
  } while(0)
#else
#define SPROF_BEGIN() (void)0
#define SPROF_END(r) (void)0
#endif

// Refill policy of the persistent trace waves: go back for new rays once this many lanes of the wave are idle.
#ifndef REFILL_IDLE_LANES
#define REFILL_IDLE_LANES 16
#endif
// (The any-hit walks' rays are short since they start at the far end, and the refill is a third of their wave time (round-5 profile) -- but their own
//  threshold at 8 / 24 / 32 / 44 measured within +-0.3 % of 16 on every workload: fewer refills serve fewer busy lanes.  One threshold for both.)
// Dense triangle phase of the 8-wide walk: start once this many lanes have parked triangles, leave below the exit count.
#ifndef TRI_PHASE_LANES_N
#define TRI_PHASE_LANES_N 24
#endif
constexpr int TRI_PHASE_LANES      = TRI_PHASE_LANES_N;
constexpr int TRI_PHASE_EXIT_LANES = 10;
// Triangle rounds of the closest-hit walk (triRoundClosest): start one once this many lanes have parked triangles, and let
// a lane hand in at most this many triangles per round.
#ifndef TRI_ROUND_LANES
#define TRI_ROUND_LANES 20
#endif
// (Until round 5 a round also started once 4 lanes had both of their park records in use.  On the reinserted trees that trigger only cut rounds short:
//  never firing it measured atrium 684.6 -> 691.1, street 728.5 -> 739.4 Msamples/s -- profiles/r05_threshold_retune.txt -- and it is gone.)
#ifndef TRI_ROUND_LANE_CAP_N
#define TRI_ROUND_LANE_CAP_N 4  // (7 until round 4.  The publish loop runs as many passes as the busiest lane hands in: 2 / 3 / 4 / 5 / 6 / 7 measured
                                //  576 / 590 / 607 / 606 / 603 / 600 Msamples/s on the atrium, 593 / 609 / 627 / 626 / 624 / 624 on the street)
#endif
constexpr int TRI_ROUND_LANE_CAP = TRI_ROUND_LANE_CAP_N;
// Deferred alpha tests (alphaRound*): flush the wave's list at this many entries, or once this many lanes wait for it.
#ifndef ALPHA_ROUND_MIN
#define ALPHA_ROUND_MIN 32
#endif
#ifndef ALPHA_ROUND_WAITING
#define ALPHA_ROUND_WAITING 8
#endif

// Copies the top of the 8-wide BVH into this workgroup's LDS (whole block; contains a barrier).
PT_DEV uint32_t fillNodeCache(const DevScene& sc, uint4* s_nodes, uint32_t capacity = NODE_CACHE)
{
  const uint32_t n = min(uint32_t(sc.bvh8NumNodes), capacity);
  for(uint32_t i = threadIdx.x; i < n * 5u; i += blockDim.x)
    s_nodes[i] = sc.bvh8Nodes[i];
  __syncthreads();
  return n;
}

// Closest-hit candidate test shared by both BVH flavours: deterministic tie-break, object-space back-face culling,
// stochastic alpha (raytracer_interface.h.slang:76-111).
struct ClosestBest
{
  float    t, u, v;
  int      tri;
  uint32_t rnode, prim;
};
template <bool HAS_ALPHA>
PT_DEV void closestTestLoaded(const DevScene& sc, const RaySetup& r, const DevTri& T, int triIndex, ClosestBest& best, uint32_t& seed0, bool& seedLoaded,
                              const float4* misc, uint32_t slot);
template <bool HAS_ALPHA>
PT_DEV void closestTestTriangle(const DevScene& sc, const RaySetup& r, int triIndex, ClosestBest& best, uint32_t& seed0, bool& seedLoaded,
                                const float4* misc, uint32_t slot)
{
  const DevTri T = sc.tris[triIndex];
  closestTestLoaded<HAS_ALPHA>(sc, r, T, triIndex, best, seed0, seedLoaded, misc, slot);
}
template <bool HAS_ALPHA>
PT_DEV void closestTestLoaded(const DevScene& sc, const RaySetup& r, const DevTri& T, int triIndex, ClosestBest& best, uint32_t& seed0, bool& seedLoaded,
                              const float4* misc, uint32_t slot)
{
  TriHit       h;
  if(!intersectTri(xyz(T.a), xyz(T.b), xyz(T.c), r.org, r.dir, h) || !(h.t > 0.0f))
    return;
  const uint32_t rnode = __float_as_uint(T.a.w), prim = __float_as_uint(T.b.w), flags = __float_as_uint(T.c.w);
  // deterministic closest hit: smaller t wins, exact ties by (renderNode, primitive)
  bool better = h.t < best.t || (h.t == best.t && (rnode < best.rnode || (rnode == best.rnode && prim < best.prim)));
  // RAY_FLAG_CULL_BACK_FACING_TRIANGLES unless TRIANGLE_FACING_CULL_DISABLE; facing is decided in object space
  const bool front = h.front != ((flags & INST_FLIP_FACING) != 0u);
  better           = better && (front || (flags & INST_CULL_DISABLE));
  if(HAS_ALPHA && better && !(flags & (INST_FORCE_OPAQUE | INST_ALPHA_PASSES)))  // (ALPHA_PASSES: opacity 1, the draw always commits)
  {
    if(!seedLoaded)
    {
      seed0      = __float_as_uint(misc[slot].z);
      seedLoaded = true;
    }
    float opacity = getOpacityFast(sc, triIndex, mk3(1.0f - h.u - h.v, h.u, h.v));
    better        = candidateRand(seed0, int(rnode), int(prim)) <= opacity;
  }
  if(better)
  {
    best.t = h.t; best.tri = triIndex; best.u = h.u; best.v = h.v; best.rnode = rnode; best.prim = prim;
  }
}

PT_DEV uint32_t laneCountBelow(unsigned long long mask)  // set bits of `mask` in the lanes below this one
{
  return __builtin_amdgcn_mbcnt_hi(uint32_t(mask >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(mask), 0u));
}
PT_DEV float laneRead(float v, uint32_t lane) { return __int_as_float(__builtin_amdgcn_ds_bpermute(int(lane << 2), __float_as_int(v))); }
PT_DEV uint32_t laneRead(uint32_t v, uint32_t lane) { return uint32_t(__builtin_amdgcn_ds_bpermute(int(lane << 2), int(v))); }

// One round of triangle tests for the whole wave.  After a node step the parked triangles are spread very unevenly over the
// lanes (a few rays sit in front of a big leaf, most have nothing), and testing them lane by lane kept a sixth of the wave
// busy.  Here the owners publish up to 64 parked triangles to a wave-private list, every lane tests ONE of them for the ray
// of the lane that parked it (the ray comes over by lane permute), and the owners read their results back.  The closest hit
// is the minimum over accepted candidates under a total order (t, then renderNode, then primitive) and the alpha draw is a
// pure function of (seed, renderNode, primitive), so the outcome does not depend on who tests what or in which order.
// The round is split in two so that the triangle records are in flight while the wave does its next node step:
// triRoundPublish (owners publish, every lane issues the loads of its triangle) ... node step ... triRoundFinish.
struct TriRound
{
  uint32_t off, n;   // owner: first published item and how many
  uint32_t item;     // worker: triangle | owner lane << 26
  bool     has;
  DevTri   T;
};
PT_DEV void triRoundPublish(const DevScene& sc, bool active, uint32_t& pBase, uint32_t& pMask, uint32_t& qBase, uint32_t& qMask, uint32_t* waveItems, TriRound& tr)
{
  const uint32_t lane = laneId();
  // how many triangles each lane hands in, and where (exclusive prefix over the lanes from the count's bit planes)
  const uint32_t c = active ? min(leafCount(pMask) + leafCount(qMask), uint32_t(TRI_ROUND_LANE_CAP)) : 0u;  // (p / q: leaf words, pt_bvh8.h)
  const unsigned long long b0 = __ballot((c & 1u) != 0u), b1 = __ballot((c & 2u) != 0u), b2 = __ballot((c & 4u) != 0u);
  const uint32_t off   = laneCountBelow(b0) + 2u * laneCountBelow(b1) + 4u * laneCountBelow(b2);
  const uint32_t total = min(64u, uint32_t(__popcll(b0)) + 2u * uint32_t(__popcll(b1)) + 4u * uint32_t(__popcll(b2)));
  const uint32_t n     = off >= 64u ? 0u : min(c, 64u - off);
  for(uint32_t k = 0; __ballot(k < n) != 0ull; ++k)
  {
    if(k < n)
    {
      if(!leafPending(pMask)) { pBase = qBase; pMask = qMask; qMask = 0u; }
      waveItems[off + k] = leafPop(pMask, pBase) | (lane << 26);
    }
  }
  if(!leafPending(pMask)) { pBase = qBase; pMask = qMask; qMask = 0u; }
  __builtin_amdgcn_wave_barrier();
  tr.off  = off;
  tr.n    = n;
  tr.has  = lane < total;
  tr.item = tr.has ? waveItems[lane] : 0u;
  __builtin_amdgcn_wave_barrier();
  if(tr.has)
    tr.T = sc.tris[tr.item & 0x3ffffffu];
}
// Candidates on alpha-tested materials are not resolved where they are found: the test is a chain of two dependent fetches
// (alpha record, then texels) for what is usually one or two lanes of the wave.  They go on a wave-private list (owner lane,
// triangle, t, u, v) and the list is worked off by the whole wave at once -- an alpha round -- when it is long enough or
// rays are waiting for it.  A ray is not finished while it has entries on the list; until then its tmax is merely not as
// tight as it could be.  The draw is a pure function of (seed, renderNode, primitive): deferring does not change it.
PT_DEV void closestUpdate(const DevScene& sc, ClosestBest& best, float tk, float uk, float vk, uint32_t ik)
{
  // smaller t wins, exact ties by (renderNode, primitive) -- looked up only in that rare case
  bool better = tk < best.t;
  if(tk == best.t && best.tri >= 0)
  {
    const DevTri A = sc.tris[ik], B = sc.tris[best.tri];
    const uint32_t ra = __float_as_uint(A.a.w), pa = __float_as_uint(A.b.w), rb = __float_as_uint(B.a.w), pb = __float_as_uint(B.b.w);
    better = ra < rb || (ra == rb && pa < pb);
  }
  if(better)
  {
    best.t = tk; best.u = uk; best.v = vk; best.tri = int(ik);
  }
}
// Where the shadow walk of a scene with transmissive instances (k_trace_shadow MODE 3) records the transmissive candidates it
// meets: a device-wide pool of {t, u, v, triangle}, the entries of one ray chained through `next`; the list heads live in LDS,
// one per lane (= per ray in flight), next to a flag for "some candidate of this ray did not fit the pool".
struct CandRec
{
  float4*   pool;
  uint32_t* next;
  uint32_t* poolCounter;
  uint32_t  cap;
  uint32_t* waveHead;  // this wave's 64 list heads (LDS)
  uint32_t* waveOvf;   // this wave's 64 overflow flags (LDS)
  // the wave's private piece of the pool [cur, end): a device-scope atomic costs microseconds and its result is needed at once, so a
  // wave takes CAND_CHUNK entries at a time (one atomic per ~15 alpha rounds instead of one per round) and wastes what is left at its end
  uint32_t  cur, end;
};
constexpr uint32_t CAND_CHUNK = 256;  // >= 64, the most one alpha round can record
// Records the transmissive shadow candidates of the lanes with `record` (whole wave; {t, u, v, triangle} of each goes to the device-wide pool, out of
// the wave's private piece, chained to its ray -- the lane `owner` of this wave -- through the LDS list heads).
PT_DEV void recordCandidates(CandRec* rec, bool record, uint32_t owner, float t, float u, float v, uint32_t tri)
{
  const unsigned long long m = __ballot(record);
  if(m == 0ull)
    return;
  const uint32_t lane = laneId();
  const uint32_t n    = uint32_t(__popcll(m));
  if(rec->cur + n > rec->end)
  {
    const int first = __ffsll((long long)m) - 1;
    uint32_t  base  = 0u;
    if(int(lane) == first)
      base = atomicAdd(rec->poolCounter, CAND_CHUNK);
    rec->cur = uint32_t(__builtin_amdgcn_readlane(int(base), first));
    rec->end = rec->cur + CAND_CHUNK;
  }
  const uint32_t idx = rec->cur + laneCountBelow(m);
  rec->cur += n;
  if(record)
  {
    if(idx < rec->cap)
    {
      rec->pool[idx] = make_float4(t, u, v, __uint_as_float(tri));
      rec->next[idx] = __hip_atomic_exchange(&rec->waveHead[owner], idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    else
      rec->waveOvf[owner] = 1u;
  }
  __builtin_amdgcn_wave_barrier();
}
// SHADOW: any-hit semantics (an accepted candidate occludes, draw < opacity); otherwise closest-hit (draw <= opacity).
// REC (shadow only): candidates on transmissive instances are not decided here -- their effect depends on their order along
// the ray (Beer segments, raytracer_interface.h.slang:160-178) -- but recorded for k_shadow_resolve.
template <bool SHADOW, bool REC = false>
PT_DEV void alphaRound(const DevScene& sc, uint32_t seed0, uint4* waveAlpha, uint32_t& aCount, bool& aPending, ClosestBest& best, bool& occluded,
                       CandRec* rec = nullptr)
{
  const uint32_t lane = laneId();
  const bool     has  = lane < aCount;
  const uint4    e    = has ? waveAlpha[lane] : make_uint4(0u, 0u, 0u, 0u);
  __builtin_amdgcn_wave_barrier();
  const uint32_t owner = e.x >> 26, tri = e.x & 0x3ffffffu;
  const uint32_t seedSrc = laneRead(seed0, owner);
  bool           accept  = false, record = false;
  if(has)
  {
    const float    u = __uint_as_float(e.z), v = __uint_as_float(e.w);
    const uint32_t rnode = __float_as_uint(sc.tris[tri].a.w), prim = __float_as_uint(sc.tris[tri].b.w);
    if(REC && (__float_as_uint(sc.tris[tri].c.w) & INST_TRANSMISSIVE))
      record = true;
    else
    {
      const float draw = candidateRand(seedSrc, int(rnode), int(prim)), opacity = getOpacityFast(sc, int(tri), mk3(1.0f - u - v, u, v));
      accept = SHADOW ? draw < opacity : draw <= opacity;
    }
  }
  if(REC)
    recordCandidates(rec, record, owner, __uint_as_float(e.y), __uint_as_float(e.z), __uint_as_float(e.w), tri);
  unsigned long long acc = __ballot(accept);
  while(acc != 0ull)
  {
    const int k = __ffsll((long long)acc) - 1;
    acc &= acc - 1ull;
    const uint32_t item = uint32_t(__builtin_amdgcn_readlane(int(e.x), k));
    if(lane == (item >> 26))
    {
      if(SHADOW)
        occluded = true;
      else
        closestUpdate(sc, best, __int_as_float(__builtin_amdgcn_readlane(int(e.y), k)), __int_as_float(__builtin_amdgcn_readlane(int(e.z), k)),
                      __int_as_float(__builtin_amdgcn_readlane(int(e.w), k)), item & 0x3ffffffu);
    }
  }
  aCount   = 0u;
  aPending = false;
}
// puts this round's alpha candidates on the list (flushing it first when they would not fit)
template <bool SHADOW, bool REC = false>
PT_DEV void alphaDefer(const DevScene& sc, bool needAlpha, uint32_t item, float t, float u, float v, uint32_t seed0, uint4* waveAlpha, uint32_t& aCount,
                       bool& aPending, ClosestBest& best, bool& occluded, CandRec* rec = nullptr)
{
  const unsigned long long m = __ballot(needAlpha);
  if(m == 0ull)
    return;
  if(aCount + uint32_t(__popcll(m)) > 64u)
    alphaRound<SHADOW, REC>(sc, seed0, waveAlpha, aCount, aPending, best, occluded, rec);
  if(needAlpha)
    waveAlpha[aCount + laneCountBelow(m)] = make_uint4(item, __float_as_uint(t), __float_as_uint(u), __float_as_uint(v));
  aCount += uint32_t(__popcll(m));
  __builtin_amdgcn_wave_barrier();
}

template <bool HAS_ALPHA, bool COUNT>
PT_DEV void triRoundFinish(const DevScene& sc, const RaySetup& r, ClosestBest& best, uint32_t seed0, const TriRound& tr, unsigned& tris,
                           unsigned long long* waveSlots, uint4* waveAlpha, uint32_t& aCount, bool& aPending, unsigned long long* profAcc = nullptr)
{
  const unsigned long long tTest = PROF_T();
  // ---- every lane tests its triangle for the owner's ray
  const uint32_t src = tr.item >> 26, tri = tr.item & 0x3ffffffu;
  const f3       org = mk3(laneRead(r.org.x, src), laneRead(r.org.y, src), laneRead(r.org.z, src));
  const f3       dir = mk3(laneRead(r.dir.x, src), laneRead(r.dir.y, src), laneRead(r.dir.z, src));
  const float    tmaxSrc = laneRead(best.t, src);
  float          rt = INFINITE_F, ru = 0.0f, rv = 0.0f, tHit = 0.0f;
  bool           needAlpha = false;
  if(tr.has)
  {
    const DevTri& T = tr.T;
    TriHit        h;
    if(intersectTri(xyz(T.a), xyz(T.b), xyz(T.c), org, dir, h) && h.t > 0.0f && h.t <= tmaxSrc)
    {
      const uint32_t flags = __float_as_uint(T.c.w);
      // RAY_FLAG_CULL_BACK_FACING_TRIANGLES unless TRIANGLE_FACING_CULL_DISABLE; facing is decided in object space
      const bool front = h.front != ((flags & INST_FLIP_FACING) != 0u);
      if(front || (flags & INST_CULL_DISABLE))
      {
        needAlpha = HAS_ALPHA && !(flags & (INST_FORCE_OPAQUE | INST_ALPHA_PASSES));  // (ALPHA_PASSES: opacity 1, the draw always commits)
        tHit      = h.t;
        rt        = needAlpha ? -1.0f : h.t;  // negative tells the owner "deferred"
        ru        = h.u;
        rv        = h.v;
      }
    }
  }
  PROF_ADD(3, tTest);
  const unsigned long long tGather = PROF_T();
  if(COUNT) tris += tr.has ? 1u : 0u;
  if(HAS_ALPHA)
  {
    bool none = false;
    alphaDefer<false>(sc, needAlpha, tr.item, tHit, ru, rv, seed0, waveAlpha, aCount, aPending, best, none);
  }
  // ---- owners collect.  Each lane's best candidate of this round is found by a 64-bit minimum (t | testing lane) on the
  // owner's slot of the wave's LDS list; u, v and the triangle then come from the winning lane.  Two candidates of one ray
  // at exactly the same t must be ordered by (renderNode, primitive): that case (coincident geometry) takes the slow way.
  const uint32_t           lane = laneId();
  const unsigned long long mine = tr.n ? (((tr.n >= 64u ? 0ull : (1ull << tr.n)) - 1ull) << tr.off) : 0ull;  // this owner's items: lanes [off, off + n)
  if(HAS_ALPHA && (__ballot(needAlpha) & mine))
    aPending = true;
  const bool cand = rt >= 0.0f && rt < INFINITE_F;
  waveSlots[lane] = ~0ull;
  __builtin_amdgcn_wave_barrier();
  if(cand)
    (void)__hip_atomic_fetch_min(&waveSlots[src], ((unsigned long long)__float_as_uint(rt) << 32) | lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  __builtin_amdgcn_wave_barrier();
  const unsigned long long won = waveSlots[lane], wonSrc = waveSlots[src];
  __builtin_amdgcn_wave_barrier();
  const bool tie = cand && uint32_t(wonSrc >> 32) == __float_as_uint(rt) && uint32_t(wonSrc) != lane;
  if(__ballot(tie) == 0ull)
  {
    const uint32_t j  = uint32_t(won) & 63u;
    const float    uk = laneRead(ru, j), vk = laneRead(rv, j);
    const uint32_t ik = laneRead(tri, j);
    if(won != ~0ull)
      closestUpdate(sc, best, __uint_as_float(uint32_t(won >> 32)), uk, vk, ik);
  }
  else
  {
    for(uint32_t k = 0; __ballot(k < tr.n) != 0ull; ++k)
    {
      const uint32_t j  = (tr.off + k) & 63u;
      const float    tk = laneRead(rt, j), uk = laneRead(ru, j), vk = laneRead(rv, j);
      const uint32_t ik = laneRead(tri, j);
      if(k < tr.n && tk >= 0.0f && tk < INFINITE_F)
        closestUpdate(sc, best, tk, uk, vk, ik);
    }
  }
  PROF_ADD(7, tGather);
}

// Second half of a triangle round of the any-hit (shadow) walk without transmissive instances: a candidate that commits
// decides its ray (raytracer_interface.h.slang:149-179), so the owners only need to know whether any of theirs did.
template <bool HAS_ALPHA, bool COUNT, bool REC = false>
PT_DEV void triRoundFinishShadow(const DevScene& sc, const RaySetup& r, float tMax, uint32_t seed0, const TriRound& tr, bool& occluded, unsigned& tris,
                                 uint4* waveAlpha, uint32_t& aCount, bool& aPending, CandRec* rec = nullptr)
{
  const uint32_t src = tr.item >> 26;
  const f3       org = mk3(laneRead(r.org.x, src), laneRead(r.org.y, src), laneRead(r.org.z, src));
  const f3       dir = mk3(laneRead(r.dir.x, src), laneRead(r.dir.y, src), laneRead(r.dir.z, src));
  const float    tmaxSrc = laneRead(tMax, src);
  bool           commits = false, needAlpha = false, recordNow = false;
  float          hu = 0.0f, hv = 0.0f, ht = 0.0f;
  if(tr.has)
  {
    const DevTri& T = tr.T;
    TriHit        h;
    if(intersectTri(xyz(T.a), xyz(T.b), xyz(T.c), org, dir, h) && h.t > 0.0f && h.t < tmaxSrc)
    {
      const uint32_t flags = __float_as_uint(T.c.w);
      // RAY_FLAG_NONE: no culling; opaque geometry commits.  Non-transmissive alpha material: an accepted candidate multiplies
      // the transmission by getShadowTransmission() == 0 (pathtrace_functions.h.slang:256-261) whatever its position in the
      // order -- the draw is deferred to an alpha round
      // INST_ALPHA_PASSES (alphaMode OPAQUE on a non-opaque instance): the draw is known to commit.  On a transmissive instance the candidate is
      // recorded here and now (REC) -- an alpha round would do nothing else with it; on any other it occludes like an opaque one
      const bool passes = (flags & INST_ALPHA_PASSES) != 0u;
      recordNow = REC && HAS_ALPHA && passes && (flags & INST_TRANSMISSIVE) != 0u;
      needAlpha = HAS_ALPHA && !recordNow && !(flags & INST_FORCE_OPAQUE) && !(passes && !(REC && (flags & INST_TRANSMISSIVE)));
      commits   = !needAlpha && !recordNow;
      hu        = h.u;
      hv        = h.v;
      ht        = h.t;
    }
  }
  if(COUNT) tris += tr.has ? 1u : 0u;
  const unsigned long long mine = tr.n ? (((tr.n >= 64u ? 0ull : (1ull << tr.n)) - 1ull) << tr.off) : 0ull;  // this owner's items: lanes [off, off + n)
  if(REC)
    recordCandidates(rec, recordNow, src, ht, hu, hv, tr.item & 0x3ffffffu);
  if(HAS_ALPHA)
  {
    ClosestBest none{};
    alphaDefer<true, REC>(sc, needAlpha, tr.item, ht, hu, hv, seed0, waveAlpha, aCount, aPending, none, occluded, rec);
    if(__ballot(needAlpha) & mine)
      aPending = true;
  }
  if(__ballot(commits) & mine)
    occluded = true;
}

template <bool WIDE, bool HAS_ALPHA, bool COUNT>
__global__ void __launch_bounds__(TRACE_BLOCK, TRACE_MIN_WAVES) k_trace_closest(DevScene sc, PathSoA P, Queues Q, int cur, StatCounters* stats)
{
  __shared__ int      s_stack[BVH_STACK_LDS * TRACE_BLOCK];  // BVH2: 24 ints/lane; BVH8: 12 node groups x 2 ints/lane
  __shared__ uint32_t s_prefix[NSUB + 1];
  constexpr int CACHE = HAS_ALPHA ? NODE_CACHE_ALPHA : NODE_CACHE;
  __shared__ uint4    s_nodes[WIDE ? CACHE * 5 : 1];
  __shared__ unsigned long long s_slots[WIDE ? TRACE_BLOCK : 1];     // triangle rounds: 64 published triangles per wave (first half
                                                                      // of the wave's 512 bytes), then the 64 result slots
  __shared__ uint4    s_alpha[(WIDE && HAS_ALPHA) ? TRACE_BLOCK : 1];  // deferred alpha tests: 64 entries per wave

  static_assert(2 * BVH8_STACK_LDS == BVH_STACK_LDS, "both stack flavours share one LDS allocation");
  if(blockIdx.x == 0 && threadIdx.x < NSUB)
  {
    // the shade kernel of this iteration appends to these; zero them here (the kernel boundary orders the writes)
    // (the counters of the SHADOW stage -- its feed heads, the candidate pool, the overflow list -- are zeroed by k_shade, the launch
    //  between the previous iteration's shadow stage and this one's: with frames in flight below MI_PT_OVERLAP that stage runs on a
    //  second stream next to THIS kernel, which therefore must not touch them)
    Q.counters[(cur ? QC_PAIR0 : QC_PAIR1) + 2 * threadIdx.x]     = 0;
    Q.counters[(cur ? QC_PAIR0 : QC_PAIR1) + 2 * threadIdx.x + 1] = 0;
  }
  const RayQueue in = Q.active[cur];
  queuePrefix(&Q.counters[cur ? QC_PAIR1 : QC_PAIR0], s_prefix);
  WaveFeed feed;
  feedInit(feed, s_prefix[NSUB]);
  if(!feedBlockHasWork(feed))
    return;
  const uint32_t cachedNodes = WIDE ? fillNodeCache(sc, s_nodes, uint32_t(CACHE)) : 0u;
  LaneStack  st;   // BVH2 walk state
  LaneStack2 st2;  // BVH8 walk state
  int        stackOverflow[WIDE ? 2 * BVH8_STACK_PRIV : BVH_STACK_PRIV];  // scratch; only touched beyond the LDS depth
  st.lds = s_stack;  st.tid = int(threadIdx.x);  st.stride = TRACE_BLOCK;  st.sp = 0;  st.priv = stackOverflow;
  st2.lds = s_stack; st2.tid = int(threadIdx.x); st2.stride = TRACE_BLOCK; st2.sp = 0;
  st2.privBase = reinterpret_cast<uint32_t*>(stackOverflow); st2.privBits = reinterpret_cast<uint32_t*>(stackOverflow) + (WIDE ? BVH8_STACK_PRIV : 0);
  bool        active = false;
  uint32_t    slot = 0, pos = 0;
  RaySetup    r{};
  int         node = BVH_EMPTY;
  NodeGroup   G{0, 0};
  uint32_t    octinv = 0;
  uint32_t    pBase = 0, pMask = 0, qBase = 0, qMask = 0;  // parked leaf hits (8-wide walk): triangle base + leaf word (pt_bvh8.h)
  int         lastVisiting = 64;                           // lanes that visited a node in the previous step
  uint32_t    aCount   = 0;                                // deferred alpha tests on this wave's list (wave-uniform)
  bool        aPending = false;                            // ... some of them this lane's
  uint4*      waveAlpha = s_alpha + ((WIDE && HAS_ALPHA) ? (threadIdx.x & ~63u) : 0u);
  ClosestBest best{INFINITE_F, 0.0f, 0.0f, -1, 0xffffffffu, 0xffffffffu};
  uint32_t    seed0 = 0;
  bool        seedLoaded = false;
  unsigned    nodes = 0, tris = 0, rays = 0;
  // every lane keeps its NEXT ray prefetched in registers: the loads are issued one ray ahead and land while the lane walks
  bool     pValid = false;
  uint32_t pPos = 0, pSlot = QUEUE_DEAD;
  float4   pO = make_float4(0, 0, 0, 0), pD = make_float4(0, 0, 0, 0);
#ifdef TRACE_PROFILE
  unsigned long long profAcc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long profCnt[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long profStart = PROF_T();
#endif
  for(;;)
  {
    const unsigned long long tFeed = PROF_T();
    PROF_CNT(4, 1);
    PROF_CNT(5, 64 - __popcll(__ballot(active)));
    // ---- idle lanes start their prefetched ray; every lane without a prefetched ray takes the next queue index
    if(!active && pValid)
    {
      pValid = false;
      if(pSlot != QUEUE_DEAD)
      {
        slot       = pSlot;
        pos        = pPos;
        r          = makeRaySetup(xyz(pO), xyz(pD));
        best       = ClosestBest{INFINITE_F, 0.0f, 0.0f, -1, 0xffffffffu, 0xffffffffu};  // t = ray.TMax
        seed0      = __float_as_uint(pD.w);  // the alpha draws of this ray: the path's seed travels in the queue entry (dir.w)
        seedLoaded = true;
        if(WIDE)
        {
          octinv = rayOctInv(r.idir);
          G      = rootGroup(octinv);
          st2.sp = 0;
        }
        else
        {
          st.sp = 0;
          node  = sc.bvhRoot;
        }
        active = sc.bvhRoot != BVH_EMPTY;
        if(!active)
          in.aux[pos] = make_float4(INFINITE_F, __int_as_float(-1), 0.0f, 0.0f);
        if(COUNT) ++rays;
      }
    }
    PROF_ADD(4, tFeed);
    if(!feed.exhausted)
    {
      const unsigned long long tTake = PROF_T();
      uint32_t flat = feedTake(feed, !pValid, &Q.counters[QC_HEADS_TRACE]);
      PROF_ADD(5, tTake);
      const unsigned long long tIssue = PROF_T();
      if(flat != 0xffffffffu)
      {
        pPos   = queuePos(Q.subCap, s_prefix, flat);
        pSlot  = in.slot[pPos];
        pO     = in.org[pPos];
        pD     = in.dir[pPos];
        pValid = true;
      }
      PROF_ADD(6, tIssue);
    }
    const bool moreWork = !feed.exhausted || __ballot(pValid) != 0ull;
    if(__ballot(active) == 0ull)
    {
      if(!moreWork)
        break;
      continue;
    }
    PROF_ADD(0, tFeed);
    // ---- walk until enough lanes have finished to make a refill worthwhile
    for(;;)
    {
      if(WIDE)
      {
        // Triangle round, first half: once enough lanes have parked triangles (or a lane has no room left, or few lanes are
        // still walking) they are published and every lane starts loading one; the loads fly during the node step below.
        const unsigned long long tTri0 = PROF_T();
        TriRound           tr;
        bool               round = false;
        unsigned long long pend  = __ballot(active && leafPending(pMask));
        if(pend != 0ull)
        {
          const bool drain = lastVisiting < TRI_PHASE_LANES;  // few lanes left walking: nothing to wait for
          // (round 6, review item 4a: a round started on the number of TRIANGLES the wave would hand in -- 32 / 48 / 60 instead of 20 lanes holding some --
          //  measured atrium closest-hit walk 1.365 -> 1.464 / 1.387 / 1.385 ms per frame, street 5.44 -> 5.82 / 5.51 / 5.50: profiles/r06_shade_walk_ab.txt)
          round            = drain || __popcll(pend) >= TRI_ROUND_LANES;
          if(round)
          {
            PROF_CNT(2, 1);
            PROF_CNT(3, __popcll(pend));
            triRoundPublish(sc, active, pBase, pMask, qBase, qMask, reinterpret_cast<uint32_t*>(s_slots + (threadIdx.x & ~63u)), tr);
          }
        }
        PROF_ADD(2, tTri0);
        const unsigned long long tNode = PROF_T();
        // Node step: lanes with room for one more leaf record visit their next node.  Triangles are NOT tested here: a
        // memory instruction costs the CU's address unit the same 64 lane-slots whether 3 or 64 lanes are active, and
        // straight after a node visit only a few lanes have triangles.  The hits are parked (two records per lane) and
        // tested in a dense phase once enough lanes have some; the closest hit does not depend on the test order.
        bool visited = false;
        if(active && !leafPending(qMask))
        {
          if((G.bits >> 8) == 0u && st2.sp > 0)
            G = st2.pop();
          if(G.bits >> 8)
          {
            const uint32_t child = groupPopChild(G, octinv);
            if(G.bits >> 8)
              st2.push(G);
            uint32_t tBase, tMask;
            bvh8Visit(sc, r, best.t, octinv, child, G, tBase, tMask, s_nodes, cachedNodes);
            if(COUNT) ++nodes;
            visited = true;
            if(leafPending(tMask))
            {
              if(!leafPending(pMask)) { pBase = tBase; pMask = tMask; }
              else                    { qBase = tBase; qMask = tMask; }
            }
          }
        }
        PROF_ADD(1, tNode);
        PROF_CNT(0, 1);
        PROF_CNT(1, __popcll(__ballot(visited)));
        PROF_CNT(7, __popcll(__ballot(active && leafPending(qMask))));
        lastVisiting = __popcll(__ballot(visited));
        const unsigned long long tTri = PROF_T();
        if(round)
#ifdef TRACE_PROFILE
          triRoundFinish<HAS_ALPHA, COUNT>(sc, r, best, seed0, tr, tris, s_slots + (threadIdx.x & ~63u), waveAlpha, aCount, aPending, profAcc);
#else
          triRoundFinish<HAS_ALPHA, COUNT>(sc, r, best, seed0, tr, tris, s_slots + (threadIdx.x & ~63u), waveAlpha, aCount, aPending);
#endif
        if(HAS_ALPHA && aCount != 0u)
        {
          // alpha round: the list is long enough, or the walk is running dry, or rays have nothing left to do but wait for it
          const bool walked  = active && !leafPending(pMask) && (G.bits >> 8) == 0u && st2.sp == 0;
          const int  waiting = __popcll(__ballot(walked && aPending));
          if(aCount >= ALPHA_ROUND_MIN || lastVisiting < TRI_PHASE_LANES || waiting >= ALPHA_ROUND_WAITING)
          {
            const unsigned long long tAlpha = PROF_T();
            bool none = false;
            alphaRound<false>(sc, seed0, waveAlpha, aCount, aPending, best, none);
            PROF_ADD(8, tAlpha);
            PROF_CNT(8, 1);
          }
        }
        PROF_ADD(2, tTri);
        if(active && !leafPending(pMask) && (G.bits >> 8) == 0u && st2.sp == 0 && !(HAS_ALPHA && aPending))
        {
          in.aux[pos] = make_float4(best.t, __int_as_float(best.tri), best.u, best.v);
          active      = false;
        }
      }
      else if(active)
      {
        bool finished = false;
        // inner nodes first (bounded), so that most lanes arrive at a leaf together
#pragma unroll 1
        for(int k = 0; k < 4 && node >= 0; ++k)
        {
          node = bvhInnerStep(sc, r, best.t, node, st);
          if(COUNT) ++nodes;
        }
        if(node < 0 && node != BVH_EMPTY)
        {
          if(COUNT) ++tris;
          closestTestTriangle<HAS_ALPHA>(sc, r, ~node, best, seed0, seedLoaded, P.misc, slot);
          node = bvhPop(st);
        }
        finished = node == BVH_EMPTY;
        if(finished)
        {
          in.aux[pos] = make_float4(best.t, __int_as_float(best.tri), best.u, best.v);
          active      = false;
        }
      }
      const unsigned long long act = __ballot(active);
      if(act == 0ull || (moreWork && __popcll(act) <= 64 - REFILL_IDLE_LANES))
        break;
    }
  }
#ifdef TRACE_PROFILE
  if(laneId() == 0)
  {
    atomicAdd(&g_traceProf[0], profAcc[0]);
    atomicAdd(&g_traceProf[1], profAcc[1]);
    atomicAdd(&g_traceProf[2], profAcc[2]);
    atomicAdd(&g_traceProf[3], PROF_T() - profStart);
    atomicAdd(&g_traceProf[4], 1ull);
    atomicAdd(&g_traceProf[5], profAcc[4]);
    atomicAdd(&g_traceProf[6], profAcc[5]);
    atomicAdd(&g_traceProf[7], profAcc[6]);
    for(int i = 0; i < 8; ++i)
      atomicAdd(&g_traceProf[8 + i], profCnt[i]);
    atomicAdd(&g_traceProf[16], profAcc[3]);
    atomicAdd(&g_traceProf[17], profAcc[7]);
    atomicAdd(&g_traceProf[18], profAcc[8]);
    atomicAdd(&g_traceProf[19], profCnt[8]);
  }
#endif
  if(COUNT)
  {
    atomicAdd(&stats->segments, (unsigned long long)rays);
    atomicAdd(&stats->nodesClosest, (unsigned long long)nodes);
    atomicAdd(&stats->trisClosest, (unsigned long long)tris);
  }
}

//================================================================================================================================
// k_trace_primary: bounce 0 of an 8-wide-BVH scene in ONE kernel -- camera-ray generation, closest hit, and the end of every
// path whose camera ray leaves the scene.
//
// One 8x8-pixel micro-tile per wave.  The 64 camera rays of a wave are nearly parallel and visit almost the same nodes, so the
// walk is a PACKET walk: one traversal stack per wave, node and triangle records fetched ONCE per wave through the scalar cache
// (s_load, no vector-memory gather at all), every lane slab-tests the node's children with its own origin / tmax, and a child
// is entered when ANY lane hits it.  Per-lane closest hits, tie-breaks, culling and alpha draws are those of k_trace_closest,
// and box tests only ever prune, so the hit records are bit-identical.
//
// Camera rays are a pure function of the path slot, so they are generated here instead of being written to HBM by one kernel
// and read back by the next (1.8 GB + 4 GB per 32-frame 1080p batch in round 1).  A camera ray that hits nothing (and cannot hit
// the infinite plane) ends its path on the spot -- environment or backplate at full weight, exactly what the shade kernel would
// compute for it (gltf_pathtrace.slang:129-156 with the initial state lastSamplePdf = DIRAC) -- and is never queued.  The rays
// that did hit something are packed WITHIN their 256-slot chunk: a workgroup owns the chunk, survivors go to the front of the
// chunk's range of queue 0, the rest of the range is marked dead.  No atomics, the queue counts stay the static ones, and the
// bounce-0 shade launch sees waves that are full of surface hits or empty (the hit/miss part of the per-bounce sort).
//================================================================================================================================
constexpr int PACKET_STACK = 96;  // node groups per wave (tree depth bound; an overflow falls back to per-lane results below)

typedef uint32_t u32x4s __attribute__((ext_vector_type(4)));
// 80-B node / 48-B triangle through the scalar unit.  The address MUST be wave-uniform.
PT_DEV void scalarLoadNode(const uint4* nodes, uint32_t index, uint4& n0, uint4& n1, uint4& n2, uint4& n3, uint4& n4)
{
  const uint64_t addr = uint64_t(reinterpret_cast<uintptr_t>(nodes)) + uint64_t(index) * 80ull;
  const uint64_t A    = (uint64_t(uint32_t(__builtin_amdgcn_readfirstlane(uint32_t(addr >> 32)))) << 32) | uint64_t(uint32_t(__builtin_amdgcn_readfirstlane(uint32_t(addr))));  // (the builtin returns int: widen unsigned)
  u32x4s         a, b, c, d, e;
  asm volatile("s_load_dwordx4 %0, %5, 0x0\n\ts_load_dwordx4 %1, %5, 0x10\n\ts_load_dwordx4 %2, %5, 0x20\n\ts_load_dwordx4 %3, %5, 0x30\n\t"
               "s_load_dwordx4 %4, %5, 0x40\n\ts_waitcnt lgkmcnt(0)"
               : "=&s"(a), "=&s"(b), "=&s"(c), "=&s"(d), "=&s"(e)
               : "s"(A)
               : "memory");
  n0 = make_uint4(a[0], a[1], a[2], a[3]); n1 = make_uint4(b[0], b[1], b[2], b[3]); n2 = make_uint4(c[0], c[1], c[2], c[3]);
  n3 = make_uint4(d[0], d[1], d[2], d[3]); n4 = make_uint4(e[0], e[1], e[2], e[3]);
}
// Packet node of a one-octant packet: the header words of the 80-B record and, from DevScene::bvh8Planes, the near and the far
// block of each axis (eight floats = one child plane each), at the byte offsets the octant selects.  All wave-uniform.
PT_DEV void scalarLoadNodePlanes(const uint4* nodes, const float* planes, uint32_t index, uint32_t offNx, uint32_t offNy, uint32_t offNz, uint32_t offFx,
                                 uint32_t offFy, uint32_t offFz, uint4& n0, uint4& n1, f32x8s& pnx, f32x8s& pny, f32x8s& pnz, f32x8s& pfx, f32x8s& pfy,
                                 f32x8s& pfz)
{
  const uint64_t addr = uint64_t(reinterpret_cast<uintptr_t>(nodes)) + uint64_t(index) * 80ull;
  const uint64_t A    = (uint64_t(uint32_t(__builtin_amdgcn_readfirstlane(uint32_t(addr >> 32)))) << 32) | uint64_t(uint32_t(__builtin_amdgcn_readfirstlane(uint32_t(addr))));
  const uint64_t pad  = uint64_t(reinterpret_cast<uintptr_t>(planes)) + uint64_t(index) * 192ull;
  const uint64_t Pn   = (uint64_t(uint32_t(__builtin_amdgcn_readfirstlane(uint32_t(pad >> 32)))) << 32) | uint64_t(uint32_t(__builtin_amdgcn_readfirstlane(uint32_t(pad))));
  u32x4s         a, b;
  asm volatile("s_load_dwordx4 %0, %8, 0x0\n\ts_load_dwordx4 %1, %8, 0x10\n\t"
               "s_load_dwordx8 %2, %9, %10\n\ts_load_dwordx8 %3, %9, %11\n\ts_load_dwordx8 %4, %9, %12\n\t"
               "s_load_dwordx8 %5, %9, %13\n\ts_load_dwordx8 %6, %9, %14\n\ts_load_dwordx8 %7, %9, %15\n\ts_waitcnt lgkmcnt(0)"
               : "=&s"(a), "=&s"(b), "=&s"(pnx), "=&s"(pny), "=&s"(pnz), "=&s"(pfx), "=&s"(pfy), "=&s"(pfz)
               : "s"(A), "s"(Pn), "s"(offNx), "s"(offNy), "s"(offNz), "s"(offFx), "s"(offFy), "s"(offFz)
               : "memory");
  n0 = make_uint4(a[0], a[1], a[2], a[3]);
  n1 = make_uint4(b[0], b[1], b[2], b[3]);
}
// the two header words of a node alone (origin, exponents, inner mask | child base, triangle base, valid16 of the leaf triangles)
PT_DEV void scalarLoadNodeHeader(const uint4* nodes, uint32_t index, uint4& n0, uint4& n1)
{
  const uint64_t addr = uint64_t(reinterpret_cast<uintptr_t>(nodes)) + uint64_t(index) * 80ull;
  const uint64_t A    = (uint64_t(uint32_t(__builtin_amdgcn_readfirstlane(uint32_t(addr >> 32)))) << 32) | uint64_t(uint32_t(__builtin_amdgcn_readfirstlane(uint32_t(addr))));
  u32x4s         a, b;
  asm volatile("s_load_dwordx4 %0, %2, 0x0\n\ts_load_dwordx4 %1, %2, 0x10\n\ts_waitcnt lgkmcnt(0)" : "=&s"(a), "=&s"(b) : "s"(A) : "memory");
  n0 = make_uint4(a[0], a[1], a[2], a[3]);
  n1 = make_uint4(b[0], b[1], b[2], b[3]);
}
// ---- cross-lane pieces of the interval node test (pt_packet.h): lane = child * 8 + plane
template <int CTRL>
PT_DEV float dppMove(float v)  // every lane reads the lane CTRL names inside its row; all 64 lanes take part
{
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
// over the aligned group of eight lanes: quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror
PT_DEV float group8Max(float v)
{
  v = fmaxf(v, dppMove<0xB1>(v));
  v = fmaxf(v, dppMove<0x4E>(v));
  return fmaxf(v, dppMove<0x141>(v));
}
PT_DEV float group8Min(float v)
{
  v = fminf(v, dppMove<0xB1>(v));
  v = fminf(v, dppMove<0x4E>(v));
  return fminf(v, dppMove<0x141>(v));
}
PT_DEV float waveMax(float v)
{
#pragma unroll
  for(int o = 32; o >= 1; o >>= 1)
    v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
PT_DEV float waveMin(float v)
{
#pragma unroll
  for(int o = 32; o >= 1; o >>= 1)
    v = fminf(v, __shfl_xor(v, o));
  return v;
}
// bits 0, 8, ..., 56 of a ballot -> bits 0..7
PT_DEV uint32_t everyEighthBit(unsigned long long m)
{
  const uint32_t lo = uint32_t(m) & 0x01010101u, hi = uint32_t(m >> 32) & 0x01010101u;
  return ((lo * 0x01020408u) >> 24) | (((hi * 0x01020408u) >> 24) << 4);
}
PT_DEV DevTri scalarLoadTri(const DevTri* tris, uint32_t index)
{
  const uint64_t addr = uint64_t(reinterpret_cast<uintptr_t>(tris)) + uint64_t(index) * 48ull;
  const uint64_t A    = (uint64_t(uint32_t(__builtin_amdgcn_readfirstlane(uint32_t(addr >> 32)))) << 32) | uint64_t(uint32_t(__builtin_amdgcn_readfirstlane(uint32_t(addr))));  // (the builtin returns int: widen unsigned)
  u32x4s         a, b, c;
  asm volatile("s_load_dwordx4 %0, %3, 0x0\n\ts_load_dwordx4 %1, %3, 0x10\n\ts_load_dwordx4 %2, %3, 0x20\n\ts_waitcnt lgkmcnt(0)"
               : "=&s"(a), "=&s"(b), "=&s"(c)
               : "s"(A)
               : "memory");
  DevTri T;
  T.a = make_float4(__uint_as_float(a[0]), __uint_as_float(a[1]), __uint_as_float(a[2]), __uint_as_float(a[3]));
  T.b = make_float4(__uint_as_float(b[0]), __uint_as_float(b[1]), __uint_as_float(b[2]), __uint_as_float(b[3]));
  T.c = make_float4(__uint_as_float(c[0]), __uint_as_float(c[1]), __uint_as_float(c[2]), __uint_as_float(c[3]));
  return T;
}

#ifndef PRIMARY_MIN_WAVES
#define PRIMARY_MIN_WAVES 4
#endif
// INTERVAL: the launch whose packets are one pixel's samples (pixel-major slots): eligible packets take the interval node test
// (pt_packet.h), the others the per-ray byte path.  The per-ray plane path lives in the other instantiation only, so that neither
// pays the other's registers.
#ifndef PRIMARY_INTERVAL_MIN_WAVES
#define PRIMARY_INTERVAL_MIN_WAVES PRIMARY_MIN_WAVES
#endif
template <bool HAS_ALPHA, bool COUNT, bool INTERVAL>
__global__ void __launch_bounds__(256, INTERVAL ? PRIMARY_INTERVAL_MIN_WAVES : PRIMARY_MIN_WAVES) k_trace_primary(DevScene sc, FrameConsts fc, const DevScene* __restrict__ scp, const FrameConsts* __restrict__ fcp, PathSoA P,
                                                        Queues Q, const uint32_t* ownedTiles, int sampleIndex, uint32_t batchSlots, StatCounters* stats)
{
  // `sc` / `fc` (kernel arguments, SGPRs) serve the inlined generation and walk; the non-inlined environment helpers of the miss
  // path get the device-resident copies so that the arguments' addresses never escape (no scratch copy, cf. k_shade)
  __shared__ uint32_t s_wstack[4][PACKET_STACK][2];
  __shared__ uint32_t s_cnt[4];
  if(blockIdx.x == 0 && threadIdx.x < NSUB)
  {
    // queue 0 holds one (partly dead) chunk per workgroup: static counts, as k_generate writes them
    const uint32_t numChunks = batchSlots / QCHUNK;
    Q.counters[QC_PAIR0 + 2 * threadIdx.x] = QCHUNK * (numChunks / NSUB + (threadIdx.x < numChunks % NSUB ? 1u : 0u));
    // same hand-over as k_trace_closest(cur = 0): the shade kernel of this iteration appends to these
    Q.counters[QC_PAIR1 + 2 * threadIdx.x]     = 0;
    Q.counters[QC_PAIR1 + 2 * threadIdx.x + 1] = 0;
    if(threadIdx.x < 8)
      Q.counters[QC_HEADS_TRACE + threadIdx.x] = 0;
  }
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  const uint32_t chunk = blockIdx.x;                 // 256 consecutive path slots = 4 micro-tiles (batchSlots is a multiple of 256)
  const uint32_t slot  = chunk * QCHUNK + threadIdx.x;
  if(chunk * QCHUNK >= batchSlots)
    return;  // (uniform per workgroup)
  const CameraPath cp     = generateCameraPath(fc, P, ownedTiles, slot, sampleIndex);
  const bool       active = cp.valid;
  RaySetup r{};
  if(active)
    r = makeRaySetup(cp.origin, cp.direction);
  ClosestBest best{INFINITE_F, 0.0f, 0.0f, -1, 0xffffffffu, 0xffffffffu};
  uint32_t    seed0 = cp.seed;  // the alpha draws of this ray (P.misc is written below, after the walk)
  bool        seedLoaded = true;
  unsigned    nodes = 0, tris = 0;
  const unsigned long long activeMask = __ballot(active);
  if(sc.bvhRoot != BVH_EMPTY && activeMask != 0ull)
  {
    // wave-uniform walk state (kept uniform with readfirstlane so that it lives in SGPRs)
    const uint32_t firstLane = uint32_t(__ffsll((long long)activeMask) - 1);
    const uint32_t octinv    = __builtin_amdgcn_readfirstlane(uint32_t(__shfl(int(rayOctInv(r.idir)), int(firstLane))));
    uint32_t       gBase = 0, gBits = ((1u << octinv) << 8) | 1u;  // rootGroup
    int            sp = 0;
    bool           overflow = false;
    // one-octant packets (nearly all of them) test the nodes through the float planes in SGPRs (pt_bvh8.h)
    const uint32_t oct      = 7u ^ octinv;  // bit a set: direction component a is negative
    const bool     oneOctAny = sc.bvh8Planes != nullptr && __ballot(active && rayOctInv(r.idir) != octinv) == 0ull;
    const bool     oneOct    = !INTERVAL && oneOctAny;  // the per-ray plane path
    const uint32_t offNx = (oct & 1u) ? 32u : 0u, offFx = 32u - offNx, offNy = 64u + ((oct & 2u) ? 32u : 0u), offFy = 160u - offNy;
    const uint32_t offNz = 128u + ((oct & 4u) ? 32u : 0u), offFz = 288u - offNz;
    const float    sgnx = (oct & 1u) ? 1.0f : -1.0f, sgny = (oct & 2u) ? 1.0f : -1.0f, sgnz = (oct & 4u) ? 1.0f : -1.0f;
    // Interval node test (pt_packet.h) for the packets it is tight for: 64 samples of one pixel (pixel-major slots) leaving one point
    // (no lens) into one octant.  lane = child * 8 + plane tests one plane against the packet's interval ray; ~35 vector
    // instructions per node instead of 174.  A child entered without need costs a visit and changes nothing.
    bool       useInterval = false;
    PacketLane PL{};
    float      tmaxPacket = INFINITE_F;  // the largest distance any ray of the packet still accepts
    if(INTERVAL && oneOctAny)
    {
      const float ox = __shfl(r.org.x, int(firstLane)), oy = __shfl(r.org.y, int(firstLane)), oz = __shfl(r.org.z, int(firstLane));
      if(__ballot(active && (r.org.x != ox || r.org.y != oy || r.org.z != oz)) == 0ull)
      {
        const float  inf = __builtin_inff();
        PacketBounds B;
        B.omin[0] = B.omax[0] = ox; B.omin[1] = B.omax[1] = oy; B.omin[2] = B.omax[2] = oz;
        B.imin[0] = waveMin(active ? r.idir.x : inf); B.imax[0] = waveMax(active ? r.idir.x : -inf);
        B.imin[1] = waveMin(active ? r.idir.y : inf); B.imax[1] = waveMax(active ? r.idir.y : -inf);
        B.imin[2] = waveMin(active ? r.idir.z : inf); B.imax[2] = waveMax(active ? r.idir.z : -inf);
        PL          = makePacketLane(lane, oct, B);
        useInterval = true;
      }
    }
    for(;;)
    {
      if((gBits >> 8) == 0u)
      {
        if(sp == 0)
          break;
        --sp;
        gBase = __builtin_amdgcn_readfirstlane(s_wstack[wave][sp][0]);
        gBits = __builtin_amdgcn_readfirstlane(s_wstack[wave][sp][1]);
      }
      // nearest pending child of the group (priority space = slot ^ octinv)
      const uint32_t hitsP = gBits >> 8;
      const uint32_t pr    = 31u - uint32_t(__clz(int(hitsP)));
      const uint32_t cslot = pr ^ octinv;
      const uint32_t gim   = gBits & 0xffu;
      gBits &= ~(0x100u << pr);
      const uint32_t child = gBase + uint32_t(__popc(gim & ((1u << cslot) - 1u)));
      if(gBits >> 8)
      {
        if(sp < PACKET_STACK)
        {
          if(lane == 0)
          {
            s_wstack[wave][sp][0] = gBase;
            s_wstack[wave][sp][1] = gBits;
          }
          ++sp;
        }
        else
          overflow = true;
      }
      uint4    n0, n1;
      uint32_t hm = 0, hmU = 0;  // hm: children this lane's ray enters; hmU: children any ray of the packet enters
      if(INTERVAL && useInterval)
      {
        // this lane's plane of this lane's child: the node's 48 plane floats in one read, issued BEFORE the header's scalar loads
        // (whose helper waits for them) so that the two latencies of this serial walk overlap
        const float q  = PL.live ? sc.bvh8Planes[size_t(child) * 48u + PL.planeOffset] : 0.0f;
        scalarLoadNodeHeader(sc.bvh8Nodes, child, n0, n1);
        const float sx = __uint_as_float((n0.w & 0xffu) << 23), sy = __uint_as_float(((n0.w >> 8) & 0xffu) << 23), sz = __uint_as_float(((n0.w >> 16) & 0xffu) << 23);
        const float sA = PL.axis == 0u ? sx : (PL.axis == 1u ? sy : sz);
        const float pA = __uint_as_float(PL.axis == 0u ? n0.x : (PL.axis == 1u ? n0.y : n0.z));
        const float cand = packetPlaneTime(PL, q, pA, sA);
        const float inf  = __builtin_inff();
        const float tn   = group8Max((PL.live && PL.entry) ? cand : -inf);
        const float tf   = group8Min((PL.live && !PL.entry) ? cand : inf);
        hmU = everyEighthBit(__ballot(packetChildHit(tn, tf, tmaxPacket) && (lane & 7u) == 0u));
        hm  = active ? hmU : 0u;  // every ray of the packet tests the triangles of an entered leaf child
      }
      else if(!INTERVAL && oneOct)
      {
        f32x8s pnx, pny, pnz, pfx, pfy, pfz;
        scalarLoadNodePlanes(sc.bvh8Nodes, sc.bvh8Planes, child, offNx, offNy, offNz, offFx, offFy, offFz, n0, n1, pnx, pny, pnz, pfx, pfy, pfz);
        if(active)
          hm = bvh8TestChildrenPlanes(n0, pnx, pny, pnz, pfx, pfy, pfz, r, best.t, sgnx, sgny, sgnz);
      }
      else
      {
        uint4    n2, n3, n4;
        uint32_t tmaskLane = 0;
        scalarLoadNode(sc.bvh8Nodes, child, n0, n1, n2, n3, n4);
        if(active)
          bvh8TestChildren(n0, n1, n2, n3, n4, r, best.t, hm, tmaskLane);
      }
      if(COUNT && lane == firstLane) ++nodes;  // counters = records FETCHED: one per wave here, one per lane in the per-lane kernels
      // a child is entered / its triangles are tested when any lane hits its box
      if(!(INTERVAL && useInterval))
      {
#pragma unroll
        for(int i = 0; i < 8; ++i)
          hmU |= (__ballot((hm >> i) & 1u) != 0ull) ? (1u << i) : 0u;
      }
      const uint32_t imask = n0.w >> 24;
      uint32_t       hits  = hmU & imask;
      hits = (octinv & 1u) ? (((hits & 0x55u) << 1) | ((hits & 0xaau) >> 1)) : hits;
      hits = (octinv & 2u) ? (((hits & 0x33u) << 2) | ((hits & 0xccu) >> 2)) : hits;
      hits = (octinv & 4u) ? (((hits & 0x0fu) << 4) | ((hits & 0xf0u) >> 4)) : hits;
      gBase = n1.x;
      gBits = (hits << 8) | imask;
      uint32_t    leafU   = hmU & ~imask;
      const float tBefore = best.t;
      while(leafU)
      {
        const int i = __ffs(int(leafU)) - 1;
        leafU &= leafU - 1u;
        // (valid16 of the node, bvh8.hip: two bits per slot; the child's triangles follow those of the leaf children in lower slots)
        const uint32_t valid = n1.z & 0xffffu;
        const uint32_t cnt = uint32_t(__popc((valid >> (2 * i)) & 3u)), off = uint32_t(__popc(valid & ((1u << (2 * i)) - 1u)));
        for(uint32_t k = 0; k < cnt; ++k)
        {
          const uint32_t triIndex = n1.y + off + k;
          const DevTri   T        = scalarLoadTri(sc.tris, triIndex);
          if(COUNT && lane == firstLane) ++tris;
          if(active && ((hm >> i) & 1u))
          {
            closestTestLoaded<HAS_ALPHA>(sc, r, T, int(triIndex), best, seed0, seedLoaded, P.misc, slot);
          }
        }
      }
      if(INTERVAL && useInterval && __ballot(active && best.t < tBefore) != 0ull)
        tmaxPacket = waveMax(active ? best.t : -__builtin_inff());  // some ray found something nearer: the packet's bound may shrink
    }
    (void)overflow;  // PACKET_STACK covers any tree the builder emits for < 2^31 triangles at branching >= 2 per pending group
  }
  // ---- who goes on to the shade kernel: mesh hits, and rays that reach the infinite plane before anything else
  const bool meshHit = active && best.tri >= 0;
  const bool toShade = active && (meshHit || infinitePlaneT(fc, cp.origin, cp.direction, best.t) > 0.0f);
  // Survivors are packed to the front of the chunk's range of queue 0 (k_generate's placement of the chunk), the rest of the range is
  // marked dead: the bounce-0 shade launch then runs full waves (measured on the helmet workload: 6.8 ms against 7.5 ms with the
  // dead entries left where they fall; the workgroup barrier costs this kernel nothing measurable).
  const unsigned long long shadeMask = __ballot(toShade);
  if(lane == 0)
    s_cnt[wave] = uint32_t(__popcll(shadeMask));
  __syncthreads();
  uint32_t before = 0, total = 0;
#pragma unroll
  for(uint32_t w = 0; w < 4; ++w)
  {
    const uint32_t c = s_cnt[w];
    before += w < wave ? c : 0u;
    total += c;
  }
  const uint32_t pos0 = (chunk % NSUB) * Q.subCap + (chunk / NSUB) * QCHUNK;
  if(toShade)
  {
    const uint32_t pos   = pos0 + before + laneCountBelow(shadeMask);
    Q.active[0].slot[pos] = slot;
    Q.active[0].org[pos]  = make_float4(cp.origin.x, cp.origin.y, cp.origin.z, 0.0f);
    Q.active[0].dir[pos]  = make_float4(cp.direction.x, cp.direction.y, cp.direction.z, __uint_as_float(cp.seed));
    Q.active[0].aux[pos]  = make_float4(best.t, __int_as_float(best.tri), best.u, best.v);
    const float4 misc0    = make_float4(0.0f, __uint_as_float(PF_ALIVE), __uint_as_float(cp.seed), 0.0f);  // cone.width = 0
    if(fc.stateInQueue)
      Q.active[0].misc[pos] = misc0;  // the state travels with the ray (pt_scene.h: RayQueue)
    else
      P.misc[slot] = misc0;
  }
  if(threadIdx.x >= total)
    Q.active[0].slot[pos0 + threadIdx.x] = QUEUE_DEAD;
  // ---- the others end here.  What the shade kernel's miss branch would add for a first ray (gltf_pathtrace.slang:129-156: the
  // environment or the backplate at throughput 1, lastSamplePdf DIRAC) is a pure function of the direction, and evaluating it
  // here -- an atan2 / acos pair and four dependent gathers for the lanes of a wave that happen to miss, in a kernel whose waves
  // are serial scalar-load chains -- cost 1.08 ms of 5.16 per 32 helmet frames.  The path's record keeps the DIRECTION instead of a
  // radiance and k_finish_sample, which reads that record anyway and runs one dense thread per pixel, evaluates it.
  if(active && !toShade)
  {
    if(hasFlag(fc.pc.flags, MI_PT_FIRST_FRAME) && cp.frame == 0u)  // NDC depth input of a first frame (k_finish_sample)
      P.firstHit[pathSlotPixel(fc, slot)] = make_float4(cp.direction.x, cp.direction.y, cp.direction.z, 0.0f);
    P.radiance[slot] = make_float4(cp.direction.x, cp.direction.y, cp.direction.z, __uint_as_float(RADW_PRIMARY_MISS));
    if(P.guideAlbedo && fc.pc.numSamples == 1)  // (single-sample frames: nobody zeroed the guide records; this path has no first hit)
    {
      P.guideAlbedo[slot] = make_float4(0, 0, 0, 0);
      P.guideNormal[slot] = make_float4(0, 0, 0, 0);
    }
    if(P.misc)  // (multi-sample frames: the next sample's seed)
      P.misc[slot] = make_float4(0.0f, __uint_as_float(PF_NOT_SOLID | PF_PRIMARY_MISS), __uint_as_float(cp.seed), 0.0f);  // depth 0, not alive
  }
  if(COUNT)
  {
    if(lane == 0)
    {
      atomicAdd(&stats->cameraPaths, (unsigned long long)__popcll(activeMask));
      atomicAdd(&stats->segments, (unsigned long long)__popcll(activeMask));
    }
    atomicAdd(&stats->nodesPrimary, (unsigned long long)nodes);
    atomicAdd(&stats->trisPrimary, (unsigned long long)tris);
  }
}

//================================================================================================================================
// k_selection: traceSelectionRay / TraceLow (pathtrace_functions.h.slang:813-820, raytracer_interface.h.slang:124-137)
//================================================================================================================================
template <bool WIDE>
__global__ void __launch_bounds__(SEL_BLOCK) k_selection(DevScene sc, FrameConsts fc, const uint32_t* ownedTiles, uint32_t* selection)
{
  __shared__ int s_stack[BVH_STACK_LDS * SEL_BLOCK];
  uint32_t       slot = blockIdx.x * blockDim.x + threadIdx.x;
  int            px, py;
  if(slot >= uint32_t(fc.numSlots) || !slotToPixel(fc, ownedTiles, slot, px, py))
    return;
  f3 origin, direction;
  getRay(fc, mk2(float(px), float(py)), mk2(0.5f, 0.5f), origin, direction);
  const RaySetup r     = makeRaySetup(origin, direction);
  float          bestT = INFINITE_F;
  uint32_t       bestRnode = 0xffffffffu, bestPrim = 0xffffffffu;
  // RAY_FLAG_FORCE_OPAQUE, no culling: plain closest hit over every triangle
  auto testTri = [&](int triIndex) {
    const DevTri T = sc.tris[triIndex];
    TriHit       h;
    if(!intersectTri(xyz(T.a), xyz(T.b), xyz(T.c), r.org, r.dir, h) || !(h.t > 0.0f))
      return;
    const uint32_t rnode = __float_as_uint(T.a.w), prim = __float_as_uint(T.b.w);
    if(h.t < bestT || (h.t == bestT && (rnode < bestRnode || (rnode == bestRnode && prim < bestPrim))))
    {
      bestT = h.t; bestRnode = rnode; bestPrim = prim;
    }
  };
  if(sc.bvhRoot != BVH_EMPTY)
  {
    if(WIDE)
    {
      LaneStack2 st2;
      uint32_t   stackOverflow[2 * BVH8_STACK_PRIV];
      st2.lds = s_stack; st2.tid = int(threadIdx.x); st2.stride = SEL_BLOCK; st2.sp = 0;
      st2.privBase = stackOverflow; st2.privBits = stackOverflow + BVH8_STACK_PRIV;
      const uint32_t octinv = rayOctInv(r.idir);
      NodeGroup      G      = rootGroup(octinv);
      for(;;)
      {
        if((G.bits >> 8) == 0u)
        {
          if(st2.sp == 0)
            break;
          G = st2.pop();
        }
        const uint32_t child = groupPopChild(G, octinv);
        if(G.bits >> 8)
          st2.push(G);
        uint32_t tBase, tMask;
        bvh8Visit(sc, r, bestT, octinv, child, G, tBase, tMask, nullptr, 0u);
        while(leafPending(tMask))
        {
          const int k = int(leafPop(tMask, 0u));  // (offset from the node's first triangle)
          testTri(int(tBase) + k);
        }
      }
    }
    else
    {
      LaneStack st;
      int       stackOverflow[BVH_STACK_PRIV];
      st.lds = s_stack; st.tid = int(threadIdx.x); st.stride = SEL_BLOCK; st.sp = 0; st.priv = stackOverflow;
      bvhWalk(sc, r, bestT, st, [&](int triIndex, float) -> float {
        testTri(triIndex);
        return bestT;
      });
    }
  }
  selection[size_t(py) * size_t(fc.width) + size_t(px)] = (bestRnode != 0xffffffffu) ? bestRnode + 1u : 0u;
}

//================================================================================================================================
// k_shade: everything of pathTraceOneBounce / pathTrace between the two Trace calls (gltf_pathtrace.slang:104-430, 441-494)
//================================================================================================================================
// FIRST: the launch shades bounce 0 (every queued path still has its initial state, see k_generate).
// (Round 6 measured the reorder family on this kernel -- a dense pass of its own for the paths that leave the scene, continuation entries partitioned by
//  next-event technique x lobe draw at append -- profiles/r06_ser_ab.txt: lanes per instruction 27.7 -> 30.9, 8 % fewer instructions, no time saved:
//  the later-bounce launch waits on its gather chain at 3 waves per SIMD.  Removed again.)
template <bool COUNT, bool SIMPLE, bool FIRST>
__global__ void __launch_bounds__(SHADE_BLOCK, SIMPLE ? SHADE_SIMPLE_WAVES : 1) k_shade(const DevScene* __restrict__ scp, const FrameConsts* __restrict__ fcp, PathSoA P, Queues Q, int cur, int sortMode, StatCounters* stats)
{
  // The scene / frame descriptors reach the non-inlined helpers (getTexture, sampleLights, the sky) by reference.  As
  // by-value kernel arguments they would be copied to scratch (their address escapes) and every field read would become a
  // memory round trip; as device-resident structs they are read through one uniform pointer.
  // ... and as CONSTANT memory (uniformConst): a read of `sc.` / `fc.` that follows a store or a call is otherwise a vector load of a
  // uniform address -- the compiler must assume the store or the callee wrote there -- and the loop is full of both.
  const DevScene&    sc = uniformConst(*scp);
  const FrameConsts& fc = uniformConst(*fcp);
  const bool         stateInQueue = fc.stateInQueue != 0;  // misc / throughput / radiance of a living path ride in its queue entry (pt_scene.h: RayQueue)
  __shared__ uint32_t s_prefix[NSUB + 1];
  __shared__ uint32_t s_push[4];
  __shared__ float    s_srgb[256];  // sRGB decode table next to the ALU: 3 lookups per texel, up to 8 texels per tap
#ifndef SHADE_NO_DEFERRED_MISS
  // Later bounces: the paths that left the scene are not finished where they are found -- by then 5 % (atrium) to 18 % (street) of a queue's rays miss,
  // spread evenly, and nearly every wave ran the environment evaluation (physical sky + sun disc or the HDR lookup, and the MIS weight: ~900 vector
  // instructions) for two or three of its lanes -- but listed in LDS and finished by the whole block, 256 at a time with every lane busy, once that many
  // have gathered (and at the block's end).  A first ray's miss (backplate) and any miss of a frame with the infinite plane stay inline.
  // (As a pass of its own -- a scan of the whole queue for the misses -- the gain was eaten by the scan: profiles/r06_ser_ab.txt.)
  constexpr bool DEFER_MISS = !FIRST;
#else
  constexpr bool DEFER_MISS = false;
#endif
  __shared__ uint32_t s_missPos[DEFER_MISS ? 2 * SHADE_BLOCK : 1];
  __shared__ uint32_t s_missCount;
  // ... and only where the environment is the physical sky: an HDR lookup is too cheap to be worth the list (helmet 5310 -> 5285, glass 551 -> 549 Msamples/s with it,
  // atrium 715 -> 727, street 764 -> 771: profiles/r06_deferred_miss_ab.txt)
  const bool          deferMiss = DEFER_MISS && !hasFlag(fc.frameInfo.flags, MI_SCENE_USE_INFINITE_PLANE) && !hasFlag(fc.frameInfo.flags, MI_SCENE_USE_HDR_ENVIRONMENT);
  if(threadIdx.x == 0)
    s_missCount = 0;  // (the barrier of queuePrefix below orders it)
  // finishes up to 256 listed misses: exactly what the inline branch does for a ray that is not a first ray (gltf_pathtrace.slang:139-156), through the same
  // non-inlined missEnvironmentCall.  Whole block; contains barriers.
  auto finishMisses = [&](bool all) {
    __syncthreads();
    const uint32_t n = s_missCount;
    if(n == 0u || (!all && n < uint32_t(SHADE_BLOCK)))
      return;
    const uint32_t take = min(n, uint32_t(SHADE_BLOCK)), base = n - take;
    uint32_t       pos = 0;
    if(threadIdx.x < take)
      pos = s_missPos[base + threadIdx.x];
    __syncthreads();
    if(threadIdx.x == 0)
      s_missCount = base;
    if(threadIdx.x < take)
    {
      const uint32_t slot  = Q.active[cur].slot[pos];
      const float4   d4    = Q.active[cur].dir[pos];
      const float4   misc4 = stateInQueue ? Q.active[cur].misc[pos] : P.misc[slot];
      const float4   tp4   = stateInQueue ? Q.active[cur].aux2[pos] : P.throughput[slot];
      const float4   rad4  = stateInQueue ? Q.active[cur].rad[pos] : P.radiance[slot];
      f3             radiance = xyz(rad4);
      f3             envColor;
      float          mis;
      missEnvironment(sc, fc, xyz(d4), tp4.w, envColor, mis);
      radiance += xyz(tp4) * mis * envColor;
      // the record an ended path leaves behind (see the end of the round below): flags without ALIVE -- and, from the SIMPLE kernel, without INSIDE
      uint32_t flags = __float_as_uint(misc4.y) & (PF_INSIDE | PF_NOT_SOLID | (0xffu << PF_DEPTH_SHIFT) | (0xffu << PF_SCATTER_SHIFT));
      if(SIMPLE)
        flags &= ~uint32_t(PF_INSIDE);
      const bool solid = !(flags & PF_NOT_SOLID);
      P.radiance[slot] = make_float4(radiance.x, radiance.y, radiance.z, __uint_as_float(__float_as_uint(fmaxf(fabsf(rad4.w), 0.0f)) | (solid ? 0u : RADW_NOT_SOLID)));
      if(P.misc)
        P.misc[slot] = make_float4(misc4.x, __uint_as_float(flags), misc4.z, misc4.w);
    }
    __syncthreads();
  };
  // The window sort exists in the generic kernel only (key: material).  Where every material runs the same code (SIMPLE) grouping by material buys
  // nothing, and a window keyed by next-event technique (rounds 3-4: fewer instructions, fuller waves, 6-15 % SLOWER -- the key costs a dependent gather
  // and the window two barriers, and this kernel waits on gather depth, not on issue) was removed in round 5 together with its registers: the
  // later-bounce SIMPLE kernel spilled 16 VGPRs for a feature that was off (LABNOTES.md).
  constexpr bool CAN_SORT = !SIMPLE;
  __shared__ uint32_t s_order[CAN_SORT ? SORT_WINDOW : 1];                   // queue positions of the window's live entries, sorted by bin
  __shared__ uint16_t s_segCount[CAN_SORT ? SORT_SEGMENTS : 1][SORT_BINS];   // entries of a bin in one (round, wave) segment -> exclusive prefix inside the bin
  __shared__ uint32_t s_binBase[SORT_BINS + 1];                 // first sorted index of each bin; [SORT_BINS] = live entries of the window
  static_assert(SHADE_BLOCK == 256, "one table entry per thread");
  if(blockIdx.x == 0 && threadIdx.x < 8)
  {
    Q.counters[QC_HEADS_TRACE + threadIdx.x] = 0;  // for the next iteration's k_trace_closest
    // ... and for this iteration's shadow stage (k_trace_shadow / k_shadow_resolve run after this launch and, the previous iteration's,
    // before it -- also when that stage runs on its own stream next to the following k_trace_closest)
    Q.counters[QC_HEADS_SHADOW + threadIdx.x]   = 0;
    Q.counters[QC_HEADS_OVERFLOW + threadIdx.x] = 0;
    if(threadIdx.x == 0)
      Q.counters[QC_CAND_POOL] = Q.counters[QC_RESOLVE] = Q.counters[QC_OVERFLOW] = 0;
  }
  queuePrefix(&Q.counters[cur ? QC_PAIR1 : QC_PAIR0], s_prefix);
  const uint32_t count      = s_prefix[NSUB];
  const int      nxt        = cur ^ 1;
  constexpr uint32_t ROUNDS = CAN_SORT ? SORT_ROUNDS : 1u;
  constexpr uint32_t WINDOW = ROUNDS * SHADE_BLOCK;
  const uint32_t     numWindows = (count + WINDOW - 1) / WINDOW;
  if(blockIdx.x >= numWindows)
    return;  // nothing for this block (late bounces launch the full grid on short or empty queues)
  s_srgb[threadIdx.x] = sc.srgbLut[threadIdx.x];
  __syncthreads();
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  for(uint32_t win = blockIdx.x; win < numWindows; win += gridDim.x)
  {
    // ---- Per-bounce sort of the queue, one SORT_WINDOW-entry window at a time, in LDS (no extra pass over HBM, no global atomics):
    // key = dead entries last, paths that left the scene (or reach the infinite plane) before them, surface hits first and grouped
    // by material -- so that a wave shades one kind of thing, the texture / material records it gathers are shared by its lanes,
    // and the dead entries the bounce-0 kernel leaves behind cost nothing.  Stable counting sort: a lane's rank inside its
    // (round, wave) segment comes from ballots over the distinct bins of the wave (usually one to three), the segments of a bin
    // are laid out in queue order.  Paths are independent, so the processing order cannot change any result.
    uint32_t live = min(WINDOW, count - win * WINDOW);  // no sort: the window as it is, dead entries and all
    if(CAN_SORT && sortMode != 0)
    {
    uint32_t myPos[SORT_ROUNDS], myBin[SORT_ROUNDS], myRank[SORT_ROUNDS];
    for(uint32_t t = threadIdx.x; t < SORT_SEGMENTS * SORT_BINS; t += SHADE_BLOCK)
      (&s_segCount[0][0])[t] = 0;
    __syncthreads();
#pragma unroll
    for(uint32_t k = 0; k < SORT_ROUNDS; ++k)
    {
      const uint32_t i = win * WINDOW + k * SHADE_BLOCK + threadIdx.x;
      uint32_t       bin = SORT_BIN_DEAD;
      myPos[k]           = 0u;
      if(i < count)
      {
        myPos[k]            = queuePos(Q.subCap, s_prefix, i);
        const uint32_t slot = Q.active[cur].slot[myPos[k]];
        if(slot != QUEUE_DEAD)
        {
          const int tri = __float_as_int(Q.active[cur].aux[myPos[k]].y);
          // sortMode 1: surface hits / the rest / dead; 2: surface hits grouped by material as well
          if(tri < 0)
            bin = SORT_BIN_MISS;
          else
            bin = sortMode >= 2 ? uint32_t(sc.shadeTris[tri].materialID) % SORT_BIN_MISS : 0u;
        }
      }
      myBin[k] = bin;
      // rank among the lanes of this wave with the same bin, and the segment's count of that bin
      uint32_t           rank = 0;
      unsigned long long todo = __ballot(bin != SORT_BIN_DEAD);
      while(todo != 0ull)
      {
        const uint32_t           b = uint32_t(__builtin_amdgcn_readlane(int(bin), __ffsll((long long)todo) - 1));
        const unsigned long long m = __ballot(bin == b);
        if(bin == b)
          rank = laneCountBelow(m);
        if(lane == 0)
          s_segCount[k * 4 + wave][b] = uint16_t(__popcll(m));
        todo &= ~m;
      }
      myRank[k] = rank;
    }
    __syncthreads();
    if(threadIdx.x < SORT_BINS)  // exclusive prefix of the segments inside each bin (queue order), and the bin totals
    {
      uint32_t acc = 0;
      for(uint32_t sgm = 0; sgm < SORT_SEGMENTS; ++sgm)
      {
        const uint32_t c = s_segCount[sgm][threadIdx.x];
        s_segCount[sgm][threadIdx.x] = uint16_t(acc);
        acc += c;
      }
      // bins are laid out in index order: surface hits by material, then misses; dead entries are not laid out at all
      uint32_t incl = acc;
#pragma unroll
      for(int d = 1; d < SORT_BINS; d <<= 1)
      {
        const uint32_t t = uint32_t(__shfl_up(int(incl), d));
        if(threadIdx.x >= uint32_t(d))
          incl += t;
      }
      s_binBase[threadIdx.x] = incl - acc;
      if(threadIdx.x == SORT_BINS - 1)
        s_binBase[SORT_BINS] = incl;
    }
    __syncthreads();
#pragma unroll
    for(uint32_t k = 0; k < SORT_ROUNDS; ++k)
      if(myBin[k] != SORT_BIN_DEAD)
        s_order[s_binBase[myBin[k]] + s_segCount[k * 4 + wave][myBin[k]] + myRank[k]] = myPos[k];
    __syncthreads();
    live = s_binBase[SORT_BINS];
    }
   for(uint32_t round = 0; round * SHADE_BLOCK < live; ++round)
   {
    const uint32_t chunk   = win * ROUNDS + round;  // 256 processed entries append to sub-queue chunk % NSUB, like a chunk of the queue
    const uint32_t e       = round * SHADE_BLOCK + threadIdx.x;
    const bool     inRange = e < live;
    const uint32_t inPos   = !inRange ? 0u : ((CAN_SORT && sortMode != 0) ? s_order[e] : queuePos(Q.subCap, s_prefix, win * WINDOW + e));
    uint32_t       slot    = inRange ? Q.active[cur].slot[inPos] : QUEUE_DEAD;
    bool           alive = false, pushShadow = false;
    unsigned       taps = 0;
    float4         nextOrg = make_float4(0, 0, 0, 0), nextDir = make_float4(0, 0, 0, 0);
    float4         nextRad = make_float4(0, 0, 0, 0), nextMisc = make_float4(0, 0, 0, 0), nextThr = make_float4(0, 0, 0, 0);  // state of a path that goes on
    float4         shOrg = make_float4(0, 0, 0, 0), shDir = make_float4(0, 0, 0, 0), shCon = make_float4(0, 0, 0, 0), shCon2 = make_float4(0, 0, 0, 0);
    bool           catcher = false;
#ifdef SHADE_PROFILE
    const unsigned long long sprofRound0 = __builtin_amdgcn_s_memtime();
#endif
    if(inRange && slot != QUEUE_DEAD)
    {
      SPROF_BEGIN();
      const float4 hit4 = Q.active[cur].aux[inPos], o4 = Q.active[cur].org[inPos], d4 = Q.active[cur].dir[inPos];
      // the path's state: records of its queue entry (unit stride, like the ray), or -- catcher frames / MI_PT_STATE_BY_SLOT -- gathered by slot
      // (round 6: the NEXT round's entry prefetched into registers across the append -- 28 VGPRs spilled at the 168-register budget: atrium 690.7 -> 686.7,
      //  helmet 5255 -> 5068 Msamples/s; and staged through LDS by global_load_lds together with its triangle's shade record (no register held: 37 KB of LDS
      //  per block, one more barrier, every later wait of the round a full drain): atrium 722.9 -> 717.3, helmet 5216 -> 5099; and only the first TWO levels of the next
      //  window's chain held -- position, hit triangle, 32-byte shade record, ten registers: later-bounce shade 0.661 -> 0.710 ms, atrium 730.9 -> 717.2 -- profiles/r06_shade_walk_ab.txt)
      const float4 misc4 = stateInQueue ? Q.active[cur].misc[inPos] : P.misc[slot];
      const float4 tp4   = FIRST ? make_float4(1.0f, 1.0f, 1.0f, DIRAC) : (stateInQueue ? Q.active[cur].aux2[inPos] : P.throughput[slot]);
      const float4 rad4  = FIRST ? make_float4(0.0f, 0.0f, 0.0f, 0.0f) : (stateInQueue ? Q.active[cur].rad[inPos] : P.radiance[slot]);
      f3       rayOrigin = xyz(o4), rayDir = xyz(d4);
      float    coneWidth = misc4.w;
      f3       throughput = xyz(tp4), radiance = xyz(rad4);
      float    lastSamplePdf = tp4.w;
      f2       maxRoughness  = mk2(fabsf(rad4.w), misc4.x);  // the sign of radiance.w is the path's !solid (PathSoA)
      uint32_t flags = __float_as_uint(misc4.y), seed = __float_as_uint(misc4.z);
      int      surfaceDepth   = int((flags >> PF_DEPTH_SHIFT) & 0xffu);
      int      scatterBounces = int((flags >> PF_SCATTER_SHIFT) & 0xffu);
      bool     isInside = !SIMPLE && (flags & PF_INSIDE) != 0u, solid = !(flags & PF_NOT_SOLID);
      const bool firstRay = (surfaceDepth == 0);
      bool       guideWritten = false;
      const int  maxDepth = fc.pc.maxDepth;
      // the first-hit position only feeds the NDC depth of a first frame (k_finish_sample), i.e. frame 0 of the batch
      const bool needFirstHit = hasFlag(fc.pc.flags, MI_PT_FIRST_FRAME) && pathSlotFrame(fc, slot) == 0u;

      float hitT   = hit4.x;
      int   triIdx = __float_as_int(hit4.y);
      bool  done   = false;  // eBreak
      bool  deferred = false;  // a miss handed to finishMisses
      bool  earlyContinue = false;

      HitState hit;
      uint4    core0 = make_uint4(0u, 0u, 0u, 0u);
      int      rnodeID = -1, primitiveID = -1, materialID = 0;
      (void)primitiveID;
      const bool meshHit = triIdx >= 0;
      if(meshHit)
      {
        const DevShadeTri S = gat(sc.shadeTris, triIdx);
        rnodeID             = int(S.rnode);
        primitiveID         = int(S.prim);
        materialID          = S.materialID;
        if(!FIRST)  // (in flight next to the vertices: evaluateMaterial<SIMPLE, CORE> plans the base-colour fetch from it)
          core0 = gat(sc.coreTex, 5u * uint32_t(materialID));
        const MiGltfRenderNode& rn = gat(sc.nodes, rnodeID);
        // record -> vertices directly; the primitive's stream table only for the attributes that are not interleaved (uv1, colours)
        // (round 6: ONE 192-byte record per triangle -- its three vertices copied, the shade record's fields, the base-colour slot record -- so that vertices, material index
        //  and texel addresses are one round trip behind the queue entry instead of two: GPU suite green, later-bounce shade 0.659 -> 0.664 ms (atrium), 0.653 -> 0.676 (sliver
        //  atrium), helmet / street unchanged -- de-indexed vertices lose the lines neighbouring hits share.  Removed: profiles/r06_shade_walk_ab.txt)
        DevPrim rp{};
        u3      ti{0u, 0u, 0u};
        if(S.attrs & (SHADE_HAS_UV1 | SHADE_HAS_COLORS))
        {
          rp = gat(sc.prims, S.renderPrimID);
          ti = getTriangleIndices(rp, int(S.prim));
        }
        hit = getHitState(&gat(sc.geomPool, S.v0), &gat(sc.geomPool, S.v1), &gat(sc.geomPool, S.v2), S.attrs, &rp, ti,
                          mk3(1.0f - hit4.z - hit4.w, hit4.z, hit4.w), rn.worldToObject, rn.objectToWorld, rayDir);
      }
      else
        hitT = INFINITE_F;
      SPROF_END(0);

      // checkInfinitePlaneIntersection, pathtrace_functions.h.slang:556-585
      bool hitInfinitePlane = false;
      {
        const float t = infinitePlaneT(fc, rayOrigin, rayDir, hitT);
        if(t > 0.0f)
        {
          hitT             = t;
          hit.pos          = rayOrigin + rayDir * hitT;
          hit.shadowPos    = hit.pos;
          hit.nrm          = mk3(0, 1, 0);
          hit.geonrm       = mk3(0, 1, 0);
          hit.tangent      = mk3(1, 0, 0);
          hit.bitangent    = mk3(0, 0, 1);
          hitInfinitePlane = true;
        }
      }

      if(deferMiss && hitT == INFINITE_F && !firstRay)  // finished by the whole block later (finishMisses): nothing else of this entry is touched
      {
        SPROF_BEGIN();
        s_missPos[atomicAdd(&s_missCount, 1u)] = inPos;
        deferred = true;
        done     = true;
        SPROF_END(6);
      }
      else if(hitT == INFINITE_F)  // gltf_pathtrace.slang:129-156
      {
        SPROF_BEGIN();
        bool backplate = false;
        if(firstRay)  // tryPrimaryMissBackplate, pathtrace_functions.h.slang:944-971
        {
          solid             = false;
          if(needFirstHit)
            P.firstHit[pathSlotPixel(fc, slot)] = make_float4(rayDir.x, rayDir.y, rayDir.z, 0.0f);
          backplate = primaryMissBackplate(sc, fc, rayDir, radiance);
        }
        if(!backplate)
        {
          f3    envColor;
          float mis;
          missEnvironment(sc, fc, rayDir, lastSamplePdf, envColor, mis);
          radiance += throughput * mis * envColor;
        }
        done = true;
        SPROF_END(6);
      }

      if(!done)
      {
        PbrMaterial pbrMat;
        // rayConeWorldFootprint, pathtrace_functions.h.slang:174-178
        float worldFoot = (coneWidth + fc.pc.pixelAngle * hitT) / fmaxf(fabsf(dot(hit.geonrm, -rayDir)), 1e-3f);
        bool  unlit     = false;
        bool  catcherPlane = false;  // shadow-catcher hit: the rest of the bounce is skipped (eBreak / eEarlyContinue)
        if(hitInfinitePlane)
        {
          pbrMat           = defaultPbrMaterial();
          pbrMat.baseColor = mk3(fc.frameInfo.infinitePlaneBaseColor);
          pbrMat.metallic  = fc.frameInfo.infinitePlaneMetallic;
          float r          = fc.frameInfo.infinitePlaneRoughness;
          pbrMat.roughness = mk2(r * r, r * r);
          pbrMat.N = hit.nrm; pbrMat.Ng = hit.nrm; pbrMat.Nc = hit.nrm;
          pbrMat.T = hit.tangent; pbrMat.B = hit.bitangent;
          // handleShadowCatcher, pathtrace_functions.h.slang:499-554 (called from gltf_pathtrace.slang:169-187).  The reference
          // needs the shadow factor inside the bounce; here the bounce is finished speculatively (the alpha draws of
          // TraceShadow do not advance the seed, §6, so the continuation sample is the same) and k_trace_shadow applies the
          // radiance terms and drops the continuation again when the point turns out to be unshadowed.
          if(hasFlag(fc.frameInfo.flags, MI_SCENE_INFINITE_PLANE_SHADOW_CATCHER))
          {
            catcherPlane = true;
            coneWidth    = worldFoot;
            if(FIRST && needFirstHit)  // the reference leaves SampleResult::hitPosition at its 1e34 default on this path
              P.firstHit[pathSlotPixel(fc, slot)] = make_float4(1e34f, 1e34f, 1e34f, 0.0f);
            DirectLight dl;
            sampleLights(sc, fc, hit.pos, seed, dl);
            const bool traceIt = dot(dl.direction, hit.nrm) > 0.0f && dl.pdf != 0.0f;
            f3         envColor;
            float      envPdf;
            sampleEnvironment(sc, fc, rayDir, envColor, envPdf);
            const float mis         = computeEnvHitMisWeight(sc, fc, lastSamplePdf, envPdf);
            const f3    unshadowed  = throughput * mis * envColor;
            if(!traceIt)
            {
              radiance += unshadowed;
              done = true;
            }
            else
            {
              shOrg = make_float4(hit.pos.x, hit.pos.y, hit.pos.z, INFINITE_F);
              shDir = make_float4(dl.direction.x, dl.direction.y, dl.direction.z, __uint_as_float(2u));
              shCon = make_float4(envColor.x, envColor.y, envColor.z, __uint_as_float(seed));
              shCon2     = make_float4(unshadowed.x, unshadowed.y, unshadowed.z, 0.0f);
              pushShadow = true;
              catcher    = true;
              float      r1 = rnd(seed), r2 = rnd(seed), r3 = rnd(seed);
              BsdfSample sd = bsdfSampleSimple(-rayDir, mk3(r1, r2, r3), pbrMat);
              if(sd.event_type == BSDF_EVENT_ABSORB)
                done = true;
              else
              {
                f3 offsetDir = dot(sd.k2, hit.geonrm) > 0.0f ? hit.geonrm : -hit.geonrm;
                rayOrigin    = safeOffsetRay(hit.pos, offsetDir);
                rayDir       = normalize(sd.k2);  // pathTrace loop head, gltf_pathtrace.slang:447
                throughput *= sd.bsdf_over_pdf;
                lastSamplePdf = sd.pdf;
              }
            }
          }
        }
        else
        {
          const MiGltfShadeMaterial& mat = gat(sc.materials, materialID);
          MeshState                  mesh;
          mesh.N = hit.nrm; mesh.T = hit.tangent; mesh.B = hit.bitangent; mesh.Ng = hit.geonrm;
          mesh.tc0 = hit.uv0; mesh.tc1 = hit.uv1;
          mesh.isInside           = isInside;
          mesh.texGrad            = worldFoot * hit.texelDensity * fc.pc.texGradScale;
          mesh.baseColorVertexMul = hit.color;
          mesh.tex                = TexCtx{sc.texRefs, sc.texels, s_srgb, sc.texQuads};
          mesh.core0              = core0;
          SPROF_BEGIN();
          pbrMat                  = evaluateMaterial<SIMPLE, !FIRST>(sc, mat, mesh, taps);  // (!FIRST: the base colour through its core record, pt_shading.h)
          unlit                   = mat.unlit > 0;
          SPROF_END(1);
        }
        if(!catcherPlane)
        {
          if(firstRay)  // gltf_pathtrace.slang:228-264
          {
            if(needFirstHit)
              P.firstHit[pathSlotPixel(fc, slot)] = make_float4(hit.pos.x, hit.pos.y, hit.pos.z, 0.0f);
            if(P.guideAlbedo)
            {
              float4 ga = make_float4(0, 0, 0, 0), gn = ga;
              if(fc.pc.numSamples > 1)  // (the sum over the frame's samples, zeroed by sample 0: generateCameraPath)
              {
                ga = P.guideAlbedo[slot];
                gn = P.guideNormal[slot];
              }
              guideWritten        = true;
              P.guideAlbedo[slot] = make_float4(ga.x + pbrMat.baseColor.x, ga.y + pbrMat.baseColor.y, ga.z + pbrMat.baseColor.z, ga.w + 1.0f);
              P.guideNormal[slot] = make_float4(gn.x + pbrMat.N.x, gn.y + pbrMat.N.y, gn.z + pbrMat.N.z, 0.0f);
            }
          }
          maxRoughness     = mk2(fmaxf(pbrMat.roughness.x, maxRoughness.x), fmaxf(pbrMat.roughness.y, maxRoughness.y));  // :267-268
          pbrMat.roughness = maxRoughness;
          radiance += pbrMat.emissive * throughput;  // :293
          if(unlit)                                  // :298-304
          {
            radiance += pbrMat.baseColor;
            done = true;
          }

          // processVolumeSegment, pathtrace_functions.h.slang:904-939
          bool volumeContinue = false;
          if(!SIMPLE && !done && isInside)
          {
            f3    ext, scat;
            float aniso;
            unpackMedium(P.medium[slot], ext, scat, aniso);
            if(maxComp(ext) > 0.0f || maxComp(scat) > 0.0f)
            {
              // handleVolumeScatter, :605-645
              bool  scattered  = false;
              float maxScatter = maxComp(scat);
              f3    wiBefore = rayDir, originBefore = rayOrigin;
              if(maxScatter > VOLUME_MIN_SCATTER)
              {
                float maxExt      = maxComp(ext);
                float scatterDist = -logf(fmaxf(rnd(seed), VOLUME_RAND_FLOOR)) / maxExt;
                if(scatterDist < hitT)
                {
                  throughput *= mk3(1.0f) - (ext - scat) / maxExt;
                  rayOrigin     = rayOrigin + rayDir * scatterDist;
                  float r1 = rnd(seed), r2 = rnd(seed);
                  rayDir        = sampleHenyeyGreenstein(mk2(r1, r2), aniso, wiBefore);
                  lastSamplePdf = henyeyGreensteinPdf(dot(wiBefore, rayDir), aniso);
                  scattered     = true;
                }
                else
                  throughput *= exp3((mk3(maxExt) - ext) * hitT);
              }
              else
                throughput *= exp3(ext * (-hitT));
              if(scattered)
              {
                scatterBounces = min(scatterBounces + 1, 255);
                coneWidth += fc.pc.pixelAngle * length(rayOrigin - originBefore);
                // volumeScatterNEE, :651-672 (the shadow ray is deferred to k_trace_shadow; initialInside = true)
                DirectLight dl;
                sampleLights(sc, fc, rayOrigin, seed, dl);
                if(dl.pdf > 0.0f)
                {
                  float phasePdf = henyeyGreensteinPdf(dot(wiBefore, dl.direction), aniso);
                  float mis      = dl.pdf / (dl.pdf + phasePdf);
                  f3    contrib  = throughput * dl.radianceOverPdf * mis * phasePdf;
                  shOrg = make_float4(rayOrigin.x, rayOrigin.y, rayOrigin.z, dl.distance);
                  shDir = make_float4(dl.direction.x, dl.direction.y, dl.direction.z, __uint_as_float(1u));
                  shCon = make_float4(contrib.x, contrib.y, contrib.z, __uint_as_float(seed));
                  pushShadow            = true;
                }
                if(scatterBounces >= VOLUME_FREE_BUDGET)
                {
                  float rrPcont = fminf(maxComp(throughput) + RR_PCONT_FLOOR, RR_PCONT_CAP);
                  if(rnd(seed) >= rrPcont)
                    done = true;
                  else
                    throughput /= rrPcont;
                }
                volumeContinue = !done;
                rayDir         = normalize(rayDir);
              }
            }
          }

          if(!done && !volumeContinue)
          {
            coneWidth = worldFoot;  // :313
            DirectLight dl;
#ifdef MI_PT_DIAG_NO_NEE  // cost-attribution build (tools/attribution.sh): wrong image, no next-event estimation
            dl = DirectLight{};
#else
            {
              SPROF_BEGIN();
              sampleLights(sc, fc, hit.pos, seed, dl);  // :319-320
              SPROF_END(2);
            }
#endif
            bool nextEventValid = (dot(dl.direction, hit.nrm) > 0.0f || pbrMat.diffuseTransmissionFactor > 0.0f) && dl.pdf != 0.0f;
            f3   contribution   = mk3(0.0f);
            if(nextEventValid)  // :330-351
            {
              SPROF_BEGIN();
              float    r1 = rnd(seed), r2 = rnd(seed), r3 = rnd(seed);
              BsdfEval ev = bsdfEvaluate(-rayDir, dl.direction, mk3(r1, r2, r3), pbrMat);
              if(ev.pdf > 0.0f)
              {
                float mis    = (dl.pdf == DIRAC) ? 1.0f : dl.pdf / (dl.pdf + ev.pdf);
                contribution = throughput * dl.radianceOverPdf * mis * ev.bsdf;
              }
              SPROF_END(3);
            }
            {  // :357-416
              SPROF_BEGIN();
              float      r1 = rnd(seed), r2 = rnd(seed), r3 = rnd(seed);
#ifdef MI_PT_DIAG_NO_SAMPLE  // cost-attribution build: mirror direction at half weight instead of the BSDF sample
              BsdfSample sd{};
              sd.k2 = rayDir - hit.nrm * (2.0f * dot(rayDir, hit.nrm)); sd.bsdf_over_pdf = mk3(0.5f * r1 + 0.25f); sd.pdf = 1.0f + r2 + r3;
              sd.event_type = BSDF_EVENT_GLOSSY_REFLECTION;
#else
              BsdfSample sd = bsdfSample(-rayDir, mk3(r1, r2, r3), pbrMat);
#endif
              throughput *= sd.bsdf_over_pdf;
              rayDir        = sd.k2;
              lastSamplePdf = sd.pdf;
              if(sd.event_type != BSDF_EVENT_ABSORB)
              {
                f3 offsetDir = dot(rayDir, hit.geonrm) > 0.0f ? hit.geonrm : -hit.geonrm;
                rayOrigin    = safeOffsetRay(hit.pos, offsetDir);
                if(!SIMPLE && (sd.event_type & BSDF_EVENT_TRANSMISSION))
                {
                  isInside = !isInside;
                  if(isInside)  // makeVolumeMedium, pathtrace_functions.h.slang:125-132
                    P.medium[slot] = packMedium(volumeExtinctionCoefficient(pbrMat), pbrMat.scatterCoefficient, pbrMat.scatterAnisotropy);
                }
              }
              else
                surfaceDepth = maxDepth;
              SPROF_END(4);
            }
            if(nextEventValid)  // :421-426 + the TraceShadow of pathTrace :462-471, deferred to k_trace_shadow
            {
              bool forward = dot(dl.direction, hit.nrm) > 0.0f;
              f3   sOrg    = safeOffsetRay(forward ? hit.shadowPos : hit.pos, forward ? hit.geonrm : -hit.geonrm);
              shOrg = make_float4(sOrg.x, sOrg.y, sOrg.z, dl.distance);
              shDir = make_float4(dl.direction.x, dl.direction.y, dl.direction.z, __uint_as_float(0u));
              shCon = make_float4(contribution.x, contribution.y, contribution.z, __uint_as_float(seed));
              pushShadow            = true;
            }
            // Russian roulette, :476-482
            if(surfaceDepth >= RR_MIN_DEPTH)
            {
              float rrPcont = fminf(maxComp(throughput) + 0.001f, 0.95f);
              if(rnd(seed) >= rrPcont)
                done = true;
              else
                throughput /= rrPcont;
            }
            if(!done)
            {
              surfaceDepth++;
              rayDir = normalize(rayDir);
            }
          }
        }
        (void)earlyContinue;
      }

      if(FIRST && P.guideAlbedo && !guideWritten && fc.pc.numSamples == 1)  // a path without a first surface hit (miss, catcher plane): empty guides
      {
        P.guideAlbedo[slot] = make_float4(0, 0, 0, 0);
        P.guideNormal[slot] = make_float4(0, 0, 0, 0);
      }
      alive = !done && surfaceDepth < maxDepth;
      flags = (isInside ? PF_INSIDE : 0u) | (solid ? 0u : PF_NOT_SOLID) | (alive ? PF_ALIVE : 0u) | (uint32_t(min(surfaceDepth, 255)) << PF_DEPTH_SHIFT)
              | (uint32_t(scatterBounces) << PF_SCATTER_SHIFT);
      // (fmaxf: a NaN or negative maxRoughness.x from degenerate material input must not collide with RADW_PRIMARY_MISS or the flag bit)
      nextRad  = make_float4(radiance.x, radiance.y, radiance.z, __uint_as_float(__float_as_uint(fmaxf(maxRoughness.x, 0.0f)) | (solid ? 0u : RADW_NOT_SOLID)));
      nextMisc = make_float4(maxRoughness.y, __uint_as_float(flags), __uint_as_float(seed), coneWidth);
      // A path that ends here leaves its radiance where k_finish_sample reads it and its seed where the next sample of a multi-sample
      // frame picks it up; one that goes on takes its state along in its queue entry (below, once the entry's position is known).
      if((!stateInQueue || !alive) && !deferred)
      {
        P.radiance[slot] = nextRad;
        if(P.misc)  // (null unless the frame has several samples or its state lives by slot: PathSoA)
          P.misc[slot] = nextMisc;
      }
      if(alive)
      {
        nextOrg = make_float4(rayOrigin.x, rayOrigin.y, rayOrigin.z, 0.0f);
        nextDir = make_float4(rayDir.x, rayDir.y, rayDir.z, __uint_as_float(seed));  // .w: the path's seed -- the walk's alpha draws start from it
        nextThr = make_float4(throughput.x, throughput.y, throughput.z, lastSamplePdf);
        if(!stateInQueue)
          P.throughput[slot] = nextThr;
      }
      if(COUNT && taps)
        atomicAdd(&stats->textureTaps, (unsigned long long)taps);
      if(COUNT && (meshHit || hitInfinitePlane))
        atomicAdd(&stats->surfaceHits, 1ull);
    }
#ifdef SHADE_PROFILE
    const unsigned long long sprofPush0 = __builtin_amdgcn_s_memtime();
#endif
    const PushPos  pp      = queuePushBlock2(alive, pushShadow, Q.subCap, &Q.counters[(nxt ? QC_PAIR1 : QC_PAIR0) + 2 * (chunk % NSUB)], chunk % NSUB, s_push);
    const uint32_t posNext = pp.next, posShadow = pp.shadow;
    if(alive)
    {
      Q.active[nxt].slot[posNext] = slot;
      Q.active[nxt].org[posNext]  = nextOrg;
      Q.active[nxt].dir[posNext]  = nextDir;
      if(stateInQueue)
      {
        Q.active[nxt].rad[posNext]  = nextRad;
        Q.active[nxt].misc[posNext] = nextMisc;
        Q.active[nxt].aux2[posNext] = nextThr;
      }
    }
    if(pushShadow)
    {
      // where k_shadow_resolve adds this ray's term: the path's next queue entry while it lives, PathSoA::radiance once it has ended
      Q.shadow.slot[posShadow] = (stateInQueue && alive) ? (posNext | SHADOW_TARGET_QUEUE) : slot;
      Q.shadow.org[posShadow]  = shOrg;
      Q.shadow.dir[posShadow]  = shDir;
      Q.shadow.aux[posShadow]  = shCon;
      if(catcher)
        Q.shadow.aux2[posShadow] = make_float4(shCon2.x, shCon2.y, shCon2.z, __uint_as_float(alive ? posNext : 0xffffffffu));
    }
#ifdef SHADE_PROFILE
    if(!FIRST && laneId() == 0)
    {
      const unsigned long long t_ = __builtin_amdgcn_s_memtime();
      atomicAdd(&g_shadeProf[14], t_ - sprofPush0);
      atomicAdd(&g_shadeProf[15], (t_ - sprofPush0) * (unsigned long long)__popcll(__ballot(alive || pushShadow)));
    }
    const unsigned long long sprofFin0 = __builtin_amdgcn_s_memtime();
#endif
    if(DEFER_MISS && deferMiss)
      finishMisses(false);  // (once 256 have gathered)
#ifdef SHADE_PROFILE
    if(!FIRST && laneId() == 0)
    {
      const unsigned long long t_ = __builtin_amdgcn_s_memtime();
      atomicAdd(&g_shadeProf[16], t_ - sprofFin0);
      atomicAdd(&g_shadeProf[18], t_ - sprofRound0);
      atomicAdd(&g_shadeProf[19], (t_ - sprofRound0) * (unsigned long long)__popcll(__ballot(inRange && slot != QUEUE_DEAD)));
    }
#endif
   }  // rounds of the window
   if(CAN_SORT)
     __syncthreads();  // s_order / s_segCount are rebuilt for the next window
  }
  if(DEFER_MISS && deferMiss)
  {
    finishMisses(true);
    finishMisses(true);  // (at most 511 were listed)
  }
}

//================================================================================================================================
// k_trace_shadow: RayQueryRaytracer::TraceShadow (raytracer_interface.h.slang:139-187) + `pt.radiance += contribution * T`
//================================================================================================================================
// The alpha / transmission flavour of the shadow kernel carries the ordered-candidate state on top of the walk and does not
// fit the 128 VGPRs of a 1024-thread workgroup without spilling into its hot loop: it runs as three 256-thread workgroups per
// CU with a smaller LDS node cache instead.
// MODE 0: every instance is FORCE_OPAQUE; 1: alpha-tested materials but no transmissive instance (no ordered candidates at all);
// 2: transmissive instances present, ordered accumulation of getShadowTransmission inside the walk (BVH2 scenes, and the rays of
//    MODE 3 whose candidates overflowed the pool: `overflowOnly`);
// 3: transmissive instances present, 8-wide BVH: the walk of MODE 1 (triangle rounds, deferred alpha rounds, 128 VGPRs, 16 waves per
//    CU) that RECORDS the transmissive candidates it meets instead of evaluating them; k_shadow_resolve orders and evaluates them
//    as one dense pass.  The ordered state (candidate list, search state, material evaluation: 245 VGPRs, 2 waves per SIMD) no
//    longer rides on every shadow ray, and no ray is walked more than once.
template <int MODE>
struct ShadowCfg
{
  static constexpr int BLOCK = MODE == 2 ? 256 : TRACE_BLOCK;
  static constexpr int CACHE = MODE == 2 ? 320 : (MODE == 1 ? NODE_CACHE_ALPHA : (MODE == 3 ? NODE_CACHE_ALPHA - 104 : NODE_CACHE));
};

// end of a shadow ray: radiance += contribution * transmission (gltf_pathtrace.slang:462-471), or the two outcomes of
// handleShadowCatcher (pathtrace_functions.h.slang:520-534) for rays the shade kernel flagged as catcher probes
PT_DEV void shadowDeposit(const PathSoA& P, const Queues& Q, int nxt, uint32_t slot, uint32_t qpos, bool catcherRay, f3 contrib, f3 total, bool occ,
                          float catcherDarken)
{
  if(!catcherRay)
  {
    if(!occ)
    {
      // `slot` is the entry's target (pt_scene.h: SHADOW_TARGET_QUEUE): the living path's record in the next active queue, or its slot
      float4* const target = (slot & SHADOW_TARGET_QUEUE) ? &Q.active[nxt].rad[slot & ~SHADOW_TARGET_QUEUE] : &P.radiance[slot];
      float4        rad    = *target;
      rad.x += contrib.x * total.x;
      rad.y += contrib.y * total.y;
      rad.z += contrib.z * total.z;
      *target = rad;
    }
    return;
  }
  const float4 a2  = Q.shadow.aux2[qpos];
  const f3     sf  = occ ? mk3(0.0f) : total;
  float4       rad = P.radiance[slot];
  if(sf.x == 1.0f && sf.y == 1.0f && sf.z == 1.0f)
  {
    rad.x += a2.x; rad.y += a2.y; rad.z += a2.z;  // unshadowed: environment seen through the plane, path ends
    const uint32_t posNext = __float_as_uint(a2.w);
    if(posNext != 0xffffffffu)
      Q.active[nxt].slot[posNext] = QUEUE_DEAD;
  }
  else
  {
    f3 r3 = mk3(rad.x, rad.y, rad.z);
    r3 += contrib * sf;
    r3 -= contrib * (mk3(1.0f) - sf) * catcherDarken;
    rad.x = r3.x; rad.y = r3.y; rad.z = r3.z;
  }
  P.radiance[slot] = rad;
}

template <bool WIDE, int MODE, bool COUNT>
__global__ void __launch_bounds__(ShadowCfg<MODE>::BLOCK, TRACE_MIN_WAVES) k_trace_shadow(DevScene sc, const DevScene* __restrict__ scp, PathSoA P, Queues Q, int nxt, float catcherDarken, StatCounters* stats, int overflowOnly)
{
  // `sc` (kernel argument, SGPRs) serves the inlined walk; the non-inlined material helpers of the transmissive path get the
  // device-resident copy `*scp` so that the argument's address never escapes (no scratch copy, cf. k_shade)
  constexpr int  SBLOCK    = ShadowCfg<MODE>::BLOCK;
  constexpr bool HAS_ALPHA = MODE >= 1, HAS_TRANS = MODE == 2, REC = MODE == 3;
  constexpr bool DEFER = MODE == 1 || MODE == 3;  // candidates of non-opaque instances go through alpha rounds
  static_assert(!REC || WIDE, "the recording walk exists for the 8-wide BVH only");
  __shared__ int      s_stack[BVH_STACK_LDS * SBLOCK];
  __shared__ uint32_t s_prefix[NSUB + 1];
  __shared__ uint4    s_nodes[WIDE ? ShadowCfg<MODE>::CACHE * 5 : 1];
  __shared__ uint32_t s_items[(WIDE && !HAS_TRANS) ? SBLOCK : 1];  // triangle rounds (see triRoundPublish)
  __shared__ uint4    s_alpha[(WIDE && DEFER) ? SBLOCK : 1];       // deferred alpha tests (see alphaRound)
  __shared__ uint32_t s_head[REC ? SBLOCK : 1], s_ovf[REC ? SBLOCK : 1];  // per ray in flight: recorded-candidate list head, overflow flag
  __shared__ uint8_t  s_octLut[(WIDE && MI_PT_OCT_LUT) ? OCT_LUT_BYTES : 1];  // child mask -> front-to-back order per octant (pt_bvh8.h: octPermute)
  queuePrefix(&Q.counters[(nxt ? QC_PAIR1 : QC_PAIR0) + 1], s_prefix);  // the shadow tails written next to active queue `nxt`
  const RayQueue in = Q.shadow;
  const bool     subset = MODE == 2 && overflowOnly != 0;  // only the rays k_trace_shadow<MODE 3> listed in Q.overflow
  uint32_t* const heads = &Q.counters[subset ? QC_HEADS_OVERFLOW : QC_HEADS_SHADOW];
  WaveFeed feed;
  feedInit(feed, subset ? Q.counters[QC_OVERFLOW] : s_prefix[NSUB]);
  if(!feedBlockHasWork(feed))
    return;
  if(WIDE && MI_PT_OCT_LUT)
    fillOctLut(s_octLut);
  const uint32_t cachedNodes = WIDE ? fillNodeCache(sc, s_nodes, uint32_t(ShadowCfg<MODE>::CACHE)) : 0u;
  LaneStack  st;
  LaneStack2 st2;
  int        stackOverflow[WIDE ? 2 * BVH8_STACK_PRIV : BVH_STACK_PRIV];  // scratch; only touched beyond the LDS depth
  st.lds = s_stack;  st.tid = int(threadIdx.x);  st.stride = SBLOCK;  st.sp = 0;  st.priv = stackOverflow;
  st2.lds = s_stack; st2.tid = int(threadIdx.x); st2.stride = SBLOCK; st2.sp = 0;
  st2.privBase = reinterpret_cast<uint32_t*>(stackOverflow); st2.privBits = reinterpret_cast<uint32_t*>(stackOverflow) + (WIDE ? BVH8_STACK_PRIV : 0);
  // per-lane state.  phase 0: any-hit walk (opaque geometry and non-transmissive alpha resolve here, order independent);
  // phase 1: one walk per transmissive candidate, in increasing (t, renderNode, primitive) order.
  bool      active = false;
  uint32_t  slot   = 0;
  RaySetup  r{};
  int       node = BVH_EMPTY;
  NodeGroup G{0, 0};
  uint32_t  octinv = 0;
  float     tMax  = 0.0f;
  int       phase = 0;
  unsigned  nTrans = 0;
  bool      occluded = false;
  uint32_t  seed0 = 0;
  f3        contrib = mk3(0.0f);
  // phase-1 search state
  float    bT = 0.0f, bU = 0.0f, bV = 0.0f, lastT = -1.0f, prevHitT = 0.0f;
  uint32_t bRnode = 0, bPrim = 0, lastRnode = 0, lastPrim = 0;
  int      bTri = 0;
  bool     found = false, haveLast = false, isInside = false;
  f3       total = mk3(1.0f);
  unsigned nodes = 0, tris = 0, rays = 0;
  // prefetched next shadow ray
  bool     pValid = false;
  uint32_t pSlot = QUEUE_DEAD, pQPos = 0, qpos = 0;
  float4   pO = make_float4(0, 0, 0, 0), pD = make_float4(0, 0, 0, 0), pC = make_float4(0, 0, 0, 0);
  bool     catcherRay = false;
  uint32_t pBase = 0, pMask = 0, qBase = 0, qMask = 0;  // parked leaf hits of the 8-wide walk (see k_trace_closest)
  uint32_t aCount   = 0;                                // deferred alpha tests of this wave / some of them this lane's (MODE 1, 3)
  bool     aPending = false;
  bool     toOverflow = false;                          // MODE 3: this lane's ray just ended with candidates that did not fit the pool
#ifdef TRACE_PROFILE
  unsigned long long sp[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long spStart = PROF_T();
#define SPROF_ADD(i, t0) sp[i] += PROF_T() - (t0)
#define SPROF_CNT(i, n) sp[i] += (unsigned long long)(n)
#else
#define SPROF_ADD(i, t0) (void)(t0)
#define SPROF_CNT(i, n) (void)0
#endif
  uint4*   waveAlpha = s_alpha + ((WIDE && DEFER) ? (threadIdx.x & ~63u) : 0u);
  CandRec  rec{Q.candPool, Q.candNext, &Q.counters[QC_CAND_POOL], Q.candCap, s_head + (REC ? (threadIdx.x & ~63u) : 0u), s_ovf + (REC ? (threadIdx.x & ~63u) : 0u), 0u, 0u};
  (void)rec;

  auto deposit = [&](bool occ) { shadowDeposit(P, Q, nxt, slot, qpos, catcherRay, contrib, total, occ, catcherDarken); };

  // Transmissive candidates met by the any-hit walk, kept so that the common case (a few glass surfaces on the way to the
  // light) is settled by ordering this list instead of one search walk per candidate.  Separate local arrays: scratch, only
  // touched on such hits.
  constexpr unsigned SHADOW_CANDS = 8;
  float    cT[HAS_TRANS ? SHADOW_CANDS : 1], cU[HAS_TRANS ? SHADOW_CANDS : 1], cV[HAS_TRANS ? SHADOW_CANDS : 1];
  uint32_t cR[HAS_TRANS ? SHADOW_CANDS : 1], cP[HAS_TRANS ? SHADOW_CANDS : 1];
  int      cI[HAS_TRANS ? SHADOW_CANDS : 1];
  // one accepted-or-not transmissive candidate in order (raytracer_interface.h.slang:160-178)
  auto processCandidate = [&](float t, int tri, uint32_t rnode, uint32_t prim, float u, float v) {
    f3    bary    = mk3(1.0f - u - v, u, v);
    float opacity = getOpacityFast(sc, tri, bary);
    if(candidateRand(seed0, int(rnode), int(prim)) < opacity)
    {
      float segment = fmaxf(0.0f, t - prevHitT);
      f3    curT    = getShadowTransmission(*scp, int(rnode), int(prim), bary, segment, r.dir, isInside);
      prevHitT      = t;
      total *= curT;
      if(maxComp(total) <= MIN_TRANSMISSION)
        occluded = true;
    }
  };

  auto restartWalk = [&]() {
    if(WIDE)
    {
      G      = rootGroup(octinv);
      st2.sp = 0;
      pMask = qMask = 0u;
    }
    else
    {
      st.sp = 0;
      node  = sc.bvhRoot;
    }
  };
  // one shadow candidate (raytracer_interface.h.slang:149-179); returns true when the ray is decided (occluded)
  auto testTriLoaded = [&](const DevTri& T, int triIndex) {
    if(COUNT) ++tris;
    const uint32_t flags = __float_as_uint(T.c.w);
    TriHit         h;
    const bool consider = !HAS_TRANS || phase == 0 || ((flags & INST_TRANSMISSIVE) && !(flags & INST_FORCE_OPAQUE));
    if(!(consider && intersectTri(xyz(T.a), xyz(T.b), xyz(T.c), r.org, r.dir, h) && h.t > 0.0f && h.t < tMax))
      return;
    const uint32_t rnode = __float_as_uint(T.a.w), prim = __float_as_uint(T.b.w);
    if(!HAS_TRANS || phase == 0)
    {
      if(!HAS_ALPHA || (flags & INST_FORCE_OPAQUE))
        occluded = true;  // RAY_FLAG_NONE: no culling; opaque geometry commits
      else if(HAS_TRANS && (flags & INST_TRANSMISSIVE))
      {
        if(nTrans < SHADOW_CANDS)
        {
          cT[nTrans] = h.t; cI[nTrans] = triIndex; cR[nTrans] = rnode; cP[nTrans] = prim; cU[nTrans] = h.u; cV[nTrans] = h.v;
        }
        ++nTrans;
      }
      else
      {
        // non-transmissive alpha material: an accepted candidate multiplies the transmission by
        // getShadowTransmission() == 0 (pathtrace_functions.h.slang:256-261) whatever its position in the order
        float opacity = getOpacityFast(sc, triIndex, mk3(1.0f - h.u - h.v, h.u, h.v));
        if(candidateRand(seed0, int(rnode), int(prim)) < opacity)
          occluded = true;
      }
    }
    else
    {
      const bool afterLast  = !haveLast || h.t > lastT || (h.t == lastT && (rnode > lastRnode || (rnode == lastRnode && prim > lastPrim)));
      const bool beforeBest = !found || h.t < bT || (h.t == bT && (rnode < bRnode || (rnode == bRnode && prim < bPrim)));
      if(afterLast && beforeBest)
      {
        found = true; bT = h.t; bTri = triIndex; bRnode = rnode; bPrim = prim; bU = h.u; bV = h.v;
      }
    }
  };

  auto testTri = [&](int triIndex) { testTriLoaded(sc.tris[triIndex], triIndex); };

  for(;;)
  {
    const unsigned long long tOuter = PROF_T();
    SPROF_CNT(8, 1);
    SPROF_CNT(9, 64 - __popcll(__ballot(active)));
    if(!active && pValid)
    {
      pValid = false;
      if(pSlot != QUEUE_DEAD)
      {
        slot     = pSlot;
        qpos     = pQPos;
        r        = makeRaySetup(xyz(pO), xyz(pD));
        tMax     = pO.w;
        isInside = (__float_as_uint(pD.w) & 1u) != 0u;
        catcherRay = (__float_as_uint(pD.w) & 2u) != 0u;
        contrib  = xyz(pC);
        seed0    = __float_as_uint(pC.w);
        phase    = 0;
        nTrans   = 0;
        occluded = false;
        total    = mk3(1.0f);
        haveLast = false;
        found    = false;
        prevHitT = 0.0f;
        // any-hit is order independent: MI_PT_SHADOW_FAR_FIRST walks from the ray's far end (DevScene::shadowOctFlip); the ordered search of MODE 2 keeps near first
        octinv   = rayOctInv(r.idir) ^ (MODE == 2 ? 0u : sc.shadowOctFlip);
        restartWalk();
        active = true;
        if(REC)
        {
          s_head[threadIdx.x] = CAND_NIL;
          s_ovf[threadIdx.x]  = 0u;
        }
        if(COUNT) ++rays;
        if(sc.bvhRoot == BVH_EMPTY)  // nothing to hit: unoccluded
        {
          if(HAS_TRANS)
            deposit(false);
          else
            reinterpret_cast<uint32_t*>(&Q.shadow.org[qpos])[3] = CAND_NIL;
          active = false;
        }
      }
    }
    if(!feed.exhausted)
    {
      uint32_t flat = feedTake(feed, !pValid, heads);
      if(flat != 0xffffffffu)
      {
        const uint32_t pPos = subset ? Q.overflow[flat] : queuePos(Q.subCap, s_prefix, flat);
        pQPos  = pPos;
        pSlot  = in.slot[pPos];
        pO     = in.org[pPos];
        pD     = in.dir[pPos];
        pC     = in.aux[pPos];
        pValid = true;
      }
    }
    const bool moreWork = !feed.exhausted || __ballot(pValid) != 0ull;
    SPROF_ADD(0, tOuter);
    if(__ballot(active) == 0ull)
    {
      if(!moreWork)
        break;
      continue;
    }
    for(;;)
    {
      bool walkDone = false;
      const unsigned long long tInner = PROF_T();
      SPROF_CNT(10, 1);
      SPROF_CNT(11, __popcll(__ballot(active)));
      if(WIDE)
      {
        // node step + dense triangle phase, as in k_trace_closest (any-hit and the phase-1 search are order independent)
        bool visited = false;
        if(active && !leafPending(qMask) && !(DEFER && occluded))  // (an occluded ray may still wait for its deferred alpha tests)
        {
          const float walkTmax = (HAS_TRANS && phase == 1 && found) ? bT : tMax;
          if((G.bits >> 8) == 0u && st2.sp > 0)
            G = st2.pop();
          if(G.bits >> 8)
          {
            const uint32_t child = groupPopChild(G, octinv);
            if(G.bits >> 8)
              st2.push(G);
            uint32_t tBase, tMask;
            bvh8Visit(sc, r, walkTmax, octinv, child, G, tBase, tMask, s_nodes, cachedNodes, s_octLut);
            if(COUNT) ++nodes;
            visited = true;
            if(leafPending(tMask))
            {
              if(!leafPending(pMask)) { pBase = tBase; pMask = tMask; }
              else                    { qBase = tBase; qMask = tMask; }
            }
          }
        }
        SPROF_ADD(1, tInner);
        SPROF_CNT(12, __popcll(__ballot(visited)));
        const unsigned long long tTri = PROF_T();
        unsigned long long pend = __ballot(active && leafPending(pMask));
        if(!HAS_TRANS)
        {
          // without transmissive instances a committing candidate simply decides its ray: the triangle tests are spread over
          // the whole wave (triRoundPublish / triRoundFinishShadow), as in k_trace_closest
          if(pend != 0ull)
          {
            const int  visiting = __popcll(__ballot(visited));
            const bool drain    = visiting < TRI_PHASE_LANES;
            if(drain || __popcll(pend) >= TRI_ROUND_LANES)
            {
              do
              {
                TriRound tr;
                triRoundPublish(sc, active, pBase, pMask, qBase, qMask, s_items + (threadIdx.x & ~63u), tr);
                triRoundFinishShadow<HAS_ALPHA, COUNT, REC>(sc, r, tMax, seed0, tr, occluded, tris, waveAlpha, aCount, aPending, &rec);
                if(occluded) { pMask = 0u; qMask = 0u; }
                pend = __ballot(active && leafPending(pMask));
              } while(pend != 0ull && (drain || __popcll(pend) >= TRI_ROUND_LANES));
            }
          }
          SPROF_ADD(2, tTri);
          const unsigned long long tAlpha = PROF_T();
          if(DEFER && aCount != 0u)
          {
            const bool walked  = active && (occluded || ((G.bits >> 8) == 0u && st2.sp == 0 && !leafPending(pMask)));
            const int  waiting = __popcll(__ballot(walked && aPending));
            if(aCount >= ALPHA_ROUND_MIN || __popcll(__ballot(visited)) < TRI_PHASE_LANES || waiting >= ALPHA_ROUND_WAITING)
            {
              ClosestBest none{};
              SPROF_CNT(13, 1);
              SPROF_CNT(14, aCount);
              alphaRound<true, REC>(sc, seed0, waveAlpha, aCount, aPending, none, occluded, &rec);
              if(occluded) { pMask = 0u; qMask = 0u; }
            }
          }
          SPROF_ADD(3, tAlpha);
        }
        else if(pend != 0ull)
        {
          const int  visiting = __popcll(__ballot(visited));
          const bool drain    = visiting < TRI_PHASE_LANES;
          if(drain || __popcll(pend) >= TRI_PHASE_LANES || __ballot(active && leafPending(qMask)) != 0ull)
          {
            do
            {
              if(active && leafPending(pMask))
              {
                // two triangles per round, both records in flight together (as in k_trace_closest)
                const int  i0  = int(leafPop(pMask, pBase));
                const bool two = leafPending(pMask);
                const int  i1  = two ? int(leafPop(pMask, pBase)) : i0;
                const DevTri T0 = sc.tris[i0], T1 = sc.tris[i1];
                testTriLoaded(T0, i0);
                if(two && !occluded)
                  testTriLoaded(T1, i1);
                if(occluded) { pMask = 0u; qMask = 0u; }
                else if(!leafPending(pMask)) { pBase = qBase; pMask = qMask; qMask = 0u; }
              }
              pend = __ballot(active && leafPending(pMask));
            } while(pend != 0ull && (drain || __popcll(pend) >= TRI_PHASE_EXIT_LANES || __ballot(active && leafPending(qMask)) != 0ull));
          }
        }
        walkDone = active && !(DEFER && aPending) && (occluded || ((G.bits >> 8) == 0u && st2.sp == 0 && !leafPending(pMask)));
      }
      else if(active)
      {
        const float walkTmax = (HAS_TRANS && phase == 1 && found) ? bT : tMax;
#pragma unroll 1
        for(int k = 0; k < 4 && node >= 0; ++k)
        {
          node = bvhInnerStep(sc, r, walkTmax, node, st);
          if(COUNT) ++nodes;
        }
        if(node < 0 && node != BVH_EMPTY)
        {
          testTri(~node);
          node = bvhPop(st);
        }
        walkDone = occluded || node == BVH_EMPTY;
      }
      const unsigned long long tFin = PROF_T();
      if(active)
      {
        if(walkDone)
        {
          bool finished = true;
          if(HAS_TRANS && !occluded && phase == 0 && nTrans > 0 && nTrans <= SHADOW_CANDS)
          {
            // every candidate is on the list: take them in increasing (t, renderNode, primitive) order
            for(unsigned k = 0; k < nTrans && !occluded; ++k)
            {
              int best = -1;
              for(unsigned j = 0; j < nTrans; ++j)
              {
                const bool afterLast  = !haveLast || cT[j] > lastT || (cT[j] == lastT && (cR[j] > lastRnode || (cR[j] == lastRnode && cP[j] > lastPrim)));
                const bool beforeBest = best < 0 || cT[j] < cT[best] || (cT[j] == cT[best] && (cR[j] < cR[best] || (cR[j] == cR[best] && cP[j] < cP[best])));
                if(afterLast && beforeBest)
                  best = int(j);
              }
              if(best < 0)
                break;
              haveLast = true; lastT = cT[best]; lastRnode = cR[best]; lastPrim = cP[best];
              processCandidate(cT[best], cI[best], cR[best], cP[best], cU[best], cV[best]);
            }
            nTrans = 0;
          }
          if(HAS_TRANS && !occluded && !(phase == 1 && !found))
          {
            if(phase == 1)
            {
              haveLast = true; lastT = bT; lastRnode = bRnode; lastPrim = bPrim;
              processCandidate(bT, bTri, bRnode, bPrim, bU, bV);
              --nTrans;
            }
            if(!occluded && nTrans > 0)
            {
              // (re)start a search walk for the next transmissive candidate
              phase = 1;
              found = false;
              restartWalk();
              finished = false;
            }
          }
          if(finished)
          {
            if(HAS_TRANS)
              deposit(occluded);  // the ordered-search kernel settles its rays itself
            else
            {
              // The outcome goes into the ray's queue entry (org.w, the walk is done with tmax): k_shadow_resolve adds the
              // contributions as one streaming pass.  Doing it here -- a dependent read-modify-write of the path's radiance in
              // the middle of a persistent wave -- stalled the whole wave for a memory round trip whenever one of its rays
              // ended: a quarter of this kernel's time on the glass workload.
              uint32_t code = occluded ? SHADOW_OCCLUDED : CAND_NIL;
              if(REC && !occluded)
              {
                code       = s_head[threadIdx.x];
                toOverflow = s_ovf[threadIdx.x] != 0u;
              }
              if(toOverflow)  // the entry keeps its tmax for the ordered-search kernel and is flagged in dir.w (bit 2)
                reinterpret_cast<uint32_t*>(&Q.shadow.dir[qpos])[3] |= SHADOW_DIR_OVERFLOW;
              else
                reinterpret_cast<uint32_t*>(&Q.shadow.org[qpos])[3] = code;
            }
            active = false;
          }
        }
      }
      if(REC)
      {
        // rays whose candidates did not fit the pool are listed for the ordered-search kernel (rare: one device atomic per event)
        const unsigned long long mO = __ballot(toOverflow);
        if(mO != 0ull)
        {
          const int first = __ffsll((long long)mO) - 1;
          uint32_t  base  = 0u;
          if(int(laneId()) == first)
            base = atomicAdd(&Q.counters[QC_OVERFLOW], uint32_t(__popcll(mO)));
          base = uint32_t(__builtin_amdgcn_readlane(int(base), first));
          if(toOverflow)
            Q.overflow[base + laneCountBelow(mO)] = qpos;
        }
        toOverflow = false;
      }
      SPROF_ADD(4, tFin);
      const unsigned long long act = __ballot(active);
      if(act == 0ull || (moreWork && __popcll(act) <= 64 - REFILL_IDLE_LANES))
        break;
    }
  }
#ifdef TRACE_PROFILE
  if(laneId() == 0 && WIDE && DEFER)
  {
    sp[5] = PROF_T() - spStart;
    sp[6] = 1;
    for(int i = 0; i < 16; ++i)
      atomicAdd(&g_shadowProf[i], sp[i]);
  }
#endif
#undef SPROF_ADD
#undef SPROF_CNT
  if(COUNT)
  {
    atomicAdd(&stats->shadowRays, (unsigned long long)rays);
    atomicAdd(&stats->nodesShadow, (unsigned long long)nodes);
    atomicAdd(&stats->trisShadow, (unsigned long long)tris);
  }
}

//================================================================================================================================
// k_shadow_resolve: the end of every shadow ray -- `radiance += contribution * transmission` (gltf_pathtrace.slang:462-471) -- and
// the ordered half of TraceShadow for rays that met transmissive candidates (raytracer_interface.h.slang:160-178)
//================================================================================================================================
// One streaming pass over the shadow queue, one thread per entry, reading the outcome k_trace_shadow left in org.w.  Occluded rays
// cost one word.  A ray with recorded candidates (a chain in Q.candPool, typically two to four: the shells of the glass between a
// point and the light) takes them in increasing (t, renderNode, primitive) order by repeated selection over the chain -- no
// per-thread array, the chain is L2 resident -- and each accepted one multiplies the transmission by getShadowTransmission() over
// the segment since the previous accepted one, until the product drops to MIN_TRANSMISSION.  Nothing here walks the BVH.
// One recorded shadow ray: its transmissive candidates in increasing (t, renderNode, primitive) order -> transmission (raytracer_interface.h.slang:160-178),
// then the deposit.  `code` = head of the ray's candidate chain.
PT_DEV void resolveChain(const DevScene& sc, const PathSoA& P, const Queues& Q, int nxt, uint32_t qpos, uint32_t slot, float4 d4, float4 c4, uint32_t code, bool catcherRay,
                         float catcherDarken)
{
  f3             total    = mk3(1.0f);
  bool           occluded = false;
  const f3       dir      = xyz(d4);
  const uint32_t seed0    = __float_as_uint(c4.w);
  bool           isInside = (__float_as_uint(d4.w) & 1u) != 0u, haveLast = false;
  float          lastT = -1.0f, prevHitT = 0.0f;
  uint32_t       lastRnode = 0u, lastPrim = 0u;
  for(;;)
  {
    // next candidate in the order: the smallest (t, renderNode, primitive) after the last one taken
    uint32_t best = CAND_NIL, bRnode = 0u, bPrim = 0u;
    bool     bIds = false;
    float4   bC   = make_float4(0, 0, 0, 0);
    for(uint32_t k = code; k != CAND_NIL; k = Q.candNext[k])
    {
      const float4   c   = Q.candPool[k];
      const uint32_t tri = __float_as_uint(c.w);
      const bool     tieLast = haveLast && c.x == lastT, tieBest = best != CAND_NIL && c.x == bC.x;
      uint32_t       rnode = 0u, prim = 0u;
      if(tieLast || tieBest)  // exact ties only: coincident surfaces
      {
        rnode = __float_as_uint(gat(sc.tris, tri).a.w);
        prim  = __float_as_uint(gat(sc.tris, tri).b.w);
      }
      const bool afterLast  = !haveLast || c.x > lastT || (tieLast && (rnode > lastRnode || (rnode == lastRnode && prim > lastPrim)));
      bool       beforeBest = best == CAND_NIL || c.x < bC.x;
      if(tieBest)
      {
        if(!bIds)  // the best so far was taken without its ids
        {
          const uint32_t bt = __float_as_uint(bC.w);
          bRnode = __float_as_uint(gat(sc.tris, bt).a.w);
          bPrim  = __float_as_uint(gat(sc.tris, bt).b.w);
          bIds   = true;
        }
        beforeBest = rnode < bRnode || (rnode == bRnode && prim < bPrim);
      }
      if(afterLast && beforeBest)
      {
        best = k; bC = c; bRnode = rnode; bPrim = prim; bIds = tieLast || tieBest;
      }
    }
    if(best == CAND_NIL)
      break;
    const uint32_t tri = __float_as_uint(bC.w);
    const uint32_t rnode = __float_as_uint(gat(sc.tris, tri).a.w), prim = __float_as_uint(gat(sc.tris, tri).b.w);
    haveLast = true; lastT = bC.x; lastRnode = rnode; lastPrim = prim;
    const f3    bary    = mk3(1.0f - bC.y - bC.z, bC.y, bC.z);
    // (INST_ALPHA_PASSES: opacity 1, the draw always commits -- no fetch of the alpha record, no draw)
    const bool  passes  = (__float_as_uint(gat(sc.tris, tri).c.w) & INST_ALPHA_PASSES) != 0u;
    if(passes || candidateRand(seed0, int(rnode), int(prim)) < getOpacityFast(sc, int(tri), bary))
    {
      const float segment = fmaxf(0.0f, bC.x - prevHitT);
      const f3    curT    = getShadowTransmissionTri(sc, int(rnode), int(prim), int(tri), bary, segment, dir, isInside);
      prevHitT            = bC.x;
      total *= curT;
      if(maxComp(total) <= MIN_TRANSMISSION)
      {
        occluded = true;
        break;
      }
    }
  }
  if(occluded && !catcherRay)
    return;
  shadowDeposit(P, Q, nxt, slot, qpos, catcherRay, xyz(c4), total, occluded, catcherDarken);
}

// REC: the rays that carry a chain of recorded candidates are a minority spread over every wave (glass-class workload: the kernel ran 12.8 of 64 lanes, profiles/
// r06_glass_pmc_summary.json): they are listed in LDS while the block streams through its entries -- rays without a chain are settled on the spot -- and the block
// works the list off 256 rays at a time, every lane on a chain (-DRESOLVE_INLINE_CHAINS: rounds 2-5, each ray's chain where the ray is met).
template <bool REC>
__global__ void __launch_bounds__(256) k_shadow_resolve(const DevScene* __restrict__ scp, PathSoA P, Queues Q, int nxt, float catcherDarken)
{
  __shared__ uint32_t s_prefix[NSUB + 1];
#ifndef RESOLVE_INLINE_CHAINS
  constexpr bool LIST = REC;
#else
  constexpr bool LIST = false;
#endif
  __shared__ uint32_t s_chain[LIST ? 512 : 1];
  __shared__ uint32_t s_chainCount;
  if(threadIdx.x == 0)
    s_chainCount = 0;
  queuePrefix(&Q.counters[(nxt ? QC_PAIR1 : QC_PAIR0) + 1], s_prefix);
  const uint32_t  count = s_prefix[NSUB];
  const DevScene& sc    = uniformConst(*scp);
  auto workOff = [&](bool all) {  // whole block; contains barriers
    __syncthreads();
    const uint32_t n = s_chainCount;
    if(n == 0u || (!all && n < 256u))
      return;
    const uint32_t take = min(n, 256u), base = n - take;
    uint32_t       qpos = 0;
    if(threadIdx.x < take)
      qpos = s_chain[base + threadIdx.x];
    __syncthreads();
    if(threadIdx.x == 0)
      s_chainCount = base;
    if(threadIdx.x < take)
    {
      const float4 d4 = Q.shadow.dir[qpos];
      resolveChain(sc, P, Q, nxt, qpos, Q.shadow.slot[qpos], d4, Q.shadow.aux[qpos], reinterpret_cast<const uint32_t*>(&Q.shadow.org[qpos])[3], (__float_as_uint(d4.w) & 2u) != 0u,
                   catcherDarken);
    }
    __syncthreads();
  };
  const uint32_t rounds = (count + gridDim.x * blockDim.x - 1u) / (gridDim.x * blockDim.x);  // (the same for every thread of the block: workOff has barriers)
  for(uint32_t it = 0; it < rounds; ++it)
  {
    const uint32_t i = it * gridDim.x * blockDim.x + blockIdx.x * blockDim.x + threadIdx.x;
    if(i < count)
    {
      const uint32_t qpos = queuePos(Q.subCap, s_prefix, i);
      const uint32_t slot = Q.shadow.slot[qpos];
      const float4   d4   = slot != QUEUE_DEAD ? Q.shadow.dir[qpos] : make_float4(0, 0, 0, 0);
      // (dead entry; or REC: settled by the ordered-search kernel)
      if(slot != QUEUE_DEAD && !(REC && (__float_as_uint(d4.w) & SHADOW_DIR_OVERFLOW)))
      {
        const uint32_t code       = reinterpret_cast<const uint32_t*>(&Q.shadow.org[qpos])[3];
        const bool     catcherRay = (__float_as_uint(d4.w) & 2u) != 0u;
        const bool     occluded   = code == SHADOW_OCCLUDED;
        if(!occluded || catcherRay)
        {
          if(REC && !occluded && code != CAND_NIL)
          {
            if(LIST)
              s_chain[atomicAdd(&s_chainCount, 1u)] = qpos;
            else
              resolveChain(sc, P, Q, nxt, qpos, slot, d4, Q.shadow.aux[qpos], code, catcherRay, catcherDarken);
          }
          else
            shadowDeposit(P, Q, nxt, slot, qpos, catcherRay, xyz(Q.shadow.aux[qpos]), mk3(1.0f), occluded, catcherDarken);
        }
      }
    }
    if(LIST)
      workOff(false);
  }
  if(LIST)
  {
    workOff(true);
    workOff(true);
  }
}

//================================================================================================================================
// k_flush_survivors: paths still alive when the host's bounce loop stops (volume-scatter scenes: the loop is cut at maxDepth * 66 + 512
// iterations, mi_pt_api.hip) carry their radiance and seed in their queue entry (FrameConsts::stateInQueue) -- the records
// k_finish_sample and the next sample of a multi-sample frame read BY SLOT are written only when a path ends inside the loop.  This
// pass writes them for the survivors, so that a truncated path keeps its partial radiance and its current seed as it did while the
// state lived by slot.  Normally the queue is empty and the launch costs a few microseconds per sample.
//================================================================================================================================
__global__ void __launch_bounds__(256) k_flush_survivors(PathSoA P, Queues Q, int cur)
{
  __shared__ uint32_t s_prefix[NSUB + 1];
  queuePrefix(&Q.counters[cur ? QC_PAIR1 : QC_PAIR0], s_prefix);
  const uint32_t count = s_prefix[NSUB];
  for(uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x)
  {
    const uint32_t qpos = queuePos(Q.subCap, s_prefix, i);
    const uint32_t slot = Q.active[cur].slot[qpos];
    if(slot == QUEUE_DEAD)
      continue;
    P.radiance[slot] = Q.active[cur].rad[qpos];
    if(P.misc)
      P.misc[slot] = Q.active[cur].misc[qpos];
  }
}

//================================================================================================================================
// k_finish_sample: firefly clamp + per-frame mean + running-mean accumulation + NDC depth
// (gltf_pathtrace.slang:531-538, 596, 604-630)
//================================================================================================================================
#ifndef FINISH_AHEAD
#define FINISH_AHEAD 4  // records fetched ahead of the fold (build switch for the A/B)
#endif
__global__ void __launch_bounds__(256) k_finish_sample(FrameConsts fc, const DevScene* __restrict__ scp, const FrameConsts* __restrict__ fcp, PathSoA P,
                                                       const uint32_t* ownedTiles, int sampleIndex, float4* accum, float* depth, float4* albedoOut,
                                                       float4* normalOut)
{
  // One thread per PIXEL slot; the frames in flight are folded into the running mean in frame order, with exactly the
  // arithmetic of numFrames successive single-frame dispatches.
  const uint32_t pslot = blockIdx.x * blockDim.x + threadIdx.x;
  int            px, py;
  if(pslot >= uint32_t(fc.numSlots) || !slotToPixel(fc, ownedTiles, pslot, px, py))
    return;
  const size_t idx      = size_t(py) * size_t(fc.width) + size_t(px);
  const bool   lastSample = sampleIndex + 1 >= fc.pc.numSamples;
  const float  n        = float(fc.pc.numSamples);
  const bool   guides   = P.guideAlbedo && albedoOut;
  float4       acc = make_float4(0, 0, 0, 0), accA = acc, accN = acc;
  bool         loaded = false;
  // Four frames' records are fetched ahead of their (serial, in frame order) fold: with pixel-major slots a pixel's records are
  // consecutive -- the four share a 64-byte line, which a thread must ask for while it is hot, its neighbours' lines lying 16 x
  // numFrames bytes apart -- and in either layout the four gathers overlap instead of queueing behind one another.
  for(int f0 = 0; f0 < fc.numFrames; f0 += FINISH_AHEAD)
  {
   float4   radAhead[FINISH_AHEAD];
#pragma unroll
   for(int j = 0; j < FINISH_AHEAD; ++j)
     if(f0 + j < fc.numFrames)
     {
       const uint32_t sj = pathSlot(fc, pslot, uint32_t(f0 + j));
       radAhead[j]       = P.radiance[sj];
     }
#pragma unroll
   for(int j = 0; j < FINISH_AHEAD; ++j)
   {
    const int f = f0 + j;
    if(f >= fc.numFrames)
      break;
    const uint32_t slot  = pathSlot(fc, pslot, uint32_t(f));
    float4         rad4  = radAhead[j];
    // what this kernel needs of the path's flags rides in radiance.w (PathSoA): one 16-byte record per path and frame, not two
    const uint32_t radw  = __float_as_uint(rad4.w);
    const bool     solid = !(radw & RADW_NOT_SOLID);
    if(radw == RADW_PRIMARY_MISS)
    {
      // a camera ray that left the scene: rad4 is its direction (k_trace_primary); the shade kernel's miss branch for a first ray
      const f3 dir      = mk3(rad4.x, rad4.y, rad4.z);
      f3       radiance = mk3(0.0f);
      if(!primaryMissBackplate(*scp, *fcp, dir, radiance))
      {
        f3    envColor;
        float mis;
        missEnvironment(*scp, *fcp, dir, DIRAC, envColor, mis);
        radiance += mk3(1.0f) * mis * envColor;
      }
      rad4 = make_float4(radiance.x, radiance.y, radiance.z, 0.0f);
    }
    f4             r     = mk4(rad4.x, rad4.y, rad4.z, solid ? 1.0f : 0.0f);
    // A sample whose radiance is not finite is dropped (black, its coverage kept) instead of poisoning the running mean for good: with the shading arithmetic at
    // the reference's precision class a division by a denormal pdf is an infinity where IEEE gives ~1e38 -- the glass + dragon workload produced one such path in
    // 1e9 (frame 407, pixel (1736, 558): the oracle's value there is a 4.5e5 firefly the clamp below cuts to luminance 10; tools/diag_nonfinite.py) -- and the
    // reference's own OpFDiv is undefined for that divisor as well.
    if(!(fabsf(r.x) <= 3.0e38f && fabsf(r.y) <= 3.0e38f && fabsf(r.z) <= 3.0e38f))
      r = mk4(0.0f, 0.0f, 0.0f, r.w);
    float          lum   = dot(xyz(r), mk3(1.0f / 3.0f));
    if(lum > fc.pc.fireflyClampThreshold)
      r *= divExact(fc.pc.fireflyClampThreshold, lum);  // (k_finish_sample keeps IEEE division: the accumulator is the oracle's running mean of the same samples, pt_math.h)
    float4 sum = sampleIndex == 0 ? make_float4(0, 0, 0, 0) : P.pixelSum[slot];
    sum        = make_float4(sum.x + r.x, sum.y + r.y, sum.z + r.z, sum.w + r.w);
    if(!lastSample)
    {
      P.pixelSum[slot] = sum;
      continue;
    }
    const f4    pixel      = mk4(divExact(sum.x, n), divExact(sum.y, n), divExact(sum.z, n), divExact(sum.w, n));
    const bool  firstFrame = f == 0 && hasFlag(fc.pc.flags, MI_PT_FIRST_FRAME);
    const float tot = float(fc.pc.totalSamples + f * fc.pc.numSamples), after = float(fc.pc.totalSamples + (f + 1) * fc.pc.numSamples);
    if(!firstFrame && !loaded)
    {
      acc = accum[idx];
      if(guides)
      {
        accA = albedoOut[idx];
        accN = normalOut[idx];
      }
    }
    loaded = true;
    if(firstFrame)
    {
      const bool hasSolidHit = r.w > 0.0f;
      float      ndcDepth    = 1.0f;
      if(hasSolidHit)
      {
        float4 fh   = P.firstHit[pslot];  // (per pixel slot: f == 0 here)
        f4     clip = mulFull(fc.frameInfo.viewProjMatrix, mk4(fh.x, fh.y, fh.z, 1.0f));
        ndcDepth    = divExact(clip.z, clip.w);
      }
      depth[idx] = ndcDepth;
      acc        = make_float4(pixel.x, pixel.y, pixel.z, pixel.w);
    }
    else
      acc = make_float4(divExact(acc.x * tot + pixel.x * n, after), divExact(acc.y * tot + pixel.y * n, after), divExact(acc.z * tot + pixel.z * n, after),
                        divExact(acc.w * tot + pixel.w * n, after));
    if(guides)
    {
      const float4 ga = P.guideAlbedo[slot], gn = P.guideNormal[slot];
      const float4 a  = make_float4(divExact(ga.x, n), divExact(ga.y, n), divExact(ga.z, n), r.w > 0.0f ? 1.0f : 0.0f);
      // .w: second moment of this frame's pixel luminance (Rec. 709) -- with luminance(accum) it gives the temporal variance the
      // SVGF pass is guided by (denoise.hip), at no extra image
      const float  lumP = 0.2126f * pixel.x + 0.7152f * pixel.y + 0.0722f * pixel.z;
      const float4 nn   = make_float4(divExact(gn.x, n), divExact(gn.y, n), divExact(gn.z, n), lumP * lumP);
      if(firstFrame)
      {
        accA = a;
        accN = nn;
      }
      else
      {
        const float wOld = divExact(tot, after), wNew = divExact(n, after);
        accA = make_float4(accA.x * wOld + a.x * wNew, accA.y * wOld + a.y * wNew, accA.z * wOld + a.z * wNew, accA.w * wOld + a.w * wNew);
        accN = make_float4(accN.x * wOld + nn.x * wNew, accN.y * wOld + nn.y * wNew, accN.z * wOld + nn.z * wNew, accN.w * wOld + nn.w * wNew);
      }
    }
   }
  }
  if(lastSample)
  {
    accum[idx] = acc;
    if(guides)
    {
      albedoOut[idx] = accA;
      normalOut[idx] = accN;
    }
  }
}

__global__ void k_alpha_records(DevScene sc, uint32_t n, DevAlphaTri* out)
{
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i < n)
    out[i] = makeAlphaRecord(sc, sc.tris[i]);
}
__global__ void k_shade_records(DevScene sc, uint32_t n, DevShadeTri* out)
{
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i < n)
    out[i] = makeShadeRecord(sc, sc.tris[i]);
}

// bilinear footprints of one mip level (DevScene::texQuads): neighbours under the texture's wrap modes
__global__ void __launch_bounds__(256) k_texture_quads(const uint32_t* __restrict__ texels, uint4* __restrict__ quads, uint32_t offset, int w, int h, int wrapS, int wrapT)
{
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if(i >= uint32_t(w) * uint32_t(h))
    return;
  const int x = int(i % uint32_t(w)), y = int(i / uint32_t(w));
  const int x1 = wrapS == MI_WRAP_CLAMP_TO_EDGE ? min(x + 1, w - 1) : (x + 1 == w ? 0 : x + 1);
  const int y1 = wrapT == MI_WRAP_CLAMP_TO_EDGE ? min(y + 1, h - 1) : (y + 1 == h ? 0 : y + 1);
  const uint32_t* L = texels + offset;
  quads[offset + i] = make_uint4(L[uint32_t(y) * uint32_t(w) + uint32_t(x)], L[uint32_t(y) * uint32_t(w) + uint32_t(x1)],
                                 L[uint32_t(y1) * uint32_t(w) + uint32_t(x)], L[uint32_t(y1) * uint32_t(w) + uint32_t(x1)]);
}

__global__ void k_reset_counters(uint32_t* counters)
{
  for(int i = threadIdx.x; i < QC_COUNT; i += blockDim.x)
    counters[i] = 0;
}

}  // namespace

//================================================================================================================================
// host-side launch helpers
//================================================================================================================================
void launchBuildAlphaRecords(const DevScene& scene, uint32_t numTris, DevAlphaTri* out, hipStream_t s)
{
  if(numTris)
    hipLaunchKernelGGL(k_alpha_records, dim3((numTris + 255) / 256), dim3(256), 0, s, scene, numTris, out);
}
void dumpTraceProfile()
{
#ifdef SHADE_PROFILE
  {
    unsigned long long h[20] = {};
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_shadeProf), sizeof(h));
    const double tot = h[18] ? double(h[18]) : 1.0;
    const char*  names[9] = {"entry + hit state", "material + textures", "sampleLights", "bsdfEvaluate", "bsdfSample", "(unused)", "miss (inline / listed)", "append + stores", "finishMisses"};
    fprintf(stderr, "[mi_pt shade profile] later-bounce k_shade: round ticks %.4g, lanes with an entry %.1f\n", tot, h[18] ? double(h[19]) / double(h[18]) : 0.0);
    for(int r = 0; r < 9; ++r)
      if(h[2 * r])
        fprintf(stderr, "[mi_pt shade profile]   %-24s %5.1f %% of the round's wave time, %5.1f lanes inside\n", names[r], 100.0 * double(h[2 * r]) / tot,
                r == 8 ? 0.0 : double(h[2 * r + 1]) / double(h[2 * r]));
  }
#endif
#ifdef TRACE_PROFILE
  unsigned long long h[20] = {};
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_traceProf), sizeof(h));
  const double tot = double(h[3]) > 0 ? double(h[3]) : 1.0;
  fprintf(stderr, "[mi_pt trace profile] feed split: start-ray (waits for the prefetch) %.1f%% take %.1f%% issue %.1f%%\n", 100.0 * h[5] / tot, 100.0 * h[6] / tot,
          100.0 * h[7] / tot);
  fprintf(stderr, "[mi_pt trace profile] waves %llu total ticks %.4g: feed %.1f%% node %.1f%% tri %.1f%% other %.1f%%\n", h[4], tot, 100.0 * h[0] / tot,
          100.0 * h[1] / tot, 100.0 * h[2] / tot, 100.0 * (tot - h[0] - h[1] - h[2]) / tot);
  fprintf(stderr, "[mi_pt trace profile] triangle rounds: test (permutes, intersection, alpha) %.1f%% gather %.1f%% of total; alpha rounds %.1f%% of total (%llu rounds)\n",
          100.0 * h[16] / tot, 100.0 * h[17] / tot, 100.0 * h[18] / tot, h[19]);
  unsigned long long g[16] = {};
  (void)hipMemcpyFromSymbol(g, HIP_SYMBOL(g_shadowProf), sizeof(g));
  const double st = double(g[5]) > 0 ? double(g[5]) : 1.0;
  const auto   pr = [](unsigned long long a, unsigned long long b) { return b ? double(a) / double(b) : 0.0; };
  fprintf(stderr, "[mi_pt shadow profile] waves %llu total ticks %.4g: feed %.1f%% node %.1f%% tri %.1f%% alpha %.1f%% finish %.1f%% other %.1f%%\n", g[6], st,
          100.0 * g[0] / st, 100.0 * g[1] / st, 100.0 * g[2] / st, 100.0 * g[3] / st, 100.0 * g[4] / st, 100.0 * (st - g[0] - g[1] - g[2] - g[3] - g[4]) / st);
  fprintf(stderr, "[mi_pt shadow profile] outer iterations %llu (%.1f idle lanes each), inner iterations %llu (%.1f active lanes, %.1f visiting), alpha rounds %llu (%.1f entries)\n",
          g[8], pr(g[9], g[8]), g[10], pr(g[11], g[10]), pr(g[12], g[10]), g[13], pr(g[14], g[13]));
  const auto per = [](unsigned long long a, unsigned long long b) { return b ? double(a) / double(b) : 0.0; };
  fprintf(stderr, "[mi_pt trace profile] lanes: %.1f of 64 visit per node step (%llu steps, %.1f blocked); %.1f test per triangle round (%llu rounds in %llu phases); "
                  "%.1f idle at each of %llu refills\n",
          per(h[9], h[8]), h[8], per(h[15], h[8]), per(h[11], h[10]), h[10], h[14], per(h[13], h[12]), h[12]);
#endif
}
void launchBuildShadeRecords(const DevScene& scene, uint32_t numTris, DevShadeTri* out, hipStream_t s)
{
  if(numTris)
    hipLaunchKernelGGL(k_shade_records, dim3((numTris + 255) / 256), dim3(256), 0, s, scene, numTris, out);
}
// DevScene::bvh8Planes: the 48 quantised plane bytes of every node as floats, block 2 * axis + side (pt_bvh8.h).  One thread per
// plane word of the node record (12 words: lower x, y, z then upper x, y, z, two words of four children each).
__global__ void __launch_bounds__(256) k_bvh8_planes(const uint4* __restrict__ nodes, uint32_t numNodes, float* __restrict__ planes)
{
  const uint32_t id = blockIdx.x * 256u + threadIdx.x;
  if(id >= numNodes * 12u)
    return;
  const uint32_t  node = id / 12u, word = id % 12u;
  const uint32_t* rec  = reinterpret_cast<const uint32_t*>(nodes + size_t(node) * 5u) + 8u;  // n2.x ...
  const uint32_t  w    = rec[word];
  const uint32_t  pair = word >> 1, axis = pair % 3u, side = pair / 3u;
  float4          v    = make_float4(float(w & 0xffu), float((w >> 8) & 0xffu), float((w >> 16) & 0xffu), float(w >> 24));
  reinterpret_cast<float4*>(planes + size_t(node) * 48u + (2u * axis + side) * 8u)[word & 1u] = v;
}
void launchBvh8Planes(const uint4* nodes, uint32_t numNodes, float* planes, hipStream_t s)
{
  if(numNodes)
    hipLaunchKernelGGL(k_bvh8_planes, dim3((numNodes * 12u + 255u) / 256u), dim3(256), 0, s, nodes, numNodes, planes);
}
void launchTextureQuads(const uchar4* texels, uint4* quads, uint32_t offset, int width, int height, int wrapS, int wrapT, hipStream_t s)
{
  const uint32_t n = uint32_t(width) * uint32_t(height);
  hipLaunchKernelGGL(k_texture_quads, dim3((n + 255u) / 256u), dim3(256), 0, s, reinterpret_cast<const uint32_t*>(texels), quads, offset, width, height, wrapS, wrapT);
}
void launchResetCounters(const Queues& Q, hipStream_t s)
{
  hipLaunchKernelGGL(k_reset_counters, dim3(1), dim3(64), 0, s, Q.counters);
}
// the per-parameter-set constants of the physical sky (FrameConsts::skyPre), with the device's own arithmetic
__global__ void k_sky_precomp(MiSkyPhysicalParameters sky, SkyPrecomp* out)
{
  if(blockIdx.x == 0 && threadIdx.x == 0)
    *out = makeSkyPrecomp(sky);
}
void launchSkyPrecomp(const MiSkyPhysicalParameters& sky, SkyPrecomp* out, hipStream_t stream)
{
  hipLaunchKernelGGL(k_sky_precomp, dim3(1), dim3(64), 0, stream, sky, out);
}

void launchGenerate(const LaunchCtx& c, int sampleIndex)
{
  unsigned grid = (unsigned(c.fc.numSlots) * unsigned(c.fc.numFrames) + 255u) / 256u;
  hipLaunchKernelGGL(k_generate, dim3(grid), dim3(256), 0, c.stream, c.scene, c.fc, c.paths, c.queues, c.ownedTiles, sampleIndex,
                     c.collectCounters ? c.stats : nullptr);
}
namespace {
template <bool WIDE>
void launchTraceClosestT(const LaunchCtx& c, int cur)
{
  dim3 grid(c.persistentBlocks * 256u / TRACE_BLOCK), block(TRACE_BLOCK);
  if(c.hasAlphaClosest)
  {
    if(c.collectCounters)
      hipLaunchKernelGGL((k_trace_closest<WIDE, true, true>), grid, block, 0, c.stream, c.scene, c.paths, c.queues, cur, c.stats);
    else
      hipLaunchKernelGGL((k_trace_closest<WIDE, true, false>), grid, block, 0, c.stream, c.scene, c.paths, c.queues, cur, c.stats);
  }
  else
  {
    if(c.collectCounters)
      hipLaunchKernelGGL((k_trace_closest<WIDE, false, true>), grid, block, 0, c.stream, c.scene, c.paths, c.queues, cur, c.stats);
    else
      hipLaunchKernelGGL((k_trace_closest<WIDE, false, false>), grid, block, 0, c.stream, c.scene, c.paths, c.queues, cur, c.stats);
  }
}
template <bool WIDE>
void launchTraceShadowT(const LaunchCtx& c, int nxt)
{
  const float darken = c.fc.frameInfo.shadowCatcherDarkenAmount;
  const bool  record = WIDE && c.hasAlpha && c.hasTransmissive && c.queues.candPool != nullptr;
  const int   mode   = !c.hasAlpha ? 0 : (c.hasTransmissive ? (record ? 3 : 2) : 1);
  const unsigned bs  = mode == 2 ? unsigned(ShadowCfg<2>::BLOCK) : unsigned(ShadowCfg<0>::BLOCK);
  dim3 grid(c.persistentBlocks * 256u / bs), block(bs);
#define MI_LAUNCH_SHADOW(M, C, OVF) \
  hipLaunchKernelGGL((k_trace_shadow<WIDE, M, C>), grid, block, 0, c.stream, c.scene, c.sceneDev, c.paths, c.queues, nxt, darken, c.stats, OVF)
  // the walk leaves every ray's outcome in its queue entry; k_shadow_resolve adds the contributions (and, after the recording
  // walk, evaluates the recorded transmissive candidates in order) as one streaming pass
  const dim3 rgrid(c.persistentBlocks), rblock(256);
  if(mode == 0)      { if(c.collectCounters) MI_LAUNCH_SHADOW(0, true, 0); else MI_LAUNCH_SHADOW(0, false, 0); }
  else if(mode == 1) { if(c.collectCounters) MI_LAUNCH_SHADOW(1, true, 0); else MI_LAUNCH_SHADOW(1, false, 0); }
  else if(mode == 2) { if(c.collectCounters) MI_LAUNCH_SHADOW(2, true, 0); else MI_LAUNCH_SHADOW(2, false, 0); }
  else
  {
    if constexpr(WIDE)
    {
      if(c.collectCounters) MI_LAUNCH_SHADOW(3, true, 0); else MI_LAUNCH_SHADOW(3, false, 0);
    }
  }
  if(mode == 3)
  {
    hipLaunchKernelGGL(k_shadow_resolve<true>, rgrid, rblock, 0, c.stream, c.sceneDev, c.paths, c.queues, nxt, darken);
    if constexpr(WIDE)
    {
      // normally on an empty list: the rays whose candidates did not fit the pool, by ordered search
      grid  = dim3(c.persistentBlocks * 256u / unsigned(ShadowCfg<2>::BLOCK));
      block = dim3(unsigned(ShadowCfg<2>::BLOCK));
      if(c.collectCounters) MI_LAUNCH_SHADOW(2, true, 1); else MI_LAUNCH_SHADOW(2, false, 1);
    }
  }
  else if(mode != 2)
    hipLaunchKernelGGL(k_shadow_resolve<false>, rgrid, rblock, 0, c.stream, c.sceneDev, c.paths, c.queues, nxt, darken);
#undef MI_LAUNCH_SHADOW
}
}  // namespace
void launchTracePrimary(const LaunchCtx& c, int sampleIndex)
{
  const uint32_t batchSlots = uint32_t(c.fc.numSlots) * uint32_t(c.fc.numFrames);
  dim3           grid(batchSlots / 256u), block(256);  // one workgroup per 256-slot chunk (numSlots is a multiple of 256)
  // packets of one pixel's samples (pixel-major slots) take the interval node test: its own instantiation
  const bool interval = c.fc.slotLayout == 1 && c.scene.packetInterval != 0 && c.scene.bvh8Planes != nullptr;
#define MI_LAUNCH_PRIMARY(A, C)                                                                                                                   \
  do                                                                                                                                              \
  {                                                                                                                                               \
    if(interval)                                                                                                                                  \
      hipLaunchKernelGGL((k_trace_primary<A, C, true>), grid, block, 0, c.stream, c.scene, c.fc, c.sceneDev, c.fcDev, c.paths, c.queues, c.ownedTiles, sampleIndex, batchSlots, c.stats);  \
    else                                                                                                                                          \
      hipLaunchKernelGGL((k_trace_primary<A, C, false>), grid, block, 0, c.stream, c.scene, c.fc, c.sceneDev, c.fcDev, c.paths, c.queues, c.ownedTiles, sampleIndex, batchSlots, c.stats); \
  } while(0)
  if(c.hasAlphaClosest)
  {
    if(c.collectCounters) MI_LAUNCH_PRIMARY(true, true); else MI_LAUNCH_PRIMARY(true, false);
  }
  else
  {
    if(c.collectCounters) MI_LAUNCH_PRIMARY(false, true); else MI_LAUNCH_PRIMARY(false, false);
  }
#undef MI_LAUNCH_PRIMARY
}
void launchTraceClosest(const LaunchCtx& c, int cur)
{
  if(c.wide)
    launchTraceClosestT<true>(c, cur);
  else
    launchTraceClosestT<false>(c, cur);
}
void launchShade(const LaunchCtx& c, int cur, bool first)
{
  dim3 grid(c.persistentBlocks), block(SHADE_BLOCK);
#define MI_LAUNCH_SHADE(C, S, F) \
  hipLaunchKernelGGL((k_shade<C, S, F>), grid, block, 0, c.stream, c.sceneDev, c.fcDev, c.paths, c.queues, cur, (S ? 0 : c.sortMode), c.stats)
#define MI_LAUNCH_SHADE_F(C, S) do { if(first) MI_LAUNCH_SHADE(C, S, true); else MI_LAUNCH_SHADE(C, S, false); } while(0)
  if(c.simpleMaterials)
  {
    if(c.collectCounters) MI_LAUNCH_SHADE_F(true, true); else MI_LAUNCH_SHADE_F(false, true);
  }
  else
  {
    if(c.collectCounters) MI_LAUNCH_SHADE_F(true, false); else MI_LAUNCH_SHADE_F(false, false);
  }
#undef MI_LAUNCH_SHADE_F
#undef MI_LAUNCH_SHADE
}
void launchTraceShadow(const LaunchCtx& c, int nxt)
{
  if(c.wide)
    launchTraceShadowT<true>(c, nxt);
  else
    launchTraceShadowT<false>(c, nxt);
}
void launchFlushSurvivors(const LaunchCtx& c, int cur)
{
  hipLaunchKernelGGL(k_flush_survivors, dim3(c.persistentBlocks), dim3(256), 0, c.stream, c.paths, c.queues, cur);
}
void launchFinishSample(const LaunchCtx& c, int sampleIndex, float4* accum, float* depth, float4* albedo, float4* normal)
{
  unsigned grid = (unsigned(c.fc.numSlots) + 255u) / 256u;
  hipLaunchKernelGGL(k_finish_sample, dim3(grid), dim3(256), 0, c.stream, c.fc, c.sceneDev, c.fcDev, c.paths, c.ownedTiles, sampleIndex, accum, depth, albedo, normal);
}
void launchSelection(const LaunchCtx& c, uint32_t* selection)
{
  unsigned grid = (unsigned(c.fc.numSlots) + SEL_BLOCK - 1) / SEL_BLOCK;
  if(c.wide)
    hipLaunchKernelGGL(k_selection<true>, dim3(grid), dim3(SEL_BLOCK), 0, c.stream, c.scene, c.fc, c.ownedTiles, selection);
  else
    hipLaunchKernelGGL(k_selection<false>, dim3(grid), dim3(SEL_BLOCK), 0, c.stream, c.scene, c.fc, c.ownedTiles, selection);
}

}  // namespace pt
