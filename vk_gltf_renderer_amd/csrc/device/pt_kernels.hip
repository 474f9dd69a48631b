// Wavefront path-trace kernels for gfx950.  One frame = numSamples x { generate -> [trace -> shade -> shadow]* -> finish }.
// The megakernel of the reference (shaders/gltf_pathtrace.slang:87-671, one thread per pixel running the whole bounce
// loop around hardware ray queries) is split at its two Trace calls so that traversal (memory/latency bound, small
// register footprint, LDS stack) and shading (ALU bound, large live state) run as separate persistent grids over
// compacted queues of path slots.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include "pt_kernels.h"
#include "pt_shading.h"
#include "pt_bvh.h"

namespace pt {

namespace {

constexpr int TRACE_BLOCK = 256;
constexpr int SHADE_BLOCK = 256;

// ---- wave-aggregated queue append ------------------------------------------------------------------------------------------
PT_DEV uint32_t laneId() { return __lane_id(); }
PT_DEV void queuePush(bool pred, uint32_t* queue, uint32_t* counter, uint32_t value)
{
  unsigned long long mask = __ballot(pred);
  if(mask == 0ull)
    return;
  uint32_t lane   = laneId();
  uint32_t leader = uint32_t(__ffsll((long long)mask) - 1);
  uint32_t base   = 0;
  if(lane == leader)
    base = atomicAdd(counter, uint32_t(__popcll(mask)));
  base = uint32_t(__shfl(int(base), int(leader)));
  if(pred)
  {
    uint32_t rank = uint32_t(__popcll(mask & ((1ull << lane) - 1ull)));
    queue[base + rank] = value;
  }
}

// ---- slot <-> pixel -----------------------------------------------------------------------------------------------------------
// Slots are tile-major; inside a tile 8x8 micro-tiles, so one 64-lane wave owns one 8x8 pixel block (coherent camera rays).
PT_DEV bool slotToPixel(const FrameConsts& fc, const uint32_t* ownedTiles, uint32_t slot, int& px, int& py)
{
  const uint32_t T2    = uint32_t(fc.tileSize * fc.tileSize);
  uint32_t       tile  = ownedTiles[slot / T2];
  uint32_t       w     = slot % T2;
  uint32_t       micro = w >> 6, lane = w & 63u;
  uint32_t       mpr   = uint32_t(fc.tileSize) >> 3;
  px                   = int((tile % uint32_t(fc.tilesX)) * uint32_t(fc.tileSize) + (micro % mpr) * 8u + (lane & 7u));
  py                   = int((tile / uint32_t(fc.tilesX)) * uint32_t(fc.tileSize) + (micro / mpr) * 8u + (lane >> 3));
  return px < fc.width && py < fc.height;
}

// ---- camera (pathtrace_functions.h.slang:784-811, gltf_pathtrace.slang:502-529) ------------------------------------------
PT_DEV void getRay(const FrameConsts& fc, f2 samplePos, f2 offset, f3& origin, f3& direction)
{
  const MiSceneFrameInfo& fi = fc.frameInfo;
  f2 clip = mk2((samplePos.x + offset.x) / float(fc.width) * 2.0f - 1.0f, (samplePos.y + offset.y) / float(fc.height) * 2.0f - 1.0f);
  f4 view = mulFull(fi.projInv, mk4(clip.x, clip.y, -1.0f, 1.0f));
  view    = view / view.w;
  if(hasFlag(fi.flags, MI_SCENE_IS_ORTHOGRAPHIC))
  {
    origin    = xyz(mulFull(fi.viewInv, view));
    direction = normalize(xyz(mulFull(fi.viewInv, mk4(0, 0, -1, 0))));
  }
  else
  {
    origin    = mk3(fi.viewInv[12], fi.viewInv[13], fi.viewInv[14]);
    direction = normalize(xyz(mulFull(fi.viewInv, view)) - origin);
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
// k_generate: seed, AA jitter, camera ray, thin-lens DoF, path-state reset  (gltf_pathtrace.slang:546-596, 502-529)
//================================================================================================================================
__global__ void __launch_bounds__(256) k_generate(DevScene sc, FrameConsts fc, PathSoA P, Queues Q, const uint32_t* ownedTiles, int sampleIndex,
                                                   StatCounters* stats)
{
  uint32_t slot  = blockIdx.x * blockDim.x + threadIdx.x;
  bool     valid = slot < uint32_t(fc.numSlots);
  int      px = 0, py = 0;
  if(valid)
    valid = slotToPixel(fc, ownedTiles, slot, px, py);
  if(valid)
  {
    uint32_t seed;
    f2       jitter;
    if(sampleIndex == 0)
    {
      seed     = xxhash32(uint32_t(px), uint32_t(py), uint32_t(fc.pc.frameCount));
      float u1 = rnd(seed), u2 = rnd(seed);
      // sampleGaussian (Box-Muller), pathtrace_functions.h.slang:784-789
      float r     = sqrtf(-2.0f * logf(fmaxf(1e-38f, u1)));
      float theta = 2.0f * K_PI * u2;
      jitter      = mk2(0.5f + ANTIALIASING_STANDARD_DEVIATION * (r * cosf(theta)), 0.5f + ANTIALIASING_STANDARD_DEVIATION * (r * sinf(theta)));
      P.pixelSum[slot] = make_float4(0, 0, 0, 0);
      if(P.guideAlbedo)
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
    f3 origin, direction;
    getRay(fc, mk2(float(px), float(py)), jitter, origin, direction);
    if(!hasFlag(fc.frameInfo.flags, MI_SCENE_IS_ORTHOGRAPHIC))
    {
      const float* V          = fc.frameInfo.viewInv;
      f3           focalPoint = direction * fc.pc.focalDistance;
      float        cam_r1     = rnd(seed) * K_TWO_PI;
      float        cam_r2     = rnd(seed) * fc.pc.aperture;
      f3           cam_right  = mk3(V[0], V[4], V[8]);  // Slang mul(viewMatrixI, float4(1,0,0,0)) = M^T e0
      f3           cam_up     = mk3(V[1], V[5], V[9]);
      f3           aperturePos = (cam_right * cosf(cam_r1) + cam_up * sinf(cam_r1)) * sqrtf(cam_r2);
      f3           finalDir    = normalize(focalPoint - aperturePos);
      origin += aperturePos;
      direction = finalDir;
    }
    direction          = normalize(direction);  // pathTrace loop head, gltf_pathtrace.slang:447
    P.rayOrg[slot]     = make_float4(origin.x, origin.y, origin.z, INFINITE_F);
    P.rayDir[slot]     = make_float4(direction.x, direction.y, direction.z, 0.0f);  // cone.width = 0
    P.throughput[slot] = make_float4(1.0f, 1.0f, 1.0f, DIRAC);
    P.radiance[slot]   = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    P.misc[slot]       = make_float4(0.0f, __uint_as_float(PF_ALIVE), __uint_as_float(seed), 0.0f);
    P.medium[slot]     = make_uint4(0, 0, 0, 0);
    P.firstHit[slot]   = make_float4(1e34f, 1e34f, 1e34f, 0.0f);
    if(stats)
      atomicAdd(&stats->cameraPaths, 1ull);
  }
  queuePush(valid, Q.active[0], &Q.counters[QC_ACTIVE0], slot);
}

//================================================================================================================================
// k_trace_closest: RayQueryRaytracer::Trace (raytracer_interface.h.slang:69-122) on the software BVH
//================================================================================================================================
template <bool HAS_ALPHA, bool COUNT>
__global__ void __launch_bounds__(TRACE_BLOCK) k_trace_closest(DevScene sc, PathSoA P, Queues Q, int cur, StatCounters* stats)
{
  __shared__ int s_stack[BVH_STACK_LDS * TRACE_BLOCK];
  if(blockIdx.x == 0 && threadIdx.x == 0)
  {
    // the shade kernel of this iteration appends to these; zero them here (kernel boundary orders the write)
    Q.counters[cur ^ 1]    = 0;
    Q.counters[QC_SHADOW]  = 0;
  }
  const uint32_t count = Q.counters[cur];
  LaneStack      st;
  st.lds    = s_stack;
  st.tid    = int(threadIdx.x);
  st.stride = TRACE_BLOCK;
  for(uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x)
  {
    const uint32_t slot = Q.active[cur][i];
    const float4   o4 = P.rayOrg[slot], d4 = P.rayDir[slot];
    const RaySetup r     = makeRaySetup(xyz(o4), xyz(d4));
    float          bestT = INFINITE_F;  // ray.TMax
    int            bestTri = -1;
    float          bestU = 0.0f, bestV = 0.0f;
    uint32_t       bestRnode = 0xffffffffu, bestPrim = 0xffffffffu;
    uint32_t       seed0 = 0;
    bool           seedLoaded = false;
    unsigned       nodes = 0, tris = 0;
    bvhWalk(sc, r, bestT, st, [&](int triIndex, float tmax) -> float {
      if(COUNT) ++tris;
      const DevTri T = sc.tris[triIndex];
      TriHit       h;
      if(!intersectTri(xyz(T.a), xyz(T.b), xyz(T.c), r.org, r.dir, h) || !(h.t > 0.0f))
        return tmax;
      const uint32_t rnode = __float_as_uint(T.a.w), prim = __float_as_uint(T.b.w), flags = __float_as_uint(T.c.w);
      // deterministic closest hit: smaller t wins, exact ties by (renderNode, primitive)
      if(!(h.t < bestT || (h.t == bestT && (rnode < bestRnode || (rnode == bestRnode && prim < bestPrim)))))
        return tmax;
      // RAY_FLAG_CULL_BACK_FACING_TRIANGLES unless TRIANGLE_FACING_CULL_DISABLE; facing is decided in object space
      const bool front = h.front != ((flags & INST_FLIP_FACING) != 0u);
      if(!front && !(flags & INST_CULL_DISABLE))
        return tmax;
      if(HAS_ALPHA && !(flags & INST_FORCE_OPAQUE))
      {
        if(!seedLoaded)
        {
          seed0      = __float_as_uint(P.misc[slot].z);
          seedLoaded = true;
        }
        float opacity = getOpacity(sc, int(rnode), int(prim), mk3(1.0f - h.u - h.v, h.u, h.v));
        if(!(candidateRand(seed0, int(rnode), int(prim)) <= opacity))
          return tmax;
      }
      bestT = h.t; bestTri = triIndex; bestU = h.u; bestV = h.v; bestRnode = rnode; bestPrim = prim;
      return bestT;
    }, nodes);
    P.hit[slot] = make_float4(bestT, __int_as_float(bestTri), bestU, bestV);
    if(COUNT)
    {
      atomicAdd(&stats->segments, 1ull);
      atomicAdd(&stats->nodesClosest, (unsigned long long)nodes);
      atomicAdd(&stats->trisClosest, (unsigned long long)tris);
    }
  }
}

//================================================================================================================================
// k_selection: traceSelectionRay / TraceLow (pathtrace_functions.h.slang:813-820, raytracer_interface.h.slang:124-137)
//================================================================================================================================
__global__ void __launch_bounds__(TRACE_BLOCK) k_selection(DevScene sc, FrameConsts fc, const uint32_t* ownedTiles, uint32_t* selection)
{
  __shared__ int s_stack[BVH_STACK_LDS * TRACE_BLOCK];
  uint32_t       slot = blockIdx.x * blockDim.x + threadIdx.x;
  int            px, py;
  if(slot >= uint32_t(fc.numSlots) || !slotToPixel(fc, ownedTiles, slot, px, py))
    return;
  LaneStack st;
  st.lds    = s_stack;
  st.tid    = int(threadIdx.x);
  st.stride = TRACE_BLOCK;
  f3 origin, direction;
  getRay(fc, mk2(float(px), float(py)), mk2(0.5f, 0.5f), origin, direction);
  const RaySetup r     = makeRaySetup(origin, direction);
  float          bestT = INFINITE_F;
  uint32_t       bestRnode = 0xffffffffu, bestPrim = 0xffffffffu;
  unsigned       nodes = 0;
  bvhWalk(sc, r, bestT, st, [&](int triIndex, float tmax) -> float {
    const DevTri T = sc.tris[triIndex];
    TriHit       h;
    if(!intersectTri(xyz(T.a), xyz(T.b), xyz(T.c), r.org, r.dir, h) || !(h.t > 0.0f))
      return tmax;
    const uint32_t rnode = __float_as_uint(T.a.w), prim = __float_as_uint(T.b.w);
    if(!(h.t < bestT || (h.t == bestT && (rnode < bestRnode || (rnode == bestRnode && prim < bestPrim)))))
      return tmax;
    bestT = h.t; bestRnode = rnode; bestPrim = prim;
    return bestT;
  }, nodes);
  selection[size_t(py) * size_t(fc.width) + size_t(px)] = (bestRnode != 0xffffffffu) ? bestRnode + 1u : 0u;
}

//================================================================================================================================
// k_shade: everything of pathTraceOneBounce / pathTrace between the two Trace calls (gltf_pathtrace.slang:104-430, 441-494)
//================================================================================================================================
template <bool COUNT>
__global__ void __launch_bounds__(SHADE_BLOCK) k_shade(DevScene sc, FrameConsts fc, PathSoA P, Queues Q, int cur, StatCounters* stats)
{
  const uint32_t count = Q.counters[cur];
  const int      nxt   = cur ^ 1;
  const uint32_t iters = (count + gridDim.x * blockDim.x - 1) / (gridDim.x * blockDim.x);
  for(uint32_t it = 0; it < iters; ++it)
  {
    const uint32_t i      = it * gridDim.x * blockDim.x + blockIdx.x * blockDim.x + threadIdx.x;
    const bool     inRange = i < count;
    uint32_t       slot = 0;
    bool           alive = false, pushShadow = false;
    unsigned       taps = 0;
    if(inRange)
    {
      slot = Q.active[cur][i];
      const float4 hit4 = P.hit[slot], o4 = P.rayOrg[slot], d4 = P.rayDir[slot], tp4 = P.throughput[slot], rad4 = P.radiance[slot], misc4 = P.misc[slot];
      f3       rayOrigin = xyz(o4), rayDir = xyz(d4);
      float    coneWidth = d4.w;
      f3       throughput = xyz(tp4), radiance = xyz(rad4);
      float    lastSamplePdf = tp4.w;
      f2       maxRoughness  = mk2(rad4.w, misc4.x);
      uint32_t flags = __float_as_uint(misc4.y), seed = __float_as_uint(misc4.z);
      int      surfaceDepth   = int((flags >> PF_DEPTH_SHIFT) & 0xffu);
      int      scatterBounces = int((flags >> PF_SCATTER_SHIFT) & 0xffu);
      bool     isInside = (flags & PF_INSIDE) != 0u, solid = !(flags & PF_NOT_SOLID);
      const bool firstRay = (surfaceDepth == 0);
      const int  maxDepth = fc.pc.maxDepth;

      float hitT   = hit4.x;
      int   triIdx = __float_as_int(hit4.y);
      bool  done   = false;  // eBreak
      bool  earlyContinue = false;

      HitState hit;
      int      rnodeID = -1, primitiveID = -1, materialID = 0;
      const bool meshHit = triIdx >= 0;
      if(meshHit)
      {
        const DevTri T = sc.tris[triIdx];
        rnodeID        = int(__float_as_uint(T.a.w));
        primitiveID    = int(__float_as_uint(T.b.w));
        const MiGltfRenderNode& rn = sc.nodes[rnodeID];
        materialID                 = max(0, rn.materialID);
        const DevPrim rp           = sc.prims[rn.renderPrimID];
        hit = getHitState(rp, mk3(1.0f - hit4.z - hit4.w, hit4.z, hit4.w), rn.worldToObject, rn.objectToWorld, primitiveID, rayDir);
      }
      else
        hitT = INFINITE_F;

      // checkInfinitePlaneIntersection, pathtrace_functions.h.slang:556-585
      bool hitInfinitePlane = false;
      if(hasFlag(fc.frameInfo.flags, MI_SCENE_USE_INFINITE_PLANE))
      {
        float planeHeight = fc.frameInfo.infinitePlaneDistance;
        float Dn          = rayDir.y;
        if(rayOrigin.y > planeHeight && fabsf(Dn) > 1e-6f)
        {
          float t = (-rayOrigin.y + planeHeight) / Dn;
          if(t > 0.0f && t < hitT)
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
      }

      if(hitT == INFINITE_F)  // gltf_pathtrace.slang:129-156
      {
        bool backplate = false;
        if(firstRay)  // tryPrimaryMissBackplate, pathtrace_functions.h.slang:944-971
        {
          solid             = false;
          P.firstHit[slot]  = make_float4(rayDir.x, rayDir.y, rayDir.z, 0.0f);
          if(hasFlag(fc.frameInfo.flags, MI_SCENE_USE_SOLID_BACKGROUND))
          {
            radiance  = mk3(fc.frameInfo.backgroundColor);
            backplate = true;
          }
          else if(hasFlag(fc.frameInfo.flags, MI_SCENE_USE_HDR_ENVIRONMENT) && fc.frameInfo.envBlur > 0.0f)
          {
            f3 dir    = rotateAxis(rayDir, mk3(0, 1, 0), -fc.frameInfo.envRotation);
            radiance  = smoothHDRBlur(sc, getSphericalUv(dir), fc.frameInfo.envBlur) * fc.frameInfo.envIntensity;
            backplate = true;
          }
        }
        if(!backplate)
        {
          f3    envColor;
          float envPdf;
          sampleEnvironment(sc, fc, rayDir, envColor, envPdf);
          float mis = computeEnvHitMisWeight(sc, fc, lastSamplePdf, envPdf);
          radiance += throughput * mis * envColor;
        }
        done = true;
      }

      if(!done)
      {
        PbrMaterial pbrMat;
        // rayConeWorldFootprint, pathtrace_functions.h.slang:174-178
        float worldFoot = (coneWidth + fc.pc.pixelAngle * hitT) / fmaxf(fabsf(dot(hit.geonrm, -rayDir)), 1e-3f);
        bool  unlit     = false;
        if(hitInfinitePlane)
        {
          pbrMat           = defaultPbrMaterial();
          pbrMat.baseColor = mk3(fc.frameInfo.infinitePlaneBaseColor);
          pbrMat.metallic  = fc.frameInfo.infinitePlaneMetallic;
          float r          = fc.frameInfo.infinitePlaneRoughness;
          pbrMat.roughness = mk2(r * r, r * r);
          pbrMat.N = hit.nrm; pbrMat.Ng = hit.nrm; pbrMat.Nc = hit.nrm;
          pbrMat.T = hit.tangent; pbrMat.B = hit.bitangent;
          // the shadow-catcher variant needs the shadow result inside the bounce; not supported by this wavefront split yet
        }
        else
        {
          const MiGltfShadeMaterial& mat = sc.materials[materialID];
          MeshState                  mesh;
          mesh.N = hit.nrm; mesh.T = hit.tangent; mesh.B = hit.bitangent; mesh.Ng = hit.geonrm;
          mesh.tc0 = hit.uv0; mesh.tc1 = hit.uv1;
          mesh.isInside           = isInside;
          mesh.texGrad            = worldFoot * hit.texelDensity * fc.pc.texGradScale;
          mesh.baseColorVertexMul = hit.color;
          pbrMat                  = evaluateMaterial(sc, mat, mesh, taps);
          unlit                   = mat.unlit > 0;
        }
        if(firstRay)  // gltf_pathtrace.slang:228-264
        {
          P.firstHit[slot] = make_float4(hit.pos.x, hit.pos.y, hit.pos.z, 0.0f);
          if(P.guideAlbedo)
          {
            float4 ga = P.guideAlbedo[slot], gn = P.guideNormal[slot];
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
        if(!done && isInside)
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
                P.shadowOrg[slot]     = make_float4(rayOrigin.x, rayOrigin.y, rayOrigin.z, dl.distance);
                P.shadowDir[slot]     = make_float4(dl.direction.x, dl.direction.y, dl.direction.z, __uint_as_float(1u));
                P.shadowContrib[slot] = make_float4(contrib.x, contrib.y, contrib.z, __uint_as_float(seed));
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
          sampleLights(sc, fc, hit.pos, seed, dl);  // :319-320
          bool nextEventValid = (dot(dl.direction, hit.nrm) > 0.0f || pbrMat.diffuseTransmissionFactor > 0.0f) && dl.pdf != 0.0f;
          f3   contribution   = mk3(0.0f);
          if(nextEventValid)  // :330-351
          {
            float    r1 = rnd(seed), r2 = rnd(seed), r3 = rnd(seed);
            BsdfEval ev = bsdfEvaluate(-rayDir, dl.direction, mk3(r1, r2, r3), pbrMat);
            if(ev.pdf > 0.0f)
            {
              float mis    = (dl.pdf == DIRAC) ? 1.0f : dl.pdf / (dl.pdf + ev.pdf);
              contribution = throughput * dl.radianceOverPdf * mis * ev.bsdf;
            }
          }
          {  // :357-416
            float      r1 = rnd(seed), r2 = rnd(seed), r3 = rnd(seed);
            BsdfSample sd = bsdfSample(-rayDir, mk3(r1, r2, r3), pbrMat);
            throughput *= sd.bsdf_over_pdf;
            rayDir        = sd.k2;
            lastSamplePdf = sd.pdf;
            if(sd.event_type != BSDF_EVENT_ABSORB)
            {
              f3 offsetDir = dot(rayDir, hit.geonrm) > 0.0f ? hit.geonrm : -hit.geonrm;
              rayOrigin    = safeOffsetRay(hit.pos, offsetDir);
              if(sd.event_type & BSDF_EVENT_TRANSMISSION)
              {
                isInside = !isInside;
                if(isInside)  // makeVolumeMedium, pathtrace_functions.h.slang:125-132
                  P.medium[slot] = packMedium(volumeExtinctionCoefficient(pbrMat), pbrMat.scatterCoefficient, pbrMat.scatterAnisotropy);
              }
            }
            else
              surfaceDepth = maxDepth;
          }
          if(nextEventValid)  // :421-426 + the TraceShadow of pathTrace :462-471, deferred to k_trace_shadow
          {
            bool forward = dot(dl.direction, hit.nrm) > 0.0f;
            f3   sOrg    = safeOffsetRay(forward ? hit.shadowPos : hit.pos, forward ? hit.geonrm : -hit.geonrm);
            P.shadowOrg[slot]     = make_float4(sOrg.x, sOrg.y, sOrg.z, dl.distance);
            P.shadowDir[slot]     = make_float4(dl.direction.x, dl.direction.y, dl.direction.z, __uint_as_float(0u));
            P.shadowContrib[slot] = make_float4(contribution.x, contribution.y, contribution.z, __uint_as_float(seed));
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
        (void)earlyContinue;
      }

      alive = !done && surfaceDepth < maxDepth;
      flags = (isInside ? PF_INSIDE : 0u) | (solid ? 0u : PF_NOT_SOLID) | (alive ? PF_ALIVE : 0u) | (uint32_t(min(surfaceDepth, 255)) << PF_DEPTH_SHIFT)
              | (uint32_t(scatterBounces) << PF_SCATTER_SHIFT);
      P.radiance[slot] = make_float4(radiance.x, radiance.y, radiance.z, maxRoughness.x);
      P.misc[slot]     = make_float4(maxRoughness.y, __uint_as_float(flags), __uint_as_float(seed), 0.0f);
      if(alive)
      {
        P.rayOrg[slot]     = make_float4(rayOrigin.x, rayOrigin.y, rayOrigin.z, INFINITE_F);
        P.rayDir[slot]     = make_float4(rayDir.x, rayDir.y, rayDir.z, coneWidth);
        P.throughput[slot] = make_float4(throughput.x, throughput.y, throughput.z, lastSamplePdf);
      }
      if(COUNT && taps)
        atomicAdd(&stats->textureTaps, (unsigned long long)taps);
    }
    queuePush(alive, Q.active[nxt], &Q.counters[nxt], slot);
    queuePush(pushShadow, Q.shadow, &Q.counters[QC_SHADOW], slot);
  }
}

//================================================================================================================================
// k_trace_shadow: RayQueryRaytracer::TraceShadow (raytracer_interface.h.slang:139-187) + `pt.radiance += contribution * T`
//================================================================================================================================
template <bool HAS_ALPHA, bool COUNT>
__global__ void __launch_bounds__(TRACE_BLOCK) k_trace_shadow(DevScene sc, PathSoA P, Queues Q, StatCounters* stats)
{
  __shared__ int s_stack[BVH_STACK_LDS * TRACE_BLOCK];
  const uint32_t count = Q.counters[QC_SHADOW];
  LaneStack      st;
  st.lds    = s_stack;
  st.tid    = int(threadIdx.x);
  st.stride = TRACE_BLOCK;
  for(uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x)
  {
    const uint32_t slot = Q.shadow[i];
    const float4   o4 = P.shadowOrg[slot], d4 = P.shadowDir[slot], c4 = P.shadowContrib[slot];
    const RaySetup r    = makeRaySetup(xyz(o4), xyz(d4));
    const float    tMax = o4.w;
    unsigned       nodes = 0, tris = 0;
    bool           occluded = false;
    unsigned       nonOpaque = 0;
    // pass 1: any opaque-instance triangle in (0, tMax) terminates (RAY_FLAG_NONE: no culling)
    bvhWalk(sc, r, tMax, st, [&](int triIndex, float tmax) -> float {
      if(COUNT) ++tris;
      const DevTri T = sc.tris[triIndex];
      TriHit       h;
      if(!intersectTri(xyz(T.a), xyz(T.b), xyz(T.c), r.org, r.dir, h) || !(h.t > 0.0f) || !(h.t < tMax))
        return tmax;
      if(!HAS_ALPHA || (__float_as_uint(T.c.w) & INST_FORCE_OPAQUE))
      {
        occluded = true;
        return -1.0f;
      }
      ++nonOpaque;
      return tmax;
    }, nodes);
    f3 total = occluded ? mk3(0.0f) : mk3(1.0f);
    if(HAS_ALPHA && !occluded && nonOpaque > 0)
    {
      // pass 2: non-opaque candidates in increasing (t, renderNode, primitive) order, one walk per candidate
      const uint32_t seed0    = __float_as_uint(c4.w);
      bool           isInside = (__float_as_uint(d4.w) & 1u) != 0u;
      float          prevHitT = 0.0f;
      float          lastT = -1.0f;
      uint32_t       lastRnode = 0, lastPrim = 0;
      bool           haveLast = false;
      for(unsigned n = 0; n < nonOpaque; ++n)
      {
        float    bT = tMax;
        uint32_t bRnode = 0xffffffffu, bPrim = 0xffffffffu;
        float    bU = 0.0f, bV = 0.0f;
        bool     found = false;
        bvhWalk(sc, r, tMax, st, [&](int triIndex, float tmax) -> float {
          if(COUNT) ++tris;
          const DevTri T = sc.tris[triIndex];
          if(__float_as_uint(T.c.w) & INST_FORCE_OPAQUE)
            return tmax;
          TriHit h;
          if(!intersectTri(xyz(T.a), xyz(T.b), xyz(T.c), r.org, r.dir, h) || !(h.t > 0.0f) || !(h.t < tMax))
            return tmax;
          const uint32_t rnode = __float_as_uint(T.a.w), prim = __float_as_uint(T.b.w);
          if(haveLast && !(h.t > lastT || (h.t == lastT && (rnode > lastRnode || (rnode == lastRnode && prim > lastPrim)))))
            return tmax;  // already processed
          if(found && !(h.t < bT || (h.t == bT && (rnode < bRnode || (rnode == bRnode && prim < bPrim)))))
            return tmax;
          found = true; bT = h.t; bRnode = rnode; bPrim = prim; bU = h.u; bV = h.v;
          return bT;
        }, nodes);
        if(!found)
          break;
        haveLast = true; lastT = bT; lastRnode = bRnode; lastPrim = bPrim;
        f3    bary    = mk3(1.0f - bU - bV, bU, bV);
        float opacity = getOpacity(sc, int(bRnode), int(bPrim), bary);
        if(candidateRand(seed0, int(bRnode), int(bPrim)) < opacity)
        {
          float segment = fmaxf(0.0f, bT - prevHitT);
          f3    cur     = getShadowTransmission(sc, int(bRnode), int(bPrim), bary, segment, r.dir, isInside);
          prevHitT      = bT;
          total *= cur;
          if(maxComp(total) <= MIN_TRANSMISSION)
          {
            total = mk3(0.0f);
            break;
          }
        }
      }
    }
    if(total.x != 0.0f || total.y != 0.0f || total.z != 0.0f)
    {
      float4 rad = P.radiance[slot];
      rad.x += c4.x * total.x;
      rad.y += c4.y * total.y;
      rad.z += c4.z * total.z;
      P.radiance[slot] = rad;
    }
    if(COUNT)
    {
      atomicAdd(&stats->shadowRays, 1ull);
      atomicAdd(&stats->nodesShadow, (unsigned long long)nodes);
      atomicAdd(&stats->trisShadow, (unsigned long long)tris);
    }
  }
}

//================================================================================================================================
// k_finish_sample: firefly clamp + per-frame mean + running-mean accumulation + NDC depth
// (gltf_pathtrace.slang:531-538, 596, 604-630)
//================================================================================================================================
__global__ void __launch_bounds__(256) k_finish_sample(FrameConsts fc, PathSoA P, const uint32_t* ownedTiles, int sampleIndex, float4* accum, float* depth,
                                                       float4* albedoOut, float4* normalOut)
{
  uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
  int      px, py;
  if(slot >= uint32_t(fc.numSlots) || !slotToPixel(fc, ownedTiles, slot, px, py))
    return;
  const float4   rad4  = P.radiance[slot];
  const uint32_t flags = __float_as_uint(P.misc[slot].y);
  const bool     solid = !(flags & PF_NOT_SOLID);
  f4             r     = mk4(rad4.x, rad4.y, rad4.z, solid ? 1.0f : 0.0f);
  float          lum   = dot(xyz(r), mk3(1.0f / 3.0f));
  if(lum > fc.pc.fireflyClampThreshold)
    r *= fc.pc.fireflyClampThreshold / lum;
  float4 sum = P.pixelSum[slot];
  sum        = make_float4(sum.x + r.x, sum.y + r.y, sum.z + r.z, sum.w + r.w);
  if(sampleIndex + 1 < fc.pc.numSamples)
  {
    P.pixelSum[slot] = sum;
    return;
  }
  const float  n     = float(fc.pc.numSamples);
  const f4     pixel = mk4(sum.x, sum.y, sum.z, sum.w) / n;
  const size_t idx   = size_t(py) * size_t(fc.width) + size_t(px);
  const bool   firstFrame = hasFlag(fc.pc.flags, MI_PT_FIRST_FRAME);
  if(firstFrame)
  {
    const bool hasSolidHit = r.w > 0.0f;
    float      ndcDepth    = 1.0f;
    if(hasSolidHit)
    {
      float4 fh   = P.firstHit[slot];
      f4     clip = mulFull(fc.frameInfo.viewProjMatrix, mk4(fh.x, fh.y, fh.z, 1.0f));
      ndcDepth    = clip.z / clip.w;
    }
    depth[idx] = ndcDepth;
    accum[idx] = make_float4(pixel.x, pixel.y, pixel.z, pixel.w);
  }
  else
  {
    const float  tot = float(fc.pc.totalSamples), after = float(fc.pc.totalSamples + fc.pc.numSamples);
    const float4 old = accum[idx];
    accum[idx] = make_float4((old.x * tot + pixel.x * n) / after, (old.y * tot + pixel.y * n) / after, (old.z * tot + pixel.z * n) / after,
                             (old.w * tot + pixel.w * n) / after);
  }
  if(P.guideAlbedo && albedoOut)
  {
    const float4 ga = P.guideAlbedo[slot], gn = P.guideNormal[slot];
    const float4 a  = make_float4(ga.x / n, ga.y / n, ga.z / n, r.w > 0.0f ? 1.0f : 0.0f);
    const float4 nn = make_float4(gn.x / n, gn.y / n, gn.z / n, 0.0f);
    if(firstFrame)
    {
      albedoOut[idx] = a;
      normalOut[idx] = nn;
    }
    else
    {
      const float  tot = float(fc.pc.totalSamples), after = float(fc.pc.totalSamples + fc.pc.numSamples);
      const float  wOld = tot / after, wNew = n / after;
      const float4 oa = albedoOut[idx], on = normalOut[idx];
      albedoOut[idx] = make_float4(oa.x * wOld + a.x * wNew, oa.y * wOld + a.y * wNew, oa.z * wOld + a.z * wNew, oa.w * wOld + a.w * wNew);
      normalOut[idx] = make_float4(on.x * wOld + nn.x * wNew, on.y * wOld + nn.y * wNew, on.z * wOld + nn.z * wNew, 0.0f);
    }
  }
}

__global__ void k_reset_counters(uint32_t* counters)
{
  if(threadIdx.x < QC_COUNT)
    counters[threadIdx.x] = 0;
}

}  // namespace

//================================================================================================================================
// host-side launch helpers
//================================================================================================================================
void launchResetCounters(const Queues& Q, hipStream_t s)
{
  hipLaunchKernelGGL(k_reset_counters, dim3(1), dim3(64), 0, s, Q.counters);
}
void launchGenerate(const LaunchCtx& c, int sampleIndex)
{
  unsigned grid = (unsigned(c.fc.numSlots) + 255u) / 256u;
  hipLaunchKernelGGL(k_generate, dim3(grid), dim3(256), 0, c.stream, c.scene, c.fc, c.paths, c.queues, c.ownedTiles, sampleIndex,
                     c.collectCounters ? c.stats : nullptr);
}
void launchTraceClosest(const LaunchCtx& c, int cur)
{
  dim3 grid(c.persistentBlocks), block(TRACE_BLOCK);
  if(c.hasAlpha)
  {
    if(c.collectCounters)
      hipLaunchKernelGGL((k_trace_closest<true, true>), grid, block, 0, c.stream, c.scene, c.paths, c.queues, cur, c.stats);
    else
      hipLaunchKernelGGL((k_trace_closest<true, false>), grid, block, 0, c.stream, c.scene, c.paths, c.queues, cur, c.stats);
  }
  else
  {
    if(c.collectCounters)
      hipLaunchKernelGGL((k_trace_closest<false, true>), grid, block, 0, c.stream, c.scene, c.paths, c.queues, cur, c.stats);
    else
      hipLaunchKernelGGL((k_trace_closest<false, false>), grid, block, 0, c.stream, c.scene, c.paths, c.queues, cur, c.stats);
  }
}
void launchShade(const LaunchCtx& c, int cur)
{
  dim3 grid(c.persistentBlocks), block(SHADE_BLOCK);
  if(c.collectCounters)
    hipLaunchKernelGGL((k_shade<true>), grid, block, 0, c.stream, c.scene, c.fc, c.paths, c.queues, cur, c.stats);
  else
    hipLaunchKernelGGL((k_shade<false>), grid, block, 0, c.stream, c.scene, c.fc, c.paths, c.queues, cur, c.stats);
}
void launchTraceShadow(const LaunchCtx& c)
{
  dim3 grid(c.persistentBlocks), block(TRACE_BLOCK);
  if(c.hasAlpha)
  {
    if(c.collectCounters)
      hipLaunchKernelGGL((k_trace_shadow<true, true>), grid, block, 0, c.stream, c.scene, c.paths, c.queues, c.stats);
    else
      hipLaunchKernelGGL((k_trace_shadow<true, false>), grid, block, 0, c.stream, c.scene, c.paths, c.queues, c.stats);
  }
  else
  {
    if(c.collectCounters)
      hipLaunchKernelGGL((k_trace_shadow<false, true>), grid, block, 0, c.stream, c.scene, c.paths, c.queues, c.stats);
    else
      hipLaunchKernelGGL((k_trace_shadow<false, false>), grid, block, 0, c.stream, c.scene, c.paths, c.queues, c.stats);
  }
}
void launchFinishSample(const LaunchCtx& c, int sampleIndex, float4* accum, float* depth, float4* albedo, float4* normal)
{
  unsigned grid = (unsigned(c.fc.numSlots) + 255u) / 256u;
  hipLaunchKernelGGL(k_finish_sample, dim3(grid), dim3(256), 0, c.stream, c.fc, c.paths, c.ownedTiles, sampleIndex, accum, depth, albedo, normal);
}
void launchSelection(const LaunchCtx& c, uint32_t* selection)
{
  unsigned grid = (unsigned(c.fc.numSlots) + TRACE_BLOCK - 1) / TRACE_BLOCK;
  hipLaunchKernelGGL(k_selection, dim3(grid), dim3(TRACE_BLOCK), 0, c.stream, c.scene, c.fc, c.ownedTiles, selection);
}

}  // namespace pt
