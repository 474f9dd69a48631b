/*
 * mi_pt.h — C-ABI of the MI355X-native wavefront path tracer (libmi_pt.so).
 *
 * This is the drop-in boundary for ONE path of nvpro-samples/vk_gltf_renderer: the `PathTracer : BaseRenderer`
 * plugin (reference: src/renderer_base.hpp:33-55, src/renderer_pathtracer.cpp:500-614) and the Slang megakernel it
 * dispatches (reference: shaders/gltf_pathtrace.slang:546-699).  Every entry point below names the reference
 * interface it replaces.  Plain pointers and sizes only; no Vulkan, torch or C++ types cross this line.
 *
 * Conventions: every function returns 0 on success, a negative MiPtStatus on failure, never throws; the caller owns
 * all memory it passes in (it may be freed as soon as the call returns), the library owns all device memory.
 * Not thread-safe per instance (like BaseRenderer::onRender, which runs on the app thread only).
 *
 * Environment: the library's behaviour does not depend on the environment in production.  A set of MI_PT_* variables selects A/B
 * variants and diagnostics ("Run-time switches (all of them)", lists every one with its default); they are read ONCE,
 * in mi_pt_create, into the instance -- a variable that appears or changes later cannot alter a live instance's slot layout, kernels
 * or scene.
 */
#ifndef MI_PT_H
#define MI_PT_H

#include "mi_pt_shaderio.h"

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define MI_PT_API __attribute__((visibility("default")))
#else
#define MI_PT_API
#endif

typedef enum MiPtStatus
{
  MI_PT_OK            = 0,
  MI_PT_ERR_ARGUMENT  = -1,
  MI_PT_ERR_NO_DEVICE = -2, /* no HIP device: the product path never falls back to the CPU */
  MI_PT_ERR_HIP       = -3,
  MI_PT_ERR_STATE     = -4,
  MI_PT_ERR_IO        = -5
} MiPtStatus;

/* ------------------------------------------------------------------------------------------------------------------
 * Scene tables handed to the renderer.  They are what `SceneVk` uploads and what `GltfScene` points at in the
 * reference (src/gltf_scene_vk.cpp:330-349 scene-desc, :365 materials, :531 render nodes, :741-870 vertex buffers,
 * :951-1098 textures, :1354-1392 lights), restated with host pointers.
 * ---------------------------------------------------------------------------------------------------------------- */

/* One RenderPrimitive: SoA attribute streams, NULL = attribute absent (reference: VertexBuffers,
 * shaders/gltf_scene_io.h.slang:50-64; indices are always u32 triplets, src/gltf_scene_vk.cpp:816-836; COLOR_0 is
 * packed unorm4x8, :766-798). */
typedef struct MiPtRenderPrimitive
{
  const uint32_t* indices; /* 3 * triangleCount */
  uint32_t        triangleCount;
  uint32_t        vertexCount;
  const float*    positions;  /* 3 floats / vertex */
  const float*    normals;    /* 3 floats / vertex or NULL */
  const uint32_t* colors;     /* unorm4x8 / vertex or NULL */
  const float*    tangents;   /* 4 floats / vertex or NULL */
  const float*    texCoords0; /* 2 floats / vertex or NULL */
  const float*    texCoords1; /* 2 floats / vertex or NULL */
  /* Triangles [0, opaqueTriangleCount) are known to pass their material's alpha test everywhere (alpha-MASK geometry classified at
   * load, mi_scene_cut_alpha -- the OPAQUE state of the reference's opacity micro-maps, src/gltf_scene_omm.cpp): the walks treat them
   * like triangles of a FORCE_OPAQUE instance.  0 = nothing known (every triangle of a non-opaque material is alpha-tested). */
  uint32_t        opaqueTriangleCount;
  uint32_t        reserved; /* 0 */
} MiPtRenderPrimitive;

enum MiPtFilter { MI_FILTER_NEAREST = 0, MI_FILTER_LINEAR = 1 };
enum MiPtWrap { MI_WRAP_REPEAT = 0, MI_WRAP_CLAMP_TO_EDGE = 1, MI_WRAP_MIRRORED_REPEAT = 2 };

/* One glTF *texture* (image + sampler), indexed by GltfTextureInfo.index (reference: src/renderer.cpp:1883-1912
 * bindless array; sampler mapping src/gltf_scene_vk.cpp:909-947; sRGB detection :1102-1154). RGBA8, full mip chain
 * supplied by the caller (the reference blits it at upload, src/gltf_scene_vk.cpp:1247-1347). */
typedef struct MiPtTexture
{
  const uint8_t* const* levels; /* numLevels pointers, level i is max(1,width>>i) x max(1,height>>i) RGBA8 */
  int                   width, height, numLevels;
  int                   srgb; /* 1: texels are sRGB-encoded colour (alpha linear) */
  int                   magFilter, minFilter, mipmapMode; /* MiPtFilter */
  int                   wrapS, wrapT;                     /* MiPtWrap */
} MiPtTexture;

typedef struct MiPtSceneDesc
{
  const MiGltfShadeMaterial* materials;
  int                        numMaterials;
  const MiGltfTextureInfo*   textureInfos; /* [0] is the reserved "no texture" slot */
  int                        numTextureInfos;
  const MiGltfRenderNode*    renderNodes;
  int                        numRenderNodes;
  const uint8_t*             renderNodeVisible; /* numRenderNodes flags or NULL (= all visible); invisible nodes get
                                                   no geometry, reference src/gltf_scene_rtx.cpp:319-323 */
  const MiPtRenderPrimitive* renderPrimitives;
  int                        numRenderPrimitives;
  const MiGltfLight*         lights;
  int                        numLights;
  const MiPtTexture*         textures;
  int                        numTextures;
} MiPtSceneDesc;

/* HDR environment as `nvvk::HdrIbl` prepares it (reference: src/renderer.cpp:1982-2017; consumed at
 * shaders/pathtrace_functions.h.slang:436-447,474-479): lat-long RGBA32F whose alpha holds the sampling pdf, plus the
 * alias table with one entry per texel. */
typedef struct MiPtEnvironment
{
  const float*      rgba; /* width*height*4 */
  const MiEnvAccel* accel; /* width*height */
  int               width, height;
  float             integral; /* luminance integral (HdrIbl::getIntegral) */
} MiPtEnvironment;

typedef struct MiPtCreateOptions
{
  int device;          /* HIP device ordinal */
  int collectCounters; /* 1: kernels export traversal/shading counters (slower) */
  int bvhBuilder;      /* bit 0: 0 = 8-wide compressed BVH (default), 1 = plain BVH2 (A/B, tests);
                        * bit 1: 0 = PLOC clustering over the Morton order (default), 1 = Karras LBVH topology (A/B) */
  int reserved[5];
} MiPtCreateOptions;

/* Work counters behind SURVEY §8(d)'s algorithmic-bytes model; totals since the last mi_pt_reset_stats. */
typedef struct MiPtStats
{
  uint64_t cameraPaths;     /* samples started */
  uint64_t segments;        /* closest-hit rays traced */
  uint64_t shadowRays;      /* shadow rays traced */
  uint64_t nodesClosest;    /* BVH nodes visited by closest-hit rays (needs collectCounters) */
  uint64_t trisClosest;     /* triangles tested by closest-hit rays */
  uint64_t nodesShadow;
  uint64_t trisShadow;
  uint64_t textureTaps;     /* getTexture() calls */
  uint64_t bvhNodeCount;    /* static: nodes in the traversal structure */
  uint64_t bvhTriangleCount;
  uint64_t bvhNodeBytes;    /* S_node */
  uint64_t bvhTriangleBytes;/* S_tri */
  uint64_t surfaceHits;     /* segments that ended on a surface (mesh or infinite plane) and were shaded; segments - surfaceHits
                             * left the scene (environment / backplate).  Needs collectCounters. */
  uint64_t nodesPrimary;    /* node / triangle records fetched by the bounce-0 packet walk: ONE per wave (64 camera rays) and visit, */
  uint64_t trisPrimary;     /* not contained in nodesClosest / trisClosest (which count one per ray and visit) */
} MiPtStats;

/* Per-kernel device time summed over every mi_pt_render_frame since mi_pt_enable_timing(pt, 1), measured with HIP events
 * recorded on the frame's stream around each launch; resolved (one device sync) by mi_pt_get_frame_timing. */
typedef struct MiPtFrameTiming
{
  float totalMs;
  float generateMs;
  float traceClosestMs;
  float sortMs;
  float shadeMs;
  float traceShadowMs;
  float accumulateMs;
  int   traceClosestLaunches;
  int   shadeLaunches;
  int   traceShadowLaunches;
  int   bounceIterations;
  /* the bounce-0 launches on their own (they are also contained in traceClosestMs / shadeMs and the launch counts above) */
  float tracePrimaryMs;   /* k_trace_primary: camera-ray generation + packet walk + end of the paths that leave the scene */
  float shadeFirstMs;     /* k_shade<FIRST> */
  int   tracePrimaryLaunches;
  int   shadeFirstLaunches;
} MiPtFrameTiming;

typedef struct MiPt MiPt;

/* replaces PathTracer::onAttach + SceneVk::create + SceneRtx BLAS/TLAS build
 * (reference: src/renderer_pathtracer.cpp:150-260, src/gltf_scene_vk.cpp:218, src/gltf_scene_rtx.cpp:173-385).
 * Uploads the tables, builds the BVH on the device.  MI_PT_ERR_ARGUMENT: inconsistent tables, a texture pool of 2^32 texels
 * or more, or -- with the default 8-wide BVH -- 2^26 or more flattened triangles (the BVH2 walk, bvhBuilder bit 0, has no such limit). */
MI_PT_API int mi_pt_create(const MiPtSceneDesc* scene, const MiPtCreateOptions* options, MiPt** out);

/* replaces the per-frame instance update of animated / edited scenes: SceneVk::updateRenderNodesBuffer + the TLAS update of
 * SceneRtx (reference: src/gltf_scene_transform_vk.cpp:534-639, src/gltf_scene_rtx.cpp:299-385).  Takes the render-node table
 * again (same length as at creation; matrices, material ids and visibility may have changed) and rebuilds the acceleration
 * structure over the resident geometry ON THE DEVICE -- flatten, Morton sort, PLOC, 8-wide collapse: ~20 ms for 2.8 M triangles,
 * which is why instances are flattened instead of kept behind a two-level structure.  Synchronises with the work in flight;
 * the caller restarts accumulation (MI_PT_FIRST_FRAME) like the reference does after a scene change. */
MI_PT_API int mi_pt_update_render_nodes(MiPt* pt, const MiGltfRenderNode* renderNodes, int numRenderNodes, const uint8_t* renderNodeVisible);

/* replaces the light half of the per-frame scene sync of animated scenes: SceneVk::syncFromScene(eSyncLights) -> uploadLights
 * (reference: src/gltf_scene_vk.hpp:96-103, called from GltfRenderer::updateAnimation, src/renderer.cpp:2118-2131).  Takes the
 * light table again (same length as at creation; placement, colour, intensity, cone may have changed).  Synchronises with the
 * work in flight; the caller restarts accumulation. */
MI_PT_API int mi_pt_update_lights(MiPt* pt, const MiGltfLight* lights, int numLights);

/* replaces PathTracer::onDetach (reference: src/renderer_base.hpp:40) */
MI_PT_API int mi_pt_destroy(MiPt* pt);

/* replaces GltfRenderer::createHDR (reference: src/renderer.cpp:1982-2017). NULL env = no HDR loaded. */
MI_PT_API int mi_pt_set_environment(MiPt* pt, const MiPtEnvironment* env);

/* replaces PathTracer::onResize (reference: src/renderer_base.hpp:41): (re)allocates eImgRendered / eImgSelection /
 * depth and the path-state queues; resets accumulation. */
MI_PT_API int mi_pt_resize(MiPt* pt, int width, int height);

/* replaces the vkCmdUpdateBuffer of bFrameInfo / bSkyParams (reference: src/renderer.cpp:675-708) */
MI_PT_API int mi_pt_set_frame_info(MiPt* pt, const MiSceneFrameInfo* info);
MI_PT_API int mi_pt_set_sky(MiPt* pt, const MiSkyPhysicalParameters* sky);

/* Image-tile partition for multi-GPU runs (no counterpart in the reference, which is single-GPU): this instance
 * renders only tiles whose index satisfies (tileY * tilesX + tileX) % world == rank; other pixels of the accumulator
 * are left untouched (zero after resize) so a sum-reduce over ranks yields the full frame. world = 1 disables. */
MI_PT_API int mi_pt_set_tile_partition(MiPt* pt, int rank, int world, int tileSize);

/* Render into caller-owned device memory (width*height float4, e.g. a torch tensor) instead of the internal image. */
MI_PT_API int mi_pt_bind_accum(MiPt* pt, void* deviceRGBA32F);
/* The same for the images the denoiser reads next to the accumulator: first-hit albedo and normal guides (width*height float4
 * each) and the frame-0 NDC depth (width*height float).  NULL = the internal image.  A multi-GPU run binds zero-initialised
 * caller memory on every rank, sum-reduces it together with the accumulator (disjoint tiles: sum == gather) and denoises on the
 * rank that holds the sum. */
MI_PT_API int mi_pt_bind_guides(MiPt* pt, void* deviceAlbedoRGBA32F, void* deviceNormalRGBA32F, void* deviceDepthR32F);

/* replaces PathTracer::onRender = setupPushConstant + renderRayQuery (reference: src/renderer_pathtracer.cpp:500-614,
 * :1496-1574, :1404-1431): enqueues ONE frame (params->numSamples spp for every owned pixel, running-mean
 * accumulation, selection id + NDC depth when MI_PT_FIRST_FRAME is set) on `hipStream` (a hipStream_t, NULL = default
 * stream).  Asynchronous unless counters/timing are being collected. */
MI_PT_API int mi_pt_render_frame(MiPt* pt, const MiPathtraceParams* params, void* hipStream);

/* Frames in flight: the result (accumulator, guides, depth, selection) is bit-identical to `numFrames` successive
 * mi_pt_render_frame calls with params->frameCount + f, params->totalSamples + f * numSamples and MI_PT_FIRST_FRAME only
 * on f = 0 -- i.e. the frames GltfRenderer::onRender / updateFrameCounter would issue one after the other
 * (reference: src/renderer.cpp:1939-1977, src/renderer_pathtracer.cpp:1401, :1502-1505) -- but their paths share every
 * wavefront launch, so the short late-bounce queues and the launch overheads are paid once per batch.  Frames only
 * couple through the running mean, which the finish kernel folds in frame order.  1 <= numFrames <= 1024 (and frames x owned pixels < 2^31); the path-state
 * arrays grow (one synchronising reallocation) the first time a larger batch is requested: ~0.3 KB per pixel per frame. */
MI_PT_API int mi_pt_render_frames(MiPt* pt, const MiPathtraceParams* params, int numFrames, void* hipStream);

/* The same batching for a caller that keeps the reference's one-call-per-frame shape (GltfRenderer::onRender -> PathTracer::onRender once per app
 * frame, src/renderer.cpp:713-717, :1939-1977): with depth > 1, mi_pt_render_frame only RECORDS the frame; consecutive frames (frameCount + 1,
 * totalSamples + numSamples, everything else equal, same stream) are held back and issued as one mi_pt_render_frames batch when `depth` of them are
 * pending -- or as soon as any other entry point of this header is called on the instance (mi_pt_synchronize, every read / bind / set / update /
 * denoise / tonemap / statistics call): whatever a caller can observe is what depth 1 would have produced, bit for bit, only later in time.
 * A frame that does not continue the pending run (a reset: MI_PT_FIRST_FRAME, changed parameters) flushes the run and starts a new one.
 * depth 1 (the default) = every call launches its frame at once, as in the reference.  1 <= depth <= 1024. */
MI_PT_API int mi_pt_set_frame_queue(MiPt* pt, int depth);

/* Block until everything enqueued by this instance has finished. */
MI_PT_API int mi_pt_synchronize(MiPt* pt);

/* Read-backs (the reference reads gBuffers images back for screenshots: src/renderer.cpp:557-573). */
MI_PT_API int mi_pt_read_accum(MiPt* pt, float* hostRGBA32F);          /* eImgRendered   */
/* Upload an image INTO eImgRendered (width*height RGBA32F), e.g. to post-process (denoise / tonemap) a frame rendered elsewhere or
 * an accumulation restored from disk; the next frame without MI_PT_FIRST_FRAME continues the running mean from it. */
MI_PT_API int mi_pt_write_accum(MiPt* pt, const float* hostRGBA32F);
/* Denoiser guide layers accumulated while MI_PT_USE_OPTIX_DENOISER is set in params->flags (first-hit albedo.rgb + hit
 * fraction, first-hit shading normal.xyz; reference capture points: shaders/gltf_pathtrace.slang:228-264, the OptiX guide
 * images of src/optix_denoiser.hpp:128-153).  Either pointer may be NULL. */
MI_PT_API int mi_pt_read_guides(MiPt* pt, float* hostAlbedoRGBA32F, float* hostNormalRGBA32F);
MI_PT_API int mi_pt_read_selection(MiPt* pt, uint32_t* hostObjectIds); /* eImgSelection: renderNode+1, 0 = none */
MI_PT_API int mi_pt_read_depth(MiPt* pt, float* hostDepth);            /* NDC depth of frame 0 */
MI_PT_API void* mi_pt_accum_device_ptr(MiPt* pt);

/* a-trous edge-avoiding wavelet denoise of the accumulator using the first-hit albedo/normal guides captured when
 * MI_PT_USE_OPTIX_DENOISER is set (replaces OptiXDenoiser::denoiseImageBuffer I/O contract, reference:
 * src/optix_denoiser.hpp:128-153). Result in hostRGBA32F (may be NULL) and in the internal denoised image. */
MI_PT_API int mi_pt_denoise(MiPt* pt, int iterations, float sigmaColor, float sigmaNormal, float sigmaAlbedo,
                            float* hostRGBA32F, void* hipStream);

/* Variance-guided (SVGF-style) denoise of the accumulator: the same I/O contract as mi_pt_denoise, guided in addition by the
 * per-pixel variance of the mean -- from the second moment of the per-frame pixel luminance that the frames accumulate next to
 * the guides while MI_PT_USE_OPTIX_DENOISER is set (spatial estimate below 4 frames) -- and by the frame-0 depth; albedo is
 * demodulated before and re-applied after filtering.  The temporal half of SVGF is the running mean itself (the camera of a
 * progressive accumulation stands still), so there is no reprojection.  Asynchronous on hipStream unless hostRGBA32F is given. */
MI_PT_API int mi_pt_denoise_svgf(MiPt* pt, int iterations, float sigmaLuminance, float sigmaNormal, float sigmaDepth, float* hostRGBA32F, void* hipStream);
/* device address of the last denoise result (NULL before the first), valid until the next denoise / resize */
MI_PT_API const void* mi_pt_denoised_device_ptr(MiPt* pt);

/* replaces GltfRenderer::tonemap -> nvshaders::Tonemapper::runCompute (reference: src/renderer.cpp:992-1056): the HDR image
 * (source 0 = the accumulator eImgRendered, 1 = the denoised image of the last mi_pt_denoise, as the reference routes the OptiX
 * output: src/renderer.cpp:1006-1016) -> display-referred RGBA8, the eImgTonemapped image the headless run saves
 * (src/renderer.cpp:557-573).  Alpha passes through.  With tm->autoExposure the metering histogram and the exposure easing run on
 * the device too; dtSeconds is the time since the previous call (< 0: jump to the target).  hostRGBA8 may be NULL (result kept in
 * the internal image, mi_pt_tonemapped_device_ptr). */
MI_PT_API int mi_pt_tonemap(MiPt* pt, const MiTonemapperData* tm, int source, float dtSeconds, uint8_t* hostRGBA8, void* hipStream);
MI_PT_API void* mi_pt_tonemapped_device_ptr(MiPt* pt);
/* the reference's defaults: Filmic, active, exposure / brightness / contrast / saturation 1, vignette 0, autoExposure as given */
MI_PT_API void mi_pt_default_tonemapper(MiTonemapperData* tm, int autoExposure);

/* Device memory held by this instance, in bytes, split the way the reference's benchmark reports it (GltfRenderer::
 * benchmarkMemorySamples, src/renderer.cpp:530-555): "Scene" = geometry, textures, materials, environment and the acceleration
 * structure (SceneVk + SceneRtx trackers); "PathTracer" = everything the renderer owns (path state, queues, images). */
typedef struct MiPtMemory
{
  uint64_t sceneBytes;
  uint64_t rendererBytes;
  uint64_t deviceUsedBytes;  /* whole device, all processes: total - free as the driver reports it */
  uint64_t deviceTotalBytes;
  uint64_t pathStateBytes;   /* the part of rendererBytes that grows with the frames in flight: by-slot path records + the three ray queues (+ candidate pool) */
  uint64_t pathSlots;        /* owned pixel slots x frames in flight the arrays are sized for: pathStateBytes / pathSlots = bytes per path slot */
} MiPtMemory;
MI_PT_API int mi_pt_get_memory(MiPt* pt, MiPtMemory* memory);

MI_PT_API int mi_pt_get_stats(MiPt* pt, MiPtStats* stats);
MI_PT_API int mi_pt_reset_stats(MiPt* pt);
MI_PT_API int mi_pt_enable_timing(MiPt* pt, int enable);
MI_PT_API int mi_pt_get_frame_timing(MiPt* pt, MiPtFrameTiming* timing);

/* Human-readable description of the last failure on this thread ("" if none). */
MI_PT_API const char* mi_pt_last_error(void);
MI_PT_API const char* mi_pt_version(void);
/* Layout version of the public structs of this header.  It is bumped whenever a struct a caller allocates (MiPtMemory, MiPtStats, MiPtFrameTiming,
 * MiPathtraceParams, ...) grows or changes: the library writes every field of the struct it was compiled with, so a caller built against an older
 * header must refuse to run -- `if(mi_pt_abi_version() != MI_PT_ABI_VERSION) fail` right after loading the library.
 * 6: MiPtMemory grew pathStateBytes / pathSlots (round 5); mi_pt_render_frames refuses maxDepth 0; mi_pt_set_frame_queue. */
#define MI_PT_ABI_VERSION 6
MI_PT_API int mi_pt_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MI_PT_H */
