/*
 * mi_host.h — C-ABI of the host-side scene front end (libmi_host.so): the callers of the path-trace hot path.
 * It produces exactly the tables mi_pt_create() / mi_pt_set_environment() / mi_pt_set_frame_info() consume, from the
 * same inputs the reference's application layer uses (a .gltf/.glb file, a Radiance .hdr file, a camera).
 * Reference counterparts: nvvkgltf::Scene::load (src/gltf_scene.cpp:298), GltfRenderer::createHDR
 * (src/renderer.cpp:1982), the SceneFrameInfo fill (src/renderer.cpp:675-705) and PathTracer::setupPushConstant
 * (src/renderer_pathtracer.cpp:1496-1574).
 */
#ifndef MI_HOST_H
#define MI_HOST_H

#include "mi_pt.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct MiScene MiScene;
typedef struct MiHdr   MiHdr;

/* reference: nvutils::CameraManipulator::Camera as filled by toManipulatorCamera (src/gltf_camera_utils.hpp:35-55) */
typedef struct MiCamera
{
  float eye[3], center[3], up[3];
  float fovDegrees; /* vertical */
  float znear, zfar;
  int   orthographic;
  float xmag, ymag;
} MiCamera;

MI_PT_API int                  mi_scene_load(const char* path, MiScene** out);
MI_PT_API void                 mi_scene_destroy(MiScene* scene);
MI_PT_API const MiPtSceneDesc* mi_scene_desc(const MiScene* scene);
MI_PT_API int                  mi_scene_num_cameras(const MiScene* scene);
MI_PT_API int                  mi_scene_camera(const MiScene* scene, int index, MiCamera* out);
MI_PT_API void                 mi_scene_bounds(const MiScene* scene, float bmin[3], float bmax[3]);
MI_PT_API uint64_t             mi_scene_num_triangles(const MiScene* scene);
/* recomputeTangents(model, forceCreation, mikktspace) (reference: src/gltf_create_tangent.hpp:28-40, the UI's "Recreate Tangents"
 * items src/ui_renderer.cpp:855-875): simple UV-gradient tangents (mikktspace = 0) or Mikkelsen's tangent space with vertex
 * splitting at UV seams / mirrored UVs.  Returns the number of vertices added by the splitting (>= 0) or a negative MiPtStatus;
 * the MiPtSceneDesc of the scene changes (fetch mi_scene_desc again and re-create the renderer, as the reference re-creates
 * SceneVk / SceneRtx after a split). */
MI_PT_API int                  mi_scene_recompute_tangents(MiScene* scene, int forceCreation, int mikktspace);
/* The raw per-corner output of the Mikkelsen tangent-space computation for a triangle list (4 floats per corner: unit tangent, +1 /
 * -1 orientation), exposed so that tests can compare it with the reference's third_party/MikkTSpace on the same arrays. */
MI_PT_API int                  mi_mikktspace(const float* positions, const float* normals, const float* texCoords, uint32_t numVertices,
                                             const uint32_t* indices, uint32_t numTriangles, float* cornerTangents);

/* Load-time bake for alpha-MASK geometry, this renderer's counterpart of the reference's opacity micro-map bake
 * (src/gltf_scene_omm.cpp; UI switch "Use OMM" src/ui_renderer.cpp): every alpha-MASK triangle is cut adaptively along a
 * subdivisions x subdivisions barycentric grid (rounded up to 2, 4, 8 or 16; 4 is a good default) and the pieces on which the alpha test cannot pass -- no texel a fetch inside them may
 * touch reaches alphaCutoff -- are dropped, so that rays through the empty part of a leaf card meet no candidate at all; the pieces on
 * which it cannot FAIL are moved to the front of their primitive and counted in MiPtRenderPrimitive::opaqueTriangleCount (the walks skip
 * the alpha test for them).  The
 * image is unchanged up to float rounding of the interpolated vertices (and up to sub-ulp T-junction cracks where a merged coarse piece
 * meets refined ones: ~1e-7 of a card's area, csrc/host/alpha_cut.cpp); the selection image (TraceLow treats every triangle as
 * opaque) reports what is seen through a removed part instead of the alpha-tested instance itself.  Returns the number of (sub-)triangles dropped (>= 0)
 * or a negative MiPtStatus; the MiPtSceneDesc changes (fetch mi_scene_desc again, create the renderer afterwards). */
MI_PT_API int64_t              mi_scene_cut_alpha(MiScene* scene, int subdivisions);

/* Keyframe animation of node transforms (reference: nvvkgltf::AnimationSystem, src/gltf_scene_animation.hpp:93-122; AnimationInfo
 * src/gltf_scene.hpp:159-189; driven per frame by GltfRenderer::updateAnimation, src/renderer.cpp:2065-2170).  Translation /
 * rotation / scale channels with LINEAR, STEP and CUBICSPLINE samplers; morph weights, skins and KHR_animation_pointer are not
 * evaluated.  mi_scene_update_animation poses the scene at `time` (seconds on the clip's own axis, [start, end] as reported by
 * mi_scene_animation_info) and rewrites the matrices of the render-node table and the light placements of mi_scene_desc() in
 * place -- same pointers, same counts -- ready for mi_pt_update_render_nodes() + mi_pt_update_lights().  Returns 1 when
 * something moved, 0 when no channel covered `time`, or a negative MiPtStatus. */
MI_PT_API int                  mi_scene_num_animations(const MiScene* scene);
MI_PT_API int                  mi_scene_animation_info(const MiScene* scene, int index, float* start, float* end, char* name, int nameCapacity);
MI_PT_API int                  mi_scene_update_animation(MiScene* scene, int index, float time);

MI_PT_API int                    mi_hdr_load(const char* path, MiHdr** out);
MI_PT_API int                    mi_hdr_from_pixels(int width, int height, const float* rgb, MiHdr** out);
MI_PT_API void                   mi_hdr_destroy(MiHdr* hdr);
MI_PT_API const MiPtEnvironment* mi_hdr_env(const MiHdr* hdr);

/* Defaults of `SkyPhysicalParameters{}` (reference: src/renderer.cpp:1328). */
MI_PT_API void mi_default_sky(MiSkyPhysicalParameters* sky);
/* Defaults of PathtracePushConstant + PathTracer members (reference: shaders/shaderio.h:179-196,
 * src/renderer_pathtracer.cpp:60-67). */
MI_PT_API void mi_default_params(MiPathtraceParams* params);
/* Fills view/proj matrices, imageSize, flags (orthographic bit) and zeroes/defaults the rest exactly as
 * GltfRenderer::onRender does with default Settings; also returns pixelAngle and the auto-focus focal distance
 * (reference: src/renderer.cpp:675-705, src/renderer_pathtracer.cpp:1508-1512,1570-1571). */
MI_PT_API void mi_camera_frame_info(const MiCamera* camera, int width, int height, MiSceneFrameInfo* info, float* pixelAngle,
                                    float* focalDistance);

MI_PT_API const char* mi_host_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
