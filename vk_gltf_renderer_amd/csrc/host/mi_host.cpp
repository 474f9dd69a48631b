// C-ABI wrapper of the host scene front end (see include/mi_host.h).
#include "mi_host.h"
#include "mikktspace_tangents.hpp"

#include <cmath>
#include <cstring>
#include <string>

#include "gltf_scene.hpp"

namespace {
thread_local std::string g_hostError;
}

struct MiScene
{
  mihost::GltfScene scene;
};
struct MiHdr
{
  mihost::HdrEnvironment hdr;
};

extern "C" {

const char* mi_host_last_error(void)
{
  return g_hostError.c_str();
}

int mi_scene_load(const char* path, MiScene** out)
{
  if(!path || !out)
  {
    g_hostError = "mi_scene_load: null argument";
    return MI_PT_ERR_ARGUMENT;
  }
  MiScene* s = new MiScene();
  try
  {
    if(!s->scene.load(path))
    {
      g_hostError = s->scene.error();
      delete s;
      return MI_PT_ERR_IO;
    }
  }
  catch(const std::exception& e)
  {
    g_hostError = e.what();
    delete s;
    return MI_PT_ERR_IO;
  }
  *out = s;
  return MI_PT_OK;
}
void mi_scene_destroy(MiScene* scene)
{
  delete scene;
}
int mi_scene_recompute_tangents(MiScene* scene, int forceCreation, int mikktspace)
{
  if(!scene)
  {
    g_hostError = "mi_scene_recompute_tangents: null scene";
    return MI_PT_ERR_ARGUMENT;
  }
  try
  {
    return int(scene->scene.recomputeTangents(forceCreation != 0, mikktspace != 0));
  }
  catch(const std::exception& e)
  {
    g_hostError = e.what();
    return MI_PT_ERR_IO;
  }
}
int64_t mi_scene_cut_alpha(MiScene* scene, int subdivisions)
{
  if(!scene)
  {
    g_hostError = "mi_scene_cut_alpha: null scene";
    return MI_PT_ERR_ARGUMENT;
  }
  try
  {
    return int64_t(scene->scene.cutAlphaMasked(subdivisions));
  }
  catch(const std::exception& e)
  {
    g_hostError = e.what();
    return MI_PT_ERR_IO;
  }
}
int mi_scene_num_animations(const MiScene* scene)
{
  return scene ? const_cast<MiScene*>(scene)->scene.numAnimations() : 0;
}
int mi_scene_animation_info(const MiScene* scene, int index, float* start, float* end, char* name, int nameCapacity)
{
  if(!scene || index < 0 || index >= const_cast<MiScene*>(scene)->scene.numAnimations())
  {
    g_hostError = "mi_scene_animation_info: no such animation";
    return MI_PT_ERR_ARGUMENT;
  }
  const mihost::AnimationInfo& info = const_cast<MiScene*>(scene)->scene.animationInfo(index);
  if(start)
    *start = info.start;
  if(end)
    *end = info.end;
  if(name && nameCapacity > 0)
  {
    strncpy(name, info.name.c_str(), size_t(nameCapacity) - 1);
    name[nameCapacity - 1] = 0;
  }
  return MI_PT_OK;
}
int mi_scene_update_animation(MiScene* scene, int index, float time)
{
  if(!scene || index < 0 || index >= scene->scene.numAnimations() || !std::isfinite(time))
  {
    g_hostError = "mi_scene_update_animation: no such animation, or a non-finite time";
    return MI_PT_ERR_ARGUMENT;
  }
  try
  {
    scene->scene.animationInfo(index).currentTime = time;
    return scene->scene.updateAnimation(index) ? 1 : 0;
  }
  catch(const std::exception& e)
  {
    g_hostError = e.what();
    return MI_PT_ERR_IO;
  }
}
int mi_mikktspace(const float* positions, const float* normals, const float* texCoords, uint32_t numVertices, const uint32_t* indices,
                  uint32_t numTriangles, float* cornerTangents)
{
  if(!positions || !normals || !texCoords || !indices || !cornerTangents)
  {
    g_hostError = "mi_mikktspace: null argument";
    return MI_PT_ERR_ARGUMENT;
  }
  for(size_t c = 0; c < size_t(numTriangles) * 3; ++c)
    if(indices[c] >= numVertices)
    {
      g_hostError = "mi_mikktspace: index out of range";
      return MI_PT_ERR_ARGUMENT;
    }
  try
  {
    std::vector<float> out;
    mihost::mikkTangentSpaces(positions, normals, texCoords, indices, numTriangles, out);
    std::memcpy(cornerTangents, out.data(), out.size() * sizeof(float));
  }
  catch(const std::exception& e)
  {
    g_hostError = e.what();
    return MI_PT_ERR_IO;
  }
  return MI_PT_OK;
}
const MiPtSceneDesc* mi_scene_desc(const MiScene* scene)
{
  return scene ? &scene->scene.desc() : nullptr;
}
int mi_scene_num_cameras(const MiScene* scene)
{
  return scene ? int(scene->scene.cameras().size()) : 0;
}
int mi_scene_camera(const MiScene* scene, int index, MiCamera* out)
{
  if(!scene || !out || index < 0 || index >= int(scene->scene.cameras().size()))
    return MI_PT_ERR_ARGUMENT;
  const mihost::RenderCamera& c = scene->scene.cameras()[size_t(index)];
  for(int i = 0; i < 3; ++i)
  {
    out->eye[i]    = float(c.eye[i]);
    out->center[i] = float(c.center[i]);
    out->up[i]     = float(c.up[i]);
  }
  out->znear        = float(c.znear);
  out->zfar         = float(c.zfar);
  out->orthographic = c.type == mihost::RenderCamera::eOrthographic ? 1 : 0;
  out->xmag         = float(c.xmag);
  out->ymag         = float(c.ymag);
  // reference: toManipulatorCamera, src/gltf_camera_utils.hpp:35-55
  out->fovDegrees = out->orthographic ? 45.0f : float(c.yfov * 180.0 / M_PI);
  return MI_PT_OK;
}
void mi_scene_bounds(const MiScene* scene, float bmin[3], float bmax[3])
{
  scene->scene.bounds(bmin, bmax);
}
uint64_t mi_scene_num_triangles(const MiScene* scene)
{
  return scene ? scene->scene.numTriangles() : 0;
}

int mi_hdr_load(const char* path, MiHdr** out)
{
  if(!path || !out)
    return MI_PT_ERR_ARGUMENT;
  MiHdr* h = new MiHdr();
  if(!h->hdr.load(path))
  {
    g_hostError = h->hdr.error();
    delete h;
    return MI_PT_ERR_IO;
  }
  *out = h;
  return MI_PT_OK;
}
int mi_hdr_from_pixels(int width, int height, const float* rgb, MiHdr** out)
{
  if(width <= 0 || height <= 0 || !rgb || !out)
    return MI_PT_ERR_ARGUMENT;
  MiHdr* h = new MiHdr();
  h->hdr.setPixels(width, height, rgb);
  *out = h;
  return MI_PT_OK;
}
void mi_hdr_destroy(MiHdr* hdr)
{
  delete hdr;
}
const MiPtEnvironment* mi_hdr_env(const MiHdr* hdr)
{
  return hdr ? &hdr->hdr.env() : nullptr;
}

void mi_default_sky(MiSkyPhysicalParameters* sky)
{
  MiSkyPhysicalParameters s{};
  s.rgbUnitConversion[0] = s.rgbUnitConversion[1] = s.rgbUnitConversion[2] = 1.0f / 80000.0f;
  s.multiplier                                                             = 0.1f;
  s.haze                                                                   = 0.1f;
  s.redblueshift                                                           = 0.1f;
  s.saturation                                                             = 1.0f;
  s.horizonHeight                                                          = 0.0f;
  s.groundColor[0] = s.groundColor[1] = s.groundColor[2] = 0.4f;
  s.horizonBlur                                          = 0.3f;
  s.nightColor[0] = s.nightColor[1] = s.nightColor[2] = 0.0f;
  s.sunDiskIntensity                                   = 1.0f;
  s.sunDirection[0] = s.sunDirection[1] = s.sunDirection[2] = 0.5773502691896258f;
  s.sunDiskScale                                             = 1.0f;
  s.sunGlowIntensity                                         = 1.0f;
  s.yIsUp                                                    = 1;
  *sky                                                       = s;
}

void mi_default_params(MiPathtraceParams* params)
{
  MiPathtraceParams p{};
  p.maxDepth              = 5;
  p.frameCount            = 0;
  p.fireflyClampThreshold = 10.0f;
  p.texGradScale          = 1.0f;
  p.numSamples            = 1;
  p.totalSamples          = 0;
  p.focalDistance         = 0.0f;
  p.aperture              = 0.0f;
  p.flags                 = 0;
  p.pixelAngle            = 0.0f;
  *params                 = p;
}

void mi_camera_frame_info(const MiCamera* cam, int width, int height, MiSceneFrameInfo* info, float* pixelAngle, float* focalDistance)
{
  mx::vec3 eye{cam->eye[0], cam->eye[1], cam->eye[2]}, center{cam->center[0], cam->center[1], cam->center[2]},
      up{cam->up[0], cam->up[1], cam->up[2]};
  mx::mat4 view   = mx::lookAt(eye, center, up);
  float    aspect = float(width) / float(height);
  mx::mat4 proj;
  if(cam->orthographic)
  {
    // nvutils::CameraManipulator orthographic: magnitudes scaled to the viewport aspect
    float xm = cam->xmag, ym = cam->ymag;
    proj = mx::orthoVk(-xm, xm, -ym, ym, cam->znear, cam->zfar);
  }
  else
    proj = mx::perspectiveVk(cam->fovDegrees * float(M_PI) / 180.0f, aspect, cam->znear, cam->zfar);
  mx::mat4 projInv  = mx::inverse(proj);
  mx::mat4 viewInv  = mx::inverse(view);
  mx::mat4 viewProj = mx::mul(proj, view);
  MiSceneFrameInfo f;
  memset(&f, 0, sizeof(f));
  memcpy(f.viewMatrix, view.m, 64);
  memcpy(f.projInv, projInv.m, 64);
  memcpy(f.viewInv, viewInv.m, 64);
  memcpy(f.viewProjMatrix, viewProj.m, 64);
  memcpy(f.prevMVP, viewProj.m, 64);
  f.imageSize[0] = float(width);
  f.imageSize[1] = float(height);
  f.flags        = cam->orthographic ? MI_SCENE_IS_ORTHOGRAPHIC : 0;
  // Settings defaults (reference: src/resources.hpp:82-131)
  f.envRotation               = 0.0f;
  f.envBlur                   = 0.0f;
  f.envIntensity              = 1.0f;
  f.visualization             = 0;
  f.infinitePlaneDistance     = 0.0f;
  f.infinitePlaneBaseColor[0] = f.infinitePlaneBaseColor[1] = f.infinitePlaneBaseColor[2] = 0.5f;
  f.infinitePlaneMetallic                                                                 = 0.0f;
  f.infinitePlaneRoughness                                                                = 0.5f;
  f.shadowCatcherDarkenAmount                                                             = 0.0f;
  *info                                                                                   = f;
  if(pixelAngle)
    *pixelAngle = 2.0f * std::fabs(projInv.at(1, 1)) / std::max(float(height), 1.0f);
  if(focalDistance)
    *focalDistance = mx::length(eye - center);
}
}
