#include "renderer_pathtracer.hpp"

#include <cmath>
#include <cstdio>

PathTracer::PathTracer()
{
  // defaults of PathtracePushConstant + PathTracer members (reference: shaders/shaderio.h:179-196,
  // src/renderer_pathtracer.cpp:60-67)
  m_pushConst.maxDepth              = 5;
  m_pushConst.frameCount            = 0;
  m_pushConst.fireflyClampThreshold = 10.0f;
  m_pushConst.texGradScale          = 1.0f;
  m_pushConst.numSamples            = 1;
  m_pushConst.totalSamples          = 0;
  m_pushConst.focalDistance         = 0.0f;
  m_pushConst.aperture              = 0.0f;
  m_pushConst.flags                 = 0;
  m_pushConst.pixelAngle            = 0.0f;
}

PathTracer::~PathTracer()
{
  if(m_pt)
    mi_pt_destroy(m_pt);
}

void PathTracer::registerParameters(ParameterRegistry* r)
{
  r->add("ptMaxDepth", "Maximum depth of the ray", &m_pushConst.maxDepth);
  r->add("ptSamples", "Number of samples per pixel per frame (disables adaptive sampling)", &m_pushConst.numSamples);
  r->add("ptFireflyClamp", "Firefly clamp threshold", &m_pushConst.fireflyClampThreshold);
  r->add("ptTexGradScale", "Ray-footprint gradient scale", &m_pushConst.texGradScale);
  r->add("ptAperture", "Aperture for depth of field", &m_pushConst.aperture);
  r->add("ptFocalDistance", "Focal distance (disables auto focus)", &m_pushConst.focalDistance);
  r->add("ptAutoFocus", "Focus on the camera's interest point", &m_autoFocus);
  r->add("ptAdaptiveSampling", "Accepted for CLI compatibility; adaptive sampling is not implemented (always off)", &m_adaptiveSampling);
}

void PathTracer::onAttach(Resources& res, void* profiler)
{
  BaseRenderer::onAttach(res, profiler);
  onSceneInvalidated(res);
}

void PathTracer::onSceneInvalidated(Resources& res)
{
  if(m_pt)
  {
    mi_pt_destroy(m_pt);
    m_pt = nullptr;
  }
  if(!res.scene)
    return;
  MiPtCreateOptions opt{};
  opt.device = res.device;
  if(mi_pt_create(mi_scene_desc(res.scene), &opt, &m_pt) != MI_PT_OK)
  {
    m_error = mi_pt_last_error();
    fprintf(stderr, "PathTracer: mi_pt_create failed: %s\n", m_error.c_str());
    m_pt = nullptr;
    return;
  }
  if(res.hdrIbl)
    mi_pt_set_environment(m_pt, mi_hdr_env(res.hdrIbl));
  if(res.renderSize.width > 0)
    mi_pt_resize(m_pt, int(res.renderSize.width), int(res.renderSize.height));
}

void PathTracer::onDetach(Resources&)
{
  if(m_pt)
    mi_pt_destroy(m_pt);
  m_pt = nullptr;
}

void PathTracer::onResize(StreamHandle, const Extent2D& size, Resources&)
{
  if(m_pt && mi_pt_resize(m_pt, int(size.width), int(size.height)) != MI_PT_OK)
    m_error = mi_pt_last_error();
}

void PathTracer::setupPushConstant(Resources& res, const Extent2D& renderingSize)
{
  if(res.frameCount == 0)
    m_totalSamplesAccumulated = 0;  // reset sample counter when scene/camera changes
  if(m_autoFocus)
  {
    const float dx = res.camera.eye[0] - res.camera.center[0], dy = res.camera.eye[1] - res.camera.center[1], dz = res.camera.eye[2] - res.camera.center[2];
    m_pushConst.focalDistance = std::sqrt(dx * dx + dy * dy + dz * dz);
  }
  m_pushConst.frameCount   = res.frameCount;
  m_pushConst.flags        = (res.frameCount == 0 ? MI_PT_FIRST_FRAME : 0);
  m_pushConst.totalSamples = m_totalSamplesAccumulated;
  // pixelAngle = 2 |projInv[1][1]| / viewportHeight
  m_pushConst.pixelAngle = 2.0f * std::fabs(res.frameInfo.projInv[5]) / std::max(float(renderingSize.height), 1.0f);
}

void PathTracer::updateStatistics()
{
  m_totalSamplesAccumulated += m_pushConst.numSamples;
}

void PathTracer::onRender(StreamHandle cmd, Resources& res)
{
  if(!m_pt)
    return;
  setupPushConstant(res, res.renderSize);
  mi_pt_set_frame_info(m_pt, &res.frameInfo);
  mi_pt_set_sky(m_pt, &res.skyParams);
  if(mi_pt_render_frame(m_pt, &m_pushConst, cmd) != MI_PT_OK)
  {
    m_error = mi_pt_last_error();
    fprintf(stderr, "PathTracer::onRender: %s\n", m_error.c_str());
    return;
  }
  updateStatistics();
}

bool PathTracer::readRendered(float* rgba) const
{
  return m_pt && mi_pt_read_accum(m_pt, rgba) == MI_PT_OK;
}
bool PathTracer::readSelection(uint32_t* ids) const
{
  return m_pt && mi_pt_read_selection(m_pt, ids) == MI_PT_OK;
}
