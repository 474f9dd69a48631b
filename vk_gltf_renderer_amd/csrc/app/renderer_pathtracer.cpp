#include "renderer_pathtracer.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>

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
  // an explicit sample count switches the adaptive controller off (reference: src/renderer_pathtracer.cpp:120-123)
  r->addCallback("ptSamples", "Number of samples per pixel per frame (disables adaptive sampling)", 1, [this](const std::vector<std::string>& a) {
    m_pushConst.numSamples = std::atoi(a[0].c_str());
    m_adaptiveSampling     = false;
  });
  r->add("ptFireflyClamp", "Firefly clamp threshold", &m_pushConst.fireflyClampThreshold);
  r->add("ptTexGradScale", "Ray-footprint gradient scale", &m_pushConst.texGradScale);
  r->add("ptAperture", "Aperture for depth of field", &m_pushConst.aperture);
  r->add("ptFocalDistance", "Focal distance (disables auto focus)", &m_pushConst.focalDistance);
  r->add("ptAutoFocus", "Focus on the camera's interest point", &m_autoFocus);
  r->add("ptAdaptiveSampling", "Adjust the samples per pixel and frame to the performance target", &m_adaptiveSampling);
  // the denoiser keeps the OptiX adapter's switches (reference: src/optix_denoiser.cpp:735-741)
  r->add("optixEnable", "Denoiser: enable (HIP variance-guided a-trous in place of the OptiX denoiser)", &m_denoiser.enable);
  r->add("optixAutoDenoiseEnabled", "Denoiser: auto-denoise every N frames", &m_denoiser.autoDenoiseEnabled);
  r->add("optixAutoDenoiseInterval", "Denoiser: auto-denoise interval (frames)", &m_denoiser.autoDenoiseInterval);
  r->add("denoiseMethod", "Denoiser: [a-trous:0, variance-guided:1]", &m_denoiser.method);
  r->add("ptPerformanceTarget", "Performance target [Interactive:0, Balanced:1, Quality:2, MaxQuality:3]", &m_performanceTarget);
  // (our own) traversal / shading counters of the C-ABI: slower kernels, printed as HEADLESS_COUNTERS at the end of a headless run
  r->add("ptCounters", "Collect traversal counters (MiPtStats; slower) and print them at the end of a headless run", &m_collectCounters);
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
  opt.collectCounters = m_collectCounters ? 1 : 0;
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
  // the guide layers are captured while the denoiser is enabled (reference: USE_OPTIX_DENOISER flag, src/renderer_pathtracer.cpp:1534-1550)
  m_pushConst.flags        = (res.frameCount == 0 ? MI_PT_FIRST_FRAME : 0) | (m_denoiser.enable ? MI_PT_USE_OPTIX_DENOISER : 0);
  if(res.frameCount == 0)
    m_hasDenoisedOutput = false;
  m_pushConst.totalSamples = m_totalSamplesAccumulated;
  // pixelAngle = 2 |projInv[1][1]| / viewportHeight
  m_pushConst.pixelAngle = 2.0f * std::fabs(res.frameInfo.projInv[5]) / std::max(float(renderingSize.height), 1.0f);
}

// Adaptive sampling (reference: PathTracer::updateAdaptiveSampling, src/renderer_pathtracer.cpp:1326-1374): a one-step controller
// on the device time of the previous frame's path-trace pass -- more samples per frame while there is 20 % of headroom under the
// target, fewer when 10 % over it; 1 sample again whenever accumulation restarts; hands off for the first frames after a restart.
void PathTracer::updateAdaptiveSampling(Resources& res)
{
  if(!m_adaptiveSampling || !m_pt)
    return;
  if(res.frameCount == 0)
  {
    m_pushConst.numSamples = kMinSamplesPerPixel;
    return;
  }
  if(res.frameCount < 5 || m_lastFrameDeviceMs <= 0.0)
    return;
  const double target = targetFrameTimeMs();
  if(m_lastFrameDeviceMs < target * 0.8 && m_pushConst.numSamples < kMaxSamplesPerPixel)
    ++m_pushConst.numSamples;
  else if(m_lastFrameDeviceMs > target * 1.1 && m_pushConst.numSamples > kMinSamplesPerPixel)
    --m_pushConst.numSamples;
  m_pushConst.numSamples = std::min(std::max(m_pushConst.numSamples, kMinSamplesPerPixel), kMaxSamplesPerPixel);
}

bool PathTracer::denoiseOneShot()
{
  if(!m_pt)
    return false;
  const int rc = m_denoiser.method == 0 ? mi_pt_denoise(m_pt, 5, 0.6f, 64.0f, 0.2f, nullptr, nullptr) : mi_pt_denoise_svgf(m_pt, 5, 4.0f, 128.0f, 1.0f, nullptr, nullptr);
  if(rc != MI_PT_OK)
  {
    m_error = mi_pt_last_error();
    fprintf(stderr, "PathTracer::denoiseOneShot: %s\n", m_error.c_str());
    return false;
  }
  m_hasDenoisedOutput = true;
  m_denoisedAtSamples = m_totalSamplesAccumulated;
  ++m_denoiseCount;
  return true;
}

// Auto-denoise cadence: once per crossed multiple of the interval (every frame when the interval is 1); the tracking restarts
// with the accumulation.  Unlike the reference's (src/optix_denoiser.hpp:78-97) there is no one-frame lag: the pass is enqueued on
// the frame's stream behind the frame it denoises.
void PathTracer::updateDenoiser(Resources& res)
{
  if(!m_denoiser.enable || !m_denoiser.autoDenoiseEnabled || m_denoiser.autoDenoiseInterval <= 0)
    return;
  const int frame = res.frameCount + 1;  // frames accumulated so far
  if(frame < m_lastAutoDenoiseFrame)
    m_lastAutoDenoiseFrame = 0;
  const bool crossed = frame / m_denoiser.autoDenoiseInterval > m_lastAutoDenoiseFrame / m_denoiser.autoDenoiseInterval && frame % m_denoiser.autoDenoiseInterval == 0;
  if((crossed || m_denoiser.autoDenoiseInterval == 1) && denoiseOneShot())
    m_lastAutoDenoiseFrame = frame;
}

void PathTracer::updateStatistics()
{
  m_totalSamplesAccumulated += m_pushConst.numSamples * m_framesLastCall;
}

void PathTracer::onRender(StreamHandle cmd, Resources& res)
{
  if(!m_pt)
    return;
  updateAdaptiveSampling(res);  // reference order: src/renderer_pathtracer.cpp:552-553, before the push constants are set up
  setupPushConstant(res, res.renderSize);
  mi_pt_set_frame_info(m_pt, &res.frameInfo);
  mi_pt_set_sky(m_pt, &res.skyParams);
  if(m_adaptiveSampling)
    mi_pt_enable_timing(m_pt, 1);  // (re)starts the per-kernel event timers: the controller needs this frame's device time
  m_framesLastCall = (m_adaptiveSampling || m_denoiser.enable) ? 1 : m_framesThisCall;
  if((m_framesLastCall == 1 ? mi_pt_render_frame(m_pt, &m_pushConst, cmd) : mi_pt_render_frames(m_pt, &m_pushConst, m_framesLastCall, cmd)) != MI_PT_OK)
  {
    m_framesLastCall = 1;
    m_error = mi_pt_last_error();
    fprintf(stderr, "PathTracer::onRender: %s\n", m_error.c_str());
    return;
  }
  if(m_adaptiveSampling)
  {
    MiPtFrameTiming t{};
    if(mi_pt_get_frame_timing(m_pt, &t) == MI_PT_OK)  // (synchronises: the reference reads its GPU timer one frame late instead)
      m_lastFrameDeviceMs = t.totalMs;
  }
  updateStatistics();
  updateDenoiser(res);
}

bool PathTracer::readRendered(float* rgba) const
{
  return m_pt && mi_pt_read_accum(m_pt, rgba) == MI_PT_OK;
}
bool PathTracer::readSelection(uint32_t* ids) const
{
  return m_pt && mi_pt_read_selection(m_pt, ids) == MI_PT_OK;
}
