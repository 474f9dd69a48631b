// class PathTracer : BaseRenderer — the host plugin of the path-trace mode, re-authored on top of the C-ABI
// (reference: src/renderer_pathtracer.{hpp,cpp}; Vulkan pipelines / Slang variants / DLSS / OptiX are not carried over).
#pragma once
#include <string>

#include "mi_pt.h"
#include "parameter_registry.hpp"
#include "renderer_base.hpp"
#include "resources.hpp"

class PathTracer : public BaseRenderer
{
public:
  PathTracer();
  ~PathTracer() override;

  void onAttach(Resources& resources, void* profiler) override;  // creates the device scene + BVH (SceneVk/SceneRtx role)
  void onDetach(Resources& resources) override;
  void onResize(StreamHandle cmd, const Extent2D& size, Resources& resources) override;
  void onRender(StreamHandle cmd, Resources& resources) override;  // ONE frame: numSamples spp into eImgRendered
  void onSceneInvalidated(Resources& resources) override;
  [[nodiscard]] bool onUIRender(Resources&) override { return false; }

  void registerParameters(ParameterRegistry* registry);  // reference: src/renderer_pathtracer.cpp:116-140
  [[nodiscard]] bool isDlssEnabled() const { return false; }

  // read-backs used by the headless save path (reference: src/renderer.cpp:557-573)
  bool readRendered(float* rgba) const;
  bool readSelection(uint32_t* ids) const;
  const std::string& lastError() const { return m_error; }
  MiPt* handle() const { return m_pt; }
  bool  adaptiveSampling() const { return m_adaptiveSampling; }
  bool  collectsCounters() const { return m_collectCounters; }
  int   totalSamples() const { return m_totalSamplesAccumulated; }

  // The denoiser that takes the OptiX adapter's place (reference: OptiXDenoiser::Settings, src/optix_denoiser.hpp:134-140; same
  // parameter names): off by default, auto-denoise at frames 50, 100, 150, ... once enabled.
  struct DenoiserSettings
  {
    bool enable{false};
    bool autoDenoiseEnabled{true};
    int  autoDenoiseInterval{50};
    int  method{1};  // ours: 0 = plain a-trous (mi_pt_denoise), 1 = variance-guided (mi_pt_denoise_svgf)
  };
  bool isDenoiserEnabled() const { return m_denoiser.enable; }
  bool hasValidDenoisedOutput() const { return m_hasDenoisedOutput; }  // reference: src/optix_denoiser.hpp:196
  bool denoiseOneShot();                                               // reference: OptiXDenoiser::denoiseOneShot
  int  denoiseCount() const { return m_denoiseCount; }
  bool denoisedIsCurrent() const { return m_hasDenoisedOutput && m_denoisedAtSamples == m_totalSamplesAccumulated; }

  // frames the next onRender traces in one go (1 = the reference's behaviour); what it actually did (1 whenever the adaptive
  // controller or the denoiser cadence need every frame's boundary)
  void setFramesThisCall(int n) { m_framesThisCall = n < 1 ? 1 : n; }
  int  framesLastCall() const { return m_framesLastCall; }

  MiPathtraceParams m_pushConst{};  // read by benchmarkFrameInfo() in the reference (src/renderer.cpp:526)

private:
  void setupPushConstant(Resources& resources, const Extent2D& renderingSize);  // reference: :1496-1574
  void updateStatistics();                                                      // reference: :1377-1402
  void updateAdaptiveSampling(Resources& resources);                            // reference: :1326-1374
  void updateDenoiser(Resources& resources);                                    // reference: src/optix_denoiser.cpp:751-805
  // reference: src/renderer_pathtracer.hpp:167-195 (Interactive 60 / Balanced 30 / Quality 15 / MaxQuality 10 frames per second)
  double targetFrameTimeMs() const
  {
    static const double fps[4] = {60.0, 30.0, 15.0, 10.0};
    return 1000.0 / fps[(m_performanceTarget >= 0 && m_performanceTarget < 4) ? m_performanceTarget : 1];
  }
  static constexpr int kMinSamplesPerPixel = 1, kMaxSamplesPerPixel = 100;

  MiPt*       m_pt{nullptr};
  bool        m_autoFocus{true};           // reference: src/renderer_pathtracer.hpp:89
  bool        m_adaptiveSampling{true};    // reference default: on, until --ptSamples is given (src/renderer_pathtracer.hpp:161)
  int         m_performanceTarget{1};      // Balanced
  bool        m_collectCounters{false};    // --ptCounters (our own): MiPtCreateOptions::collectCounters
  double      m_lastFrameDeviceMs{0.0};
  int         m_totalSamplesAccumulated{0};
  int         m_framesThisCall{1}, m_framesLastCall{1};
  DenoiserSettings m_denoiser;
  int         m_lastAutoDenoiseFrame{0};
  int         m_denoiseCount{0};
  int         m_denoisedAtSamples{-1};
  bool        m_hasDenoisedOutput{false};
  std::string m_error;
};
