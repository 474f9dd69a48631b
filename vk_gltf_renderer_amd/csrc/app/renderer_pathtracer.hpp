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
  int   totalSamples() const { return m_totalSamplesAccumulated; }

  MiPathtraceParams m_pushConst{};  // read by benchmarkFrameInfo() in the reference (src/renderer.cpp:526)

private:
  void setupPushConstant(Resources& resources, const Extent2D& renderingSize);  // reference: :1496-1574
  void updateStatistics();                                                      // reference: :1377-1402
  void updateAdaptiveSampling(Resources& resources);                            // reference: :1326-1374
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
  double      m_lastFrameDeviceMs{0.0};
  int         m_totalSamplesAccumulated{0};
  std::string m_error;
};
