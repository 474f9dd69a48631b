// Abstract renderer plugin interface — the drop-in boundary of the path-trace mode.
// Same ten virtuals as the reference's `class BaseRenderer` (src/renderer_base.hpp:33-55) with the Vulkan types
// replaced by their HIP-side meaning: VkCommandBuffer -> hipStream_t (as void*), VkExtent2D -> Extent2D,
// nvvk::ProfilerGpuTimer* -> the C-ABI's own HIP-event timing (mi_pt_enable_timing).
#pragma once
#include <cstdint>

struct Resources;
struct Extent2D
{
  uint32_t width = 0, height = 0;
};
using StreamHandle = void*;  // a hipStream_t; nullptr = default stream

class BaseRenderer
{
public:
  BaseRenderer()          = default;
  virtual ~BaseRenderer() = default;

  virtual void onAttach(Resources& /*resources*/, void* profiler) { m_profiler = profiler; }
  virtual void onDetach(Resources& /*resources*/) {}
  virtual void onResize(StreamHandle /*cmd*/, const Extent2D& /*size*/, Resources& /*resources*/) {}
  virtual void onRender(StreamHandle /*cmd*/, Resources& /*resources*/) {}
  virtual void onUIMenu() {}
  virtual void onSceneInvalidated(Resources& /*resources*/) {}
  [[nodiscard]] virtual bool onUIRender(Resources&) { return false; }

  virtual void compileShader(Resources& /*resources*/, bool /*fromFile*/ = true) {}
  virtual void createPipeline(Resources& /*resources*/) {}
  virtual void freeRecordCommandBuffer(Resources& /*resources*/) {}

protected:
  void* m_profiler{nullptr};
};
