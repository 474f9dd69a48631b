// GltfRenderer — the slice of the reference's application element (src/renderer.{hpp,cpp}) that feeds the path tracer in
// headless runs: scene/HDR creation, per-frame SceneFrameInfo, frame counter, accumulation reset, tonemap + image save.
#pragma once
#include <string>
#include <vector>

#include "benchmarking.hpp"
#include "parameter_registry.hpp"
#include "renderer_pathtracer.hpp"
#include "resources.hpp"

class GltfRenderer
{
public:
  GltfRenderer();
  ~GltfRenderer();
  void registerParameters(ParameterRegistry* registry);
  bool createScene(const std::string& sceneFile);  // reference: src/renderer.cpp:1238
  bool createHDR(const std::string& hdrFile);      // reference: src/renderer.cpp:1982
  void onAttach(const Extent2D& size);
  void onRender(StreamHandle cmd, bool headless, uint32_t headlessFrames, int batch = 1);  // reference: src/renderer.cpp:588-742
  void onLastHeadlessFrame(uint32_t headlessFrames);                        // reference: src/renderer.cpp:762-767
  bool updateAnimation();                                                    // reference: :2065-2170
  void resetFrame() { m_resources.frameCount = -1; }                        // reference: :1939-1942
  // Scripted benchmark sequences (reference: nvutils::ParameterSequencer driven from src/main.cpp:85-160 with --benchmark 1
  // --sequencefile / --sequencestring; docs/benchmarking.md "Scripted sequencer"): every `SEQUENCE "name"` block sets parameters
  // through the registry, renders --sequenceresetframes warm-up frames and --sequenceframes measured ones, and logs the timer
  // block (averaged over the last --sequenceaverages frames) and the memory snapshot the benchmark scripts parse.
  int  runSequences(const std::string& script, ParameterRegistry* registry);
  bool selectCamera(int index);  // --gltfCamera
  int  renderSystemIndex() const { return m_seqRenderSystem; }
  Resources&   resources() { return m_resources; }
  PathTracer&  pathTracer() { return m_pathTracer; }
  BenchmarkController& benchmark() { return m_benchmark; }

  // save helpers: PNG via zlib, Radiance .hdr dump (the tonemapper itself runs on the device: mi_pt_tonemap)
  static bool savePng(const std::string& path, const unsigned char* rgba8, int w, int h);
  static bool saveJpg(const std::string& path, const unsigned char* rgba8, int w, int h, int quality);
  static bool saveHdr(const std::string& path, const float* rgba, int w, int h);

private:
  bool updateFrameCounter();  // reference: src/renderer.cpp:1959-1977
  BenchmarkController::HeadlessFrameInfo benchmarkFrameInfo(uint32_t frames) const;
  void saveHeadlessOutputImage();

  Resources           m_resources;
  PathTracer          m_pathTracer;
  BenchmarkOptions    m_benchmarkOptions;
  BenchmarkController m_benchmark{m_benchmarkOptions};
  MiCamera            m_refCamera{};
  bool                m_haveRefCamera{false};
  int                 m_envSystem{0};
  int                 m_recomputeTangents{0};
  bool                m_useOpacityMicromap{true};
  int                 m_alphaCut{4};
  int                 m_animClip{-1};
  float               m_animTime{0.0f};
  // sequencer state (set through the registry by the script)
  int                 m_seqFrames{256}, m_seqAverages{64}, m_seqResetFrames{0}, m_seqRenderSystem{0}, m_gltfCamera{0};
  bool                m_seqFlag{false};
};
