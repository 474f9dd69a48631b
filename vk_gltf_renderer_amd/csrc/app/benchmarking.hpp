// BenchmarkController — headless timing + telemetry, output-compatible with the reference
// (src/benchmarking.{hpp,cpp}: HEADLESS_START / HEADLESS_PROGRESS / HEADLESS_SUMMARY lines and
// `BENCHMARK_JSON {...,"schema":1}` records consumed by utils/benchmark/benchmark_results.py).
#pragma once
#include <chrono>
#include <cstdint>
#include <string>

#include "renderer_base.hpp"

struct BenchmarkOptions  // reference: src/benchmarking.hpp:45-53
{
  bool        enabled{false};
  int         gltfCameraIndex{0};
  std::string screenshotFilename;
};

class BenchmarkController
{
public:
  struct HeadlessFrameInfo  // reference: src/benchmarking.hpp:78-84
  {
    uint32_t totalFrames{0};
    int      maxFrames{0};
    int      ptSamples{1};
    Extent2D imageSize{};
  };
  explicit BenchmarkController(BenchmarkOptions& options) : m_options(options) {}
  [[nodiscard]] bool isBenchmarkMode() const { return m_options.enabled; }
  static void alignMaxFramesForHeadless(int& maxFrames, uint32_t headlessFrames);
  void beginHeadlessTimingIfNeeded(bool isHeadless, const HeadlessFrameInfo& info);
  void updateHeadlessProgressIfNeeded(const HeadlessFrameInfo& info);
  void logHeadlessSummary(const HeadlessFrameInfo& info);
  void finishHeadlessTiming();

private:
  // docs/benchmarking.md:27 ("every 50 frames or 5 seconds") and :40 ("the first completed frame is excluded")
  static constexpr uint32_t kLogEveryFrames = 50;
  static constexpr double   kLogEveryMs     = 5000.0;
  static constexpr uint32_t kWarmupFrames   = 1;
  using Clock = std::chrono::steady_clock;
  static double msSince(Clock::time_point t) { return std::chrono::duration<double, std::milli>(Clock::now() - t).count(); }

  BenchmarkOptions& m_options;
  // a headless run = the time the loop started, the time its warm-up ended, and how many frames have completed
  bool              m_running{false}, m_warmedUp{false};
  Clock::time_point m_loopStart{}, m_warmupEnd{};
  uint32_t          m_framesDone{0}, m_warmupCount{0};
  double            m_lastLogMs{0.0};
};
