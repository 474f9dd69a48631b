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
  static constexpr uint32_t kHeadlessLogEveryNFrames  = 50;      // reference: src/benchmarking.hpp:126-128
  static constexpr double   kHeadlessLogMinIntervalMs = 5000.0;
  static constexpr uint32_t kHeadlessWarmupFrames     = 1;
  using Clock = std::chrono::steady_clock;
  static double msSince(Clock::time_point t) { return std::chrono::duration<double, std::milli>(Clock::now() - t).count(); }

  BenchmarkOptions& m_options;
  Clock::time_point m_headlessWallTimer{}, m_headlessMeasuredTimer{};
  bool              m_headlessTimingActive{false}, m_headlessMeasuredTimingActive{false};
  uint32_t          m_headlessFramesDone{0}, m_headlessMeasuredStartFrame{0};
  double            m_headlessLastProgressLogMs{0.0};
};
