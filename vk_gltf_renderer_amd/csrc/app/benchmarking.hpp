// BenchmarkController — headless timing + telemetry, output-compatible with the reference
// (src/benchmarking.{hpp,cpp}: HEADLESS_START / HEADLESS_PROGRESS / HEADLESS_SUMMARY lines and
// `BENCHMARK_JSON {...,"schema":1}` records consumed by utils/benchmark/benchmark_results.py).
#pragma once
#include <chrono>
#include <cstdint>
#include <string>
#include <vector>

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

  // Scripted-sequencer telemetry (docs/benchmarking.md "Log parsing"; consumer: utils/benchmark/benchmark_results.py parse_benchmark):
  // the timer block of one sequence -- what the reference's ProfilerManager logs after each SEQUENCE -- and its memory snapshot
  // (legacy BENCHMARK_ADV block + BENCHMARK_JSON sequence_memory record; ids count up so the two can be joined).
  struct TimerStat  // microseconds
  {
    std::string name;
    double      gpuAvg{0}, gpuMin{0}, gpuMax{0}, gpuLast{0}, cpuAvg{0}, cpuMin{0}, cpuMax{0}, cpuLast{0};
  };
  struct MemorySample  // reference: src/benchmarking.hpp MemorySample
  {
    std::string category;
    uint64_t    hostUsed{0}, deviceUsed{0}, deviceAllocated{0};
  };
  void     emitParameterSequence(const std::string& name, const std::vector<TimerStat>& timers);
  void     emitSequenceMemory(const std::vector<MemorySample>& samples);
  uint32_t sequenceId() const { return m_sequenceId; }

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
  uint32_t          m_sequenceId{0};
};
