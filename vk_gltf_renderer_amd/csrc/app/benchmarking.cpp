// See benchmarking.hpp.  Formulas follow src/benchmarking.cpp:162-304 of the reference line by line; the JSON records carry
// the same keys (nlohmann::json prints object keys alphabetically, reproduced here so logs diff cleanly).
#include "benchmarking.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <map>
#include <sstream>

namespace {

double roundTo(double value, double scale) { return std::round(value * scale) / scale; }

struct JsonRecord
{
  std::map<std::string, std::string> kv;  // alphabetical, like nlohmann::json
  void str(const std::string& k, const std::string& v) { kv[k] = "\"" + v + "\""; }
  void num(const std::string& k, double v)
  {
    char buf[64];
    if(v == std::floor(v) && std::fabs(v) < 1e15)
      snprintf(buf, sizeof(buf), "%.1f", v);
    else
      snprintf(buf, sizeof(buf), "%.15g", v);
    kv[k] = buf;
  }
  void integer(const std::string& k, long long v) { kv[k] = std::to_string(v); }
  void emit()
  {
    integer("schema", 1);
    std::ostringstream o;
    o << "BENCHMARK_JSON {";
    bool first = true;
    for(const auto& e : kv)
    {
      o << (first ? "" : ",") << "\"" << e.first << "\":" << e.second;
      first = false;
    }
    o << "}";
    printf("%s\n", o.str().c_str());
    fflush(stdout);
  }
};

}  // namespace

void BenchmarkController::alignMaxFramesForHeadless(int& maxFrames, uint32_t headlessFrames)
{
  const int minMaxFrames = static_cast<int>(headlessFrames);
  if(maxFrames < minMaxFrames)
  {
    printf("maxFrames (%d) is less than headless --frames (%u); setting maxFrames to %u\n", maxFrames, headlessFrames, headlessFrames);
    maxFrames = minMaxFrames;
  }
}

void BenchmarkController::beginHeadlessTimingIfNeeded(bool isHeadless, const HeadlessFrameInfo& info)
{
  if(!isHeadless || m_headlessTimingActive)
    return;
  m_headlessWallTimer            = Clock::now();
  m_headlessMeasuredTimer        = Clock::now();
  m_headlessTimingActive         = true;
  m_headlessMeasuredTimingActive = false;
  m_headlessFramesDone           = 0;
  m_headlessMeasuredStartFrame   = 0;
  m_headlessLastProgressLogMs    = 0.0;
  printf("HEADLESS_START frames=%u maxFrames=%d ptSamples=%d\n", info.totalFrames, info.maxFrames, info.ptSamples);
  JsonRecord r;
  r.str("type", "headless_start");
  r.integer("frames", info.totalFrames);
  r.integer("maxFrames", info.maxFrames);
  r.integer("ptSamples", info.ptSamples);
  r.emit();
}

void BenchmarkController::updateHeadlessProgressIfNeeded(const HeadlessFrameInfo& info)
{
  if(!m_headlessTimingActive)
    return;
  ++m_headlessFramesDone;
  const double elapsedMs   = msSince(m_headlessWallTimer);
  const bool   onInterval  = (m_headlessFramesDone % kHeadlessLogEveryNFrames) == 0;
  const bool   onTime      = (elapsedMs - m_headlessLastProgressLogMs) >= kHeadlessLogMinIntervalMs;
  const bool   firstOrLast = m_headlessFramesDone == 1 || m_headlessFramesDone >= info.totalFrames;
  if(!m_headlessMeasuredTimingActive && m_headlessFramesDone >= kHeadlessWarmupFrames)
  {
    m_headlessMeasuredTimer        = Clock::now();
    m_headlessMeasuredTimingActive = true;
    m_headlessMeasuredStartFrame   = m_headlessFramesDone;
  }
  if(!firstOrLast && !onInterval && !onTime)
    return;
  const float  pct = info.totalFrames > 0 ? 100.0F * static_cast<float>(m_headlessFramesDone) / static_cast<float>(info.totalFrames) : 0.0F;
  const double msPerFrame = m_headlessFramesDone > 0 ? elapsedMs / static_cast<double>(m_headlessFramesDone) : 0.0;
  printf("HEADLESS_PROGRESS app_frame %u/%u (%.0f%%) elapsed_ms=%.1f ms_per_frame=%.2f\n", m_headlessFramesDone, info.totalFrames, pct, elapsedMs,
         msPerFrame);
  JsonRecord r;
  r.str("type", "headless_progress");
  r.integer("app_frame", m_headlessFramesDone);
  r.integer("frames", info.totalFrames);
  r.num("percent", roundTo(pct, 1000.0));
  r.num("elapsed_ms", roundTo(elapsedMs, 1000.0));
  r.num("ms_per_frame", roundTo(msPerFrame, 1000.0));
  r.emit();
  m_headlessLastProgressLogMs = elapsedMs;
}

void BenchmarkController::logHeadlessSummary(const HeadlessFrameInfo& info)
{
  if(!m_headlessTimingActive)
    return;
  const double   totalWallMs     = msSince(m_headlessWallTimer);
  const double   totalMsPerFrame = info.totalFrames > 0 ? totalWallMs / static_cast<double>(info.totalFrames) : 0.0;
  const uint32_t completedFrames = std::min(m_headlessFramesDone, info.totalFrames);
  uint32_t       warmupFrames = 0, measuredFrames = completedFrames;
  double         measuredWallMs = totalWallMs;
  if(m_headlessMeasuredTimingActive && completedFrames >= m_headlessMeasuredStartFrame)
  {
    warmupFrames   = m_headlessMeasuredStartFrame;
    measuredFrames = completedFrames - m_headlessMeasuredStartFrame;
    measuredWallMs = measuredFrames > 0 ? msSince(m_headlessMeasuredTimer) : 0.0;
  }
  const double measuredWallSec    = measuredWallMs / 1000.0;
  const double measuredMsPerFrame = measuredFrames > 0 ? measuredWallMs / static_cast<double>(measuredFrames) : 0.0;
  const int    accumFrames        = std::min(static_cast<int>(info.totalFrames), std::max(info.maxFrames, 0));
  const int    measuredAccumFrames = std::clamp(accumFrames - static_cast<int>(warmupFrames), 0, static_cast<int>(measuredFrames));
  const int    effectiveSpp         = accumFrames * std::max(info.ptSamples, 1);
  const int    measuredEffectiveSpp = measuredAccumFrames * std::max(info.ptSamples, 1);
  const uint64_t pixels = static_cast<uint64_t>(info.imageSize.width) * static_cast<uint64_t>(info.imageSize.height);
  const double measuredSamples = static_cast<double>(pixels) * static_cast<double>(measuredEffectiveSpp);
  const double throughputMSps  = measuredWallSec > 0.0 ? measuredSamples / measuredWallSec / 1e6 : 0.0;
  const double sppPerSec       = measuredWallSec > 0.0 ? static_cast<double>(measuredEffectiveSpp) / measuredWallSec : 0.0;
  printf("HEADLESS_SUMMARY frames=%u maxFrames=%d ptSamples=%d effective_spp=%d measured_effective_spp=%d "
         "resolution=%ux%u wall_ms=%.3f ms_per_frame=%.3f total_wall_ms=%.3f total_ms_per_frame=%.3f "
         "warmup_frames=%u measured_frames=%u throughput_MSps=%.3f spp_per_sec=%.2f\n",
         info.totalFrames, info.maxFrames, info.ptSamples, effectiveSpp, measuredEffectiveSpp, info.imageSize.width, info.imageSize.height,
         measuredWallMs, measuredMsPerFrame, totalWallMs, totalMsPerFrame, warmupFrames, measuredFrames, throughputMSps, sppPerSec);
  JsonRecord r;
  r.str("type", "headless_summary");
  r.integer("frames", info.totalFrames);
  r.integer("maxFrames", info.maxFrames);
  r.integer("ptSamples", info.ptSamples);
  r.integer("effective_spp", effectiveSpp);
  r.integer("measured_effective_spp", measuredEffectiveSpp);
  r.integer("resolution_w", info.imageSize.width);
  r.integer("resolution_h", info.imageSize.height);
  r.num("wall_ms", roundTo(measuredWallMs, 1000.0));
  r.num("ms_per_frame", roundTo(measuredMsPerFrame, 1000.0));
  r.num("total_wall_ms", roundTo(totalWallMs, 1000.0));
  r.num("total_ms_per_frame", roundTo(totalMsPerFrame, 1000.0));
  r.integer("warmup_frames", warmupFrames);
  r.integer("measured_frames", measuredFrames);
  r.num("throughput_MSps", roundTo(throughputMSps, 1000.0));
  r.num("spp_per_sec", roundTo(sppPerSec, 100.0));
  r.emit();
}

void BenchmarkController::finishHeadlessTiming()
{
  m_headlessTimingActive         = false;
  m_headlessMeasuredTimingActive = false;
}
