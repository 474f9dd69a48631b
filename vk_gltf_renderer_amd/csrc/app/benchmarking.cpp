// Headless timing + telemetry (see benchmarking.hpp).  Written from the LOG SCHEMA the reference documents, not from its source:
//   docs/benchmarking.md:27-47   the HEADLESS_START / HEADLESS_PROGRESS / HEADLESS_SUMMARY lines and what every field means
//   utils/benchmark/benchmark_results.py  the consumer: `BENCHMARK_JSON {...,"schema":1}` records with type headless_start /
//                                headless_progress / headless_summary (parse_headless_summary reads the summary's keys)
// Design here: a run is three time stamps (loop start, end of warm-up, now) and a frame counter; everything that gets printed is
// derived from them in one place (Window / Derived below) and goes through one field list that feeds both the text line and the
// JSON record, so the two cannot drift apart.
#include "benchmarking.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <string>
#include <vector>

namespace {

// One output field: printed as `name=value` in the text line and as `"name":value` in the JSON record.
struct Field
{
  std::string name, text, json;
};
std::string fmt(const char* f, double v)
{
  char b[64];
  snprintf(b, sizeof(b), f, v);
  return b;
}
Field whole(const char* name, long long v) { return {name, std::to_string(v), std::to_string(v)}; }
// `digits` decimals in the record (the consumer parses floats; whole values keep a ".0" so they stay floats), `textFmt` in the line
Field real(const char* name, double v, int digits, const char* textFmt)
{
  const double s = std::pow(10.0, digits), r = std::round(v * s) / s;
  std::string  j = (r == std::floor(r) && std::fabs(r) < 1e15) ? fmt("%.1f", r) : fmt("%.15g", r);
  return {name, fmt(textFmt, v), j};
}
void emitRecord(const char* type, std::vector<Field> fields)
{
  fields.push_back({"schema", "", "1"});
  fields.push_back({"type", "", std::string("\"") + type + "\""});
  // keys in alphabetical order, like the reference's JSON library prints them: logs of the two programs then diff cleanly
  std::sort(fields.begin(), fields.end(), [](const Field& a, const Field& b) { return a.name < b.name; });
  std::string line = "BENCHMARK_JSON {";
  for(size_t i = 0; i < fields.size(); ++i)
    line += (i ? ",\"" : "\"") + fields[i].name + "\":" + fields[i].json;
  printf("%s}\n", line.c_str());
  fflush(stdout);
}
std::string textOf(const std::vector<Field>& fields)
{
  std::string s;
  for(const Field& f : fields)
    s += " " + f.name + "=" + f.text;
  return s;
}

}  // namespace

void BenchmarkController::alignMaxFramesForHeadless(int& maxFrames, uint32_t headlessFrames)
{
  // docs/benchmarking.md:39: in headless mode --maxFrames is raised to at least --frames so that every app frame accumulates
  if(int64_t(maxFrames) >= int64_t(headlessFrames))
    return;
  printf("maxFrames (%d) is less than headless --frames (%u); setting maxFrames to %u\n", maxFrames, headlessFrames, headlessFrames);
  maxFrames = int(headlessFrames);
}

void BenchmarkController::beginHeadlessTimingIfNeeded(bool isHeadless, const HeadlessFrameInfo& info)
{
  if(!isHeadless || m_running)
    return;
  m_running     = true;
  m_loopStart   = Clock::now();
  m_warmupEnd   = m_loopStart;
  m_warmedUp    = false;
  m_framesDone  = 0;
  m_warmupCount = 0;
  m_lastLogMs   = 0.0;
  const std::vector<Field> f = {whole("frames", info.totalFrames), whole("maxFrames", info.maxFrames), whole("ptSamples", info.ptSamples)};
  printf("HEADLESS_START%s\n", textOf(f).c_str());
  emitRecord("headless_start", f);
}

void BenchmarkController::updateHeadlessProgressIfNeeded(const HeadlessFrameInfo& info)
{
  if(!m_running)
    return;
  const uint32_t done = ++m_framesDone;
  const double   ms   = msSince(m_loopStart);
  // the measured window opens once the warm-up frames are complete (docs/benchmarking.md:40: the first completed frame carries
  // one-time setup and is not charged to throughput)
  if(!m_warmedUp && done >= kWarmupFrames)
  {
    m_warmedUp    = true;
    m_warmupEnd   = Clock::now();
    m_warmupCount = done;
  }
  // a line for the first and the last frame, every 50th frame, and whenever 5 s went by without one (docs/benchmarking.md:27)
  const bool due = done == 1 || done >= info.totalFrames || done % kLogEveryFrames == 0 || ms - m_lastLogMs >= kLogEveryMs;
  if(!due)
    return;
  m_lastLogMs            = ms;
  const double percent   = info.totalFrames ? 100.0 * double(done) / double(info.totalFrames) : 0.0;
  const double perFrame  = ms / double(done);
  printf("HEADLESS_PROGRESS app_frame %u/%u (%.0f%%) elapsed_ms=%.1f ms_per_frame=%.2f\n", done, info.totalFrames, float(percent), ms, perFrame);
  emitRecord("headless_progress", {whole("app_frame", done), whole("frames", info.totalFrames), real("percent", float(percent), 3, "%.0f"),
                                   real("elapsed_ms", ms, 3, "%.1f"), real("ms_per_frame", perFrame, 3, "%.2f")});
}

void BenchmarkController::logHeadlessSummary(const HeadlessFrameInfo& info)
{
  if(!m_running)
    return;
  // ---- the two windows: the whole loop, and the part after the warm-up
  const double   totalMs  = msSince(m_loopStart);
  const uint32_t finished = std::min(m_framesDone, info.totalFrames);
  const bool     split    = m_warmedUp && finished >= m_warmupCount;
  const uint32_t warmup   = split ? m_warmupCount : 0u;
  const uint32_t measured = finished - warmup;
  const double   windowMs = !split ? totalMs : (measured ? msSince(m_warmupEnd) : 0.0);
  // ---- accumulation: every app frame up to maxFrames adds ptSamples samples per pixel (docs/benchmarking.md:43-44)
  const int spp          = std::max(info.ptSamples, 1);
  const int accumulating = std::min(int(info.totalFrames), std::max(info.maxFrames, 0));
  const int inWindow     = std::max(0, std::min(accumulating - int(warmup), int(measured)));
  // ---- rates over the measured window (docs/benchmarking.md:45-46)
  const double seconds = windowMs * 1e-3;
  const double pixels  = double(info.imageSize.width) * double(info.imageSize.height);
  const double msps    = seconds > 0.0 ? pixels * double(inWindow * spp) / seconds * 1e-6 : 0.0;
  const double sppRate = seconds > 0.0 ? double(inWindow * spp) / seconds : 0.0;

  const std::vector<Field> head = {whole("frames", info.totalFrames), whole("maxFrames", info.maxFrames), whole("ptSamples", info.ptSamples),
                                   whole("effective_spp", accumulating * spp), whole("measured_effective_spp", inWindow * spp)};
  const std::vector<Field> tail = {real("wall_ms", windowMs, 3, "%.3f"),
                                   real("ms_per_frame", measured ? windowMs / double(measured) : 0.0, 3, "%.3f"),
                                   real("total_wall_ms", totalMs, 3, "%.3f"),
                                   real("total_ms_per_frame", info.totalFrames ? totalMs / double(info.totalFrames) : 0.0, 3, "%.3f"),
                                   whole("warmup_frames", warmup),
                                   whole("measured_frames", measured),
                                   real("throughput_MSps", msps, 3, "%.3f"),
                                   real("spp_per_sec", sppRate, 2, "%.2f")};
  // the text line names the resolution WxH, the record splits it into two keys
  printf("HEADLESS_SUMMARY%s resolution=%ux%u%s\n", textOf(head).c_str(), info.imageSize.width, info.imageSize.height, textOf(tail).c_str());
  std::vector<Field> record = head;
  record.push_back(whole("resolution_w", info.imageSize.width));
  record.push_back(whole("resolution_h", info.imageSize.height));
  record.insert(record.end(), tail.begin(), tail.end());
  emitRecord("headless_summary", record);
}

void BenchmarkController::finishHeadlessTiming()
{
  m_running  = false;
  m_warmedUp = false;
}


// ---- scripted sequencer ------------------------------------------------------------------------------------------------------
// `ParameterSequence N "name" = { Timer "stage"; GPU; avg a; min b; max c; last d; CPU; avg a; ... }`, times in whole
// microseconds: the shape utils/benchmark/benchmark_results.py parse_benchmark() reads (its regular expressions are the contract).
void BenchmarkController::emitParameterSequence(const std::string& name, const std::vector<TimerStat>& timers)
{
  printf("ParameterSequence %u \"%s\" = {\n", m_sequenceId, name.c_str());
  for(const TimerStat& t : timers)
    printf(" Timer \"%s\"; GPU; avg %lld; min %lld; max %lld; last %lld; CPU; avg %lld; min %lld; max %lld; last %lld;\n", t.name.c_str(),
           (long long)std::llround(t.gpuAvg), (long long)std::llround(t.gpuMin), (long long)std::llround(t.gpuMax), (long long)std::llround(t.gpuLast),
           (long long)std::llround(t.cpuAvg), (long long)std::llround(t.cpuMin), (long long)std::llround(t.cpuMax), (long long)std::llround(t.cpuLast));
  printf("}\n");
  fflush(stdout);
}

void BenchmarkController::emitSequenceMemory(const std::vector<MemorySample>& samples)
{
  printf("BENCHMARK_ADV %u {\n", m_sequenceId);
  for(const MemorySample& m : samples)
    printf(" Memory %s; Host used \t%llu; Device Used \t%llu; Device Allocated \t%llu; (bytes)\n", m.category.c_str(), (unsigned long long)m.hostUsed,
           (unsigned long long)m.deviceUsed, (unsigned long long)m.deviceAllocated);
  printf("}\n");
  std::string list = "[";
  for(size_t i = 0; i < samples.size(); ++i)
  {
    const MemorySample& m = samples[i];
    list += std::string(i ? "," : "") + "{\"category\":\"" + m.category + "\",\"device_allocated\":" + std::to_string(m.deviceAllocated)
            + ",\"device_used\":" + std::to_string(m.deviceUsed) + ",\"host_used\":" + std::to_string(m.hostUsed) + "}";
  }
  list += "]";
  emitRecord("sequence_memory", {whole("id", (long long)m_sequenceId), Field{"memory", "", list}});
  ++m_sequenceId;
}
