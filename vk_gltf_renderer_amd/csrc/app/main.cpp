// Headless entry of the path-trace mode — the CLI subset of the reference's src/main.cpp:70-462 that a benchmark run uses:
//   mi_gltf_renderer --headless --size 1920 1080 --scenefile scene.glb --hdrfile std_env.hdr --frames N --maxFrames N
//                    --ptSamples S --ptAdaptiveSampling 0 --renderSystem 0 --envSystem 1 [--output out.png]
// (docs/benchmarking.md:16-23).  Positional arguments ending in .gltf/.glb/.hdr are accepted like in the reference.
#include <cstdio>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <vector>
#include <cstring>
#include <stdexcept>
#include <string>

#include "renderer.hpp"

int main(int argc, char** argv)
{
  GltfRenderer      app;
  ParameterRegistry registry;
  std::string       sceneFile, hdrFile = "std_env.hdr", outputFile;
  int               size[2] = {1280, 720};
  std::string       sequenceFile, sequenceString, saveSelftest;
  int               frames = 1, framesInFlight = 32;
  bool              headless = false, vvl = false, selftest = false, benchmark = false;
  registry.add("scenefile", "Input scene filename (.gltf / .glb)", &sceneFile);
  registry.add("hdrfile", "Input HDR filename", &hdrFile);
  registry.add("output", "Headless output image (.png or .hdr)", &outputFile);
  registry.addVec2("size", "Render size: width height", size);
  registry.add("frames", "Number of frames to render in headless mode", &frames);
  registry.add("headless", "Run without a window", &headless, true);
  registry.add("framesInFlight", "Headless: app frames traced as one set of launches (same image as one by one; 1 = like the reference)", &framesInFlight);
  registry.add("benchmark", "Benchmark mode: run the scripted sequences of --sequencefile / --sequencestring", &benchmark);
  registry.add("sequencefile", "Benchmark script (.cfg) with SEQUENCE blocks", &sequenceFile);
  registry.add("sequencestring", "Benchmark script given on the command line", &sequenceString);
  registry.add("vvl", "accepted and ignored (Vulkan validation layers)", &vvl, true);
  registry.add("saveSelftest", "Write <prefix>.png and <prefix>.jpg of a synthetic image (writer check, no GPU needed)", &saveSelftest);
  registry.add("benchmarkSelftest", "Print a fabricated headless log (format check, no GPU needed)", &selftest, true);
  app.registerParameters(&registry);
  std::vector<std::string> positional;
  try
  {
    positional = registry.parse(argc, argv);
  }
  catch(const std::exception& e)
  {
    fprintf(stderr, "%s\nusage:\n%s", e.what(), registry.usage().c_str());
    return 2;
  }
  for(const std::string& p : positional)
  {
    if(p.size() > 4 && p.substr(p.size() - 4) == ".hdr")
      hdrFile = p;
    else
      sceneFile = p;
  }
  if(!saveSelftest.empty())
  {
    const int                  w = 83, h = 61;  // not multiples of 8: exercises the edge blocks
    std::vector<unsigned char> img(size_t(w) * h * 4);
    for(int y = 0; y < h; ++y)
      for(int x = 0; x < w; ++x)
      {
        unsigned char* p = &img[(size_t(y) * w + x) * 4];
        p[0] = (unsigned char)(127 + 120 * std::sin(x / 9.0));
        p[1] = (unsigned char)(127 + 120 * std::cos(y / 7.0 + x / 23.0));
        p[2] = (unsigned char)((x * 3 + y * 2) % 256);
        p[3] = 255;
      }
    return (GltfRenderer::savePng(saveSelftest + ".png", img.data(), w, h) && GltfRenderer::saveJpg(saveSelftest + ".jpg", img.data(), w, h, 90)) ? 0 : 1;
  }
  if(selftest)
  {
    BenchmarkController::HeadlessFrameInfo info;
    info.totalFrames = 3; info.maxFrames = 3; info.ptSamples = 2; info.imageSize = {64, 32};
    app.benchmark().beginHeadlessTimingIfNeeded(true, info);
    for(int i = 0; i < 3; ++i)
      app.benchmark().updateHeadlessProgressIfNeeded(info);
    app.benchmark().logHeadlessSummary(info);
    return 0;
  }
  if(benchmark)
  {
    // scripted sequencer (reference: src/main.cpp:138-160; "Benchmark mode requires --sequencefile or --sequencestring")
    std::string script = sequenceString;
    if(!sequenceFile.empty())
    {
      FILE* f = fopen(sequenceFile.c_str(), "rb");
      if(!f)
      {
        fprintf(stderr, "cannot read %s\n", sequenceFile.c_str());
        return 2;
      }
      char   buf[4096];
      size_t n;
      while((n = fread(buf, 1, sizeof(buf), f)) > 0)
        script.append(buf, n);
      fclose(f);
    }
    if(script.empty() || sceneFile.empty())
    {
      fprintf(stderr, "Benchmark mode requires --scenefile and --sequencefile or --sequencestring\n");
      return 2;
    }
    app.onAttach(Extent2D{uint32_t(size[0]), uint32_t(size[1])});
    if(!app.createScene(sceneFile))
      return 1;
    app.createHDR(hdrFile);
    return app.runSequences(script, &registry);
  }
  if(!headless || app.renderSystemIndex() != 0)
  {
    fprintf(stderr, "only `--headless --renderSystem 0` (path tracer) and `--benchmark 1` (scripted sequences) are implemented\n");
    return 2;
  }
  if(sceneFile.empty())
  {
    fprintf(stderr, "no --scenefile given\n");
    return 2;
  }
  // alignMaxFramesForHeadless (reference: src/main.cpp:133-136)
  BenchmarkController::alignMaxFramesForHeadless(app.resources().settings.maxFrames, uint32_t(frames));
  app.resources().headlessOutputPath = outputFile;
  app.onAttach(Extent2D{uint32_t(size[0]), uint32_t(size[1])});
  if(!app.createScene(sceneFile))
    return 1;
  if(!app.createHDR(hdrFile) && app.resources().settings.envSystem == EnvSystem::eHdr)
    return 1;
  // frame 0 on its own (the benchmark's warm-up frame, docs/benchmarking.md:40), the rest in batches
  for(int f = 0; f < frames;)
  {
    const int batch = f == 0 ? 1 : std::max(1, std::min(framesInFlight, frames - f));
    const int before = app.resources().frameCount;
    const auto t0 = std::chrono::steady_clock::now();
    app.onRender(nullptr, true, uint32_t(frames), batch);  // (headless: returns after the batch's device work)
    const int done = std::max(1, app.resources().frameCount - before);
    // (our own line) wall time of every batch: the steady state can be read off without the first batches' one-off costs (allocations, first touch)
    printf("HEADLESS_BATCH first_frame=%d frames=%d ms=%.3f\n", f, done, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    f += done;
  }
  app.onLastHeadlessFrame(uint32_t(frames));
  if(app.pathTracer().collectsCounters() && app.pathTracer().handle())
  {
    MiPtStats st{};
    if(mi_pt_get_stats(app.pathTracer().handle(), &st) == MI_PT_OK)
    {
      const double secondary = double(st.segments) - double(st.cameraPaths);
      printf("HEADLESS_COUNTERS camera_paths=%llu segments=%llu shadow_rays=%llu nodes_closest=%llu tris_closest=%llu nodes_shadow=%llu tris_shadow=%llu nodes_primary=%llu bvh_nodes=%llu "
             "node_visits_per_secondary_ray=%.3f triangle_tests_per_secondary_ray=%.3f node_visits_per_shadow_ray=%.3f\n",
             (unsigned long long)st.cameraPaths, (unsigned long long)st.segments, (unsigned long long)st.shadowRays, (unsigned long long)st.nodesClosest,
             (unsigned long long)st.trisClosest, (unsigned long long)st.nodesShadow, (unsigned long long)st.trisShadow, (unsigned long long)st.nodesPrimary,
             (unsigned long long)st.bvhNodeCount, secondary > 0 ? double(st.nodesClosest) / secondary : 0.0, secondary > 0 ? double(st.trisClosest) / secondary : 0.0,
             st.shadowRays ? double(st.nodesShadow) / double(st.shadowRays) : 0.0);
    }
  }
  return 0;
}
