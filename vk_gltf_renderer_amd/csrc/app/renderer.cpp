#include "renderer.hpp"

#include <zlib.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <stdexcept>
#include <cstdio>
#include <cstring>

GltfRenderer::GltfRenderer()
{
  mi_default_sky(&m_resources.skyParams);  // reference: skyParams = {} at src/renderer.cpp:1328
  mi_pt_default_tonemapper(&m_resources.tonemapperData, 1);  // reference: tonemapperData{.autoExposure = 1}, src/resources.hpp:212
}

GltfRenderer::~GltfRenderer()
{
  m_pathTracer.onDetach(m_resources);
  if(m_resources.scene)
    mi_scene_destroy(m_resources.scene);
  if(m_resources.hdrIbl)
    mi_hdr_destroy(m_resources.hdrIbl);
}

void GltfRenderer::registerParameters(ParameterRegistry* r)
{
  Settings& s = m_resources.settings;
  r->add("envSystem", "Environment: [Sky:0, HDR:1]", &m_envSystem);
  r->add("maxFrames", "Maximum number of iterations", &s.maxFrames);
  r->add("hdrEnvIntensity", "HDR environment intensity", &s.hdrEnvIntensity);
  r->add("hdrEnvRotation", "HDR environment rotation", &s.hdrEnvRotation);
  r->add("hdrBlur", "HDR environment blur", &s.hdrBlur);
  r->add("useSolidBackground", "Use a solid background color", &s.useSolidBackground);
  r->add("useInfinitePlane", "Ground plane", &s.useInfinitePlane);
  r->add("isShadowCatcher", "Ground plane only catches shadows", &s.isShadowCatcher);
  r->add("infinitePlaneDistance", "Ground plane height", &s.infinitePlaneDistance);
  r->add("device", "HIP device ordinal", &m_resources.device);
  r->add("recomputeTangents", "Recreate all tangents after loading: [off:0, UV gradient:1, MikkTSpace:2]", &m_recomputeTangents);
  // tonemapper (reference: src/renderer.cpp:173-179 -- same names, same members)
  MiTonemapperData& tm = m_resources.tonemapperData;
  r->add("tmMethod", "Tonemapper method: [Filmic:0, Uncharted:1, Clip:2, ACES:3, AgX:4, KhronosPBR:5]", &tm.method);
  r->add("tmExposure", "Tonemapper exposure", &tm.exposure);
  r->add("tmGamma", "Tonemapper brightness", &tm.brightness);
  r->add("tmContrast", "Tonemapper contrast", &tm.contrast);
  r->add("tmSaturation", "Tonemapper saturation", &tm.saturation);
  r->add("tmWhitePoint", "Tonemapper vignette", &tm.vignette);
  r->add("tmAutoExposure", "Tonemapper auto exposure [0, 1]", &tm.autoExposure);
  // what the benchmark scripts set per sequence (utils/benchmark/*.cfg)
  r->add("renderSystem", "Renderer [Pathtracer:0, Rasterizer:1]; only the path tracer exists on this path", &m_seqRenderSystem);
  r->add("sequenceframes", "Sequencer: frames to run this step", &m_seqFrames);
  r->add("sequenceaverages", "Sequencer: frames averaged for the timer report", &m_seqAverages);
  r->add("sequenceresetframes", "Sequencer: warm-up frames after the parameter changes", &m_seqResetFrames);
  r->addCallback("gltfCamera", "Select the scene camera", 1, [this](const std::vector<std::string>& a) { m_gltfCamera = std::atoi(a[0].c_str()); selectCamera(m_gltfCamera); });
  r->addCallback("resetFrame", "Restart the accumulation", 0, [this](const std::vector<std::string>&) { resetFrame(); });
  r->addCallback("updateData", "Re-upload scene data (here: restart the accumulation)", 0, [this](const std::vector<std::string>&) { resetFrame(); });
  r->addCallback("fitScene", "Frame the scene (accepted; the scene camera is kept)", 0, [](const std::vector<std::string>&) {});
  m_pathTracer.registerParameters(r);
}

bool GltfRenderer::selectCamera(int index)
{
  if(!m_resources.scene || mi_scene_camera(m_resources.scene, index, &m_resources.camera) != MI_PT_OK)
    return false;
  resetFrame();
  return true;
}

int GltfRenderer::runSequences(const std::string& script, ParameterRegistry* registry)
{
  // tokens: whitespace separated, "quoted strings" kept whole, '#' starts a comment
  struct Sequence
  {
    std::string              name;
    std::vector<std::string> tokens;
  };
  std::vector<Sequence> sequences;
  {
    std::vector<std::string> tok;
    std::string              cur;
    bool                     quoted = false, comment = false, have = false;
    auto flush = [&] { if(have) tok.push_back(cur); cur.clear(); have = false; };
    for(char ch : script)
    {
      if(comment) { if(ch == '\n') comment = false; continue; }
      if(quoted) { if(ch == '"') quoted = false; else cur += ch; continue; }
      if(ch == '"') { quoted = true; have = true; continue; }
      if(ch == '#' && !have) { comment = true; continue; }
      if(ch == ' ' || ch == '\t' || ch == '\n' || ch == '\r') { flush(); continue; }
      cur += ch; have = true;
    }
    flush();
    for(size_t i = 0; i < tok.size(); ++i)
    {
      if(tok[i] == "SEQUENCE" && i + 1 < tok.size())
        sequences.push_back({tok[++i], {}});
      else if(!sequences.empty())
        sequences.back().tokens.push_back(tok[i]);
    }
  }
  if(sequences.empty())
  {
    fprintf(stderr, "benchmark script holds no SEQUENCE block\n");
    return 2;
  }
  for(const Sequence& sq : sequences)
  {
    m_seqFrames = 256; m_seqAverages = 64; m_seqResetFrames = 0;
    try
    {
      registry->parseTokens(sq.tokens);
    }
    catch(const std::exception& e)
    {
      fprintf(stderr, "SEQUENCE \"%s\": %s\n", sq.name.c_str(), e.what());
      return 2;
    }
    m_resources.settings.envSystem = m_envSystem == 1 ? EnvSystem::eHdr : EnvSystem::eSky;
    std::vector<BenchmarkController::TimerStat> timers;
    m_resources.settings.renderSystem = m_seqRenderSystem == 0 ? RenderingMode::ePathtracer : RenderingMode::eRasterizer;
    if(m_resources.settings.renderSystem != RenderingMode::ePathtracer)
      printf("SEQUENCE \"%s\": only the path tracer exists on this path, no timers\n", sq.name.c_str());
    else
    {
      std::vector<double> gpu, cpu;
      const int total = std::max(m_seqResetFrames, 0) + std::max(m_seqFrames, 1);
      for(int f = 0; f < total; ++f)
      {
        // a converged accumulation (frameCount == maxFrames) renders nothing: restart it, the sequence measures frames
        if(m_resources.frameCount + 1 >= m_resources.settings.maxFrames)
          resetFrame();
        mi_pt_enable_timing(m_pathTracer.handle(), 1);
        const auto t0 = std::chrono::steady_clock::now();
        onRender(nullptr, false, 0);
        const double    cpuUs = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        MiPtFrameTiming t{};
        mi_pt_get_frame_timing(m_pathTracer.handle(), &t);  // synchronises
        if(f >= std::max(m_seqResetFrames, 0))
        {
          gpu.push_back(double(t.totalMs) * 1000.0);
          cpu.push_back(cpuUs);
        }
      }
      mi_pt_enable_timing(m_pathTracer.handle(), 0);
      const size_t n = std::min(gpu.size(), size_t(std::max(m_seqAverages, 1)));
      BenchmarkController::TimerStat st;
      st.name = "PathTracer::onRender";  // the stage the benchmark scripts look for first (benchmark_results.py:274)
      st.gpuMin = st.cpuMin = 1e30;
      for(size_t i = gpu.size() - n; i < gpu.size(); ++i)
      {
        st.gpuAvg += gpu[i] / double(n); st.gpuMin = std::min(st.gpuMin, gpu[i]); st.gpuMax = std::max(st.gpuMax, gpu[i]);
        st.cpuAvg += cpu[i] / double(n); st.cpuMin = std::min(st.cpuMin, cpu[i]); st.cpuMax = std::max(st.cpuMax, cpu[i]);
      }
      st.gpuLast = gpu.back();
      st.cpuLast = cpu.back();
      timers.push_back(st);
    }
    m_benchmark.emitParameterSequence(sq.name, timers);
    // GltfRenderer::benchmarkMemorySamples (reference: src/renderer.cpp:530-555)
    MiPtMemory mem{};
    if(m_pathTracer.handle())
      mi_pt_get_memory(m_pathTracer.handle(), &mem);
    m_benchmark.emitSequenceMemory({{"Scene", 0, mem.sceneBytes, mem.sceneBytes}, {"PathTracer", 0, mem.rendererBytes, mem.rendererBytes}});
  }
  return 0;
}

bool GltfRenderer::createScene(const std::string& sceneFile)
{
  if(m_resources.scene)
    mi_scene_destroy(m_resources.scene);
  m_resources.scene = nullptr;
  if(mi_scene_load(sceneFile.c_str(), &m_resources.scene) != MI_PT_OK)
  {
    fprintf(stderr, "createScene: %s\n", mi_host_last_error());
    return false;
  }
  // the UI's "Recreate Tangents" / "Recreate Tangents - MikkTSpace" items as a start-up option (reference: src/ui_renderer.cpp:855-875)
  if(m_recomputeTangents == 1 || m_recomputeTangents == 2)
  {
    const int added = mi_scene_recompute_tangents(m_resources.scene, 1, m_recomputeTangents == 2 ? 1 : 0);
    if(added < 0)
      fprintf(stderr, "recomputeTangents: %s\n", mi_host_last_error());
    else if(m_recomputeTangents == 2)
      printf("MikkTSpace: %d vertices added for tangent discontinuities\n", added);
  }
  // addSceneCamerasToWidget: first glTF camera -> manipulator (reference: src/gltf_camera_utils.hpp:62-90)
  mi_scene_camera(m_resources.scene, 0, &m_resources.camera);
  resetFrame();
  m_pathTracer.onSceneInvalidated(m_resources);
  return m_pathTracer.handle() != nullptr;
}

bool GltfRenderer::createHDR(const std::string& hdrFile)
{
  if(m_resources.hdrIbl)
    mi_hdr_destroy(m_resources.hdrIbl);
  m_resources.hdrIbl = nullptr;
  if(hdrFile.empty())
    return true;
  if(mi_hdr_load(hdrFile.c_str(), &m_resources.hdrIbl) != MI_PT_OK)
  {
    fprintf(stderr, "createHDR: %s\n", mi_host_last_error());
    return false;
  }
  if(m_pathTracer.handle())
    mi_pt_set_environment(m_pathTracer.handle(), mi_hdr_env(m_resources.hdrIbl));
  return true;
}

void GltfRenderer::onAttach(const Extent2D& size)
{
  m_resources.renderSize = size;
  m_resources.settings.envSystem = m_envSystem == 1 ? EnvSystem::eHdr : EnvSystem::eSky;
  m_pathTracer.onAttach(m_resources, nullptr);
}

bool GltfRenderer::updateFrameCounter()
{
  const MiCamera& cur = m_resources.camera;
  if(!m_haveRefCamera || memcmp(&m_refCamera, &cur, sizeof(MiCamera)) != 0)
  {
    resetFrame();
    m_refCamera     = cur;
    m_haveRefCamera = true;
  }
  if(m_resources.frameCount >= m_resources.settings.maxFrames)
    return false;
  m_resources.frameCount++;
  return true;
}

BenchmarkController::HeadlessFrameInfo GltfRenderer::benchmarkFrameInfo(uint32_t frames) const
{
  BenchmarkController::HeadlessFrameInfo info;
  info.totalFrames = frames;
  info.maxFrames   = m_resources.settings.maxFrames;
  info.ptSamples   = m_pathTracer.m_pushConst.numSamples;
  info.imageSize   = m_resources.renderSize;
  return info;
}

void GltfRenderer::onRender(StreamHandle cmd, bool headless, uint32_t headlessFrames, int batch)
{
  m_benchmark.beginHeadlessTimingIfNeeded(headless, benchmarkFrameInfo(headlessFrames));
  // `batch` app frames in one go (headless runs, --framesInFlight): the library traces them as one set of wavefront launches and
  // folds them into the accumulator in frame order -- the image of `batch` successive onRender calls, bit for bit
  // (mi_pt_render_frames).  Only frames that would all accumulate (below maxFrames) are batched.
  batch = std::max(1, std::min(batch, m_resources.settings.maxFrames - (m_resources.frameCount + 1)));
  m_pathTracer.setFramesThisCall(batch);
  int done = 1;  // app frames this call stands for (a frame beyond maxFrames renders nothing but still counts as an app frame)
  if(updateFrameCounter())
  {
    // fill SceneFrameInfo (reference: src/renderer.cpp:675-705)
    const Settings& s = m_resources.settings;
    float           pixelAngle = 0, focal = 0;
    mi_camera_frame_info(&m_resources.camera, int(m_resources.renderSize.width), int(m_resources.renderSize.height), &m_resources.frameInfo, &pixelAngle,
                         &focal);
    MiSceneFrameInfo& f = m_resources.frameInfo;
    f.flags |= (s.useSolidBackground ? MI_SCENE_USE_SOLID_BACKGROUND : 0) | (s.envSystem == EnvSystem::eHdr ? MI_SCENE_USE_HDR_ENVIRONMENT : 0)
               | (s.useInfinitePlane ? MI_SCENE_USE_INFINITE_PLANE : 0)
               | ((s.useInfinitePlane && s.isShadowCatcher) ? MI_SCENE_INFINITE_PLANE_SHADOW_CATCHER : 0);
    f.envRotation  = s.hdrEnvRotation;
    f.envBlur      = s.hdrBlur;
    f.envIntensity = s.hdrEnvIntensity;
    memcpy(f.backgroundColor, s.solidBackgroundColor, sizeof(f.backgroundColor));
    f.infinitePlaneDistance = s.infinitePlaneDistance;
    memcpy(f.infinitePlaneBaseColor, s.infinitePlaneBaseColor, sizeof(f.infinitePlaneBaseColor));
    f.infinitePlaneMetallic     = s.infinitePlaneMetallic;
    f.infinitePlaneRoughness    = s.infinitePlaneRoughness;
    f.shadowCatcherDarkenAmount = std::max(s.shadowCatcherDarkness, 0.0f);
    m_resources.skyParams.yIsUp = m_resources.camera.up[1] > 0.5f;  // reference: src/renderer.cpp:707
    m_pathTracer.onRender(cmd, m_resources);
    done = m_pathTracer.framesLastCall();
    m_resources.frameCount += done - 1;
    if(headless)
      mi_pt_synchronize(m_pathTracer.handle());  // the reference's headless loop waits for each frame's submission
  }
  if(headless)
    for(int i = 0; i < done; ++i)
      m_benchmark.updateHeadlessProgressIfNeeded(benchmarkFrameInfo(headlessFrames));
}

void GltfRenderer::onLastHeadlessFrame(uint32_t headlessFrames)
{
  m_benchmark.logHeadlessSummary(benchmarkFrameInfo(headlessFrames));
  if(m_pathTracer.adaptiveSampling())  // (our own line: the summary's effective_spp assumes a fixed --ptSamples, like the reference's)
    printf("ADAPTIVE_SAMPLING samples_per_frame_at_end=%d total_samples=%d\n", m_pathTracer.m_pushConst.numSamples, m_pathTracer.totalSamples());
  m_benchmark.finishHeadlessTiming();
  saveHeadlessOutputImage();
}

bool GltfRenderer::savePng(const std::string& path, const unsigned char* rgba8, int w, int h)
{
  std::vector<unsigned char> raw(size_t(h) * (size_t(w) * 4 + 1));
  for(int y = 0; y < h; ++y)
  {
    raw[size_t(y) * (size_t(w) * 4 + 1)] = 0;
    memcpy(&raw[size_t(y) * (size_t(w) * 4 + 1) + 1], rgba8 + size_t(y) * size_t(w) * 4, size_t(w) * 4);
  }
  uLongf                     clen = compressBound(uLong(raw.size()));
  std::vector<unsigned char> comp(clen);
  if(compress2(comp.data(), &clen, raw.data(), uLong(raw.size()), 6) != Z_OK)
    return false;
  FILE* f = fopen(path.c_str(), "wb");
  if(!f)
    return false;
  auto be32 = [](unsigned v, unsigned char* p) { p[0] = (unsigned char)(v >> 24); p[1] = (unsigned char)(v >> 16); p[2] = (unsigned char)(v >> 8); p[3] = (unsigned char)v; };
  auto chunk = [&](const char* tag, const unsigned char* data, unsigned len) {
    unsigned char hdr[8];
    be32(len, hdr);
    memcpy(hdr + 4, tag, 4);
    fwrite(hdr, 1, 8, f);
    if(len)
      fwrite(data, 1, len, f);
    uLong crc = crc32(0, reinterpret_cast<const Bytef*>(tag), 4);
    if(len)
      crc = crc32(crc, data, len);
    unsigned char c[4];
    be32(unsigned(crc), c);
    fwrite(c, 1, 4, f);
  };
  const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  fwrite(sig, 1, 8, f);
  unsigned char ihdr[13];
  be32(unsigned(w), ihdr);
  be32(unsigned(h), ihdr + 4);
  ihdr[8] = 8; ihdr[9] = 6; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;
  chunk("IHDR", ihdr, 13);
  chunk("IDAT", comp.data(), unsigned(clen));
  chunk("IEND", nullptr, 0);
  fclose(f);
  return true;
}

bool GltfRenderer::saveHdr(const std::string& path, const float* rgba, int w, int h)
{
  FILE* f = fopen(path.c_str(), "wb");
  if(!f)
    return false;
  fprintf(f, "#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n", h, w);
  // Flat (uncompressed) RGBE scanlines.  Readers take a scanline that STARTS with the bytes 2, 2, <128 for a new-style run-length
  // header when 8 <= width < 32768, so such a first pixel is nudged (green mantissa 2 -> 3: one part in 256 of one pixel).
  auto clean = [](float v) { return (v > 0.0f && std::isfinite(v)) ? v : 0.0f; };  // negative / NaN / inf components would be undefined to convert
  std::vector<unsigned char> row(size_t(w) * 4);
  for(int y = 0; y < h; ++y)
  {
    for(int x = 0; x < w; ++x)
    {
      const float* p = rgba + (size_t(y) * size_t(w) + size_t(x)) * 4;
      const float  r = clean(p[0]), g = clean(p[1]), b = clean(p[2]);
      float        m = std::max(r, std::max(g, b));
      unsigned char* o = &row[size_t(x) * 4];
      if(m < 1e-32f)
        o[0] = o[1] = o[2] = o[3] = 0;
      else
      {
        int   e;
        float s = std::frexp(m, &e) * 256.0f / m;
        auto  q = [&](float v) { return (unsigned char)std::min(255.0f, v * s); };
        o[0] = q(r); o[1] = q(g); o[2] = q(b); o[3] = (unsigned char)std::clamp(e + 128, 0, 255);
      }
    }
    if(w >= 8 && w < 32768 && row[0] == 2 && row[1] == 2 && row[2] < 128)
      row[1] = 3;
    fwrite(row.data(), 1, row.size(), f);
  }
  fclose(f);
  return true;
}

void GltfRenderer::saveHeadlessOutputImage()
{
  const int w = int(m_resources.renderSize.width), h = int(m_resources.renderSize.height);
  if(w <= 0 || !m_pathTracer.handle())
    return;
  std::string out = m_resources.headlessOutputPath.empty() ? std::string("mi_gltf_renderer.png") : m_resources.headlessOutputPath;
  if(out.size() > 4 && out.substr(out.size() - 4) == ".hdr")
  {  // eImgRendered as it is (reference: src/ui_renderer.cpp:1187-1195)
    std::vector<float> rgba(size_t(w) * size_t(h) * 4);
    if(!m_pathTracer.readRendered(rgba.data()) || !saveHdr(out, rgba.data(), w, h))
      return;
  }
  else
  {  // eImgTonemapped: GltfRenderer::tonemap on the device (reference: src/renderer.cpp:557-573, :992-1056)
    std::vector<unsigned char> ldr(size_t(w) * size_t(h) * 4);
    // the denoised image is shown in place of the rendered one when there is one (reference: src/renderer.cpp:1006-1016); a
    // headless run with the denoiser on denoises its final frame if the cadence did not land on it
    if(m_pathTracer.isDenoiserEnabled() && !m_pathTracer.denoisedIsCurrent())
      m_pathTracer.denoiseOneShot();
    const int source = (m_pathTracer.isDenoiserEnabled() && m_pathTracer.hasValidDenoisedOutput()) ? 1 : 0;
    if(source)
      printf("DENOISER passes=%d final_image=denoised\n", m_pathTracer.denoiseCount());
    if(mi_pt_tonemap(m_pathTracer.handle(), &m_resources.tonemapperData, source, -1.0f, ldr.data(), nullptr) != MI_PT_OK)
    {
      fprintf(stderr, "tonemap: %s\n", mi_pt_last_error());
      return;
    }
    for(size_t i = 3; i < ldr.size(); i += 4)
      ldr[i] = 255;  // saved opaque, like the reference's screenshot path
    if(!savePng(out, ldr.data(), w, h))
      return;
  }
  printf("Saved headless output image: %s\n", out.c_str());
}
