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
  // the reference consumes baked opacity micro-maps (EXT_mesh_opacity_micromap) when --useOpacityMicromap is on (src/main.cpp:114-115);
  // the counterpart here is baked at load time from the alpha texture (mi_scene_cut_alpha)
  r->add("useOpacityMicromap", "Bake alpha-MASK geometry at load time (see --alphaCut)", &m_useOpacityMicromap);
  r->add("alphaCut", "Alpha bake: subdivisions per triangle edge [2..16], 0 = off", &m_alphaCut);
  // animation playback (the reference drives AnimationControl from its UI strip only; these switches are this port's headless handle)
  AnimationControl& ac = m_resources.animationControl;
  r->add("animation", "Animation clip index", &ac.currentAnimation);
  r->add("animStep", "Animation: seconds advanced per app frame (0 = paused); every step restarts the accumulation", &ac.stepSeconds);
  r->add("animSpeed", "Animation: playback multiplier", &ac.speed);
  r->add("animTime", "Animation: pose the clip at this time once (scrub), then render the still scene", &ac.scrubTime);
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
  if(m_useOpacityMicromap && m_alphaCut > 0)
  {
    const long long dropped = mi_scene_cut_alpha(m_resources.scene, m_alphaCut);
    if(dropped < 0)
      fprintf(stderr, "alphaCut: %s\n", mi_host_last_error());
    else if(dropped > 0)
      printf("alphaCut: %lld (sub-)triangles of alpha-MASK geometry dropped\n", dropped);
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

// Update the scene animation (reference: src/renderer.cpp:2065-2170): advance the clip, evaluate its channels, recompute the world
// matrices, sync render nodes and lights to the device, refresh the acceleration structure.  Returns true when the scene changed.
bool GltfRenderer::updateAnimation()
{
  AnimationControl& ac = m_resources.animationControl;
  MiScene*          sc = m_resources.scene;
  if(!sc || !m_pathTracer.handle())
    return false;
  const int nAnim = mi_scene_num_animations(sc);
  if(nAnim <= 0)
    return false;
  if(ac.currentAnimation < 0 || ac.currentAnimation >= nAnim)
    ac.currentAnimation = 0;
  float start = 0, end = 0;
  mi_scene_animation_info(sc, ac.currentAnimation, &start, &end, nullptr, 0);
  if(!(end > start))  // hasPlayableAnimation: a clip with a positive duration
    return false;
  if(m_animClip != ac.currentAnimation)
  {
    m_animClip = ac.currentAnimation;
    m_animTime = start;
  }
  if(ac.scrubTime >= 0.0f)  // scrubTo: clamp, pause, evaluate once
  {
    m_animTime   = std::min(std::max(ac.scrubTime, start), end);
    ac.scrubTime = -1.0f;
    ac.play      = false;
    ac.runOnce   = true;
  }
  else if(ac.stepSeconds != 0.0f)
    ac.play = true;
  if(!ac.doAnimation())
    return false;
  if(ac.isReset())
    m_animTime = start;
  else if(ac.play)
  {
    // AnimationInfo::incrementTime with loop = true (reference: src/gltf_scene.hpp:166-188)
    const float duration = end - start;
    float       wrapped  = std::fmod(m_animTime + ac.deltaTime() - start, duration);
    if(wrapped < 0.0f)
      wrapped += duration;
    m_animTime = start + wrapped;
  }
  ac.clearStates();
  if(ac.stepSeconds == 0.0f)
    ac.play = false;
  if(mi_scene_update_animation(sc, ac.currentAnimation, m_animTime) <= 0)
    return false;
  const MiPtSceneDesc* d = mi_scene_desc(sc);
  if(mi_pt_update_render_nodes(m_pathTracer.handle(), d->renderNodes, d->numRenderNodes, d->renderNodeVisible) != MI_PT_OK
     || mi_pt_update_lights(m_pathTracer.handle(), d->lights, d->numLights) != MI_PT_OK)
  {
    fprintf(stderr, "updateAnimation: %s\n", mi_pt_last_error());
    return false;
  }
  return true;
}

void GltfRenderer::onRender(StreamHandle cmd, bool headless, uint32_t headlessFrames, int batch)
{
  m_benchmark.beginHeadlessTimingIfNeeded(headless, benchmarkFrameInfo(headlessFrames));
  if(updateAnimation())  // reference: src/renderer.cpp:657-662
    resetFrame();
  if(m_resources.animationControl.play)
    batch = 1;  // the scene changes between app frames
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

// Baseline JPEG (ITU T.81: sequential DCT, Huffman coding, 8-bit, YCbCr 4:4:4, the Annex K example tables; quantisation tables
// scaled by `quality` the IJG way) -- the reference's default headless output is `<executable>.jpg` (src/renderer.cpp:557-573,
// written by nvapp's image writer); here it is written directly.
bool GltfRenderer::saveJpg(const std::string& path, const unsigned char* rgba8, int w, int h, int quality)
{
  static const uint8_t zigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                     41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                     30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
  static const uint8_t baseQ[2][64] = {{16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,  14, 13, 16, 24, 40,  57,
                                        69, 56, 14, 17, 22,  29,  51,  87,  80, 62, 18, 22, 37,  56,  68,  109, 103, 77, 24, 35, 55,  64,
                                        81, 104, 113, 92, 49, 64,  78,  87,  103, 121, 120, 101, 72, 92,  95,  98,  112, 100, 103, 99},
                                       {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99,
                                        99, 99, 47, 66, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                                        99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99}};
  // Huffman table specifications (number of codes of each length 1..16, then the symbols in code order)
  static const uint8_t dcBits[2][16] = {{0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0}, {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0}};
  static const uint8_t dcVals[12]    = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
  static const uint8_t acBits[2][16] = {{0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d}, {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77}};
  static const uint8_t acVals[2][162] = {
      {0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1,
       0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26,
       0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56,
       0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85,
       0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa,
       0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6,
       0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9,
       0xfa},
      {0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42,
       0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19,
       0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55,
       0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83,
       0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8,
       0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4,
       0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9,
       0xfa}};
  if(w <= 0 || h <= 0 || w > 65535 || h > 65535)
    return false;
  // the symbols of an AC table must be exactly end-of-block, the 16-zero run and (run, size) for size 1..10: a typing slip in the
  // lists above would otherwise surface as a corrupt file for some image only
  for(int t = 0; t < 2; ++t)
  {
    bool seen[256] = {};
    for(uint8_t v : acVals[t])
      seen[v] = true;
    for(int v = 0; v < 256; ++v)
      if(seen[v] != (v == 0x00 || v == 0xf0 || ((v & 15) >= 1 && (v & 15) <= 10)))
        return false;
  }
  quality             = std::min(std::max(quality, 1), 100);
  const int scale     = quality < 50 ? 5000 / quality : 200 - 2 * quality;
  uint8_t   q[2][64];
  for(int t = 0; t < 2; ++t)
    for(int i = 0; i < 64; ++i)
      q[t][i] = uint8_t(std::min(std::max((int(baseQ[t][i]) * scale + 50) / 100, 1), 255));
  // canonical codes from the specifications
  struct Code
  {
    uint16_t code[256];
    uint8_t  len[256];
  };
  auto build = [](const uint8_t* bits, const uint8_t* vals, Code& c) {
    memset(&c, 0, sizeof(c));
    uint16_t code = 0;
    int      k    = 0;
    for(int l = 1; l <= 16; ++l)
    {
      for(int i = 0; i < bits[l - 1]; ++i, ++k)
      {
        c.code[vals[k]] = code++;
        c.len[vals[k]]  = uint8_t(l);
      }
      code <<= 1;
    }
  };
  Code dc[2], ac[2];
  for(int t = 0; t < 2; ++t)
  {
    build(dcBits[t], dcVals, dc[t]);
    build(acBits[t], acVals[t], ac[t]);
  }
  std::vector<uint8_t> out;
  auto put16 = [&](int v) { out.push_back(uint8_t(v >> 8)); out.push_back(uint8_t(v)); };
  out.push_back(0xff); out.push_back(0xd8);                                                 // SOI
  const uint8_t app0[] = {0xff, 0xe0, 0, 16, 'J', 'F', 'I', 'F', 0, 1, 1, 0, 0, 1, 0, 1, 0, 0};  // JFIF 1.1, aspect 1:1
  out.insert(out.end(), app0, app0 + sizeof(app0));
  for(int t = 0; t < 2; ++t)  // DQT (tables are stored in zigzag order)
  {
    out.push_back(0xff); out.push_back(0xdb); put16(67); out.push_back(uint8_t(t));
    for(int i = 0; i < 64; ++i)
      out.push_back(q[t][zigzag[i]]);
  }
  out.push_back(0xff); out.push_back(0xc0); put16(17); out.push_back(8); put16(h); put16(w); out.push_back(3);  // SOF0
  for(int c = 0; c < 3; ++c)
  {
    out.push_back(uint8_t(c + 1)); out.push_back(0x11); out.push_back(uint8_t(c ? 1 : 0));
  }
  for(int t = 0; t < 2; ++t)  // DHT: DC then AC of each table pair
  {
    out.push_back(0xff); out.push_back(0xc4); put16(2 + 1 + 16 + 12); out.push_back(uint8_t(t));
    out.insert(out.end(), dcBits[t], dcBits[t] + 16);
    out.insert(out.end(), dcVals, dcVals + 12);
    out.push_back(0xff); out.push_back(0xc4); put16(2 + 1 + 16 + 162); out.push_back(uint8_t(0x10 | t));
    out.insert(out.end(), acBits[t], acBits[t] + 16);
    out.insert(out.end(), acVals[t], acVals[t] + 162);
  }
  const uint8_t sos[] = {0xff, 0xda, 0, 12, 3, 1, 0x00, 2, 0x11, 3, 0x11, 0, 63, 0};
  out.insert(out.end(), sos, sos + sizeof(sos));
  // entropy-coded data
  uint32_t acc = 0;
  int      nacc = 0;
  auto emit = [&](uint32_t code, int len) {
    acc = (acc << len) | (code & ((1u << len) - 1u));
    nacc += len;
    while(nacc >= 8)
    {
      const uint8_t b = uint8_t(acc >> (nacc - 8));
      out.push_back(b);
      if(b == 0xff)
        out.push_back(0);  // byte stuffing
      nacc -= 8;
    }
  };
  float cosTab[8][8];
  for(int u = 0; u < 8; ++u)
    for(int x = 0; x < 8; ++x)
      cosTab[u][x] = float(std::cos((2 * x + 1) * u * 3.14159265358979323846 / 16.0)) * (u == 0 ? float(std::sqrt(0.125)) : 0.5f);
  int prevDc[3] = {0, 0, 0};
  for(int by = 0; by < h; by += 8)
    for(int bx = 0; bx < w; bx += 8)
    {
      float block[3][64];
      for(int y = 0; y < 8; ++y)
        for(int x = 0; x < 8; ++x)
        {
          const unsigned char* p = rgba8 + (size_t(std::min(by + y, h - 1)) * size_t(w) + size_t(std::min(bx + x, w - 1))) * 4;  // edge replication
          const float          r = p[0], g = p[1], b = p[2];
          block[0][y * 8 + x]    = 0.299f * r + 0.587f * g + 0.114f * b - 128.0f;
          block[1][y * 8 + x]    = -0.168736f * r - 0.331264f * g + 0.5f * b;
          block[2][y * 8 + x]    = 0.5f * r - 0.418688f * g - 0.081312f * b;
        }
      for(int c = 0; c < 3; ++c)
      {
        const int t = c ? 1 : 0;
        float     tmp[64], coef[64];
        for(int y = 0; y < 8; ++y)  // rows, then columns
          for(int u = 0; u < 8; ++u)
          {
            float s = 0;
            for(int x = 0; x < 8; ++x)
              s += block[c][y * 8 + x] * cosTab[u][x];
            tmp[y * 8 + u] = s;
          }
        for(int u = 0; u < 8; ++u)
          for(int v = 0; v < 8; ++v)
          {
            float s = 0;
            for(int y = 0; y < 8; ++y)
              s += tmp[y * 8 + u] * cosTab[v][y];
            coef[v * 8 + u] = s;
          }
        int zz[64];
        for(int i = 0; i < 64; ++i)
          zz[i] = int(std::lround(coef[zigzag[i]] / float(q[t][zigzag[i]])));
        auto category = [](int v) { int a = v < 0 ? -v : v, n = 0; while(a) { ++n; a >>= 1; } return n; };
        auto bitsOf   = [](int v, int n) { return uint32_t(v < 0 ? v + (1 << n) - 1 : v); };
        const int diff = zz[0] - prevDc[c];
        prevDc[c]      = zz[0];
        int n          = category(diff);
        emit(dc[t].code[n], dc[t].len[n]);
        if(n)
          emit(bitsOf(diff, n), n);
        int run = 0, last = 63;
        while(last > 0 && zz[last] == 0)
          --last;
        for(int i = 1; i <= last; ++i)
        {
          if(zz[i] == 0)
          {
            ++run;
            continue;
          }
          while(run > 15)
          {
            emit(ac[t].code[0xf0], ac[t].len[0xf0]);
            run -= 16;
          }
          n             = std::min(category(zz[i]), 10);
          const int val = std::min(std::max(zz[i], -1023), 1023);
          emit(ac[t].code[(run << 4) | n], ac[t].len[(run << 4) | n]);
          emit(bitsOf(val, n), n);
          run = 0;
        }
        if(last < 63)
          emit(ac[t].code[0x00], ac[t].len[0x00]);  // end of block
      }
    }
  if(nacc)
    emit(0x7f, 8 - nacc);  // pad the last byte with ones
  out.push_back(0xff); out.push_back(0xd9);  // EOI
  FILE* f = fopen(path.c_str(), "wb");
  if(!f)
    return false;
  const bool ok = fwrite(out.data(), 1, out.size(), f) == out.size();
  fclose(f);
  return ok;
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
  // default: <executable name>.jpg in the working directory (reference: src/renderer.cpp:559-561)
  std::string out = m_resources.headlessOutputPath.empty() ? std::string("mi_gltf_renderer.jpg") : m_resources.headlessOutputPath;
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
    const bool jpeg = (out.size() > 4 && out.substr(out.size() - 4) == ".jpg") || (out.size() > 5 && out.substr(out.size() - 5) == ".jpeg");
    if(!(jpeg ? saveJpg(out, ldr.data(), w, h, 90) : savePng(out, ldr.data(), w, h)))
      return;
  }
  printf("Saved headless output image: %s\n", out.c_str());
}
