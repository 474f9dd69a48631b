// Shared renderer state handed to every BaseRenderer call — the slice of the reference's `struct Resources` /
// `struct Settings` (src/resources.hpp:82-131, :167-276) that the path-trace mode reads, minus Vulkan handles.
#pragma once
#include <memory>
#include <string>

#include "mi_host.h"
#include "mi_pt.h"

enum class RenderingMode { ePathtracer, eRasterizer };
enum class EnvSystem { eSky = 0, eHdr = 1 };  // reference: shaderio::EnvSystem, shaders/shaderio.h:45-49

struct Settings  // reference defaults: src/resources.hpp:82-131
{
  RenderingMode renderSystem           = RenderingMode::ePathtracer;
  EnvSystem     envSystem              = EnvSystem::eSky;
  float         hdrEnvIntensity        = 1.0f;
  float         hdrEnvRotation         = 0.0f;
  float         hdrBlur                = 0.0f;
  bool          useSolidBackground     = false;
  float         solidBackgroundColor[3] = {0.0f, 0.0f, 0.0f};
  int           maxFrames              = 500;
  bool          useInfinitePlane       = false;
  bool          isShadowCatcher        = true;
  float         infinitePlaneDistance  = 0.0f;
  float         infinitePlaneBaseColor[3] = {0.5f, 0.5f, 0.5f};
  float         infinitePlaneMetallic  = 0.0f;
  float         infinitePlaneRoughness = 0.5f;
  float         shadowCatcherDarkness  = 0.0f;
};

// reference: AnimationControl (src/ui_animation.hpp:51-85).  Interactive playback there advances by the UI's frame time; a
// headless run here advances by a fixed `stepSeconds` per app frame so that a run is reproducible (0 = paused), or is scrubbed
// to `scrubTime` once (scrubTo: pause + one evaluation).
struct AnimationControl
{
  bool  play             = false;
  bool  runOnce          = false;
  bool  reset            = false;
  float speed            = 1.0f;
  int   currentAnimation = 0;
  float stepSeconds      = 0.0f;
  float scrubTime        = -1.0f;
  bool  doAnimation() const { return play || runOnce || reset; }
  float deltaTime() const { return (runOnce ? 1.0f / 60.0f : stepSeconds) * speed; }
  bool  isReset() const { return reset; }
  void  clearStates() { runOnce = reset = false; }
};

struct Resources
{
  MiScene*                scene{nullptr};  // nvvkgltf::Scene + SceneVk tables (libmi_host)
  MiHdr*                  hdrIbl{nullptr}; // nvvk::HdrIbl
  MiSkyPhysicalParameters skyParams{};
  MiTonemapperData        tonemapperData{};  // reference: src/resources.hpp:212
  MiCamera                camera{};        // nvutils::CameraManipulator state
  MiSceneFrameInfo        frameInfo{};     // what GltfRenderer::onRender uploads into bFrameInfo each frame
  Extent2D                renderSize{};    // gBuffers.getSize()
  std::string             headlessOutputPath;
  int                     frameCount{0};
  int                     device{0};
  Settings                settings;
  AnimationControl        animationControl;
};
