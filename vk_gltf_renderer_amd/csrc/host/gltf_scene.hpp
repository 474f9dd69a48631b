// CPU scene front end: glTF 2.0 (.gltf / .glb) -> flattened RenderNode / RenderPrimitive / material / light / camera
// tables, i.e. the slice of `nvvkgltf::Scene` + `MaterialCache` + `SceneVk` table construction that feeds the path
// tracer (reference: src/gltf_scene.cpp:298-330 load, :1350-1470 parseScene, :2139-2165 buildPrimitiveKeyMap,
// :2269-2300 lights, :2338-2429 render nodes + EXT_mesh_gpu_instancing, :1561-1594 default camera;
// src/gltf_material_cache.cpp:103-260; src/gltf_scene_vk.cpp:493-501, :741-870, :909-947, :1102-1154, :1354-1392).
// Keyframe animation of node transforms (gltf_scene_animation.cpp here; reference: src/gltf_scene_animation.cpp:355-700) feeds
// mi_pt_update_render_nodes / mi_pt_update_lights.  Editing, saving, merging, skinning, morph targets, KHR_animation_pointer
// and the variants UI are out of scope.
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <vector>

#include "image_loader.hpp"
#include "json.hpp"
#include "mathx.hpp"
#include "mi_pt.h"

namespace mihost {

struct RenderCamera  // reference: nvvkgltf::RenderCamera, src/gltf_scene.hpp
{
  enum Type { ePerspective, eOrthographic } type = ePerspective;
  double eye[3]    = {0, 0, 0};
  double center[3] = {0, 0, 0};
  double up[3]     = {0, 1, 0};
  double yfov = 0.785398, xmag = 1, ymag = 1, znear = 0.1, zfar = 100;
};

struct RenderPrimitiveData
{
  std::vector<uint32_t> indices;
  std::vector<float>    positions, normals, tangents, texCoords0, texCoords1;
  std::vector<uint32_t> colors;
  int                   meshID = -1;
  uint32_t              vertexCount = 0;
  uint32_t              opaqueTriangles = 0;  // triangles [0, n) cannot fail their material's alpha test (cutAlphaMasked)
};

struct TextureData
{
  std::vector<std::vector<uint8_t>> levels;
  std::vector<const uint8_t*>       levelPtrs;
  int                               width = 1, height = 1;
  bool                              srgb = false;
  int magFilter = MI_FILTER_LINEAR, minFilter = MI_FILTER_LINEAR, mipmapMode = MI_FILTER_LINEAR;
  int wrapS = MI_WRAP_REPEAT, wrapT = MI_WRAP_REPEAT;
};

// reference: nvvkgltf::AnimationInfo (src/gltf_scene.hpp:159-189)
struct AnimationInfo
{
  std::string name;
  float       start = 3.402823466e+38f, end = -3.402823466e+38f, currentTime = 0.0f;
  float       reset() { return currentTime = start; }
  float       incrementTime(float deltaTime, bool loop = true);
};

class GltfScene
{
public:
  // Returns false and fills error() on failure.
  bool load(const std::string& filename);
  const std::string& error() const { return m_error; }

  const MiPtSceneDesc& desc() const { return m_desc; }

  const std::vector<MiGltfShadeMaterial>& materials() const { return m_materials; }
  const std::vector<MiGltfTextureInfo>&   textureInfos() const { return m_textureInfos; }
  const std::vector<MiGltfRenderNode>&    renderNodes() const { return m_renderNodes; }
  const std::vector<RenderPrimitiveData>& renderPrimitives() const { return m_primData; }
  const std::vector<MiGltfLight>&         lights() const { return m_lights; }
  const std::vector<RenderCamera>&        cameras() const { return m_cameras; }
  const std::vector<TextureData>&         textures() const { return m_textures; }
  uint64_t                                numTriangles() const { return m_numTriangles; }
  // recomputeTangents (reference: src/gltf_create_tangent.hpp:40; the UI's "Recreate Tangents" items): UV-gradient tangents, or
  // Mikkelsen's with vertex splitting at tangent discontinuities.  Returns the number of vertices the splitting added; desc() is
  // rebuilt (its pointers change).
  uint32_t recomputeTangents(bool forceCreation, bool mikktspace);
  // Animation clips (translation / rotation / scale channels; LINEAR, STEP, CUBICSPLINE).  updateAnimation evaluates clip `index`
  // at its info's currentTime, recomputes the world matrices and rewrites the matrices of renderNodes() and the placement of
  // lights() IN PLACE (same order, same count, desc() pointers stay valid).  Returns true when something moved.
  int            numAnimations() const { return int(m_animations.size()); }
  AnimationInfo& animationInfo(int index) { return m_animations[size_t(index)].info; }
  bool           updateAnimation(int index);
  const std::vector<uint8_t>& renderNodeVisible() const { return m_renderNodeVisible; }
  // Load-time bake for alpha-MASK geometry (alpha_cut.cpp; the counterpart of the reference's opacity micro-map bake,
  // src/gltf_scene_omm.cpp): triangles are cut into subdivisions x subdivisions sub-triangles and those on which the alpha test
  // cannot pass are dropped.  Returns the number of (sub-)triangles dropped; desc() is rebuilt (its pointers change).
  struct AlphaCutStats
  {
    uint64_t trianglesRemoved = 0, trianglesSplit = 0, subTrianglesDropped = 0, trianglesOpaque = 0;
  };
  uint64_t             cutAlphaMasked(int subdivisions);
  const AlphaCutStats& alphaCutStats() const { return m_alphaCutStats; }
  void bounds(float bmin[3], float bmax[3]) const;
  float boundsRadius() const;

private:
  bool parse(const std::string& baseDir);
  bool readAccessorFloats(int accessor, int expectedComponents, std::vector<float>& out, int* outComponents = nullptr);
  bool readAccessorUints(int accessor, std::vector<uint32_t>& out);
  const uint8_t* bufferViewData(int bufferView, size_t& size, size_t& stride);
  void buildMaterials();
  void buildTextures(const std::string& baseDir);
  void buildPrimitives(std::map<std::string, int>& primMap);
  void traverse(int nodeID, const mx::mat4& parent, bool parentVisible, const std::map<std::string, int>& primMap);
  void traverseCameras(int nodeID, const mx::mat4& parent);
  void finalizeDesc();
  uint16_t addTextureInfo(const mijson::Value& texInfo);
  mx::mat4 localMatrix(int nodeID) const;
  void     parseAnimations();
  void     placeLight(MiGltfLight& info, const mx::mat4& world) const;

  struct AnimationSampler
  {
    enum Interpolation { eLinear, eStep, eCubicSpline } interpolation = eLinear;
    std::vector<float> inputs, outputs;  // outputs: `components` floats per keyframe (x3 for CUBICSPLINE: in-tangent, value, out-tangent)
    int                components = 0;
  };
  struct AnimationChannel
  {
    enum Path { eTranslation, eRotation, eScale } path = eTranslation;
    int node = -1, sampler = 0;
  };
  struct Animation
  {
    AnimationInfo                 info;
    std::vector<AnimationSampler> samplers;
    std::vector<AnimationChannel> channels;
  };
  struct NodePose  // the TRS an animation wrote; nodes never touched keep the document's matrix / TRS
  {
    bool  animated = false;
    float t[3] = {0, 0, 0}, q[4] = {0, 0, 0, 1}, s[3] = {1, 1, 1};
  };
  struct RenderNodeSource
  {
    int              node = -1, instance = -1;  // glTF node, EXT_mesh_gpu_instancing instance (-1: none)
    std::vector<int> path;                      // the nodes from a scene root down to `node`: a file whose nodes have several parents is
                                                // instantiated once per path at load (traverse), and posed per path by updateAnimation
  };
  std::vector<int>              m_curPath;      // traverse(): the path to the node being visited
  std::vector<std::vector<int>> m_lightPath;    // per light, like RenderNodeSource::path
  std::vector<Animation>        m_animations;
  std::vector<NodePose>         m_nodePose;
  std::vector<RenderNodeSource> m_renderNodeSource;
  std::vector<int>              m_lightNode, m_roots;
  AlphaCutStats                 m_alphaCutStats;
  bool                          m_alphaCutDone = false;
  std::vector<uint8_t>          m_onPath;  // nodes on the current traversal path (cycle guard)

  mijson::Value                      m_doc;
  bool decompressMeshopt();  // EXT / KHR_meshopt_compression buffer views -> their fallback regions (meshopt_decoder.hpp)
  std::vector<std::vector<uint8_t>>  m_buffers;
  std::string                        m_error;
  std::vector<MiGltfShadeMaterial>   m_materials;
  std::vector<MiGltfTextureInfo>     m_textureInfos;
  std::vector<MiGltfRenderNode>      m_renderNodes;
  std::vector<uint8_t>               m_renderNodeVisible;
  std::vector<RenderPrimitiveData>   m_primData;
  std::vector<MiPtRenderPrimitive>   m_prims;
  std::vector<MiGltfLight>           m_lights;
  std::vector<RenderCamera>          m_cameras;
  std::vector<TextureData>           m_textures;
  std::vector<MiPtTexture>           m_textureDescs;
  std::map<int, std::vector<mx::mat4>> m_gpuInstanceLocalMatrices;
  uint64_t                           m_numTriangles = 0;
  float                              m_bmin[3] = {0, 0, 0}, m_bmax[3] = {0, 0, 0};
  MiPtSceneDesc                      m_desc{};
};

// Radiance .hdr (RGBE) environment + importance-sampling table, the slice of `nvvk::HdrIbl::loadEnvironment` the
// path tracer consumes (reference call site: src/renderer.cpp:1982-2017; device consumers:
// shaders/pathtrace_functions.h.slang:436-447,474-479). The implementation lives in nvpro_core2 (not in
// /root/reference); the construction here follows its published scheme: per-texel importance = luminance x solid
// angle, Vose alias table, normalised pdf stored in the texture's alpha.
class HdrEnvironment
{
public:
  bool load(const std::string& filename);
  // Build from caller-provided linear RGB floats (width*height*3), e.g. a synthetic sky.
  void setPixels(int width, int height, const float* rgb);
  const std::string&     error() const { return m_error; }
  const MiPtEnvironment& env() const { return m_env; }
  int                    width() const { return m_env.width; }
  int                    height() const { return m_env.height; }

private:
  void buildAccel();
  std::vector<float>      m_rgba;
  std::vector<MiEnvAccel> m_accel;
  MiPtEnvironment         m_env{};
  std::string             m_error;
};

}  // namespace mihost
