// Keyframe animation of node transforms: glTF 2.0 `animations` with translation / rotation / scale channels and LINEAR, STEP
// and CUBICSPLINE samplers (glTF 2.0 specification, section 3.11 and appendix C), evaluated into per-node poses, from which
// the world matrices, the render-node table and the light placements are recomputed in place.  Behaviour follows the
// reference's AnimationSystem (src/gltf_scene_animation.cpp:84-175 parse, :355-478 update / segment search, :484-700
// interpolation): a sampler needs two keyframes, a channel is applied only while the time lies inside its keyframe range, the
// clip's [start, end] is the hull of all sampler inputs, rotations are slerped (LINEAR) or spline-evaluated and normalised.
// Morph-target weights, skins and KHR_animation_pointer channels are skipped: they change vertex data or material tables, which
// is outside the instance-update boundary (mi_pt_update_render_nodes).
#include <algorithm>
#include <cmath>
#include <map>
#include <cstring>
#include <limits>

#include "gltf_scene.hpp"

namespace mihost {

using mijson::Value;

float AnimationInfo::incrementTime(float deltaTime, bool loop)  // reference: src/gltf_scene.hpp:166-188
{
  currentTime += deltaTime;
  if(loop)
  {
    const float duration = end - start;
    if(!(duration > 0.0f))  // a clip with one keyframe (or none): fmod(x, 0) is NaN and would stick -- the clip has one pose
      return currentTime = start;
    float       wrapped  = std::fmod(currentTime - start, duration);
    if(wrapped < 0.0f)
      wrapped += duration;
    currentTime = start + wrapped;
  }
  else if(currentTime > end)
    currentTime = end;
  return currentTime;
}

void GltfScene::parseAnimations()
{
  const Value& anims = m_doc["animations"];
  for(size_t a = 0; a < anims.size(); ++a)
  {
    const Value& ga = anims[a];
    Animation    anim;
    anim.info.name = ga["name"].string("Animation" + std::to_string(a));
    const Value& samplers = ga["samplers"];
    for(size_t i = 0; i < samplers.size(); ++i)
    {
      const Value&     gs = samplers[i];
      AnimationSampler sm;
      const std::string ip = gs["interpolation"].string("LINEAR");
      sm.interpolation     = ip == "STEP" ? AnimationSampler::eStep : (ip == "CUBICSPLINE" ? AnimationSampler::eCubicSpline : AnimationSampler::eLinear);
      if(!gs["input"].isNumber() || !readAccessorFloats(gs["input"].integer(-1), 1, sm.inputs))
        sm.inputs.clear();
      if(!gs["output"].isNumber() || !readAccessorFloats(gs["output"].integer(-1), 0, sm.outputs, &sm.components) || sm.components <= 0)
      {
        sm.inputs.clear();
        sm.outputs.clear();
        sm.components = 1;
      }
      // key times and values of a usable sampler are finite (a NaN time would pass every range test below and pose the node at NaN)
      bool finite = true;
      for(float t : sm.inputs)
        finite = finite && std::isfinite(t);
      for(float v : sm.outputs)
        finite = finite && std::isfinite(v);
      if(!finite)
      {
        sm.inputs.clear();
        sm.outputs.clear();
      }
      for(float t : sm.inputs)
      {
        anim.info.start = std::min(anim.info.start, t);
        anim.info.end   = std::max(anim.info.end, t);
      }
      anim.samplers.push_back(std::move(sm));
    }
    const Value& channels = ga["channels"];
    for(size_t i = 0; i < channels.size(); ++i)
    {
      const Value&      gc   = channels[i];
      const std::string path = gc["target"]["path"].string("");
      AnimationChannel  ch;
      if(path == "translation")
        ch.path = AnimationChannel::eTranslation;
      else if(path == "rotation")
        ch.path = AnimationChannel::eRotation;
      else if(path == "scale")
        ch.path = AnimationChannel::eScale;
      else
        continue;  // weights / pointer
      ch.node    = gc["target"]["node"].integer(-1);
      ch.sampler = gc["sampler"].integer(-1);
      if(ch.node < 0 || size_t(ch.node) >= m_nodePose.size() || ch.sampler < 0 || size_t(ch.sampler) >= anim.samplers.size())
        continue;
      const int need = ch.path == AnimationChannel::eRotation ? 4 : 3;
      if(anim.samplers[size_t(ch.sampler)].components != need)
        continue;
      anim.channels.push_back(ch);
    }
    if(anim.info.start > anim.info.end)  // no keyframes at all
      anim.info.start = anim.info.end = 0.0f;
    anim.info.currentTime = 0.0f;
    m_animations.push_back(std::move(anim));
  }
}

namespace {

// glm::slerp followed by glm::normalize (shortest path; nearly parallel quaternions are lerped)
void slerpNormalized(const float* a, const float* b, float t, float* out)
{
  float z[4]     = {b[0], b[1], b[2], b[3]};
  float cosTheta = a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
  if(cosTheta < 0.0f)
  {
    for(float& c : z)
      c = -c;
    cosTheta = -cosTheta;
  }
  if(cosTheta > 1.0f - std::numeric_limits<float>::epsilon())
  {
    for(int i = 0; i < 4; ++i)
      out[i] = a[i] + t * (z[i] - a[i]);
  }
  else
  {
    const float angle = std::acos(cosTheta);
    const float wa = std::sin((1.0f - t) * angle), wb = std::sin(t * angle), inv = 1.0f / std::sin(angle);
    for(int i = 0; i < 4; ++i)
      out[i] = (wa * a[i] + wb * z[i]) * inv;
  }
  const float len = std::sqrt(out[0] * out[0] + out[1] * out[1] + out[2] * out[2] + out[3] * out[3]);
  if(len > 0.0f)
    for(int i = 0; i < 4; ++i)
      out[i] /= len;
}

}  // namespace

bool GltfScene::updateAnimation(int index)
{
  if(index < 0 || size_t(index) >= m_animations.size())
    return false;
  const Animation& anim = m_animations[size_t(index)];
  const float      time = anim.info.currentTime;
  bool             any  = false;

  for(const AnimationChannel& ch : anim.channels)
  {
    const AnimationSampler& sm = anim.samplers[size_t(ch.sampler)];
    const size_t            nk = sm.inputs.size();
    if(nk < 2)
      continue;
    // the segment [i, i+1] that holds `time` (first keyframe strictly after it, minus one)
    auto it = std::upper_bound(sm.inputs.begin(), sm.inputs.end(), time);
    if(it == sm.inputs.begin())
      continue;
    size_t i = size_t(it - sm.inputs.begin()) - 1;
    if(i + 1 >= nk)
      i = nk - 2;
    const float t0 = sm.inputs[i], t1 = sm.inputs[i + 1];
    if(!(time >= t0 && time <= t1))
      continue;
    const float keyDelta = t1 - t0;
    const float t        = std::fabs(keyDelta) < std::numeric_limits<float>::epsilon() ? 0.0f : std::min(std::max((time - t0) / keyDelta, 0.0f), 1.0f);
    const int   nc       = sm.components;
    const size_t numOut  = sm.outputs.size() / size_t(nc);
    float       v[4]     = {0, 0, 0, 1};
    bool        have     = false;
    switch(sm.interpolation)
    {
      case AnimationSampler::eLinear:
        if(i + 1 < numOut)
        {
          const float* a = &sm.outputs[i * size_t(nc)];
          const float* b = a + nc;
          if(ch.path == AnimationChannel::eRotation)
            slerpNormalized(a, b, t, v);
          else
            for(int c = 0; c < nc; ++c)
              v[c] = a[c] * (1.0f - t) + b[c] * t;  // glm::mix
          have = true;
        }
        break;
      case AnimationSampler::eStep:
        if(i < numOut)
        {
          memcpy(v, &sm.outputs[i * size_t(nc)], sizeof(float) * size_t(nc));
          have = true;
        }
        break;
      case AnimationSampler::eCubicSpline:
        if(numOut > (i + 1) * 3 + 1)
        {
          // cubic Hermite spline, glTF 2.0 appendix C: per keyframe (in-tangent a, value v, out-tangent b)
          const float  t2 = t * t, t3 = t2 * t;
          const float  cV1 = -2 * t3 + 3 * t2, cV0 = 1 - cV1, cA = keyDelta * (t3 - t2), cB = keyDelta * (t3 - 2 * t2 + t);
          const float* k0 = &sm.outputs[(i * 3) * size_t(nc)];
          const float* k1 = &sm.outputs[((i + 1) * 3) * size_t(nc)];
          for(int c = 0; c < nc; ++c)
            v[c] = k0[nc + c] * cV0 + k1[c] * cA + k0[2 * nc + c] * cB + k1[nc + c] * cV1;
          if(ch.path == AnimationChannel::eRotation)
          {
            const float len = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]);
            if(len > 0.0f)
              for(float& c : v)
                c /= len;
          }
          have = true;
        }
        break;
    }
    if(!have)
      continue;
    NodePose& pose = m_nodePose[size_t(ch.node)];
    if(!pose.animated)
    {
      // first touch: start from the document's TRS
      const Value& node = m_doc["nodes"][size_t(ch.node)];
      auto         get  = [&](const char* key, int n, float* out) {
        const Value& a = node[key];
        if(a.isArray() && a.arr.size() == size_t(n))
          for(int c = 0; c < n; ++c)
            out[c] = float(a.arr[size_t(c)].number());
      };
      get("translation", 3, pose.t);
      get("rotation", 4, pose.q);
      get("scale", 3, pose.s);
      pose.animated = true;
    }
    float* dst = ch.path == AnimationChannel::eTranslation ? pose.t : (ch.path == AnimationChannel::eRotation ? pose.q : pose.s);
    memcpy(dst, v, sizeof(float) * size_t(nc));
    any = true;
  }
  if(!any)
    return false;

  // World matrices (reference: Scene::updateNodeWorldMatrices), per PATH from a scene root: glTF hierarchies are strict trees, but
  // load-time traversal instantiates a node of a non-conforming file under every parent it is listed by -- each such render node
  // (and light) keeps the path it was reached by and is posed along it, with the multiplication order of traverse().
  std::map<std::vector<int>, mx::mat4> cache;  // (distinct paths are few: render nodes of one glTF node share theirs)
  auto worldOf = [&](const std::vector<int>& path) {
    auto it = cache.find(path);
    if(it != cache.end())
      return it->second;
    mx::mat4 w = mx::identity();
    for(size_t i = 0; i < path.size(); ++i)
      w = i == 0 ? localMatrix(path[i]) : mx::mul(w, localMatrix(path[i]));
    cache.emplace(path, w);
    return w;
  };
  for(size_t n = 0; n < m_renderNodes.size(); ++n)
  {
    const RenderNodeSource& src = m_renderNodeSource[n];
    mx::mat4                w   = worldOf(src.path);
    if(src.instance >= 0)
      w = mx::mul(w, m_gpuInstanceLocalMatrices.at(src.node)[size_t(src.instance)]);
    MiGltfRenderNode& rn = m_renderNodes[n];
    memcpy(rn.objectToWorld, w.m, sizeof(rn.objectToWorld));
    const mx::mat4 inv = mx::inverse(w);
    memcpy(rn.worldToObject, inv.m, sizeof(rn.worldToObject));
  }
  for(size_t l = 0; l < m_lights.size(); ++l)
    placeLight(m_lights[l], worldOf(m_lightPath[l]));
  return true;
}

}  // namespace mihost
