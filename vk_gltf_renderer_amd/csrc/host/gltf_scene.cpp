// See gltf_scene.hpp for scope and reference citations.
#include "gltf_scene.hpp"
#include "meshopt_decoder.hpp"
#include "mikktspace_tangents.hpp"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <functional>
#include <sstream>

namespace mihost {

using mijson::Value;

namespace {

bool readFile(const std::string& path, std::vector<uint8_t>& out)
{
  std::ifstream f(path, std::ios::binary);
  if(!f)
    return false;
  f.seekg(0, std::ios::end);
  std::streamoff n = f.tellg();
  f.seekg(0, std::ios::beg);
  out.resize(size_t(n));
  if(n > 0)
    f.read(reinterpret_cast<char*>(out.data()), n);
  return bool(f);
}

std::string dirOf(const std::string& path)
{
  size_t p = path.find_last_of("/\\");
  return p == std::string::npos ? std::string(".") : path.substr(0, p);
}

bool decodeBase64(const std::string& in, size_t start, std::vector<uint8_t>& out)
{
  static int8_t table[256];
  static bool   init = false;
  if(!init)
  {
    memset(table, -1, sizeof(table));
    const char* chars = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
    for(int i = 0; i < 64; ++i)
      table[uint8_t(chars[i])] = int8_t(i);
    init = true;
  }
  out.clear();
  uint32_t acc  = 0;
  int      bits = 0;
  for(size_t i = start; i < in.size(); ++i)
  {
    char c = in[i];
    if(c == '=')
      break;
    int8_t v = table[uint8_t(c)];
    if(v < 0)
      continue;
    acc = (acc << 6) | uint32_t(v);
    bits += 6;
    if(bits >= 8)
    {
      bits -= 8;
      out.push_back(uint8_t((acc >> bits) & 0xFF));
    }
  }
  return true;
}

std::string uriDecode(const std::string& s)
{
  std::string r;
  for(size_t i = 0; i < s.size(); ++i)
  {
    if(s[i] == '%' && i + 2 < s.size())
    {
      r += char(strtol(s.substr(i + 1, 2).c_str(), nullptr, 16));
      i += 2;
    }
    else
      r += s[i];
  }
  return r;
}

bool loadUri(const std::string& uri, const std::string& baseDir, std::vector<uint8_t>& out)
{
  if(uri.rfind("data:", 0) == 0)
  {
    size_t comma = uri.find(',');
    if(comma == std::string::npos)
      return false;
    return decodeBase64(uri, comma + 1, out);
  }
  return readFile(baseDir + "/" + uriDecode(uri), out);
}

float getFloat(const Value& o, const char* key, float def)
{
  const Value* v = o.find(key);
  return (v && v->isNumber()) ? float(v->num) : def;
}
int getInt(const Value& o, const char* key, int def)
{
  const Value* v = o.find(key);
  return (v && v->isNumber()) ? v->integer() : def;
}
void getFloats(const Value& o, const char* key, int n, float* out)
{
  const Value* v = o.find(key);
  if(v && v->isArray() && int(v->arr.size()) >= n)
    for(int i = 0; i < n; ++i)
      out[i] = float(v->arr[size_t(i)].number());
}
const Value& ext(const Value& o, const char* name)
{
  return o["extensions"][name];
}

int componentCount(const std::string& type)
{
  if(type == "SCALAR") return 1;
  if(type == "VEC2") return 2;
  if(type == "VEC3") return 3;
  if(type == "VEC4") return 4;
  if(type == "MAT2") return 4;
  if(type == "MAT3") return 9;
  if(type == "MAT4") return 16;
  return 0;
}
int componentSize(int componentType)
{
  switch(componentType)
  {
    case 5120: case 5121: return 1;
    case 5122: case 5123: return 2;
    case 5125: case 5126: return 4;
  }
  return 0;
}

// reference: src/tinygltf_utils.hpp:756-775 (normalised integer -> float)
float decodeComponent(const uint8_t* p, int componentType, bool normalized)
{
  switch(componentType)
  {
    case 5120: {
      int8_t v;
      memcpy(&v, p, 1);
      return normalized ? std::max(float(v) / 127.0f, -1.0f) : float(v);
    }
    case 5121:
      return normalized ? float(*p) / 255.0f : float(*p);
    case 5122: {
      int16_t v;
      memcpy(&v, p, 2);
      return normalized ? std::max(float(v) / 32767.0f, -1.0f) : float(v);
    }
    case 5123: {
      uint16_t v;
      memcpy(&v, p, 2);
      return normalized ? float(v) / 65535.0f : float(v);
    }
    case 5125: {
      uint32_t v;
      memcpy(&v, p, 4);
      return float(v);
    }
    case 5126: {
      float v;
      memcpy(&v, p, 4);
      return v;
    }
  }
  return 0.0f;
}
uint32_t decodeUint(const uint8_t* p, int componentType)
{
  switch(componentType)
  {
    case 5121: return *p;
    case 5123: {
      uint16_t v;
      memcpy(&v, p, 2);
      return v;
    }
    case 5125: {
      uint32_t v;
      memcpy(&v, p, 4);
      return v;
    }
    case 5120: return uint32_t(int8_t(*p));
    case 5122: {
      int16_t v;
      memcpy(&v, p, 2);
      return uint32_t(v);
    }
  }
  return 0;
}

// glm::packUnorm4x8: round(clamp(c,0,1) * 255)
uint32_t packUnorm4x8(const float c[4])
{
  uint32_t r = 0;
  for(int i = 0; i < 4; ++i)
  {
    float    v = std::min(std::max(c[i], 0.0f), 1.0f);
    uint32_t q = uint32_t(std::round(v * 255.0f));
    r |= q << (8 * i);
  }
  return r;
}

mx::mat4 nodeLocalMatrix(const Value& node)  // reference: src/tinygltf_utils.cpp:641-654
{
  const Value& m = node["matrix"];
  if(m.isArray() && m.arr.size() == 16)
  {
    mx::mat4 r{};
    for(int i = 0; i < 16; ++i)
      r.m[i] = float(m.arr[size_t(i)].number());
    return r;
  }
  float t[3] = {0, 0, 0}, q[4] = {0, 0, 0, 1}, s[3] = {1, 1, 1};
  getFloats(node, "translation", 3, t);
  getFloats(node, "rotation", 4, q);
  getFloats(node, "scale", 3, s);
  return mx::mul(mx::mul(mx::translate({t[0], t[1], t[2]}), mx::fromQuat(q[0], q[1], q[2], q[3])), mx::scale({s[0], s[1], s[2]}));
}

// reference: src/tinygltf_utils.cpp:656-665 — attributes in std::map (lexicographic) order, then the indices accessor.
std::string primitiveKey(const Value& prim)
{
  std::map<std::string, int> attrs;
  for(const auto& kv : prim["attributes"].obj)
    attrs[kv.first] = kv.second.integer();
  std::stringstream o;
  for(const auto& kv : attrs)
    o << kv.first << ":" << kv.second << " ";
  o << "indices:" << getInt(prim, "indices", -1);
  return o.str();
}

// createMissingTangentsForModel's trigger rule (reference: src/gltf_scene.cpp:2431-2448): the primitive's material has a
// normal texture and the primitive carries no TANGENT attribute.
bool needsGeneratedTangents(const Value& doc, const Value& prim)
{
  const Value& mats  = doc["materials"];
  const int    count = int(mats.size());
  int          mat   = getInt(prim, "material", -1);
  if(mat < 0 || mat >= count)
    mat = 0;
  if(mat >= count)
    return false;
  if(getInt(mats[size_t(mat)]["normalTexture"], "index", -1) < 0)
    return false;
  return !prim["attributes"].has("TANGENT");
}
// Such primitives get a TANGENT accessor of their own in the reference, i.e. a different primitive key than the same
// geometry used without a normal map.
std::string primitiveKeyWithTangents(const Value& doc, const Value& prim)
{
  return needsGeneratedTangents(doc, prim) ? primitiveKey(prim) + " TANGENT:generated" : primitiveKey(prim);
}

// shaderio::makeFastTangent as restated on the device (pt_bsdf.h): orthonormal basis of Duff et al. 2017
void makeFastTangent(const float n[3], float t[4])
{
  float sign = std::copysign(1.0f, n[2]);
  float a    = -1.0f / (sign + n[2]);
  float b    = n[0] * n[1] * a;
  t[0] = 1.0f + sign * n[0] * n[0] * a; t[1] = sign * b; t[2] = -sign * n[0]; t[3] = 1.0f;
}

// tinygltf::utils::simpleCreateTangents (reference: src/tinygltf_utils.cpp:878-998; Lengyel, FGED2 ch. 7): per-face UV-space
// tangent accumulated on the three vertices, handedness of the LAST face touching a vertex (taken against the normal of
// the face's first vertex, as the reference does), Gram-Schmidt against the vertex normal, fast-tangent fallback.
void createSimpleTangents(RenderPrimitiveData& d)
{
  const size_t nv = d.vertexCount, nf = d.indices.size() / 3;
  d.tangents.assign(nv * 4, 0.0f);
  const bool         hasUV = !d.texCoords0.empty(), hasNormal = !d.normals.empty();
  std::vector<float> geoNormal(hasNormal ? 0 : nv * 3, 0.0f);
  auto P = [&](uint32_t i) { return &d.positions[size_t(i) * 3]; };
  for(size_t f = 0; f < nf; ++f)
  {
    const uint32_t i0 = d.indices[f * 3], i1 = d.indices[f * 3 + 1], i2 = d.indices[f * 3 + 2];
    const float *  p0 = P(i0), *p1 = P(i1), *p2 = P(i2);
    const float    e1[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]}, e2[3] = {p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2]};
    float          n0[3];
    if(hasNormal)
    {
      n0[0] = d.normals[size_t(i0) * 3]; n0[1] = d.normals[size_t(i0) * 3 + 1]; n0[2] = d.normals[size_t(i0) * 3 + 2];
    }
    else
    {
      float c[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
      float l    = std::sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
      for(int a = 0; a < 3; ++a)
        n0[a] = c[a] / l;
      for(uint32_t v : {i0, i1, i2})
        for(int a = 0; a < 3; ++a)
          geoNormal[size_t(v) * 3 + a] = n0[a];
    }
    if(hasUV)
    {
      const float* u0 = &d.texCoords0[size_t(i0) * 2];
      const float* u1 = &d.texCoords0[size_t(i1) * 2];
      const float* u2 = &d.texCoords0[size_t(i2) * 2];
      const float  d1[2] = {u1[0] - u0[0], u1[1] - u0[1]}, d2[2] = {u2[0] - u0[0], u2[1] - u0[1]};
      float        fct = 1.0f;
      const float  a   = d1[0] * d2[1] - d2[0] * d1[1];
      if(std::fabs(a) > 0.0f)
        fct = 1.0f / a;
      float tg[3], bt[3];
      for(int k = 0; k < 3; ++k)
      {
        tg[k] = fct * (d2[1] * e1[k] - d1[1] * e2[k]);
        bt[k] = fct * (d2[0] * e1[k] - d1[0] * e2[k]);
      }
      const float cx[3] = {tg[1] * bt[2] - tg[2] * bt[1], tg[2] * bt[0] - tg[0] * bt[2], tg[0] * bt[1] - tg[1] * bt[0]};
      const float hand  = (cx[0] * n0[0] + cx[1] * n0[1] + cx[2] * n0[2]) > 0.0f ? 1.0f : -1.0f;
      for(uint32_t v : {i0, i1, i2})
      {
        float* t = &d.tangents[size_t(v) * 4];
        t[0] += tg[0]; t[1] += tg[1]; t[2] += tg[2];
        t[3] = hand;
      }
    }
    else
    {
      float t[4];
      makeFastTangent(n0, t);
      for(uint32_t v : {i0, i1, i2})
        std::memcpy(&d.tangents[size_t(v) * 4], t, sizeof(t));
    }
  }
  for(size_t v = 0; v < nv; ++v)
  {
    float*       t = &d.tangents[v * 4];
    const float* n = hasNormal ? &d.normals[v * 3] : &geoNormal[v * 3];
    const float  dn = n[0] * t[0] + n[1] * t[1] + n[2] * t[2];
    float        o[3] = {t[0] - dn * n[0], t[1] - dn * n[1], t[2] - dn * n[2]};
    const float  l    = std::sqrt(o[0] * o[0] + o[1] * o[1] + o[2] * o[2]);
    for(int a = 0; a < 3; ++a)
      o[a] /= l;  // glm::normalize: a zero vector yields NaN, caught below
    const float l2 = o[0] * o[0] + o[1] * o[1] + o[2] * o[2];
    if(!(l2 >= 0.1f) || std::isnan(o[0]) || std::isnan(o[1]) || std::isnan(o[2]))
    {
      float ft[4];
      makeFastTangent(n, ft);
      o[0] = ft[0]; o[1] = ft[1]; o[2] = ft[2];
    }
    t[0] = o[0]; t[1] = o[1]; t[2] = o[2];
  }
}

// recomputeTangents(model, force, mikktspace = true) for one primitive (reference: createTangentsMikkTSpace,
// src/gltf_create_tangent.cpp:512-612).  Per-corner Mikkelsen tangents; a corner's tangent is kept when it is not nearly parallel
// to the vertex normal (|t . n| < 0.9) with the handedness flipped for this renderer's bitangent convention, else the fast tangent
// of the normal (:169-187); corners of a vertex whose tangents are compatible (within ~11 degrees, same handedness, or one of them
// degenerate: :194-218) share the vertex, the others get copies of it (:353-399), in corner order.  Returns the number of
// vertices added.
uint32_t recomputeTangentsMikk(RenderPrimitiveData& d)
{
  if(d.positions.empty() || d.normals.empty() || d.texCoords0.empty() || d.indices.size() < 3)
    return 0;
  const size_t nv = d.vertexCount, nc = d.indices.size() - d.indices.size() % 3;
  for(size_t c = 0; c < nc; ++c)
    if(d.indices[c] >= nv)
      return 0;
  std::vector<float> raw;
  mikkTangentSpaces(d.positions.data(), d.normals.data(), d.texCoords0.data(), d.indices.data(), nc / 3, raw);
  std::vector<float> corner(nc * 4);
  for(size_t c = 0; c < nc; ++c)
  {
    const float* n = &d.normals[size_t(d.indices[c]) * 3];
    const float* t = &raw[c * 4];
    float*       o = &corner[c * 4];
    if(std::fabs(t[0] * n[0] + t[1] * n[1] + t[2] * n[2]) < 0.9f)
    {
      o[0] = t[0]; o[1] = t[1]; o[2] = t[2]; o[3] = -t[3];
    }
    else
      makeFastTangent(n, o);
  }
  auto compatible = [](const float* a, const float* b) {
    const float la = std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]), lb = std::sqrt(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]);
    if(la < 1e-6f || lb < 1e-6f)
      return true;
    if((a[0] / la) * (b[0] / lb) + (a[1] / la) * (b[1] / lb) + (a[2] / la) * (b[2] / lb) < 0.98f)
      return false;
    return !(a[3] * b[3] < 0.0f);
  };
  // corners per vertex, in corner order
  std::vector<uint32_t> first(nv + 1, 0), list(nc);
  for(size_t c = 0; c < nc; ++c)
    ++first[size_t(d.indices[c]) + 1];
  for(size_t v = 0; v < nv; ++v)
    first[v + 1] += first[v];
  {
    std::vector<uint32_t> fill(first.begin(), first.end() - 1);
    for(size_t c = 0; c < nc; ++c)
      list[fill[d.indices[c]]++] = uint32_t(c);
  }
  // vertex v keeps the tangent of its first corner; every later corner joins the first group (the vertex itself, then its copies in
  // creation order) whose tangent it is compatible with, or founds a new copy
  d.tangents.assign(nv * 4, 0.0f);
  std::vector<uint32_t> copyOf;  // original vertex of every added vertex
  std::vector<float>    copyTangent;
  for(size_t v = 0; v < nv; ++v)
  {
    std::vector<uint32_t> groupVertex;  // vertex index of each tangent group of v
    for(uint32_t k = first[v]; k < first[v + 1]; ++k)
    {
      const uint32_t c  = list[k];
      const float*   t  = &corner[size_t(c) * 4];
      bool           ok = false;
      for(uint32_t g : groupVertex)
      {
        const float* gt = g < nv ? &d.tangents[size_t(g) * 4] : &copyTangent[size_t(g - nv) * 4];
        if(compatible(gt, t))
        {
          d.indices[c] = g;
          ok           = true;
          break;
        }
      }
      if(ok)
        continue;
      if(groupVertex.empty())
      {
        std::memcpy(&d.tangents[v * 4], t, 16);
        groupVertex.push_back(uint32_t(v));
      }
      else
      {
        const uint32_t g = uint32_t(nv + copyOf.size());
        copyOf.push_back(uint32_t(v));
        copyTangent.insert(copyTangent.end(), t, t + 4);
        groupVertex.push_back(g);
        d.indices[c] = g;
      }
    }
  }
  // append the copies to every attribute stream
  auto grow = [&](auto& stream, size_t width) {
    if(stream.empty())
      return;
    stream.reserve((nv + copyOf.size()) * width);
    for(uint32_t src : copyOf)
      for(size_t k = 0; k < width; ++k)
        stream.push_back(stream[size_t(src) * width + k]);
  };
  grow(d.positions, 3);
  grow(d.normals, 3);
  grow(d.texCoords0, 2);
  grow(d.texCoords1, 2);
  grow(d.colors, 1);
  d.tangents.insert(d.tangents.end(), copyTangent.begin(), copyTangent.end());
  d.vertexCount = uint32_t(nv + copyOf.size());
  return uint32_t(copyOf.size());
}

}  // namespace

uint32_t GltfScene::recomputeTangents(bool forceCreation, bool mikktspace)
{
  // reference: recomputeTangents / collectPrimitivesForTangents (src/gltf_create_tangent.cpp:619-677): primitives with positions,
  // normals and TEXCOORD_0; those without a TANGENT stream only when creation is forced
  uint32_t added = 0;
  for(RenderPrimitiveData& d : m_primData)
  {
    if(d.positions.empty() || d.normals.empty() || d.texCoords0.empty())
      continue;
    if(d.tangents.empty() && !forceCreation)
      continue;
    if(mikktspace)
      added += recomputeTangentsMikk(d);
    else
      createSimpleTangents(d);
  }
  finalizeDesc();  // the streams may have been reallocated
  return added;
}

//----------------------------------------------------------------------------------------------------------------------
// Scene files are untrusted input: every byte count / offset / element count read from the JSON goes through these two helpers
// before it is used in pointer arithmetic.
namespace {
// JSON number -> size_t; false for negative, non-finite, fractional-overflowing or > 2^53 values (a cast of those is undefined)
bool toSize(double v, size_t& out)
{
  if(!(v >= 0.0) || !(v <= 9007199254740992.0))
    return false;
  out = size_t(v);
  return true;
}
bool sizeField(const Value& obj, const char* key, size_t& out)
{
  const Value& v = obj[key];
  out            = 0;
  return !v.isNumber() || toSize(v.number(0), out);
}
// [off, off + (count - 1) * stride + elem) inside [0, size), without wrapping
bool rangeFits(size_t size, size_t off, size_t count, size_t stride, size_t elem)
{
  if(count == 0)
    return off <= size;
  if(elem > size || off > size - elem)
    return false;
  const size_t room = size - elem - off;  // what (count - 1) * stride may use
  return stride == 0 ? true : (count - 1) <= room / stride;
}
}  // namespace

const uint8_t* GltfScene::bufferViewData(int bufferView, size_t& size, size_t& stride)
{
  size = stride = 0;
  if(bufferView < 0)
    return nullptr;
  const Value& bv = m_doc["bufferViews"][size_t(bufferView)];
  if(!bv.isObject())
    return nullptr;
  int    buffer = getInt(bv, "buffer", -1);
  size_t offset = 0;
  if(!sizeField(bv, "byteOffset", offset) || !sizeField(bv, "byteLength", size) || !sizeField(bv, "byteStride", stride))
    return nullptr;
  if(buffer < 0 || size_t(buffer) >= m_buffers.size())
    return nullptr;
  const size_t bufSize = m_buffers[size_t(buffer)].size();
  if(offset > bufSize || size > bufSize - offset)
    return nullptr;
  return m_buffers[size_t(buffer)].data() + offset;
}

// Accessor decode honouring byteStride, normalisation and sparse overrides
// (reference: src/tinygltf_utils.hpp:832-960).
bool GltfScene::readAccessorFloats(int accessor, int expectedComponents, std::vector<float>& out, int* outComponents)
{
  const Value& acc = m_doc["accessors"][size_t(accessor)];
  if(!acc.isObject())
    return false;
  const int    nc         = componentCount(acc["type"].string());
  const int    ctype      = getInt(acc, "componentType", 0);
  const int    csize      = componentSize(ctype);
  size_t       count      = 0;
  const bool   normalized = acc["normalized"].boolean(false);
  if(nc == 0 || csize == 0 || !sizeField(acc, "count", count) || count > (size_t(1) << 32))
    return false;
  if(outComponents)
    *outComponents = nc;
  const int oc = expectedComponents > 0 ? expectedComponents : nc;
  out.assign(count * size_t(oc), 0.0f);
  if(expectedComponents == 4 && nc == 3)  // vec3 -> vec4 promotion pads with 1 (colours)
    for(size_t i = 0; i < count; ++i)
      out[i * 4 + 3] = 1.0f;

  int bvIndex = getInt(acc, "bufferView", -1);
  if(bvIndex >= 0)
  {
    size_t         size = 0, stride = 0;
    const uint8_t* base = bufferViewData(bvIndex, size, stride);
    if(!base)
      return false;
    size_t off = 0;
    if(!sizeField(acc, "byteOffset", off))
      return false;
    if(stride == 0)
      stride = size_t(nc) * size_t(csize);
    if(!rangeFits(size, off, count, stride, size_t(nc) * size_t(csize)))
      return false;
    for(size_t i = 0; i < count; ++i)
    {
      const uint8_t* p = base + off + i * stride;
      for(int c = 0; c < std::min(nc, oc); ++c)
        out[i * size_t(oc) + size_t(c)] = decodeComponent(p + size_t(c) * size_t(csize), ctype, normalized);
    }
  }
  const Value& sparse = acc["sparse"];
  if(sparse.isObject())
  {
    size_t       scount = 0;
    if(!sizeField(sparse, "count", scount) || scount > count)  // glTF 2.0: sparse.count <= accessor.count
      return false;
    const Value& sidx   = sparse["indices"];
    const Value& sval   = sparse["values"];
    size_t       isz = 0, istride = 0, vsz = 0, vstride = 0;
    const uint8_t* ibase = bufferViewData(getInt(sidx, "bufferView", -1), isz, istride);
    const uint8_t* vbase = bufferViewData(getInt(sval, "bufferView", -1), vsz, vstride);
    if(!ibase || !vbase)
      return false;
    int    ictype = getInt(sidx, "componentType", 5125);
    size_t ioff = 0, voff = 0;
    const size_t isize = size_t(componentSize(ictype)), vsize = size_t(nc) * size_t(csize);
    // the index and value arrays are tightly packed (glTF 2.0 §3.6.2.3): both must lie inside their buffer views
    if(isize == 0 || !sizeField(sidx, "byteOffset", ioff) || !sizeField(sval, "byteOffset", voff) || !rangeFits(isz, ioff, scount, isize, isize)
       || !rangeFits(vsz, voff, scount, vsize, vsize))
      return false;
    for(size_t s = 0; s < scount; ++s)
    {
      uint32_t target = decodeUint(ibase + ioff + s * size_t(componentSize(ictype)), ictype);
      if(target >= count)
        continue;
      const uint8_t* p = vbase + voff + s * size_t(nc) * size_t(csize);
      for(int c = 0; c < std::min(nc, oc); ++c)
        out[size_t(target) * size_t(oc) + size_t(c)] = decodeComponent(p + size_t(c) * size_t(csize), ctype, normalized);
    }
  }
  return true;
}

bool GltfScene::readAccessorUints(int accessor, std::vector<uint32_t>& out)
{
  const Value& acc = m_doc["accessors"][size_t(accessor)];
  if(!acc.isObject())
    return false;
  const int    ctype = getInt(acc, "componentType", 0);
  const int    csize = componentSize(ctype);
  size_t       count = 0, off = 0;
  size_t       size = 0, stride = 0;
  const uint8_t* base = bufferViewData(getInt(acc, "bufferView", -1), size, stride);
  if(!base || csize == 0 || !sizeField(acc, "count", count) || count > (size_t(1) << 32) || !sizeField(acc, "byteOffset", off))
    return false;
  if(stride == 0)
    stride = size_t(csize);
  if(!rangeFits(size, off, count, stride, size_t(csize)))
    return false;
  out.resize(count);
  for(size_t i = 0; i < count; ++i)
    out[i] = decodeUint(base + off + i * stride, ctype);
  return true;
}

//----------------------------------------------------------------------------------------------------------------------
bool GltfScene::load(const std::string& filename)
{
  std::vector<uint8_t> file;
  if(!readFile(filename, file))
  {
    m_error = "cannot read " + filename;
    return false;
  }
  std::string          jsonText;
  std::vector<uint8_t> glbBin;
  bool                 isGlb = file.size() >= 12 && memcmp(file.data(), "glTF", 4) == 0;
  if(isGlb)
  {
    size_t pos = 12;
    while(pos + 8 <= file.size())
    {
      uint32_t len, type;
      memcpy(&len, &file[pos], 4);
      memcpy(&type, &file[pos + 4], 4);
      pos += 8;
      if(pos + len > file.size())
        break;
      if(type == 0x4E4F534A)  // JSON
        jsonText.assign(reinterpret_cast<const char*>(&file[pos]), len);
      else if(type == 0x004E4942 && glbBin.empty())  // BIN
        glbBin.assign(file.begin() + long(pos), file.begin() + long(pos + len));
      pos += (len + 3u) & ~3u;
    }
  }
  else
    jsonText.assign(reinterpret_cast<const char*>(file.data()), file.size());

  try
  {
    m_doc = mijson::parse(jsonText);
  }
  catch(const std::exception& e)
  {
    m_error = e.what();
    return false;
  }
  // What the loader implements of the reference's list (src/gltf_scene.cpp:213-254, + EXT_texture_webp of src/renderer.cpp:783): a file that REQUIRES
  // anything else is refused like there (SceneValidator::validateModelExtensions, src/gltf_scene_validator.cpp:295-322), a merely USED one gets a warning.
  // Not implemented and therefore refused when required: KHR_draco_mesh_compression (a build option of the reference).  KHR_texture_basisu counts as
  // supported: its KTX2 container is read (image_loader.cpp), a BasisLZ / UASTC payload inside is an undecodable IMAGE, not an unloadable file.
  {
    static const char* const supported[] = {
        "EXT_mesh_gpu_instancing", "EXT_mesh_opacity_micromap", "EXT_meshopt_compression", "EXT_texture_webp", "KHR_animation_pointer", "KHR_interactivity", "KHR_lights_punctual",
        "KHR_materials_anisotropy", "KHR_materials_clearcoat", "KHR_materials_diffuse_transmission", "KHR_materials_dispersion", "KHR_materials_displacement",
        "KHR_materials_emissive_strength", "KHR_materials_ior", "KHR_materials_iridescence", "KHR_materials_pbrSpecularGlossiness", "KHR_materials_retroreflection",
        "KHR_materials_sheen", "KHR_materials_specular", "KHR_materials_transmission", "KHR_materials_unlit", "KHR_materials_variants",
        "KHR_materials_volume_scatter", "KHR_materials_volume", "KHR_mesh_quantization", "KHR_meshopt_compression", "KHR_node_hoverability", "KHR_node_selectability", "KHR_node_visibility",
        "KHR_texture_basisu", "KHR_texture_transform", "MSFT_texture_dds", "NV_attributes_iray"};
    auto isSupported = [&](const std::string& e) {
      for(const char* s : supported)
        if(e == s)
          return true;
      return false;
    };
    const Value& required = m_doc["extensionsRequired"];
    for(size_t i = 0; i < required.size(); ++i)
      if(required[i].isString() && !isSupported(required[i].str))
      {
        m_error = "Required extension unsupported : " + required[i].str;
        return false;
      }
    const Value& used = m_doc["extensionsUsed"];
    for(size_t i = 0; i < used.size(); ++i)
      if(used[i].isString() && !isSupported(used[i].str))
        fprintf(stderr, "[mihost] Used extension unsupported : %s\n", used[i].str.c_str());
  }
  const std::string baseDir = dirOf(filename);
  m_buffers.clear();
  const Value& buffers = m_doc["buffers"];
  for(size_t i = 0; i < buffers.size(); ++i)
  {
    std::vector<uint8_t> data;
    const Value&         uri = buffers[i]["uri"];
    if(uri.isString())
    {
      if(!loadUri(uri.str, baseDir, data))
      {
        m_error = "cannot load buffer " + uri.str;
        return false;
      }
    }
    else if(i == 0 && isGlb)
      data = glbBin;
    m_buffers.push_back(std::move(data));
  }
  if(!decompressMeshopt())
    return false;
  return parse(baseDir);
}

// EXT_meshopt_compression / KHR_meshopt_compression (reference: Scene::decompressMeshoptExtension, src/gltf_scene.cpp:372-470, which hands the streams
// to meshoptimizer): every buffer view that carries the extension is decoded INTO the region of the buffer it nominally views -- the "fallback"
// buffer, which in a compressed-only file has a length and no data -- and the accessors then read it like any other.  Sizes are untrusted.
bool GltfScene::decompressMeshopt()
{
  const Value& views   = m_doc["bufferViews"];
  const Value& buffers = m_doc["buffers"];
  // bytes that came WITH the file, counted once before any fallback buffer is materialised: the bound on those buffers is a multiple
  // of this figure, and it must not grow with every fallback already filled (k empty fallbacks would compound to 80^k)
  double loaded = 0.0, fallbackTotal = 0.0;
  for(const std::vector<uint8_t>& b : m_buffers)
    loaded += double(b.size());
  for(size_t i = 0; i < views.size(); ++i)
  {
    const Value* e = &ext(views[i], "KHR_meshopt_compression");
    if(!e->isObject())
      e = &ext(views[i], "EXT_meshopt_compression");
    if(!e->isObject())
      continue;
    auto bad = [&](const std::string& what) {
      m_error = "meshopt_compression decompression failed: bufferView " + std::to_string(i) + ": " + what;
      return false;
    };
    const double srcBuffer = (*e)["buffer"].number(-1.0), srcOffset = (*e)["byteOffset"].number(0.0), srcLength = (*e)["byteLength"].number(-1.0);
    const double stride = (*e)["byteStride"].number(-1.0), count = (*e)["count"].number(-1.0);
    const double dstBuffer = views[i]["buffer"].number(-1.0), dstOffset = views[i]["byteOffset"].number(0.0), dstLength = views[i]["byteLength"].number(-1.0);
    const double limit = 1.0e12;  // (well inside what a double counts exactly and a size_t holds)
    if(!(srcBuffer >= 0 && srcBuffer < double(m_buffers.size()) && dstBuffer >= 0 && dstBuffer < double(m_buffers.size())))
      return bad("buffer index out of range");
    if(!(srcOffset >= 0 && srcLength >= 0 && srcOffset + srcLength <= double(m_buffers[size_t(srcBuffer)].size())))
      return bad("the compressed bytes lie outside their buffer");
    if(!(stride > 0 && stride <= 256 && count >= 0 && count < limit && dstOffset >= 0 && dstLength >= 0 && dstLength < limit && count * stride <= dstLength))
      return bad("count x byteStride exceeds the buffer view");
    std::vector<uint8_t>& dst = m_buffers[size_t(dstBuffer)];
    if(dst.empty())
    {
      // a fallback buffer without data: give it its declared length (bounded: what the views into it can address)
      // ... and by what the streams of this file can possibly expand to: the densest vertex stream spends 4 header bytes on a plane of 256 zero
      // differences (64 : 1), an index stream at least one byte per 12-byte triangle
      // (`loaded`: the file's own bytes; the bound holds for ALL fallback buffers together)
      const double declared = buffers[size_t(dstBuffer)]["byteLength"].number(0.0);
      if(!(declared >= 0 && fallbackTotal + declared <= 80.0 * loaded + 4096.0))
        return bad("fallback buffer length out of proportion to the file");
      fallbackTotal += declared;
      dst.assign(size_t(declared), 0);
    }
    if(!(dstOffset + dstLength <= double(dst.size())))
      return bad("the buffer view lies outside its buffer");
    if(size_t(srcBuffer) == size_t(dstBuffer) && srcOffset < dstOffset + dstLength && dstOffset < srcOffset + srcLength)
      return bad("the compressed bytes overlap the region they decode into");
    const uint8_t* src = m_buffers[size_t(srcBuffer)].data() + size_t(srcOffset);
    uint8_t*       out = dst.data() + size_t(dstOffset);
    const size_t   n = size_t(count), st = size_t(stride);
    const std::string mode = (*e)["mode"].isString() ? (*e)["mode"].str : std::string(), filter = (*e)["filter"].isString() ? (*e)["filter"].str : std::string("NONE");
    std::string       err;
    bool              ok = false;
    if(mode == "ATTRIBUTES")
      ok = meshopt::decodeVertexBuffer(out, n, st, src, size_t(srcLength), err);
    else if(mode == "TRIANGLES")
      ok = meshopt::decodeIndexBuffer(out, n, st, src, size_t(srcLength), err);
    else if(mode == "INDICES")
      ok = meshopt::decodeIndexSequence(out, n, st, src, size_t(srcLength), err);
    else
      err = "unknown mode \"" + mode + "\"";
    if(ok && filter == "OCTAHEDRAL")
      ok = meshopt::filterOctahedral(out, n, st, err);
    else if(ok && filter == "QUATERNION")
      ok = meshopt::filterQuaternion(out, n, st, err);
    else if(ok && filter == "EXPONENTIAL")
      ok = meshopt::filterExponential(out, n, st, err);
    else if(ok && filter != "NONE")
    {
      ok  = false;
      err = "unknown filter \"" + filter + "\"";
    }
    if(!ok)
      return bad(err);
  }
  return true;
}

//----------------------------------------------------------------------------------------------------------------------
uint16_t GltfScene::addTextureInfo(const Value& tinfo)  // reference: src/gltf_material_cache.cpp:60-98
{
  if(!tinfo.isObject())
    return 0;
  int index = getInt(tinfo, "index", -1);
  if(index < 0)
    return 0;
  MiGltfTextureInfo ti{};
  ti.index    = index;
  ti.texCoord = std::min(getInt(tinfo, "texCoord", 0), 1);
  // KHR_texture_transform (reference: src/tinygltf_utils.hpp:60-78 updateTransform, packing
  // src/gltf_material_cache.cpp:81-84)
  float offset[2] = {0, 0}, scale[2] = {1, 1}, rotation = 0;
  const Value& tt = ext(tinfo, "KHR_texture_transform");
  if(tt.isObject())
  {
    getFloats(tt, "offset", 2, offset);
    getFloats(tt, "scale", 2, scale);
    rotation = getFloat(tt, "rotation", 0.0f);
  }
  float cosR = std::cos(rotation), sinR = std::sin(rotation);
  // glm::mat3 uvTransform columns: c0 = (sx cosR, sx sinR, tx), c1 = (-sy sinR, sy cosR, ty), c2 = (0,0,1)
  float c0[3] = {scale[0] * cosR, scale[0] * sinR, offset[0]};
  float c1[3] = {-scale[1] * sinR, scale[1] * cosR, offset[1]};
  ti.uvTransform[0] = c0[0];
  ti.uvTransform[1] = c1[0];
  ti.uvTransform[2] = c0[1];
  ti.uvTransform[3] = c1[1];
  ti.uvTransform[4] = c0[2];
  ti.uvTransform[5] = c1[2];
  uint16_t idx      = uint16_t(m_textureInfos.size());
  m_textureInfos.push_back(ti);
  return idx;
}

void GltfScene::buildMaterials()  // reference: src/gltf_material_cache.cpp:103-260 + defaults src/tinygltf_utils.hpp:50-250
{
  m_textureInfos.clear();
  MiGltfTextureInfo sentinel{};
  sentinel.uvTransform[0] = 1.0f;  // float3x2(1): identity on the 2x2 block
  sentinel.uvTransform[3] = 1.0f;
  sentinel.index          = -1;
  m_textureInfos.push_back(sentinel);
  m_materials.clear();

  const Value& mats  = m_doc["materials"];
  size_t       count = std::max<size_t>(mats.size(), 1);  // at least one material (reference: src/gltf_scene.cpp:1391-1395)
  for(size_t i = 0; i < count; ++i)
  {
    static const Value emptyObj = [] {
      Value v;
      v.type = Value::Object;
      return v;
    }();
    const Value& src = i < mats.size() ? mats[i] : emptyObj;
    MiGltfShadeMaterial d;
    memset(&d, 0, sizeof(d));
    // struct defaults (reference: shaders/gltf_scene_io.h.slang:147-310)
    for(int c = 0; c < 4; ++c) d.pbrBaseColorFactor[c] = 1.0f;
    d.normalTextureScale = 1.0f;
    d.pbrRoughnessFactor = 1.0f;
    d.pbrMetallicFactor  = 1.0f;
    d.alphaCutoff        = 0.5f;
    d.occlusionStrength  = 1.0f;
    for(int c = 0; c < 3; ++c) d.attenuationColor[c] = 1.0f;
    d.ior = 1.5f;
    for(int c = 0; c < 3; ++c) d.specularColorFactor[c] = 1.0f;
    d.iridescenceThicknessMinimum = 100.0f;
    d.iridescenceThicknessMaximum = 400.0f;
    d.iridescenceIor              = 1.3f;
    for(int c = 0; c < 4; ++c) d.pbrDiffuseFactor[c] = 1.0f;
    for(int c = 0; c < 3; ++c) d.pbrSpecularFactor[c] = 1.0f;
    d.pbrGlossinessFactor = 1.0f;
    for(int c = 0; c < 3; ++c) d.diffuseTransmissionColor[c] = 1.0f;

    std::string alphaMode = src["alphaMode"].string("OPAQUE");
    d.alphaMode           = alphaMode == "OPAQUE" ? 0 : (alphaMode == "MASK" ? 1 : 2);
    d.alphaCutoff         = getFloat(src, "alphaCutoff", 0.5f);
    d.doubleSided         = src["doubleSided"].boolean(false) ? 1 : 0;
    const Value& pbr      = src["pbrMetallicRoughness"];
    getFloats(pbr, "baseColorFactor", 4, d.pbrBaseColorFactor);
    d.pbrMetallicFactor  = getFloat(pbr, "metallicFactor", 1.0f);
    d.pbrRoughnessFactor = getFloat(pbr, "roughnessFactor", 1.0f);
    d.normalTextureScale = getFloat(src["normalTexture"], "scale", 1.0f);
    d.occlusionStrength  = getFloat(src["occlusionTexture"], "strength", 1.0f);
    getFloats(src, "emissiveFactor", 3, d.emissiveFactor);

    d.emissiveTexture             = addTextureInfo(src["emissiveTexture"]);
    d.normalTexture               = addTextureInfo(src["normalTexture"]);
    d.pbrBaseColorTexture         = addTextureInfo(pbr["baseColorTexture"]);
    d.pbrMetallicRoughnessTexture = addTextureInfo(pbr["metallicRoughnessTexture"]);
    d.occlusionTexture            = addTextureInfo(src["occlusionTexture"]);

    const Value& tr       = ext(src, "KHR_materials_transmission");
    d.transmissionFactor  = getFloat(tr, "transmissionFactor", 0.0f);
    d.transmissionTexture = addTextureInfo(tr["transmissionTexture"]);

    d.ior = getFloat(ext(src, "KHR_materials_ior"), "ior", 1.5f);

    const Value& vol = ext(src, "KHR_materials_volume");
    getFloats(vol, "attenuationColor", 3, d.attenuationColor);
    d.thicknessFactor     = getFloat(vol, "thicknessFactor", 0.0f);
    d.attenuationDistance = getFloat(vol, "attenuationDistance", FLT_MAX);
    d.thicknessTexture    = addTextureInfo(vol["thicknessTexture"]);

    const Value& cc             = ext(src, "KHR_materials_clearcoat");
    d.clearcoatFactor           = getFloat(cc, "clearcoatFactor", 0.0f);
    d.clearcoatRoughness        = getFloat(cc, "clearcoatRoughnessFactor", 0.0f);
    d.clearcoatRoughnessTexture = addTextureInfo(cc["clearcoatRoughnessTexture"]);
    d.clearcoatTexture          = addTextureInfo(cc["clearcoatTexture"]);
    d.clearcoatNormalTexture    = addTextureInfo(cc["clearcoatNormalTexture"]);

    const Value& sp  = ext(src, "KHR_materials_specular");
    d.specularFactor = getFloat(sp, "specularFactor", 1.0f);
    getFloats(sp, "specularColorFactor", 3, d.specularColorFactor);
    d.specularTexture      = addTextureInfo(sp["specularTexture"]);
    d.specularColorTexture = addTextureInfo(sp["specularColorTexture"]);

    float strength = getFloat(ext(src, "KHR_materials_emissive_strength"), "emissiveStrength", 1.0f);
    for(int c = 0; c < 3; ++c)
      d.emissiveFactor[c] *= strength;

    d.unlit = ext(src, "KHR_materials_unlit").isObject() ? 1 : 0;

    const Value& ir               = ext(src, "KHR_materials_iridescence");
    d.iridescenceFactor           = getFloat(ir, "iridescenceFactor", 0.0f);
    d.iridescenceIor              = getFloat(ir, "iridescenceIor", 1.3f);
    d.iridescenceThicknessMinimum = getFloat(ir, "iridescenceThicknessMinimum", 100.0f);
    d.iridescenceThicknessMaximum = getFloat(ir, "iridescenceThicknessMaximum", 400.0f);
    d.iridescenceTexture          = addTextureInfo(ir["iridescenceTexture"]);
    d.iridescenceThicknessTexture = addTextureInfo(ir["iridescenceThicknessTexture"]);

    const Value& an         = ext(src, "KHR_materials_anisotropy");
    float        anRotation = getFloat(an, "anisotropyRotation", 0.0f);
    d.anisotropyRotation[0] = std::sin(anRotation);
    d.anisotropyRotation[1] = std::cos(anRotation);
    d.anisotropyStrength    = getFloat(an, "anisotropyStrength", 0.0f);
    d.anisotropyTexture     = addTextureInfo(an["anisotropyTexture"]);

    const Value& sh = ext(src, "KHR_materials_sheen");
    getFloats(sh, "sheenColorFactor", 3, d.sheenColorFactor);
    d.sheenRoughnessFactor  = getFloat(sh, "sheenRoughnessFactor", 0.0f);
    d.sheenColorTexture     = addTextureInfo(sh["sheenColorTexture"]);
    d.sheenRoughnessTexture = addTextureInfo(sh["sheenRoughnessTexture"]);

    d.dispersion = getFloat(ext(src, "KHR_materials_dispersion"), "dispersion", 0.0f);

    const Value& sg = ext(src, "KHR_materials_pbrSpecularGlossiness");
    if(sg.isObject())
    {
      d.pbrModel = MI_PBR_SPECULAR_GLOSSINESS;
      getFloats(sg, "diffuseFactor", 4, d.pbrDiffuseFactor);
      getFloats(sg, "specularFactor", 3, d.pbrSpecularFactor);
      d.pbrGlossinessFactor = getFloat(sg, "glossinessFactor", 1.0f);
    }
    d.pbrDiffuseTexture            = addTextureInfo(sg["diffuseTexture"]);
    d.pbrSpecularGlossinessTexture = addTextureInfo(sg["specularGlossinessTexture"]);

    const Value& dt             = ext(src, "KHR_materials_diffuse_transmission");
    d.diffuseTransmissionFactor = getFloat(dt, "diffuseTransmissionFactor", 0.0f);
    getFloats(dt, "diffuseTransmissionColorFactor", 3, d.diffuseTransmissionColor);
    d.diffuseTransmissionTexture      = addTextureInfo(dt["diffuseTransmissionTexture"]);
    d.diffuseTransmissionColorTexture = addTextureInfo(dt["diffuseTransmissionColorTexture"]);

    const Value& rr          = ext(src, "KHR_materials_retroreflection");
    d.retroreflectionFactor  = getFloat(rr, "retroreflectionFactor", 0.0f);
    d.retroreflectionTexture = addTextureInfo(rr["retroreflectionTexture"]);

    const Value& vs = ext(src, "KHR_materials_volume_scatter");
    getFloats(vs, "multiscatterColor", 3, d.multiscatterColorFactor);
    d.scatterAnisotropy = getFloat(vs, "scatterAnisotropy", 0.0f);

    m_materials.push_back(d);
  }
}

//----------------------------------------------------------------------------------------------------------------------
void GltfScene::buildTextures(const std::string& baseDir)
{
  // sRGB image set (reference: src/gltf_scene_vk.cpp:1102-1154)
  std::vector<bool> srgbImage(m_doc["images"].size(), false);
  // Effective image of a texture: `source`, overridden by the `source` of EXT_texture_webp, MSFT_texture_dds,
  // KHR_texture_basisu in that order, the last one present winning (reference: src/tinygltf_utils.cpp:42-47, :718-732)
  auto textureImageOf = [&](const Value& t) -> int {
    if(!t.isObject())
      return -1;
    int img = getInt(t, "source", -1);
    for(const char* name : {"EXT_texture_webp", "MSFT_texture_dds", "KHR_texture_basisu"})
    {
      const Value& e = ext(t, name);
      if(e.isObject())
        img = getInt(e, "source", img);
    }
    return img;
  };
  auto textureImage = [&](int texID) -> int { return textureImageOf(m_doc["textures"][size_t(texID)]); };
  auto markSrgb = [&](const Value& tinfo) {
    if(!tinfo.isObject())
      return;
    int texID = getInt(tinfo, "index", -1);
    if(texID < 0)
      return;
    int img = textureImage(texID);
    if(img >= 0 && size_t(img) < srgbImage.size())
      srgbImage[size_t(img)] = true;
  };
  const Value& mats = m_doc["materials"];
  for(size_t i = 0; i < mats.size(); ++i)
  {
    const Value& m = mats[i];
    markSrgb(m["pbrMetallicRoughness"]["baseColorTexture"]);
    markSrgb(m["emissiveTexture"]);
    markSrgb(ext(m, "KHR_materials_specular")["specularColorTexture"]);
    markSrgb(ext(m, "KHR_materials_sheen")["sheenColorTexture"]);
    markSrgb(ext(m, "KHR_materials_pbrSpecularGlossiness")["diffuseTexture"]);
    markSrgb(ext(m, "KHR_materials_pbrSpecularGlossiness")["specularGlossinessTexture"]);
  }
  const Value& texs = m_doc["textures"];
  for(size_t i = 0; i < texs.size(); ++i)
    if(texs[i]["extras"]["gamma"].number(0.0) > 1.0)
    {
      int img = textureImageOf(texs[i]);
      if(img >= 0 && size_t(img) < srgbImage.size())
        srgbImage[size_t(img)] = true;
    }

  // Decode every image once.
  const Value&       images = m_doc["images"];
  std::vector<Image> decoded(images.size());
  std::vector<bool>  decodedOk(images.size(), false);
  for(size_t i = 0; i < images.size(); ++i)
  {
    std::vector<uint8_t> bytes;
    const Value&         img = images[i];
    bool                 ok  = false;
    if(img["uri"].isString())
      ok = loadUri(img["uri"].str, baseDir, bytes);
    else if(img["bufferView"].isNumber())
    {
      size_t         size = 0, stride = 0;
      const uint8_t* p = bufferViewData(img["bufferView"].integer(), size, stride);
      if(p)
      {
        bytes.assign(p, p + size);
        ok = true;
      }
    }
    std::string err;
    if(!ok || !decodeImage(bytes.data(), bytes.size(), decoded[i], &err))
    {
      fprintf(stderr, "[mihost] image %zu not decodable (%s): using 1x1 magenta\n", i, err.c_str());
      decoded[i] = magentaImage();  // reference: src/gltf_scene_vk.cpp:1057-1060
    }
    else
      decodedOk[i] = true;
  }

  m_textures.clear();
  m_textures.resize(texs.size());
  for(size_t i = 0; i < texs.size(); ++i)
  {
    TextureData& t   = m_textures[i];
    int          img = textureImageOf(texs[i]);
    // The containers this front end does not decode (KTX2 / Basis, WebP, BC6H / BC7 DDS): where the file also carries the
    // core PNG / JPEG `source` as its fallback, that one is used instead of the magenta placeholder (the reference decodes
    // the extension image itself).  The sRGB classification goes with the image actually used.
    const int core = getInt(texs[i], "source", -1);
    if(img != core && img >= 0 && size_t(img) < decoded.size() && !decodedOk[size_t(img)] && core >= 0 && size_t(core) < decoded.size()
       && decodedOk[size_t(core)])
    {
      if(srgbImage[size_t(img)])
        srgbImage[size_t(core)] = true;
      img = core;
    }
    Image        base = (img >= 0 && size_t(img) < decoded.size()) ? decoded[size_t(img)] : magentaImage();
    t.srgb            = (img >= 0 && size_t(img) < srgbImage.size()) ? bool(srgbImage[size_t(img)]) : false;
    t.width           = base.width;
    t.height          = base.height;
    buildMipChain(base, t.srgb, t.levels);
    // sampler (reference: src/gltf_scene_vk.cpp:909-947; note mipmapMode follows magFilter)
    int s = getInt(texs[i], "sampler", -1);
    if(s >= 0)
    {
      const Value& smp  = m_doc["samplers"][size_t(s)];
      auto         filt = [](int v) { return (v == 9728 || v == 9984 || v == 9986) ? MI_FILTER_NEAREST : MI_FILTER_LINEAR; };
      auto         wrap = [](int v) { return v == 33071 ? MI_WRAP_CLAMP_TO_EDGE : (v == 33648 ? MI_WRAP_MIRRORED_REPEAT : MI_WRAP_REPEAT); };
      int          minF = getInt(smp, "minFilter", -1), magF = getInt(smp, "magFilter", -1);
      if(minF > -1)
        t.minFilter = filt(minF);
      if(magF > -1)
      {
        t.magFilter  = filt(magF);
        t.mipmapMode = filt(magF);
      }
      t.wrapS = wrap(getInt(smp, "wrapS", 10497));
      t.wrapT = wrap(getInt(smp, "wrapT", 10497));
    }
  }
}

//----------------------------------------------------------------------------------------------------------------------
void GltfScene::buildPrimitives(std::map<std::string, int>& primMap)  // reference: src/gltf_scene.cpp:2139-2165
{
  m_primData.clear();
  const Value& meshes = m_doc["meshes"];
  for(size_t i = 0; i < meshes.size(); ++i)
  {
    const Value& prims = meshes[i]["primitives"];
    for(size_t j = 0; j < prims.size(); ++j)
    {
      const Value& prim = prims[j];
      std::string  key  = primitiveKeyWithTangents(m_doc, prim);
      if(primMap.count(key))
        continue;
      primMap[key] = int(m_primData.size());
      RenderPrimitiveData d;
      d.meshID            = int(i);
      const Value& attrs  = prim["attributes"];
      int          posAcc = getInt(attrs, "POSITION", -1);
      if(posAcc >= 0)
        readAccessorFloats(posAcc, 3, d.positions);
      d.vertexCount = uint32_t(d.positions.size() / 3);
      if(attrs.has("NORMAL"))
        readAccessorFloats(attrs["NORMAL"].integer(), 3, d.normals);
      if(attrs.has("TEXCOORD_0"))
        readAccessorFloats(attrs["TEXCOORD_0"].integer(), 2, d.texCoords0);
      if(attrs.has("TEXCOORD_1"))
        readAccessorFloats(attrs["TEXCOORD_1"].integer(), 2, d.texCoords1);
      if(attrs.has("TANGENT"))
        readAccessorFloats(attrs["TANGENT"].integer(), 4, d.tangents);
      if(attrs.has("COLOR_0"))  // reference: src/gltf_scene_vk.cpp:766-798
      {
        std::vector<float> col;
        readAccessorFloats(attrs["COLOR_0"].integer(), 4, col);
        d.colors.resize(col.size() / 4);
        for(size_t v = 0; v < d.colors.size(); ++v)
          d.colors[v] = packUnorm4x8(&col[v * 4]);
      }
      // attribute streams shorter than POSITION are treated as absent
      auto fit = [&](std::vector<float>& a, size_t nc) {
        if(!a.empty() && a.size() != size_t(d.vertexCount) * nc)
          a.clear();
      };
      fit(d.normals, 3);
      fit(d.texCoords0, 2);
      fit(d.texCoords1, 2);
      fit(d.tangents, 4);
      if(!d.colors.empty() && d.colors.size() != d.vertexCount)
        d.colors.clear();

      int idxAcc = getInt(prim, "indices", -1);
      if(idxAcc >= 0)
        readAccessorUints(idxAcc, d.indices);
      else  // reference: src/gltf_scene_vk.cpp:823-830
      {
        d.indices.resize(d.vertexCount);
        for(uint32_t v = 0; v < d.vertexCount; ++v)
          d.indices[v] = v;
      }
      int mode = getInt(prim, "mode", 4);
      if(mode != 4)
        d.indices.clear();  // only triangle lists reach the acceleration structure
      d.indices.resize(d.indices.size() / 3 * 3);
      for(uint32_t& ix : d.indices)  // defensive: never index past the vertex streams
        if(ix >= d.vertexCount)
          ix = 0;
      if(d.tangents.empty() && d.vertexCount > 0 && needsGeneratedTangents(m_doc, prim))
        createSimpleTangents(d);
      m_primData.push_back(std::move(d));
    }
  }
}

//----------------------------------------------------------------------------------------------------------------------
void GltfScene::traverse(int nodeID, const mx::mat4& parent, bool parentVisible, const std::map<std::string, int>& primMap)
{
  const Value& node = m_doc["nodes"][size_t(nodeID)];
  if(!node.isObject())
    return;
  // glTF node hierarchies are strict trees (specification 3.5.2); a file that closes a cycle must not recurse without end
  if(size_t(nodeID) >= m_onPath.size() || m_onPath[size_t(nodeID)])
    return;
  m_onPath[size_t(nodeID)] = 1;
  m_curPath.push_back(nodeID);
  mx::mat4 world = mx::mul(parent, localMatrix(nodeID));
  // KHR_node_visibility cascades (reference: src/gltf_scene.cpp:1907-1948)
  bool         visible = parentVisible;
  const Value& vis     = ext(node, "KHR_node_visibility");
  if(vis.isObject() && vis["visible"].type == Value::Bool && !vis["visible"].b)
    visible = false;

  // light (reference: src/gltf_scene.cpp:2269-2300, src/gltf_scene_vk.cpp:1354-1392)
  const Value& lightExt = ext(node, "KHR_lights_punctual");
  if(lightExt.isObject() && lightExt["light"].isNumber())
  {
    int          li     = lightExt["light"].integer();
    const Value& lights = ext(m_doc, "KHR_lights_punctual")["lights"];
    if(li >= 0 && size_t(li) < lights.size())
    {
      const Value& gl = lights[size_t(li)];
      MiGltfLight  info{};
      placeLight(info, world);
      info.innerAngle   = getFloat(gl["spot"], "innerConeAngle", 0.0f);
      info.outerAngle   = getFloat(gl["spot"], "outerConeAngle", 0.7853981633974483f);
      info.color[0] = info.color[1] = info.color[2] = 1.0f;
      getFloats(gl, "color", 3, info.color);
      info.intensity       = getFloat(gl, "intensity", 1.0f);
      std::string type     = gl["type"].string("directional");
      info.type            = type == "point" ? MI_LIGHT_POINT : (type == "spot" ? MI_LIGHT_SPOT : MI_LIGHT_DIRECTIONAL);
      info.radius          = getFloat(gl["extras"], "radius", 0.0f);
      if(info.type == MI_LIGHT_DIRECTIONAL)
      {
        const double sunDistance  = 149597870.0;  // km
        info.angularSizeOrInvRange = float(2.0 * std::atan(double(info.radius) / sunDistance));
      }
      else
      {
        double range               = gl["range"].number(0.0);
        info.angularSizeOrInvRange = range > 0.0 ? 1.0f / float(range) : 0.0f;
      }
      m_lights.push_back(info);
      m_lightNode.push_back(nodeID);
      m_lightPath.push_back(m_curPath);
    }
  }

  // mesh -> one RenderNode per primitive (reference: src/gltf_scene.cpp:2338-2429)
  int meshID = getInt(node, "mesh", -1);
  if(meshID >= 0 && size_t(meshID) < m_doc["meshes"].size())
  {
    const Value& prims = m_doc["meshes"][size_t(meshID)]["primitives"];
    // EXT_mesh_gpu_instancing local matrices, shared by all primitives of the node
    const std::vector<mx::mat4>* instances = nullptr;
    const Value&                 inst      = ext(node, "EXT_mesh_gpu_instancing");
    if(inst.isObject())
    {
      auto it = m_gpuInstanceLocalMatrices.find(nodeID);
      if(it == m_gpuInstanceLocalMatrices.end())
      {
        std::vector<float> t, r, s;
        const Value&       a = inst["attributes"];
        if(a.has("TRANSLATION")) readAccessorFloats(a["TRANSLATION"].integer(), 3, t);
        if(a.has("ROTATION")) readAccessorFloats(a["ROTATION"].integer(), 4, r);
        if(a.has("SCALE")) readAccessorFloats(a["SCALE"].integer(), 3, s);
        size_t                n = std::max({t.size() / 3, r.size() / 4, s.size() / 3});
        std::vector<mx::mat4> mats(n);
        for(size_t i = 0; i < n; ++i)
        {
          mx::vec3 tt = i < t.size() / 3 ? mx::vec3{t[3 * i], t[3 * i + 1], t[3 * i + 2]} : mx::vec3{0, 0, 0};
          mx::vec3 ss = i < s.size() / 3 ? mx::vec3{s[3 * i], s[3 * i + 1], s[3 * i + 2]} : mx::vec3{1, 1, 1};
          mx::mat4 rr = i < r.size() / 4 ? mx::fromQuat(r[4 * i], r[4 * i + 1], r[4 * i + 2], r[4 * i + 3]) : mx::identity();
          mats[i]     = mx::mul(mx::mul(mx::translate(tt), rr), mx::scale(ss));
        }
        it = m_gpuInstanceLocalMatrices.emplace(nodeID, std::move(mats)).first;
      }
      instances = &it->second;
    }
    for(size_t p = 0; p < prims.size(); ++p)
    {
      const Value& prim = prims[p];
      auto         it   = primMap.find(primitiveKeyWithTangents(m_doc, prim));
      if(it == primMap.end())
        continue;
      int rprimID = it->second;
      // material: KHR_materials_variants mapping for variant 0, else max(0, material)
      // (reference: src/gltf_scene.cpp:2749-2769)
      int          materialID = std::max(0, getInt(prim, "material", -1));
      const Value& variants   = ext(prim, "KHR_materials_variants");
      if(variants.isObject())
      {
        const Value& mappings = variants["mappings"];
        bool         found    = false;
        for(size_t m = 0; m < mappings.size() && !found; ++m)
          for(size_t v = 0; v < mappings[m]["variants"].size(); ++v)
            if(mappings[m]["variants"][v].integer(-1) == 0)
            {
              materialID = mappings[m]["material"].integer(0);
              found      = true;
              break;
            }
      }
      materialID = std::min(materialID, int(m_materials.size()) - 1);
      auto addNode = [&](const mx::mat4& w, int instance) {
        MiGltfRenderNode rn{};
        memcpy(rn.objectToWorld, w.m, sizeof(rn.objectToWorld));
        mx::mat4 inv = mx::inverse(w);  // reference: src/gltf_scene_vk.cpp:493-501
        memcpy(rn.worldToObject, inv.m, sizeof(rn.worldToObject));
        rn.materialID   = materialID;
        rn.renderPrimID = rprimID;
        m_renderNodes.push_back(rn);
        m_renderNodeVisible.push_back(visible ? 1 : 0);
        m_renderNodeSource.push_back({nodeID, instance, m_curPath});
        m_numTriangles += m_primData[size_t(rprimID)].indices.size() / 3;
      };
      if(instances)
        for(size_t i = 0; i < instances->size(); ++i)
          addNode(mx::mul(world, (*instances)[i]), int(i));
      else
        addNode(world, -1);
    }
  }

  const Value& children = node["children"];
  for(size_t c = 0; c < children.size(); ++c)
    traverse(children[c].integer(-1), world, visible, primMap);
  m_onPath[size_t(nodeID)] = 0;
  m_curPath.pop_back();
}

mx::mat4 GltfScene::localMatrix(int nodeID) const
{
  // a node that carries "matrix" keeps it, animated or not (reference: src/tinygltf_utils.cpp:641-654)
  const Value& m = m_doc["nodes"][size_t(nodeID)]["matrix"];
  if(size_t(nodeID) < m_nodePose.size() && m_nodePose[size_t(nodeID)].animated && !(m.isArray() && m.arr.size() == 16))
  {
    const NodePose& p = m_nodePose[size_t(nodeID)];
    return mx::mul(mx::mul(mx::translate({p.t[0], p.t[1], p.t[2]}), mx::fromQuat(p.q[0], p.q[1], p.q[2], p.q[3])), mx::scale({p.s[0], p.s[1], p.s[2]}));
  }
  return nodeLocalMatrix(m_doc["nodes"][size_t(nodeID)]);
}

// reference: src/gltf_scene.cpp:2269-2300 (position = the node's origin, direction = its -Z axis)
void GltfScene::placeLight(MiGltfLight& info, const mx::mat4& world) const
{
  info.position[0]  = world.at(3, 0);
  info.position[1]  = world.at(3, 1);
  info.position[2]  = world.at(3, 2);
  info.direction[0] = -world.at(2, 0);
  info.direction[1] = -world.at(2, 1);
  info.direction[2] = -world.at(2, 2);
}

void GltfScene::traverseCameras(int nodeID, const mx::mat4& parent)  // reference: src/gltf_scene.cpp:2215-2267
{
  const Value& node = m_doc["nodes"][size_t(nodeID)];
  if(!node.isObject() || size_t(nodeID) >= m_onPath.size() || m_onPath[size_t(nodeID)])
    return;
  m_onPath[size_t(nodeID)] = 1;
  mx::mat4 world = mx::mul(parent, localMatrix(nodeID));
  int      camID = getInt(node, "camera", -1);
  if(camID >= 0 && size_t(camID) < m_doc["cameras"].size())
  {
    const Value& tcam = m_doc["cameras"][size_t(camID)];
    RenderCamera cam;
    if(tcam["type"].string("perspective") == "perspective")
    {
      const Value& p = tcam["perspective"];
      cam.type       = RenderCamera::ePerspective;
      cam.znear      = p["znear"].number(0.1);
      cam.zfar       = p["zfar"].number(0.0);
      cam.yfov       = p["yfov"].number(0.785398);
    }
    else
    {
      const Value& o = tcam["orthographic"];
      cam.type       = RenderCamera::eOrthographic;
      cam.znear      = o["znear"].number(0.1);
      cam.zfar       = o["zfar"].number(0.0);
      cam.xmag       = o["xmag"].number(1.0);
      cam.ymag       = o["ymag"].number(1.0);
    }
    float radius = boundsRadius();
    if(cam.zfar <= cam.znear)
      cam.zfar = std::max(cam.znear * 2.0, 4.0 * double(radius));
    // extractCameraVectors (reference: src/gltf_scene.cpp:2170-2182)
    mx::vec3 eye{world.at(3, 0), world.at(3, 1), world.at(3, 2)};
    mx::vec3 forward{-world.at(2, 0), -world.at(2, 1), -world.at(2, 2)};
    mx::vec3 sceneCenter{(m_bmin[0] + m_bmax[0]) * 0.5f, (m_bmin[1] + m_bmax[1]) * 0.5f, (m_bmin[2] + m_bmax[2]) * 0.5f};
    float    proj   = std::fabs(mx::dot(sceneCenter - eye, forward));
    mx::vec3 center = eye + forward * proj;
    cam.eye[0] = eye.x; cam.eye[1] = eye.y; cam.eye[2] = eye.z;
    cam.center[0] = center.x; cam.center[1] = center.y; cam.center[2] = center.z;
    cam.up[0] = 0; cam.up[1] = 1; cam.up[2] = 0;
    const Value& extras = node["extras"];
    auto getD3 = [&](const char* key, double* out) {
      const Value& v = extras[key];
      if(v.isArray() && v.arr.size() >= 3)
        for(int i = 0; i < 3; ++i)
          out[i] = v.arr[size_t(i)].number();
    };
    if(extras.isObject())
    {
      getD3("camera::eye", cam.eye);
      getD3("camera::center", cam.center);
      getD3("camera::up", cam.up);
    }
    m_cameras.push_back(cam);
  }
  const Value& children = node["children"];
  for(size_t c = 0; c < children.size(); ++c)
    traverseCameras(children[c].integer(-1), world);
  m_onPath[size_t(nodeID)] = 0;
}

void GltfScene::bounds(float bmin[3], float bmax[3]) const
{
  memcpy(bmin, m_bmin, sizeof(m_bmin));
  memcpy(bmax, m_bmax, sizeof(m_bmax));
}

float GltfScene::boundsRadius() const
{
  float d[3] = {m_bmax[0] - m_bmin[0], m_bmax[1] - m_bmin[1], m_bmax[2] - m_bmin[2]};
  return 0.5f * std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
}

//----------------------------------------------------------------------------------------------------------------------
bool GltfScene::parse(const std::string& baseDir)  // reference: src/gltf_scene.cpp:1350-1470
{
  m_renderNodes.clear();
  m_renderNodeVisible.clear();
  m_lights.clear();
  m_cameras.clear();
  m_gpuInstanceLocalMatrices.clear();
  m_renderNodeSource.clear();
  m_lightNode.clear();
  m_lightPath.clear();
  m_curPath.clear();
  m_roots.clear();
  m_animations.clear();
  m_nodePose.assign(m_doc["nodes"].size(), NodePose{});
  m_onPath.assign(m_doc["nodes"].size(), 0);
  m_alphaCutDone  = false;
  m_alphaCutStats = AlphaCutStats{};
  m_numTriangles = 0;

  if(m_doc["nodes"].size() == 0)
  {
    m_error = "No nodes in the glTF file";
    return false;
  }
  buildMaterials();
  buildTextures(baseDir);
  std::map<std::string, int> primMap;
  buildPrimitives(primMap);

  int          sceneID = getInt(m_doc, "scene", 0);
  const Value& scenes  = m_doc["scenes"];
  std::vector<int> roots;
  if(scenes.size() > 0)
  {
    const Value& nodes = scenes[size_t(std::max(0, std::min(sceneID, int(scenes.size()) - 1)))]["nodes"];
    for(size_t i = 0; i < nodes.size(); ++i)
      roots.push_back(nodes[i].integer());
  }
  else  // no scene: every parentless node is a root
  {
    std::vector<bool> hasParent(m_doc["nodes"].size(), false);
    for(size_t i = 0; i < m_doc["nodes"].size(); ++i)
      for(size_t c = 0; c < m_doc["nodes"][i]["children"].size(); ++c)
        hasParent[size_t(m_doc["nodes"][i]["children"][c].integer())] = true;
    for(size_t i = 0; i < hasParent.size(); ++i)
      if(!hasParent[i])
        roots.push_back(int(i));
  }
  for(int r : roots)
    traverse(r, mx::identity(), true, primMap);
  m_roots = roots;
  parseAnimations();

  // scene bounds over visible render nodes (reference: src/gltf_scene.cpp:2303-2336)
  bool  any     = false;
  float bmin[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, bmax[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for(size_t n = 0; n < m_renderNodes.size(); ++n)
  {
    const RenderPrimitiveData& pd = m_primData[size_t(m_renderNodes[n].renderPrimID)];
    if(pd.positions.empty())
      continue;
    float pmin[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, pmax[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for(size_t v = 0; v < pd.positions.size() / 3; ++v)
    {
      const float* p = &pd.positions[3 * v];
      if(!(std::isfinite(p[0]) && std::isfinite(p[1]) && std::isfinite(p[2])))
        continue;  // vertex buffers are untrusted bytes: a damaged vertex must not blow up the scene bounds (camera, z range)
      for(int c = 0; c < 3; ++c)
      {
        pmin[c] = std::min(pmin[c], p[c]);
        pmax[c] = std::max(pmax[c], p[c]);
      }
    }
    if(pmin[0] > pmax[0])
      continue;  // no finite vertex at all
    mx::mat4 w;
    memcpy(w.m, m_renderNodes[n].objectToWorld, sizeof(w.m));
    for(int corner = 0; corner < 8; ++corner)
    {
      mx::vec3 p  = {(corner & 1) ? pmax[0] : pmin[0], (corner & 2) ? pmax[1] : pmin[1], (corner & 4) ? pmax[2] : pmin[2]};
      mx::vec3 wp = mx::transformPoint(w, p);
      float    a[3] = {wp.x, wp.y, wp.z};
      if(!(std::fabs(a[0]) < 1e18f && std::fabs(a[1]) < 1e18f && std::fabs(a[2]) < 1e18f))
        continue;  // an instance blown up by its matrix: the device build drops such triangles too (bvh_build.hip: k_tri_setup)
      for(int c = 0; c < 3; ++c)
      {
        bmin[c] = std::min(bmin[c], a[c]);
        bmax[c] = std::max(bmax[c], a[c]);
      }
      any = true;
    }
  }
  if(!any)
    for(int c = 0; c < 3; ++c)
    {
      bmin[c] = -1;
      bmax[c] = 1;
    }
  memcpy(m_bmin, bmin, sizeof(bmin));
  memcpy(m_bmax, bmax, sizeof(bmax));

  for(int r : roots)
    traverseCameras(r, mx::identity());
  if(m_cameras.empty())  // reference: src/gltf_scene.cpp:1561-1594
  {
    RenderCamera cam;
    float        radius = boundsRadius();
    for(int c = 0; c < 3; ++c)
      cam.center[c] = 0.5 * (double(bmin[c]) + double(bmax[c]));
    cam.eye[0] = cam.center[0];
    cam.eye[1] = cam.center[1];
    cam.eye[2] = cam.center[2] + double(radius) * 2.414;
    cam.yfov   = double(45.0f * 3.14159265358979323846f / 180.0f);
    cam.zfar   = double(radius * 10.0f);
    cam.znear  = double(radius * 0.1f);
    m_cameras.push_back(cam);
  }
  finalizeDesc();
  return true;
}

void GltfScene::finalizeDesc()
{
  m_prims.resize(m_primData.size());
  for(size_t i = 0; i < m_primData.size(); ++i)
  {
    const RenderPrimitiveData& d = m_primData[i];
    MiPtRenderPrimitive&       p = m_prims[i];
    p.indices                    = d.indices.empty() ? nullptr : d.indices.data();
    p.triangleCount              = uint32_t(d.indices.size() / 3);
    p.vertexCount                = d.vertexCount;
    p.positions                  = d.positions.empty() ? nullptr : d.positions.data();
    p.normals                    = d.normals.empty() ? nullptr : d.normals.data();
    p.colors                     = d.colors.empty() ? nullptr : d.colors.data();
    p.tangents                   = d.tangents.empty() ? nullptr : d.tangents.data();
    p.texCoords0                 = d.texCoords0.empty() ? nullptr : d.texCoords0.data();
    p.texCoords1                 = d.texCoords1.empty() ? nullptr : d.texCoords1.data();
    p.opaqueTriangleCount        = d.opaqueTriangles;
    p.reserved                   = 0;
  }
  m_textureDescs.resize(m_textures.size());
  for(size_t i = 0; i < m_textures.size(); ++i)
  {
    TextureData& t = m_textures[i];
    t.levelPtrs.clear();
    for(const auto& l : t.levels)
      t.levelPtrs.push_back(l.data());
    MiPtTexture& d = m_textureDescs[i];
    d.levels       = t.levelPtrs.data();
    d.width        = t.width;
    d.height       = t.height;
    d.numLevels    = int(t.levels.size());
    d.srgb         = t.srgb ? 1 : 0;
    d.magFilter    = t.magFilter;
    d.minFilter    = t.minFilter;
    d.mipmapMode   = t.mipmapMode;
    d.wrapS        = t.wrapS;
    d.wrapT        = t.wrapT;
  }
  m_desc                     = MiPtSceneDesc{};
  m_desc.materials           = m_materials.data();
  m_desc.numMaterials        = int(m_materials.size());
  m_desc.textureInfos        = m_textureInfos.data();
  m_desc.numTextureInfos     = int(m_textureInfos.size());
  m_desc.renderNodes         = m_renderNodes.data();
  m_desc.numRenderNodes      = int(m_renderNodes.size());
  m_desc.renderNodeVisible   = m_renderNodeVisible.data();
  m_desc.renderPrimitives    = m_prims.data();
  m_desc.numRenderPrimitives = int(m_prims.size());
  m_desc.lights              = m_lights.data();
  m_desc.numLights           = int(m_lights.size());
  m_desc.textures            = m_textureDescs.data();
  m_desc.numTextures         = int(m_textureDescs.size());
}

//======================================================================================================================
// HDR environment
//======================================================================================================================
namespace {

bool readRgbe(const std::vector<uint8_t>& file, int& w, int& h, std::vector<float>& rgb, std::string& err)
{
  size_t pos = 0;
  auto   readLine = [&](std::string& line) {
    line.clear();
    while(pos < file.size() && file[pos] != '\n')
      line += char(file[pos++]);
    if(pos < file.size())
      ++pos;
    return true;
  };
  std::string line;
  readLine(line);
  if(line.rfind("#?", 0) != 0)
  {
    err = "not a Radiance HDR file";
    return false;
  }
  bool fmtOk = false;
  while(pos < file.size())
  {
    readLine(line);
    if(line.empty())
      break;
    if(line.find("FORMAT=32-bit_rle_rgbe") != std::string::npos)
      fmtOk = true;
  }
  if(!fmtOk)
  {
    err = "unsupported HDR format";
    return false;
  }
  readLine(line);
  if(sscanf(line.c_str(), "-Y %d +X %d", &h, &w) != 2 || w <= 0 || h <= 0)
  {
    err = "unsupported HDR orientation";
    return false;
  }
  rgb.assign(size_t(w) * size_t(h) * 3, 0.0f);
  std::vector<uint8_t> scan(size_t(w) * 4);
  auto toFloat = [&](const uint8_t* p, float* out) {
    if(p[3] == 0)
    {
      out[0] = out[1] = out[2] = 0.0f;
      return;
    }
    float f = std::ldexp(1.0f, int(p[3]) - (128 + 8));
    out[0]  = float(p[0]) * f;
    out[1]  = float(p[1]) * f;
    out[2]  = float(p[2]) * f;
  };
  for(int y = 0; y < h; ++y)
  {
    if(pos + 4 > file.size())
    {
      err = "truncated HDR";
      return false;
    }
    bool rle = (w >= 8 && w < 32768 && file[pos] == 2 && file[pos + 1] == 2 && (file[pos + 2] & 0x80) == 0);
    if(rle && ((int(file[pos + 2]) << 8) | file[pos + 3]) == w)
    {
      pos += 4;
      for(int c = 0; c < 4; ++c)
      {
        int x = 0;
        while(x < w)
        {
          if(pos >= file.size())
          {
            err = "truncated HDR";
            return false;
          }
          uint8_t count = file[pos++];
          if(count > 128)
          {
            count = uint8_t(count - 128);
            if(pos >= file.size() || x + count > w)
            {
              err = "bad HDR run";
              return false;
            }
            uint8_t v = file[pos++];
            for(int i = 0; i < count; ++i)
              scan[size_t(x++) * 4 + size_t(c)] = v;
          }
          else
          {
            if(count == 0 || pos + count > file.size() || x + count > w)
            {
              err = "bad HDR run";
              return false;
            }
            for(int i = 0; i < count; ++i)
              scan[size_t(x++) * 4 + size_t(c)] = file[pos++];
          }
        }
      }
    }
    else
    {
      if(pos + size_t(w) * 4 > file.size())
      {
        err = "truncated HDR";
        return false;
      }
      memcpy(scan.data(), &file[pos], size_t(w) * 4);
      pos += size_t(w) * 4;
    }
    for(int x = 0; x < w; ++x)
      toFloat(&scan[size_t(x) * 4], &rgb[(size_t(y) * size_t(w) + size_t(x)) * 3]);
  }
  return true;
}

}  // namespace

bool HdrEnvironment::load(const std::string& filename)
{
  std::vector<uint8_t> file;
  if(!readFile(filename, file))
  {
    m_error = "cannot read " + filename;
    return false;
  }
  int                w = 0, h = 0;
  std::vector<float> rgb;
  if(!readRgbe(file, w, h, rgb, m_error))
    return false;
  setPixels(w, h, rgb.data());
  return true;
}

void HdrEnvironment::setPixels(int width, int height, const float* rgb)
{
  m_rgba.assign(size_t(width) * size_t(height) * 4, 0.0f);
  for(size_t i = 0; i < size_t(width) * size_t(height); ++i)
  {
    m_rgba[4 * i + 0] = rgb[3 * i + 0];
    m_rgba[4 * i + 1] = rgb[3 * i + 1];
    m_rgba[4 * i + 2] = rgb[3 * i + 2];
  }
  m_env.width  = width;
  m_env.height = height;
  buildAccel();
  m_env.rgba  = m_rgba.data();
  m_env.accel = m_accel.data();
}

// Importance = max(r,g,b) x texel solid angle; Vose alias table; pdf (per steradian) = max(r,g,b) / integral in alpha.
void HdrEnvironment::buildAccel()
{
  const int    w = m_env.width, h = m_env.height;
  const size_t n = size_t(w) * size_t(h);
  m_accel.assign(n, MiEnvAccel{0, 1.0f});
  std::vector<float> importance(n);
  const double       stepPhi   = 2.0 * M_PI / double(w);
  const double       stepTheta = M_PI / double(h);
  double             total     = 0.0;
  for(int y = 0; y < h; ++y)
  {
    double theta0 = double(y) * stepTheta;
    double area   = (std::cos(theta0) - std::cos(theta0 + stepTheta)) * stepPhi;
    for(int x = 0; x < w; ++x)
    {
      size_t i      = size_t(y) * size_t(w) + size_t(x);
      float  m      = std::max(m_rgba[4 * i], std::max(m_rgba[4 * i + 1], m_rgba[4 * i + 2]));
      importance[i] = float(area * double(m));
      total += double(importance[i]);
    }
  }
  m_env.integral = float(total);
  if(total <= 0.0)
  {
    // black environment: uniform table, pdf of the uniform sphere
    for(size_t i = 0; i < n; ++i)
    {
      m_accel[i]        = MiEnvAccel{uint32_t(i), 1.0f};
      m_rgba[4 * i + 3] = float(1.0 / (4.0 * M_PI));
    }
    return;
  }
  const float invIntegral = float(1.0 / total);
  for(size_t i = 0; i < n; ++i)
    m_rgba[4 * i + 3] = std::max(m_rgba[4 * i], std::max(m_rgba[4 * i + 1], m_rgba[4 * i + 2])) * invIntegral;

  // Vose's alias method on q_i = importance_i * n / total
  // (the running q of a bright texel takes thousands of donations: in float32 its mass drifts by 1e-4 -- measured against numpy
  //  in tests/test_oracle_pins.py -- so the bookkeeping is done in double and only the final entries are rounded)
  std::vector<double>   q(n);
  std::vector<uint32_t> small, large;
  small.reserve(n);
  large.reserve(n);
  const double scale = double(n) / total;
  for(size_t i = 0; i < n; ++i)
  {
    q[i] = double(importance[i]) * scale;
    (q[i] < 1.0 ? small : large).push_back(uint32_t(i));
  }
  while(!small.empty() && !large.empty())
  {
    uint32_t s = small.back();
    small.pop_back();
    uint32_t l       = large.back();
    m_accel[s].q     = float(q[s]);
    m_accel[s].alias = l;
    q[l]             = (q[l] + q[s]) - 1.0;
    if(q[l] < 1.0)
    {
      large.pop_back();
      small.push_back(l);
    }
  }
  for(uint32_t l : large)
    m_accel[l] = MiEnvAccel{l, 1.0f};
  for(uint32_t s : small)
    m_accel[s] = MiEnvAccel{s, 1.0f};
}

}  // namespace mihost
