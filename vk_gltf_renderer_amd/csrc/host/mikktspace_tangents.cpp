// Per-corner tangent spaces by Mikkelsen's method ("Simulation of Wrinkled Surfaces Revisited", M. Mikkelsen 2008; the tangent basis
// Blender, Unity, Unreal, xNormal and glTF exporters agree on), for triangle lists.
//
// The reference links Mikkelsen's C implementation (third_party/MikkTSpace, called from src/gltf_create_tangent.cpp:540-560 by the
// "Recreate Tangents - MikkTSpace" menu item, src/ui_renderer.cpp:863-868).  This is an independent implementation of the same
// published algorithm for what a glTF primitive can hold (triangles only, so the quad rules and the tangent-space averaging of
// shared quad corners do not arise), written on std::vector / a hash weld instead of the original's grid sort and raw arrays, but
// with its float arithmetic step for step so that the results agree to the last bits -- tests/test_mikktspace.py compares with the
// reference's own mikktspace.c (built into oracle/_ref/ where /root/reference exists, as committed fixtures elsewhere).
//
// Steps: (1) weld corners with identical position / normal / texture coordinate; (2) set triangles with coinciding positions aside;
// (3) per triangle the first-order derivatives of the position over (s, t), their magnitudes, and whether the mapping preserves
// orientation; triangles without uv area group "with anything"; (4) neighbours across shared edges (opposite direction, lowest
// triangle number first when an edge is shared by more than two); (5) around every welded vertex, groups of edge-connected
// triangles of one orientation; (6) per group and corner the angle-weighted average of the members' derivatives projected into the
// vertex's tangent plane; (7) corners of degenerate triangles copy the space of a good triangle at the same welded vertex.
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <algorithm>
#include <unordered_map>
#include <vector>

#include "mikktspace_tangents.hpp"

namespace mihost {

namespace {

struct V3
{
  float x, y, z;
};
V3    sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
V3    add(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
V3    scale(float s, V3 v) { return {s * v.x, s * v.y, s * v.z}; }
float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
float length(V3 v) { return sqrtf(dot(v, v)); }
V3    normalize(V3 v) { return scale(1.0f / length(v), v); }
bool  notZero(float f) { return fabsf(f) > FLT_MIN; }
bool  notZero(V3 v) { return notZero(v.x) || notZero(v.y) || notZero(v.z); }
bool  same(V3 a, V3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
V3    projected(V3 v, V3 n)  // into the plane of n, renormalised unless it vanishes
{
  V3 p = sub(v, scale(dot(n, v), n));
  return notZero(p) ? normalize(p) : p;
}

enum : uint32_t
{
  DEGENERATE   = 1u,
  ANY_GROUP    = 4u,  // no usable uv derivatives: joins the first group that reaches it and takes that group's orientation
  ORIENT_KEEPS = 8u,  // the uv mapping preserves orientation
};

struct Tri
{
  int      corner[3];         // welded vertex per corner
  int      source;            // triangle number in the caller's list
  int      neighbour[3] = {-1, -1, -1};
  int      group[3]     = {-1, -1, -1};
  V3       dPds{0, 0, 0}, dPdt{0, 0, 0};
  float    magS = 0, magT = 0;
  uint32_t flags = 0;
};

struct Group
{
  int              vertex;
  bool             orientKeeps;
  std::vector<int> tris;
};

struct Space
{
  V3   s{1, 0, 0}, t{0, 1, 0};
  bool orientKeeps = false;
};

struct Mesh
{
  const float *   pos, *nrm, *uv;
  const uint32_t* idx;
  V3 P(int welded) const { const float* p = pos + 3 * size_t(idx[welded]); return {p[0], p[1], p[2]}; }
  V3 N(int welded) const { const float* p = nrm + 3 * size_t(idx[welded]); return {p[0], p[1], p[2]}; }
  V3 T(int welded) const { const float* p = uv + 2 * size_t(idx[welded]); return {p[0], p[1], 1.0f}; }
};

// which edge of the triangle joins vertices a and b: the edge number and its ends in the triangle's own winding
void edgeOf(const int c[3], int a, int b, int& e, int& from, int& to)
{
  if(c[0] == a || c[0] == b)
  {
    if(c[1] == a || c[1] == b) { e = 0; from = c[0]; to = c[1]; }
    else                       { e = 2; from = c[2]; to = c[0]; }
  }
  else { e = 1; from = c[1]; to = c[2]; }
}

}  // namespace

void mikkTangentSpaces(const float* positions, const float* normals, const float* texCoords, const uint32_t* indices, size_t numTriangles,
                       std::vector<float>& cornerTangents)
{
  cornerTangents.assign(numTriangles * 12, 0.0f);
  const size_t nCorners = numTriangles * 3;
  const Mesh   mesh{positions, normals, texCoords, indices};
  // ---- (1) weld: a corner is named by the first corner with the same position, normal and texture coordinate ("welded vertex" =
  // that corner's number; attributes are read through it).  Float equality, so +0 and -0 weld and a NaN welds with nothing.
  std::vector<int> weld(nCorners);
  {
    struct Key
    {
      float v[8];
      bool  operator==(const Key& o) const
      {
        for(int i = 0; i < 8; ++i)
          if(!(v[i] == o.v[i]))
            return false;
        return true;
      }
    };
    struct Hash
    {
      size_t operator()(const Key& k) const
      {
        uint64_t h = 1469598103934665603ull;
        for(int i = 0; i < 8; ++i)
        {
          float f = k.v[i] == 0.0f ? 0.0f : k.v[i];  // -0 hashes like +0
          uint32_t b;
          std::memcpy(&b, &f, 4);
          h = (h ^ b) * 1099511628211ull;
        }
        return size_t(h);
      }
    };
    std::unordered_map<Key, int, Hash> first;
    first.reserve(nCorners);
    for(size_t c = 0; c < nCorners; ++c)
    {
      const uint32_t v = indices[c];
      Key            k{{positions[3 * size_t(v)], positions[3 * size_t(v) + 1], positions[3 * size_t(v) + 2], normals[3 * size_t(v)], normals[3 * size_t(v) + 1],
                        normals[3 * size_t(v) + 2], texCoords[2 * size_t(v)], texCoords[2 * size_t(v) + 1]}};
      weld[c] = first.emplace(k, int(c)).first->second;
    }
  }
  // ---- (2) triangles, the degenerate ones (two corners at the same position) moved behind the good ones, both parts in input order
  std::vector<Tri> tris;
  tris.reserve(numTriangles);
  size_t good = 0;
  {
    std::vector<Tri> bad;
    for(size_t f = 0; f < numTriangles; ++f)
    {
      Tri t;
      t.source = int(f);
      for(int i = 0; i < 3; ++i)
        t.corner[i] = weld[f * 3 + i];
      const V3 p0 = mesh.P(t.corner[0]), p1 = mesh.P(t.corner[1]), p2 = mesh.P(t.corner[2]);
      if(same(p0, p1) || same(p0, p2) || same(p1, p2))
      {
        t.flags |= DEGENERATE;
        bad.push_back(t);
      }
      else
        tris.push_back(t);
    }
    good = tris.size();
    tris.insert(tris.end(), bad.begin(), bad.end());
  }
  // ---- (3) first-order derivatives per good triangle
  for(size_t f = 0; f < good; ++f)
  {
    Tri& t = tris[f];
    t.flags |= ANY_GROUP;  // until proven healthy
    const V3    v1 = mesh.P(t.corner[0]), v2 = mesh.P(t.corner[1]), v3 = mesh.P(t.corner[2]);
    const V3    t1 = mesh.T(t.corner[0]), t2 = mesh.T(t.corner[1]), t3 = mesh.T(t.corner[2]);
    const float t21x = t2.x - t1.x, t21y = t2.y - t1.y, t31x = t3.x - t1.x, t31y = t3.y - t1.y;
    const V3    d1 = sub(v2, v1), d2 = sub(v3, v1);
    const float area2 = t21x * t31y - t21y * t31x;  // signed, twice the uv area
    const V3    os = sub(scale(t31y, d1), scale(t21y, d2));
    const V3    ot = add(scale(-t31x, d1), scale(t21x, d2));
    if(area2 > 0)
      t.flags |= ORIENT_KEEPS;
    if(notZero(area2))
    {
      const float absArea = fabsf(area2), lenS = length(os), lenT = length(ot);
      const float sgn = (t.flags & ORIENT_KEEPS) ? 1.0f : -1.0f;
      if(notZero(lenS))
        t.dPds = scale(sgn / lenS, os);
      if(notZero(lenT))
        t.dPdt = scale(sgn / lenT, ot);
      t.magS = lenS / absArea;
      t.magT = lenT / absArea;
      if(notZero(t.magS) && notZero(t.magT))
        t.flags &= ~ANY_GROUP;
    }
  }
  // ---- (4) neighbours: edges keyed by their welded end points, triangles of an edge in ascending order; an edge side pairs with
  // the first later, still unpaired side that runs the other way
  {
    struct Edge
    {
      int lo, hi, tri;
    };
    std::vector<Edge> edges;
    edges.reserve(good * 3);
    for(size_t f = 0; f < good; ++f)
      for(int i = 0; i < 3; ++i)
      {
        const int a = tris[f].corner[i], b = tris[f].corner[(i + 1) % 3];
        edges.push_back({std::min(a, b), std::max(a, b), int(f)});
      }
    std::stable_sort(edges.begin(), edges.end(), [](const Edge& x, const Edge& y) {
      if(x.lo != y.lo) return x.lo < y.lo;
      if(x.hi != y.hi) return x.hi < y.hi;
      return x.tri < y.tri;
    });
    for(size_t i = 0; i < edges.size(); ++i)
    {
      const Edge& A = edges[i];
      int         eA, fromA, toA;
      edgeOf(tris[size_t(A.tri)].corner, A.lo, A.hi, eA, fromA, toA);
      if(tris[size_t(A.tri)].neighbour[eA] != -1)
        continue;
      for(size_t j = i + 1; j < edges.size() && edges[j].lo == A.lo && edges[j].hi == A.hi; ++j)
      {
        const Edge& B = edges[j];
        int         eB, fromB, toB;
        edgeOf(tris[size_t(B.tri)].corner, B.lo, B.hi, eB, fromB, toB);
        if(fromA == toB && toA == fromB && tris[size_t(B.tri)].neighbour[eB] == -1)
        {
          tris[size_t(A.tri)].neighbour[eA] = B.tri;
          tris[size_t(B.tri)].neighbour[eB] = A.tri;
          break;
        }
      }
    }
  }
  // ---- (5) groups: the edge-connected triangles of one orientation around a welded vertex.  An ANY_GROUP triangle does not found a
  // group; the first group to reach it fixes its orientation (the one order dependency of the method: triangles and corners in
  // ascending order, the neighbour across the corner's outgoing edge before the one across its incoming edge).
  std::vector<Group> groups;
  {
    struct Visit
    {
      int tri;
    };
    auto cornerOf = [&](const Tri& t, int vertex) { return t.corner[0] == vertex ? 0 : (t.corner[1] == vertex ? 1 : 2); };
    // the original recurses; an explicit stack that pushes the incoming-edge neighbour first visits in the same order
    auto grow = [&](int g, int start) {
      std::vector<int> stack{start};
      while(!stack.empty())
      {
        const int f = stack.back();
        stack.pop_back();
        Tri&      t = tris[size_t(f)];
        const int i = cornerOf(t, groups[size_t(g)].vertex);
        if(t.group[i] != -1)
          continue;  // already in this group, or in another
        if((t.flags & ANY_GROUP) && t.group[0] == -1 && t.group[1] == -1 && t.group[2] == -1)
          t.flags = (t.flags & ~ORIENT_KEEPS) | (groups[size_t(g)].orientKeeps ? ORIENT_KEEPS : 0u);
        if(((t.flags & ORIENT_KEEPS) != 0) != groups[size_t(g)].orientKeeps)
          continue;
        groups[size_t(g)].tris.push_back(f);
        t.group[i]   = g;
        const int nL = t.neighbour[i], nR = t.neighbour[i > 0 ? i - 1 : 2];
        if(nR >= 0)
          stack.push_back(nR);
        if(nL >= 0)
          stack.push_back(nL);
      }
    };
    for(size_t f = 0; f < good; ++f)
      for(int i = 0; i < 3; ++i)
      {
        Tri& t = tris[f];
        if((t.flags & ANY_GROUP) || t.group[i] != -1)
          continue;
        const int g = int(groups.size());
        groups.push_back({t.corner[i], (t.flags & ORIENT_KEEPS) != 0, {int(f)}});
        t.group[i]   = g;
        const int nL = t.neighbour[i], nR = t.neighbour[i > 0 ? i - 1 : 2];
        if(nL >= 0)
          grow(g, nL);
        if(nR >= 0)
          grow(g, nR);
      }
  }
  // ---- (6) tangent spaces.  With the default angular threshold (180 degrees) a corner's sub-group is every member of its group whose
  // projected derivatives do not point exactly the other way; sub-groups with the same members share one evaluation.
  std::vector<Space> corner(nCorners);  // by the caller's corner number (source triangle * 3 + corner)
  const float        thresholdCos = float(cos((180.0f * float(3.1415926535897932384626433832795)) / 180.0f));
  auto evaluate = [&](const std::vector<int>& members, int vertex) {
    Space res;
    res.s = {0, 0, 0};
    res.t = {0, 0, 0};
    float angleSum = 0;
    for(int f : members)
    {
      const Tri& t = tris[size_t(f)];
      if(t.flags & ANY_GROUP)
        continue;  // only healthy triangles contribute
      const int i = t.corner[0] == vertex ? 0 : (t.corner[1] == vertex ? 1 : 2);
      const V3  n = mesh.N(t.corner[i]);
      const V3  s = projected(t.dPds, n), tt = projected(t.dPdt, n);
      const V3  p0 = mesh.P(t.corner[i > 0 ? i - 1 : 2]), p1 = mesh.P(t.corner[i]), p2 = mesh.P(t.corner[i < 2 ? i + 1 : 0]);
      const V3  e1 = projected(sub(p0, p1), n), e2 = projected(sub(p2, p1), n);
      float     c  = dot(e1, e2);
      c            = c > 1 ? 1 : (c < -1 ? -1 : c);
      const float angle = float(acos(double(c)));  // the corner's angle in the tangent plane weighs the triangle
      res.s = add(res.s, scale(angle, s));
      res.t = add(res.t, scale(angle, tt));
      angleSum += angle;
    }
    if(notZero(res.s))
      res.s = normalize(res.s);
    if(notZero(res.t))
      res.t = normalize(res.t);
    (void)angleSum;
    return res;
  };
  for(const Group& G : groups)
  {
    std::vector<std::vector<int>> subGroups;
    std::vector<Space>            subSpaces;
    for(int f : G.tris)
    {
      const Tri& t = tris[size_t(f)];
      const int  i = t.group[0] == int(&G - groups.data()) ? 0 : (t.group[1] == int(&G - groups.data()) ? 1 : 2);
      const V3   n = mesh.N(t.corner[i]);
      const V3   s = projected(t.dPds, n), tt = projected(t.dPdt, n);
      std::vector<int> members;
      for(int m : G.tris)
      {
        const Tri& u   = tris[size_t(m)];
        const bool any = ((t.flags | u.flags) & ANY_GROUP) != 0;
        const V3   s2 = projected(u.dPds, n), t2 = projected(u.dPdt, n);
        if(any || m == f || (dot(s, s2) > thresholdCos && dot(tt, t2) > thresholdCos))
          members.push_back(m);
      }
      std::sort(members.begin(), members.end());
      size_t l = 0;
      while(l < subGroups.size() && subGroups[l] != members)
        ++l;
      if(l == subGroups.size())
      {
        subSpaces.push_back(evaluate(members, G.vertex));
        subGroups.push_back(std::move(members));
      }
      Space& out      = corner[size_t(t.source) * 3 + size_t(i)];
      out             = subSpaces[l];
      out.orientKeeps = G.orientKeeps;
    }
  }
  // ---- (7) a corner of a degenerate triangle takes the space of the first good corner at the same welded vertex
  for(size_t f = good; f < tris.size(); ++f)
    for(int i = 0; i < 3; ++i)
    {
      const int v = tris[f].corner[i];
      for(size_t g = 0; g < good * 3; ++g)
        if(tris[g / 3].corner[g % 3] == v)
        {
          corner[size_t(tris[f].source) * 3 + size_t(i)] = corner[size_t(tris[g / 3].source) * 3 + g % 3];
          break;
        }
    }
  for(size_t c = 0; c < nCorners; ++c)
  {
    cornerTangents[c * 4 + 0] = corner[c].s.x;
    cornerTangents[c * 4 + 1] = corner[c].s.y;
    cornerTangents[c * 4 + 2] = corner[c].s.z;
    cornerTangents[c * 4 + 3] = corner[c].orientKeeps ? 1.0f : -1.0f;
  }
}

}  // namespace mihost
