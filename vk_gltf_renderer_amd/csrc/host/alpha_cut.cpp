// "Alpha cut": a load-time bake for alpha-MASK geometry -- this renderer's counterpart of the reference's opacity micro-map bake
// (src/gltf_scene_omm.cpp: classify micro-triangles of alpha-tested triangles once, so that the traversal does not have to run
// the alpha test on them).  There is no micro-map hardware here and a software look-up in the walk did not pay (LABNOTES.md section 4);
// what does is removing the work instead of classifying it at run time: every alpha-MASK triangle is cut, adaptively, along an
// N x N barycentric grid, and the pieces on which the alpha test CANNOT pass -- no texel that a fetch inside the piece may touch
// reaches the cutoff -- are dropped from the geometry; pieces that survive whole are merged back into their parent.  A ray through the empty part of a leaf card then meets no candidate at all: no
// triangle test, no alpha record, no texel fetch, no continued traversal behind it.
// The other half of the micro-map's job (round 3): pieces on which the test cannot FAIL -- every texel a fetch inside the piece may
// touch reaches the cutoff with a margin -- are marked OPAQUE.  They are moved to the front of the primitive's index buffer and counted
// (RenderPrimitiveData::opaqueTriangles -> MiPtRenderPrimitive::opaqueTriangleCount); the device build flags them like the triangles
// of a FORCE_OPAQUE instance, so a hit in the interior of a leaf card commits without alpha record, texel fetch or deferred alpha
// round.  Only the rim of the card keeps its alpha test.  MASK opacity is 0 or 1, `rand <= 1` always accepts: the same image.
//
// The rendered function is unchanged up to what cannot be observed: the dropped regions are those where
// `rand <= opacity` (raytracer_interface.h.slang:76-111) has opacity 0, which the reference accepts only for a draw of exactly 0
// (2^-23); new vertices carry linearly interpolated attributes, i.e. the same linear functions the hit-attribute interpolation
// evaluates, up to float rounding.  Primitives the classification cannot be sure about are left alone: vertex alpha,
// MIRRORED_REPEAT, non-MASK modes, materials shared with other alpha settings.
// One caveat of the adaptive merge: a coarse cell next to refined ones leaves T-junctions -- the refined cells' edge vertices are
// float interpolations that do not lie exactly on the coarse cell's edge (the same across an original edge whose two triangles are
// refined to different depths) -- so a ray can slip through a crack of a few ulps inside the opaque part of a card, or hit both
// sides of an overlap (the closest-hit tie-break makes the latter harmless).  The measure of such rays is ~1e-7 of a card's
// area: below what the parity tests can see, but not zero; a uniform cut (no merge) does not have it.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>

#include "gltf_scene.hpp"

namespace mihost {

namespace {

// Summed-area table of "this texel may pass the cutoff": count(x0..x1, y0..y1) in O(1)
struct PassTable
{
  int                   w = 0, h = 0;
  std::vector<uint32_t> sat;  // (w + 1) x (h + 1)
  uint32_t count(int x0, int y0, int x1, int y1) const  // inclusive, inside the image
  {
    const size_t W = size_t(w) + 1;
    return sat[size_t(y1 + 1) * W + size_t(x1 + 1)] - sat[size_t(y0) * W + size_t(x1 + 1)] - sat[size_t(y1 + 1) * W + size_t(x0)] + sat[size_t(y0) * W + size_t(x0)];
  }
};

struct Cell  // a triangle of three points (i, j) of the barycentric grid
{
  int i0, j0, i1, j1, i2, j2;
};

// the texel range [a, b] (unwrapped, any integers) as at most two ranges inside [0, n) under the sampler's address mode
int wrapRange(int a, int b, int n, int mode, int out[2][2])
{
  if(mode == MI_WRAP_CLAMP_TO_EDGE)
  {
    out[0][0] = std::min(std::max(a, 0), n - 1);
    out[0][1] = std::min(std::max(b, 0), n - 1);
    return 1;
  }
  if(b - a + 1 >= n)
  {
    out[0][0] = 0;
    out[0][1] = n - 1;
    return 1;
  }
  const int m0 = ((a % n) + n) % n, m1 = ((b % n) + n) % n;
  if(m0 <= m1)
  {
    out[0][0] = m0;
    out[0][1] = m1;
    return 1;
  }
  out[0][0] = m0; out[0][1] = n - 1;
  out[1][0] = 0;  out[1][1] = m1;
  return 2;
}

}  // namespace

uint64_t GltfScene::cutAlphaMasked(int subdivisions)
{
  if(m_alphaCutDone)
    return 0;  // the bake is applied once per loaded scene: cutting the pieces again would only add triangles
  m_alphaCutDone = true;
  int N = 2;
  while(N < std::min(std::max(subdivisions, 2), 16))
    N *= 2;  // the adaptive cut halves cells: 2, 4, 8 or 16 per edge
  // one material per primitive (all render nodes that use the primitive agree), and that material qualifies
  std::vector<int> primMaterial(m_primData.size(), -1);
  for(const MiGltfRenderNode& rn : m_renderNodes)
  {
    if(rn.renderPrimID < 0 || size_t(rn.renderPrimID) >= m_primData.size())
      continue;
    int& pm = primMaterial[size_t(rn.renderPrimID)];
    const int m = std::max(0, rn.materialID);
    pm = (pm == -1 || pm == m) ? m : -2;
  }
  std::map<std::pair<int, int>, PassTable> tables, failTables;  // (texture, threshold byte)
  uint64_t removed = 0;
  for(size_t p = 0; p < m_primData.size(); ++p)
  {
    RenderPrimitiveData& d = m_primData[p];
    const int            mi = primMaterial[p];
    if(mi < 0 || size_t(mi) >= m_materials.size() || d.indices.size() < 3 || d.positions.empty() || !d.colors.empty())
      continue;
    const MiGltfShadeMaterial& mat = m_materials[size_t(mi)];
    if(mat.alphaMode != MI_ALPHA_MASK)
      continue;
    const bool     sg     = mat.pbrModel == MI_PBR_SPECULAR_GLOSSINESS;
    const uint16_t slot   = sg ? mat.pbrDiffuseTexture : mat.pbrBaseColorTexture;
    const float    factor = sg ? mat.pbrDiffuseFactor[3] : mat.pbrBaseColorFactor[3];
    if(slot == 0 || size_t(slot) >= m_textureInfos.size() || !(factor > 0.0f) || !(mat.alphaCutoff > 0.0f))
      continue;
    const MiGltfTextureInfo& info = m_textureInfos[slot];
    if(info.index < 0 || size_t(info.index) >= m_textures.size())
      continue;
    // (KHR_texture_transform plays no part: getOpacity samples with the raw interpolated texture coordinate,
    //  pathtrace_functions.h.slang:189-234, and so do the alpha records of the walks)
    const TextureData&        tex = m_textures[size_t(info.index)];
    const std::vector<float>& tc  = info.texCoord == 0 ? d.texCoords0 : d.texCoords1;
    if(tc.size() < size_t(d.vertexCount) * 2 || tex.levels.empty() || tex.wrapS == MI_WRAP_MIRRORED_REPEAT || tex.wrapT == MI_WRAP_MIRRORED_REPEAT)
      continue;
    const int w = tex.width, h = tex.height;
    if(w < 1 || h < 1 || tex.levels[0].size() < size_t(w) * size_t(h) * 4)
      continue;
    // smallest alpha byte that may reach the cutoff: factor * (a / 255), with a margin for the run-time arithmetic (bilinear weights,
    // the product) -- a texel below it can never contribute to a passing fetch
    int threshold = 256;
    for(int a = 0; a < 256; ++a)
      if(factor * (float(a) / 255.0f) * (1.0f + 1e-5f) + 1e-6f >= mat.alphaCutoff)
      {
        threshold = a;
        break;
      }
    if(threshold == 0)
      continue;  // everything passes: nothing to cut
    PassTable& T = tables[{info.index, threshold}];
    if(T.sat.empty())
    {
      T.w = w; T.h = h;
      T.sat.assign((size_t(w) + 1) * (size_t(h) + 1), 0);
      const uint8_t* px = tex.levels[0].data();
      for(int y = 0; y < h; ++y)
      {
        uint32_t row = 0;
        for(int x = 0; x < w; ++x)
        {
          row += px[(size_t(y) * size_t(w) + size_t(x)) * 4 + 3] >= threshold ? 1u : 0u;
          T.sat[size_t(y + 1) * (size_t(w) + 1) + size_t(x + 1)] = T.sat[size_t(y) * (size_t(w) + 1) + size_t(x + 1)] + row;
        }
      }
    }
    // smallest alpha byte that reaches the cutoff whatever the run-time rounding does: a fetch that only touches such texels passes
    int sure = 256;
    for(int a = 255; a >= 0; --a)
    {
      if(factor * (float(a) / 255.0f) * (1.0f - 1e-5f) - 1e-6f >= mat.alphaCutoff)
        sure = a;
      else
        break;
    }
    PassTable* F = nullptr;  // "this texel may FAIL the cutoff" (none when no byte is sure to pass: nothing can be marked opaque)
    // The OPAQUE class is for plain MASK materials only.  An opaque triangle is flagged like FORCE_OPAQUE geometry on the device, and a
    // shadow ray that meets one is occluded outright; a MASK material that also transmits (transmissionFactor above MIN_TRANSMISSION:
    // translucent foliage) must instead go through the ordered transmissive pass, which tints the shadow with getShadowTransmission
    // (raytracer_interface.h.slang:160-178).  Such materials -- and diffuse transmission, which clears FORCE_OPAQUE in the reference's
    // getInstanceFlag (src/gltf_scene_rtx.cpp:271-295) -- keep the alpha test on every surviving piece; only the pieces that cannot
    // pass are dropped.
    const bool plainMask = !(mat.transmissionFactor > 0.0f) && !(mat.diffuseTransmissionFactor > 0.0f);
    if(sure <= 255 && plainMask)
    {
      F = &failTables[{info.index, sure}];
      if(F->sat.empty())
      {
        F->w = w; F->h = h;
        F->sat.assign((size_t(w) + 1) * (size_t(h) + 1), 0);
        const uint8_t* px = tex.levels[0].data();
        for(int y = 0; y < h; ++y)
        {
          uint32_t row = 0;
          for(int x = 0; x < w; ++x)
          {
            row += px[(size_t(y) * size_t(w) + size_t(x)) * 4 + 3] < sure ? 1u : 0u;
            F->sat[size_t(y + 1) * (size_t(w) + 1) + size_t(x + 1)] = F->sat[size_t(y) * (size_t(w) + 1) + size_t(x + 1)] + row;
          }
        }
      }
    }
    // can a fetch with a uv inside the triangle (uv_a, uv_b, uv_c) touch a texel that may pass (table T) / that may fail (table F)?
    auto touches = [&](const PassTable& T, const float* ua, const float* ub, const float* uc) {
      const float fx0 = std::min({ua[0], ub[0], uc[0]}) * float(w), fx1 = std::max({ua[0], ub[0], uc[0]}) * float(w);
      const float fy0 = std::min({ua[1], ub[1], uc[1]}) * float(h), fy1 = std::max({ua[1], ub[1], uc[1]}) * float(h);
      if(!(std::fabs(fx0) < 1e6f && std::fabs(fx1) < 1e6f && std::fabs(fy0) < 1e6f && std::fabs(fy1) < 1e6f))
        return true;
      // nearest touches floor(f); bilinear floor(f - 0.5) and the next one; one more texel either side for the rounding of the run-time uv
      const int x0 = int(std::floor(fx0 - 0.5f)) - 1, x1 = int(std::floor(fx1 - 0.5f)) + 2;
      const int y0 = int(std::floor(fy0 - 0.5f)) - 1, y1 = int(std::floor(fy1 - 0.5f)) + 2;
      int rx[2][2], ry[2][2];
      const int nx = wrapRange(x0, x1, w, tex.wrapS, rx), ny = wrapRange(y0, y1, h, tex.wrapT, ry);
      for(int a = 0; a < nx; ++a)
        for(int b = 0; b < ny; ++b)
          if(T.count(rx[a][0], ry[b][0], rx[a][1], ry[b][1]) != 0)
            return true;
      return false;
    };
    // The same question for the TRIANGLE instead of its bounding box (a triangular cell fills half of its box, and a leaf's outline
    // is round): texel row by texel row, the x extent of the part of the triangle whose fetches can reach that row -- sample
    // positions fy in [r - 1.5, r + 2.5), the same one-texel slack either side as above -- against that row of the table.
    // Only asked when the box test is inconclusive.
    auto touchesExact = [&](const PassTable& T, const float* ua, const float* ub, const float* uc) {
      const float px[3] = {ua[0] * float(w), ub[0] * float(w), uc[0] * float(w)}, py[3] = {ua[1] * float(h), ub[1] * float(h), uc[1] * float(h)};
      const float fy0 = std::min({py[0], py[1], py[2]}), fy1 = std::max({py[0], py[1], py[2]});
      if(!(std::fabs(fy0) < 1e6f && std::fabs(fy1) < 1e6f && std::fabs(px[0]) < 1e6f && std::fabs(px[1]) < 1e6f && std::fabs(px[2]) < 1e6f))
        return true;
      const int r0 = int(std::floor(fy0 - 0.5f)) - 1, r1 = int(std::floor(fy1 - 0.5f)) + 2;
      if(r1 - r0 > 4096)
        return true;
      for(int r = r0; r <= r1; ++r)
      {
        const float ya = float(r) - 1.5f - 1e-3f, yb = float(r) + 2.5f + 1e-3f;
        float       xmin = 3.0e38f, xmax = -3.0e38f;
        for(int e = 0; e < 3; ++e)
        {
          const float x0 = px[e], y0 = py[e], x1 = px[(e + 1) % 3], y1 = py[(e + 1) % 3];
          if(y0 >= ya && y0 <= yb)
          {
            xmin = std::min(xmin, x0);
            xmax = std::max(xmax, x0);
          }
          for(const float yc : {ya, yb})
            if((y0 - yc) * (y1 - yc) < 0.0f)
            {
              const float x = x0 + (x1 - x0) * ((yc - y0) / (y1 - y0));
              xmin = std::min(xmin, x);
              xmax = std::max(xmax, x);
            }
        }
        if(xmin > xmax)
          continue;  // the band misses the triangle
        const int x0 = int(std::floor(xmin - 0.5f - 1e-3f)) - 1, x1 = int(std::floor(xmax - 0.5f + 1e-3f)) + 2;
        int rx[2][2], ry[2][2];
        const int nx = wrapRange(x0, x1, w, tex.wrapS, rx), ny = wrapRange(r, r, h, tex.wrapT, ry);
        for(int a = 0; a < nx; ++a)
          for(int b = 0; b < ny; ++b)
            if(T.count(rx[a][0], ry[b][0], rx[a][1], ry[b][1]) != 0)
              return true;
      }
      return false;
    };
    static const bool exactCells = getenv("MI_HOST_ALPHA_CUT_BOXES") == nullptr;  // (A/B: bounding boxes only, the round-2 classification)
    auto mayPass = [&](const float* ua, const float* ub, const float* uc) { return touches(T, ua, ub, uc) && (!exactCells || touchesExact(T, ua, ub, uc)); };
    auto mayFail = [&](const float* ua, const float* ub, const float* uc) {
      return F == nullptr || (touches(*F, ua, ub, uc) && (!exactCells || touchesExact(*F, ua, ub, uc)));
    };

    const size_t numTris = d.indices.size() / 3;
    std::vector<uint32_t> outIdx, opaqueIdx;  // triangles that keep their alpha test / that cannot fail it
    outIdx.reserve(d.indices.size());
    const bool hasN = d.normals.size() >= size_t(d.vertexCount) * 3, hasT = d.tangents.size() >= size_t(d.vertexCount) * 4;
    const bool has0 = d.texCoords0.size() >= size_t(d.vertexCount) * 2, has1 = d.texCoords1.size() >= size_t(d.vertexCount) * 2;
    auto lerp3 = [](std::vector<float>& v, int nc, uint32_t a, uint32_t b, uint32_t c, float wa, float wb, float wc) {
      for(int k = 0; k < nc; ++k)
        v.push_back(v[size_t(a) * nc + k] * wa + v[size_t(b) * nc + k] * wb + v[size_t(c) * nc + k] * wc);
    };
    std::vector<uint32_t> grid(size_t(N + 1) * size_t(N + 1));
    std::vector<Cell>     cells, opaqueCells;
    enum { DROPPED = 0, WHOLE_OPAQUE = 1, WHOLE_TESTED = 2, EMITTED = 3 };
    for(size_t t = 0; t < numTris; ++t)
    {
      const uint32_t ia = d.indices[3 * t], ib = d.indices[3 * t + 1], ic = d.indices[3 * t + 2];
      if(ia >= d.vertexCount || ib >= d.vertexCount || ic >= d.vertexCount)
      {
        outIdx.insert(outIdx.end(), {ia, ib, ic});
        continue;
      }
      const float* ua = &tc[size_t(ia) * 2];
      const float* ub = &tc[size_t(ib) * 2];
      const float* uc = &tc[size_t(ic) * 2];
      if(!mayPass(ua, ub, uc))
      {
        ++removed;  // the whole triangle is empty
        continue;
      }
      // Adaptive cut over the N x N barycentric grid (N a power of two): a cell is a triangle of three grid points (i, j) -- u = i / N
      // towards b, v = j / N towards c.  A cell on which the test cannot pass is dropped whole, one on which it cannot fail is opaque
      // whole; at the finest level a cell is kept as it is; otherwise its four children decide, and if all four come back whole and of
      // one kind the cell stays ONE triangle (the interior of a leaf stays coarse, only its rim is refined).
      auto uvAt = [&](int i, int j, float* out) {
        const float u = float(i) / float(N), v = float(j) / float(N), wgt = 1.0f - u - v;
        out[0] = ua[0] * wgt + ub[0] * u + uc[0] * v;
        out[1] = ua[1] * wgt + ub[1] * u + uc[1] * v;
      };
      cells.clear();
      opaqueCells.clear();
      int dropped = 0;
      // returns what became of the cell; for the two WHOLE kinds nothing is emitted yet (the caller emits or merges it)
      std::function<int(const Cell&, int)> visit = [&](const Cell& c, int size) -> int {
        float c0[2], c1[2], c2[2];
        uvAt(c.i0, c.j0, c0); uvAt(c.i1, c.j1, c1); uvAt(c.i2, c.j2, c2);
        if(!mayPass(c0, c1, c2))
        {
          dropped += size * size;  // in units of finest cells
          return DROPPED;
        }
        if(!mayFail(c0, c1, c2))
          return WHOLE_OPAQUE;
        if(size == 1)
          return WHOLE_TESTED;
        // children: the three corner cells and the inverted middle one, from the edge midpoints
        const int m01i = (c.i0 + c.i1) / 2, m01j = (c.j0 + c.j1) / 2, m12i = (c.i1 + c.i2) / 2, m12j = (c.j1 + c.j2) / 2, m20i = (c.i2 + c.i0) / 2, m20j = (c.j2 + c.j0) / 2;
        const Cell child[4] = {{c.i0, c.j0, m01i, m01j, m20i, m20j}, {m01i, m01j, c.i1, c.j1, m12i, m12j}, {m20i, m20j, m12i, m12j, c.i2, c.j2}, {m01i, m01j, m12i, m12j, m20i, m20j}};
        int  kind[4];
        bool allTested = true;
        for(int k = 0; k < 4; ++k)
        {
          kind[k]   = visit(child[k], size / 2);
          allTested = allTested && kind[k] == WHOLE_TESTED;
        }
        if(allTested)
          return WHOLE_TESTED;  // (four opaque children cannot happen: the parent would have been opaque)
        for(int k = 0; k < 4; ++k)
        {
          if(kind[k] == WHOLE_TESTED)
            cells.push_back(child[k]);
          else if(kind[k] == WHOLE_OPAQUE)
            opaqueCells.push_back(child[k]);
        }
        return EMITTED;
      };
      const Cell root{0, 0, N, 0, 0, N};
      const int  rootKind = visit(root, N);
      if(rootKind == WHOLE_TESTED || rootKind == WHOLE_OPAQUE)
      {
        std::vector<uint32_t>& dst = rootKind == WHOLE_OPAQUE ? opaqueIdx : outIdx;
        dst.insert(dst.end(), {ia, ib, ic});  // nothing to cut: the triangle stays as it is
        continue;
      }
      const int total = N * N, kept = total - dropped;
      // emit the kept cells; grid vertices are created on demand (the corners are the original vertices)
      std::fill(grid.begin(), grid.end(), 0xffffffffu);
      grid[0]                         = ia;
      grid[size_t(N)]                 = ib;               // (i = N, j = 0)
      grid[size_t(N) * size_t(N + 1)] = ic;               // (i = 0, j = N)
      auto vertexAt = [&](int i, int j) -> uint32_t {
        uint32_t& g = grid[size_t(j) * size_t(N + 1) + size_t(i)];
        if(g != 0xffffffffu)
          return g;
        const float u = float(i) / float(N), v = float(j) / float(N), wgt = 1.0f - u - v;
        lerp3(d.positions, 3, ia, ib, ic, wgt, u, v);
        if(hasN) lerp3(d.normals, 3, ia, ib, ic, wgt, u, v);
        if(hasT) lerp3(d.tangents, 4, ia, ib, ic, wgt, u, v);
        if(has0) lerp3(d.texCoords0, 2, ia, ib, ic, wgt, u, v);
        if(has1) lerp3(d.texCoords1, 2, ia, ib, ic, wgt, u, v);
        g = d.vertexCount++;
        return g;
      };
      for(const Cell& c : cells)
        outIdx.insert(outIdx.end(), {vertexAt(c.i0, c.j0), vertexAt(c.i1, c.j1), vertexAt(c.i2, c.j2)});
      for(const Cell& c : opaqueCells)
        opaqueIdx.insert(opaqueIdx.end(), {vertexAt(c.i0, c.j0), vertexAt(c.i1, c.j1), vertexAt(c.i2, c.j2)});
      m_alphaCutStats.subTrianglesDropped += uint64_t(total - kept);
      m_alphaCutStats.trianglesSplit += 1;
    }
    // opaque triangles first: the device build flags triangles [0, opaqueTriangles) of the primitive
    d.opaqueTriangles = uint32_t(opaqueIdx.size() / 3);
    m_alphaCutStats.trianglesOpaque += d.opaqueTriangles;
    opaqueIdx.insert(opaqueIdx.end(), outIdx.begin(), outIdx.end());
    outIdx.swap(opaqueIdx);
    d.indices.swap(outIdx);
  }
  m_alphaCutStats.trianglesRemoved += removed;
  // triangle total of the scene (every render node counts its primitive's triangles)
  m_numTriangles = 0;
  for(const MiGltfRenderNode& rn : m_renderNodes)
    if(rn.renderPrimID >= 0 && size_t(rn.renderPrimID) < m_primData.size())
      m_numTriangles += m_primData[size_t(rn.renderPrimID)].indices.size() / 3;
  finalizeDesc();
  return m_alphaCutStats.trianglesRemoved + m_alphaCutStats.subTrianglesDropped;
}

}  // namespace mihost
