// Hit attribute fetch, glTF material evaluation, alpha / shadow-transmission tests and next-event estimation: the device
// restatement of shaders/get_hit.h.slang, shaders/gltf_material_eval.h.slang, shaders/gltf_vertex_access.h.slang and the
// non-tracing parts of shaders/pathtrace_functions.h.slang.  Each function cites the lines it follows.
#pragma once
#include <cstddef>

#include "pt_light.h"

namespace pt {

constexpr float MIN_TRANSMISSION                = 0.01f;       // pathtrace_functions.h.slang:36
constexpr float ANTIALIASING_STANDARD_DEVIATION = 0.4246609f;  // :37
constexpr int   RR_MIN_DEPTH                    = 3;           // :38
constexpr float VOLUME_MIN_SCATTER              = 0.001f;      // :39
constexpr float VOLUME_RAND_FLOOR               = 1.0e-10f;    // :40
constexpr int   VOLUME_FREE_BUDGET              = 64;          // :41
constexpr float RR_PCONT_FLOOR                  = 0.001f;      // :42
constexpr float RR_PCONT_CAP                    = 0.95f;       // :43
constexpr float MICROFACET_MIN_ROUGHNESS        = 0.0014142f;  // gltf_material_eval.h.slang:51

PT_DEV bool hasFlag(int flags, int f) { return (flags & f) != 0; }

// ---- gltf_vertex_access.h.slang:30-150 ---------------------------------------------------------------------------------
struct u3
{
  uint32_t x, y, z;
};
PT_DEV u3 getTriangleIndices(const DevPrim& rp, int prim)
{
  const uint32_t* p = &gat(rp.indices, 3 * size_t(prim));
  return u3{p[0], p[1], p[2]};
}
PT_DEV f3 getVertexPosition(const DevPrim& rp, uint32_t i) { return mk3(&gat(rp.positions, 3 * size_t(i))); }
PT_DEV f4 unpackUnorm4x8(uint32_t p)
{
  return mk4(float((p >> 0) & 0xFF) / 255.0f, float((p >> 8) & 0xFF) / 255.0f, float((p >> 16) & 0xFF) / 255.0f, float((p >> 24) & 0xFF) / 255.0f);
}
PT_DEV f2 getInterpolatedVertexTexCoord(const DevPrim& rp, int channel, u3 idx, f3 b)
{
  const float* tc = channel == 0 ? rp.texCoords0 : rp.texCoords1;
  if(!tc)
    return mk2(0.0f, 0.0f);
  const float2* t2 = reinterpret_cast<const float2*>(tc);
  float2        a = gat(t2, idx.x), bb = gat(t2, idx.y), c = gat(t2, idx.z);
  return mk2(a.x, a.y) * b.x + mk2(bb.x, bb.y) * b.y + mk2(c.x, c.y) * b.z;
}
PT_DEV f4 getInterpolatedVertexColor(const DevPrim& rp, u3 idx, f3 b)
{
  if(!rp.colors)
    return mk4(1.0f);
  return unpackUnorm4x8(gat(rp.colors, idx.x)) * b.x + unpackUnorm4x8(gat(rp.colors, idx.y)) * b.y + unpackUnorm4x8(gat(rp.colors, idx.z)) * b.z;
}

// ---- get_hit.h.slang:26-173 -----------------------------------------------------------------------------------------------
struct HitState
{
  f3    pos, nrm;
  f4    color;
  f3    geonrm, shadowPos;
  f2    uv0, uv1;
  f3    tangent, bitangent;
  float texelDensity;
};
PT_DEV f3 pointOffset(f3 p, f3 p0, f3 p1, f3 p2, f3 n0, f3 n1, f3 n2, f3 bary)  // Hanika 2021 (nvshaders/ray_utils, external)
{
  f3    tmpu = p - p0, tmpv = p - p1, tmpw = p - p2;
  float dotu = fminf(0.0f, dot(tmpu, n0)), dotv = fminf(0.0f, dot(tmpv, n1)), dotw = fminf(0.0f, dot(tmpw, n2));
  tmpu -= n0 * dotu;
  tmpv -= n1 * dotv;
  tmpw -= n2 * dotw;
  return p + tmpu * bary.x + tmpv * bary.y + tmpw * bary.z;
}
// V0..V2: the triangle's interleaved vertices (DevPrim::verts); attrs: SHADE_HAS_*; rp / ti are only read when the primitive has
// uv1 or vertex colours (rp may be null otherwise).
PT_DEV HitState getHitState(const float4* V0, const float4* V1, const float4* V2, uint32_t attrs, const DevPrim* rp, u3 ti, f3 bary, const float* w2o,
                            const float* o2w, f3 worldRayDir)
{
  HitState hit;
  // 9 x 16 B in flight at once
  const float4 a0 = V0[0], a1 = V0[1], a2 = V0[2], b0 = V1[0], b1 = V1[1], b2 = V1[2], c0 = V2[0], c1 = V2[1], c2 = V2[2];
  f3 pos0 = mk3(a0.x, a0.y, a0.z), pos1 = mk3(b0.x, b0.y, b0.z), pos2 = mk3(c0.x, c0.y, c0.z);
  f3 position  = pos0 * bary.x + pos1 * bary.y + pos2 * bary.z;
  hit.pos      = mulPoint(o2w, position);
  f3 geoNormal = normalize(cross(pos1 - pos0, pos2 - pos0));
  hit.geonrm   = normalize(mulTransposed(w2o, geoNormal));
  f3 nrm0 = geoNormal, nrm1 = geoNormal, nrm2 = geoNormal, normal = geoNormal;
  if(attrs & SHADE_HAS_NORMALS)
  {
    nrm0   = mk3(a0.w, a1.x, a1.y);
    nrm1   = mk3(b0.w, b1.x, b1.y);
    nrm2   = mk3(c0.w, c1.x, c1.y);
    normal = nrm0 * bary.x + nrm1 * bary.y + nrm2 * bary.z;
  }
  hit.nrm         = normalize(mulTransposed(w2o, normal));
  bool  frontFace = dot(hit.geonrm, worldRayDir) < 0.0f;
  float sideFlip  = frontFace ? 1.0f : -1.0f;
  f3    shadowPos = pointOffset(position, pos0, pos1, pos2, nrm0 * sideFlip, nrm1 * sideFlip, nrm2 * sideFlip, bary);
  hit.shadowPos   = mulPoint(o2w, shadowPos);
  hit.uv0 = (attrs & SHADE_HAS_UV0) ? mk2(a1.z, a1.w) * bary.x + mk2(b1.z, b1.w) * bary.y + mk2(c1.z, c1.w) * bary.z : mk2(0.0f, 0.0f);
  hit.uv1 = (attrs & SHADE_HAS_UV1) ? getInterpolatedVertexTexCoord(*rp, 1, ti, bary) : mk2(0.0f, 0.0f);
  if(attrs & SHADE_HAS_UV0)
  {
    const float2 a = make_float2(a1.z, a1.w), b = make_float2(b1.z, b1.w), c = make_float2(c1.z, c1.w);
    // computeTexelDensity, get_hit.h.slang:44-56
    f3    we1 = mulVector(o2w, pos1 - pos0), we2 = mulVector(o2w, pos2 - pos0);
    float wArea = length(cross(we1, we2));
    f2    duv1 = mk2(b.x - a.x, b.y - a.y), duv2 = mk2(c.x - a.x, c.y - a.y);
    float uvArea     = fabsf(duv1.x * duv2.y - duv1.y * duv2.x);
    hit.texelDensity = sqrtf(fmaxf(uvArea, 1e-20f) / fmaxf(wArea, 1e-20f));
  }
  else
    hit.texelDensity = 0.0f;
  hit.color = (attrs & SHADE_HAS_COLORS) ? getInterpolatedVertexColor(*rp, ti, bary) : mk4(1.0f);
  f4 tng0, tng1, tng2;
  if(attrs & SHADE_HAS_TANGENTS)
  {
    tng0 = mk4(a2);
    tng1 = mk4(b2);
    tng2 = mk4(c2);
  }
  else
  {
    tng0 = tng1 = tng2 = makeFastTangent(normal);
  }
  hit.tangent   = normalize(xyz(tng0) * bary.x + xyz(tng1) * bary.y + xyz(tng2) * bary.z);
  hit.tangent   = mulVector(o2w, hit.tangent);
  hit.tangent   = normalize(hit.tangent - hit.nrm * dot(hit.nrm, hit.tangent));
  hit.bitangent = cross(hit.nrm, hit.tangent) * tng0.w;
  if(!frontFace)
    hit.geonrm = -hit.geonrm;
  if(dot(hit.geonrm, hit.nrm) < 0.0f)
  {
    hit.nrm       = -hit.nrm;
    hit.tangent   = -hit.tangent;
    hit.bitangent = -hit.bitangent;
  }
  f3 r = reflect(normalize(worldRayDir), hit.nrm);
  if(dot(r, hit.geonrm) < 0.0f)
    hit.nrm = hit.geonrm;
  return hit;
}

// ---- gltf_material_eval.h.slang -------------------------------------------------------------------------------------------
struct MeshState
{
  f3    N, T, B, Ng;
  f2    tc0, tc1;
  bool  isInside;
  float texGrad;
  f4    baseColorVertexMul;
  TexCtx tex;  // texture tables for the fetches of this hit (the shade kernel stages the sRGB table in LDS)
  uint4 core0;  // the hit material's base-colour slot record (DevScene::coreTex[5 * materialID], pt_scene.h: DevCoreTex), loaded by the caller as soon as the material index is known
};
PT_DEV bool isTexturePresent(uint16_t t) { return t > 0; }
// getTexture (gltf_material_eval.h.slang:76-110) on the flattened DevTexRef table: the arithmetic of sampleTexture ->
// sampleLevel -> fetchTexel (pt_light.h) with two dependent loads (record, texels) instead of five.
PT_DEV f4 fetchTexelRef(const TexCtx& tc, const DevTexRef& R, uint32_t levelOffset, int w, int x, int y)
{
  uchar4 p = gat(tc.texels, texelIndex(levelOffset, w, x, y));
  if(R.srgb)
    return mk4(tc.lut[p.x], tc.lut[p.y], tc.lut[p.z], float(p.w) * (1.0f / 255.0f));
  return mk4(float(p.x) * (1.0f / 255.0f), float(p.y) * (1.0f / 255.0f), float(p.z) * (1.0f / 255.0f), float(p.w) * (1.0f / 255.0f));
}
PT_DEV f4 decodeTexel(const TexCtx& tc, bool srgb, uint32_t p)  // fetchTexelRef's arithmetic on an already fetched texel
{
  const uint32_t r = p & 0xffu, g = (p >> 8) & 0xffu, b = (p >> 16) & 0xffu, a = p >> 24;
  if(srgb)
    return mk4(tc.lut[r], tc.lut[g], tc.lut[b], float(a) * (1.0f / 255.0f));
  return mk4(float(r) * (1.0f / 255.0f), float(g) * (1.0f / 255.0f), float(b) * (1.0f / 255.0f), float(a) * (1.0f / 255.0f));
}
PT_DEV f4 decodeTexelRef(const TexCtx& tc, const DevTexRef& R, uint32_t p) { return decodeTexel(tc, R.srgb != 0, p); }
// The two blends of a texture fetch with their operation order pinned (one multiply and one fma per weight pair): the fetch
// exists in several shapes -- footprint records or four texels, levels one after the other or together -- and all of them must
// return the same bits (test_texture_footprint_layout_is_bit_identical), which free contraction of `a * (1 - t) + b * t` inside
// differently inlined bodies does not give.
PT_DEV f4 bilinearBlend(f4 a, f4 b, f4 c, f4 d, float tx, float ty)
{
#pragma clang fp contract(off)
  const float ux = 1.0f - tx, uy = 1.0f - ty;
  auto        ch = [&](float p00, float p10, float p01, float p11) { return __fmaf_rn(__fmaf_rn(p11, tx, p01 * ux), ty, __fmaf_rn(p10, tx, p00 * ux) * uy); };
  return mk4(ch(a.x, b.x, c.x, d.x), ch(a.y, b.y, c.y, d.y), ch(a.z, b.z, c.z, d.z), ch(a.w, b.w, c.w, d.w));
}
PT_DEV f4 levelBlend(f4 a, f4 b, float f)
{
#pragma clang fp contract(off)
  const float u = 1.0f - f;
  return mk4(__fmaf_rn(b.x, f, a.x * u), __fmaf_rn(b.y, f, a.y * u), __fmaf_rn(b.z, f, a.z * u), __fmaf_rn(b.w, f, a.w * u));
}
// One bilinear tap of a level in two halves, so that the two taps of a trilinear fetch can have their footprint records in flight
// together: where the record lies and the weights (quadTap), and the arithmetic on the fetched record (quadFilter).
struct QuadTap
{
  uint32_t index;  // footprint record (DevScene::texQuads)
  float    tx, ty;
  bool     dupX, dupY;  // CLAMP_TO_EDGE left of / above the image: both coordinates of the pair clamp to texel 0
};
PT_DEV bool hasQuadPath(const TexCtx& tc, const DevTexRef& R) { return tc.quads && R.wrapS != MI_WRAP_MIRRORED_REPEAT && R.wrapT != MI_WRAP_MIRRORED_REPEAT; }
PT_DEV uint32_t levelOffsetWH(uint32_t level0, int W, int H, int level)
{
  uint32_t off = level0;
  for(int l = 0; l < level; ++l)
    off += uint32_t(max(1, W >> l)) * uint32_t(max(1, H >> l));
  return off;
}
PT_DEV uint32_t levelOffsetRef(const DevTexRef& R, int level) { return levelOffsetWH(R.level0, int(R.width), int(R.height), level); }
// first texel and weights of a bilinear footprint (shared by every shape of the fetch; products and differences stay what they are)
PT_DEV void bilinearCoords(f2 uv, int w, int h, int& ix, int& iy, float& tx, float& ty)
{
#pragma clang fp contract(off)
  float fx = uv.x * float(w), fy = uv.y * float(h);
  fx -= 0.5f;
  fy -= 0.5f;
  const float flx = floorf(fx), fly = floorf(fy);
  ix = int(flx);
  iy = int(fly);
  tx = fx - flx;
  ty = fy - fly;
}
PT_DEV QuadTap quadTapWH(int W, int H, int wrapS, int wrapT, f2 uv, int level, uint32_t off)  // off = levelOffsetWH(level0, W, H, level)
{
  const int      w = max(1, W >> level), h = max(1, H >> level);
  int            ix, iy;
  QuadTap        t;
  bilinearCoords(uv, w, h, ix, iy, t.tx, t.ty);
  t.index = texelIndex(off, w, wrapCoord(ix, w, wrapS), wrapCoord(iy, h, wrapT));
  t.dupX  = wrapS == MI_WRAP_CLAMP_TO_EDGE && ix < 0;
  t.dupY  = wrapT == MI_WRAP_CLAMP_TO_EDGE && iy < 0;
  return t;
}
PT_DEV QuadTap quadTap(const DevTexRef& R, f2 uv, int level, uint32_t off) { return quadTapWH(int(R.width), int(R.height), R.wrapS, R.wrapT, uv, level, off); }
PT_DEV f4 quadFilterS(const TexCtx& tc, bool srgb, float tx, float ty, bool dupX, bool dupY, const uint4 q)
{
  const f4 a = decodeTexel(tc, srgb, q.x), b = decodeTexel(tc, srgb, dupX ? q.x : q.y);
  const f4 c = decodeTexel(tc, srgb, dupY ? q.x : q.z), d = decodeTexel(tc, srgb, dupY ? (dupX ? q.x : q.y) : (dupX ? q.z : q.w));
  return bilinearBlend(a, b, c, d, tx, ty);
}
PT_DEV f4 quadFilter(const TexCtx& tc, const DevTexRef& R, const QuadTap& t, const uint4 q) { return quadFilterS(tc, R.srgb != 0, t.tx, t.ty, t.dupX, t.dupY, q); }
PT_DEV f4 sampleLevelRef(const TexCtx& tc, const DevTexRef& R, f2 uv, int level, int filter)
{
  if(filter != MI_FILTER_NEAREST && hasQuadPath(tc, R))
  {
    // the whole footprint in one 16-byte gather (DevScene::texQuads): same texels, same arithmetic, same result
    const QuadTap t = quadTap(R, uv, level, levelOffsetRef(R, level));
    return quadFilter(tc, R, t, gat(tc.quads, t.index));
  }
  const uint32_t off = levelOffsetRef(R, level);
  int   w = max(1, int(R.width) >> level), h = max(1, int(R.height) >> level);
  if(filter == MI_FILTER_NEAREST)
  {
    const float fx = uv.x * float(w), fy = uv.y * float(h);
    return fetchTexelRef(tc, R, off, w, wrapCoord(int(floorf(fx)), w, R.wrapS), wrapCoord(int(floorf(fy)), h, R.wrapT));
  }
  int   ix, iy, x0, x1, y0, y1;
  float tx, ty;
  bilinearCoords(uv, w, h, ix, iy, tx, ty);
  wrapCoordPair(ix, w, R.wrapS, x0, x1);
  wrapCoordPair(iy, h, R.wrapT, y0, y1);
  // the four texels in flight together, then decoded (fetchTexelRef's arithmetic)
  const uint32_t* T   = reinterpret_cast<const uint32_t*>(tc.texels);
  const uint32_t  p00 = gat(T, texelIndex(off, w, x0, y0)), p10 = gat(T, texelIndex(off, w, x1, y0));
  const uint32_t  p01 = gat(T, texelIndex(off, w, x0, y1)), p11 = gat(T, texelIndex(off, w, x1, y1));
  f4    a = decodeTexelRef(tc, R, p00), b = decodeTexelRef(tc, R, p10);
  f4    c = decodeTexelRef(tc, R, p01), d = decodeTexelRef(tc, R, p11);
  return bilinearBlend(a, b, c, d, tx, ty);
}
// The texture coordinate and the level of detail of a fetch, with their operation order PINNED (no contraction): the general fetch (getTextureRef) and
// the batched fetch of the core slots (coreTex*) are separate instantiations and must compute the same bits
// (test_texture_footprint_layout_is_bit_identical runs one against the other).
PT_DEV f2 texUv(const float* U, f2 t)
{
#pragma clang fp contract(off)
  return mk2(__fmaf_rn(t.y, U[2], t.x * U[0]) + U[4], __fmaf_rn(t.y, U[3], t.x * U[1]) + U[5]);
}
PT_DEV float texLodOf(float rx2, float ry2)
{
  // max(sqrt(a), sqrt(b)) as sqrt(max(a, b)): the same bits (the square root is monotone), one square root less
  const float rho = sqrtf(fmaxf(rx2, ry2));
  return rho > 0.0f ? log2f(rho) : -126.0f;
}
PT_DEV float texLod(const float* U, float texGrad, float W, float H)
{
#pragma clang fp contract(off)
  const float ax = (U[0] * texGrad) * W, bx = (U[1] * texGrad) * H, ay = (U[2] * texGrad) * W, by = (U[3] * texGrad) * H;
  return texLodOf(__fmaf_rn(bx, bx, ax * ax), __fmaf_rn(by, by, ay * ay));
}
PT_DEV float texLodIdentity(float texGrad, float W, float H)  // texLod under the identity transform: the same bits (the cross terms are exact zeros)
{
#pragma clang fp contract(off)
  const float ax = texGrad * W, by = texGrad * H;
  return texLodOf(ax * ax, by * by);
}
// The texture record of a slot in ONE round trip: the compiler would fetch `width` first (the early-out below tests it) and
// the rest behind the branch -- two dependent loads at the head of every fetch.
PT_DEV DevTexRef loadTexRef(const DevTexRef* refs, uint32_t slot)
{
#if !defined(__HIPCC__)
  return refs[slot];
#else
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
  const DevTexRef* p = refs + slot;
  u32x4            a, b;
  u32x3            c;
  asm volatile("global_load_dwordx4 %0, %3, off\n\tglobal_load_dwordx4 %1, %3, off offset:16\n\tglobal_load_dwordx3 %2, %3, off offset:32\n\ts_waitcnt vmcnt(0)"
               : "=&v"(a), "=&v"(b), "=&v"(c)
               : "v"(p)
               : "memory");
  DevTexRef R;
  uint32_t* W = reinterpret_cast<uint32_t*>(&R);
  W[0] = a[0]; W[1] = a[1]; W[2] = a[2]; W[3] = a[3]; W[4] = b[0]; W[5] = b[1]; W[6] = b[2]; W[7] = b[3]; W[8] = c[0]; W[9] = c[1]; W[10] = c[2];
  return R;
#endif
}
// (a slot whose record is fetched NEXT to the material record instead of here, behind it: coreTexPlan below -- in use for the base colour of the later bounces)
__device__ __noinline__ f4 getTextureRef(TexCtx tc, uint32_t slot, f2 tc0, f2 tc1, float texGrad)
{
#ifdef MI_PT_DIAG_NO_TEX  // cost-attribution build (tools/attribution.sh): wrong image, no texture filtering
  return mk4(0.5f + 1e-3f * (tc0.x + tc1.y + texGrad + float(slot)));
#endif
  const DevTexRef R  = loadTexRef(tc.refs, slot);
  f2              t  = R.texCoord == 0 ? tc0 : tc1;
  const float*    U  = R.uv;
  f2              uv = texUv(U, t);
  if(R.width == 0)
    return mk4(1.0f);
  float lod = 0.0f;
  if(texGrad > 0.0f)
    lod = texLod(U, texGrad, float(R.width), float(R.height));
  if(lod <= 0.0f)
    return sampleLevelRef(tc, R, uv, 0, R.magFilter);
  float maxLevel = float(int(R.numLevels) - 1);
  lod            = fminf(lod, maxLevel);
  if(R.mipmapMode == MI_FILTER_NEAREST)
    return sampleLevelRef(tc, R, uv, min(int(floorf(lod + 0.5f)), int(R.numLevels) - 1), R.minFilter);
  int   l0 = int(floorf(lod)), l1 = min(l0 + 1, int(R.numLevels) - 1);
  float f  = lod - float(l0);
  if(R.minFilter != MI_FILTER_NEAREST && hasQuadPath(tc, R) && !(f == 0.0f || l1 == l0))
  {
    // trilinear: both levels' footprint records in flight together
    // (l1 == l0 + 1 here: the coarser level starts where the finer one ends)
    const uint32_t off0 = levelOffsetRef(R, l0), off1 = off0 + uint32_t(max(1, int(R.width) >> l0)) * uint32_t(max(1, int(R.height) >> l0));
    const QuadTap  t0 = quadTap(R, uv, l0, off0), t1 = quadTap(R, uv, l1, off1);
    const uint4   q0 = gat(tc.quads, t0.index), q1 = gat(tc.quads, t1.index);
    const f4      a = quadFilter(tc, R, t0, q0), b = quadFilter(tc, R, t1, q1);
    return levelBlend(a, b, f);
  }
  f4    a  = sampleLevelRef(tc, R, uv, l0, R.minFilter);
  if(f == 0.0f || l1 == l0)
    return a;
  f4 b = sampleLevelRef(tc, R, uv, l1, R.minFilter);
  return levelBlend(a, b, f);
}
// ---- a core slot of a material fetched through its per-material record (pt_scene.h: DevCoreTex), in three steps: PLAN (which footprint records, which weights --
// the arithmetic of getTextureRef for a sampler with LINEAR filters), ISSUE (the records' gathers), FINISH (filter, blend) where evaluateMaterial uses the value.
// The same functions as the general fetch throughout, so the values are its values bit for bit.  Used for the BASE COLOUR in the later-bounce shade kernels
// (evaluateMaterial<SIMPLE, CORE>): the slot's record arrives with the material instead of one round trip behind it.
// Measured (profiles/r06_shade_walk_ab.txt): atrium +1.3 %, street +0.8 %; in the bounce-0 kernel as well: helmet -0.6 % (its code, not its path: the same with every
// slot sent the general way).  All five slots as ONE batch (ten dependent round trips of a five-map hit down to two) spills the 168-register kernel and is slower
// everywhere, the helmet included (5247 -> 4932; in two batches 5183); the record handed to a non-inlined fetch gains nothing.
#ifndef MI_PT_CORE_TEX_BATCH
#define MI_PT_CORE_TEX_BATCH 1  // (A/B: 0 = every slot through getTextureRef)
#endif
struct CoreTap
{
  uint32_t i0, i1;  // footprint records (i1: the coarser level of a trilinear fetch)
  float    tx0, ty0, tx1, ty1, f;
  uint32_t bits;    // CP_*
};
enum : uint32_t { CP_FAST = 1u, CP_TWO = 2u, CP_SRGB = 4u, CP_DUPX0 = 8u, CP_DUPY0 = 16u, CP_DUPX1 = 32u, CP_DUPY1 = 64u, CP_SLOW = 128u };
PT_DEV CoreTap coreTexPlan(const TexCtx& tc, const uint4 c, f2 tc0, f2 tc1, float texGrad)
{
  CoreTap p;
  p.i0 = p.i1 = 0u;
  p.tx0 = p.ty0 = p.tx1 = p.ty1 = p.f = 0.0f;
  p.bits = 0u;
  const int W = int(c.y & 0xffffu), H = int(c.y >> 16);
  if(W == 0)
    return p;  // no valid texture behind the slot: the value is 1
#ifdef MI_PT_DIAG_NO_TEX
  p.bits = CP_SLOW;
  return p;
#endif
  if(!(c.z & CT_FAST) || !tc.quads)
  {
    p.bits = CP_SLOW;
    return p;
  }
  const f2 t   = (c.z & CT_TEXCOORD1) ? tc1 : tc0;
  f2       uv  = t;
  float    lod = 0.0f;
  if(c.z & CT_TRANSFORM)
  {
    const float* U6 = gat(tc.refs, c.w).uv;
    const float  U[6] = {U6[0], U6[1], U6[2], U6[3], U6[4], U6[5]};
    uv = texUv(U, t);
    if(texGrad > 0.0f)
      lod = texLod(U, texGrad, float(W), float(H));
  }
  else if(texGrad > 0.0f)
    lod = texLodIdentity(texGrad, float(W), float(H));
  const int wrapS = int((c.z >> CT_WRAPS_SHIFT) & 3u), wrapT = int((c.z >> CT_WRAPT_SHIFT) & 3u), levels = int((c.z >> CT_LEVELS_SHIFT) & 0xffu);
  int   l0  = 0;
  bool  two = false;
  float f   = 0.0f;
  if(!(lod <= 0.0f))
  {
    lod = fminf(lod, float(levels - 1));
    if(!(c.z & CT_MIP_LINEAR))
      l0 = min(int(floorf(lod + 0.5f)), levels - 1);
    else
    {
      l0           = int(floorf(lod));
      const int l1 = min(l0 + 1, levels - 1);
      f            = lod - float(l0);
      two          = !(f == 0.0f || l1 == l0);
    }
  }
  const uint32_t off0 = levelOffsetWH(c.x, W, H, l0);
  const QuadTap  t0   = quadTapWH(W, H, wrapS, wrapT, uv, l0, off0);
  p.i0 = t0.index; p.tx0 = t0.tx; p.ty0 = t0.ty;
  p.bits = CP_FAST | ((c.z & CT_SRGB) ? CP_SRGB : 0u) | (t0.dupX ? CP_DUPX0 : 0u) | (t0.dupY ? CP_DUPY0 : 0u);
  if(two)
  {
    // (l1 == l0 + 1 here: the coarser level starts where the finer one ends)
    const uint32_t off1 = off0 + uint32_t(max(1, W >> l0)) * uint32_t(max(1, H >> l0));
    const QuadTap  t1   = quadTapWH(W, H, wrapS, wrapT, uv, l0 + 1, off1);
    p.i1 = t1.index; p.tx1 = t1.tx; p.ty1 = t1.ty; p.f = f;
    p.bits |= CP_TWO | (t1.dupX ? CP_DUPX1 : 0u) | (t1.dupY ? CP_DUPY1 : 0u);
  }
  return p;
}
PT_DEV void coreTexIssue(const TexCtx& tc, const CoreTap& p, uint4& q0, uint4& q1)
{
  q0 = q1 = make_uint4(0u, 0u, 0u, 0u);
  if(p.bits & CP_FAST)
  {
    q0 = gat(tc.quads, p.i0);
    if(p.bits & CP_TWO)
      q1 = gat(tc.quads, p.i1);
  }
}
PT_DEV f4 coreTexFinish(const TexCtx& tc, const CoreTap& p, const uint4& q0, const uint4& q1, uint32_t slot, f2 tc0, f2 tc1, float texGrad)
{
  if(p.bits & CP_SLOW)
    return getTextureRef(tc, slot, tc0, tc1, texGrad);
  if(!(p.bits & CP_FAST))
    return mk4(1.0f);
  const bool srgb = (p.bits & CP_SRGB) != 0u;
  const f4   a    = quadFilterS(tc, srgb, p.tx0, p.ty0, (p.bits & CP_DUPX0) != 0u, (p.bits & CP_DUPY0) != 0u, q0);
  if(!(p.bits & CP_TWO))
    return a;
  const f4 b = quadFilterS(tc, srgb, p.tx1, p.ty1, (p.bits & CP_DUPX1) != 0u, (p.bits & CP_DUPY1) != 0u, q1);
  return levelBlend(a, b, p.f);
}
PT_DEV f3 multiToSingleScatterAlbedo(f3 rho)  // :125-129
{
  f3 t = mk3(4.09712f) + rho * 4.20863f - sqrt3(mk3(9.59217f) + rho * 41.6808f + rho * rho * 17.7126f);
  return mk3(1.0f) - t * t;
}
PT_DEV f3 convertSGToMR(f3 diffuseColor, f3 specularColor, float glossiness, float& metallic, f2& roughness)  // :136-161
{
  const float dielectricSpecular = 0.04f;
  float       specularIntensity  = fmaxf(specularColor.x, fmaxf(specularColor.y, specularColor.z));
  metallic                       = smoothstepf(dielectricSpecular + 0.01f, dielectricSpecular + 0.05f, specularIntensity);
  f3 baseColor;
  if(metallic > 0.0f)
    baseColor = specularColor;
  else
  {
    baseColor = diffuseColor / (1.0f - dielectricSpecular * (1.0f - metallic));
    baseColor = clamp3(baseColor, 0.0f, 1.0f);
  }
  float r   = 1.0f - glossiness;
  roughness = mk2(r * r, r * r);
  return baseColor;
}
// SIMPLE: the scene has no material with transmission, diffuse transmission, clearcoat, sheen, iridescence, anisotropy or
// retroreflection (isSimpleMaterial below, decided once per scene like the reference's scene-aware shader variants,
// src/renderer_pathtracer.cpp feature macros): those inputs keep their neutral defaults as compile-time constants and
// the lobes, volume tracking and texture fetches that depend on them fold away.
template <bool SIMPLE, bool CORE = false>
PT_DEV PbrMaterial evaluateMaterial(const DevScene& sc, const MiGltfShadeMaterial& m, const MeshState& st, unsigned& taps)  // :168-457
{
#define TEX(slot) (++taps, getTextureRef(st.tex, slot, st.tc0, st.tc1, st.texGrad))
  PbrMaterial p = defaultPbrMaterial();
  // the five core map slots in one load (they are adjacent, 8-byte aligned): read where they are used, each was a 2-byte load and a
  // wait of its own in front of its texture fetch
  static_assert(offsetof(MiGltfShadeMaterial, pbrBaseColorTexture) % 8 == 0 && offsetof(MiGltfShadeMaterial, occlusionTexture) == offsetof(MiGltfShadeMaterial, pbrBaseColorTexture) + 8,
                "core texture slots: base colour, normal, metallic-roughness, emissive, occlusion");
  const uint32_t* slotWords = reinterpret_cast<const uint32_t*>(&m.pbrBaseColorTexture);
  const uint32_t  sw0 = slotWords[0], sw1 = slotWords[1], sw2 = slotWords[2];
  const uint16_t  texBaseColor = uint16_t(sw0 & 0xffffu), texNormal = uint16_t(sw0 >> 16), texMetallicRoughness = uint16_t(sw1 & 0xffffu),
                 texEmissive = uint16_t(sw1 >> 16), texOcclusion = uint16_t(sw2 & 0xffffu);
  // the base colour through its core record (slot 0 of the material's five, pt_scene.h: DevCoreTex): planned and in flight before anything else of the material is read
  CoreTap cp0{};
  uint4   cq0 = make_uint4(0u, 0u, 0u, 0u), cq1 = cq0;
  if(CORE && MI_PT_CORE_TEX_BATCH)
  {
    const bool  present = m.pbrModel != MI_PBR_SPECULAR_GLOSSINESS && isTexturePresent(texBaseColor);  // (the specular-glossiness model does not read it)
    const uint4 c       = st.core0;
    cp0                 = coreTexPlan(st.tex, present ? c : make_uint4(0u, 0u, 0u, 0u), st.tc0, st.tc1, st.texGrad);
    coreTexIssue(st.tex, cp0, cq0, cq1);
  }
  if(m.pbrModel == MI_PBR_SPECULAR_GLOSSINESS)
  {
    f4    diffuse    = mk4(m.pbrDiffuseFactor[0], m.pbrDiffuseFactor[1], m.pbrDiffuseFactor[2], m.pbrDiffuseFactor[3]) * st.baseColorVertexMul;
    float glossiness = m.pbrGlossinessFactor;
    f3    specular   = mk3(m.pbrSpecularFactor);
    if(isTexturePresent(m.pbrDiffuseTexture))
      diffuse *= TEX(m.pbrDiffuseTexture);
    if(isTexturePresent(m.pbrSpecularGlossinessTexture))
    {
      f4 s = TEX(m.pbrSpecularGlossinessTexture);
      specular *= xyz(s);
      glossiness *= s.w;
    }
    p.baseColor = convertSGToMR(xyz(diffuse), specular, glossiness, p.metallic, p.roughness);
    p.opacity   = diffuse.w;
  }
  else
  {
    f4 baseColor = mk4(m.pbrBaseColorFactor[0], m.pbrBaseColorFactor[1], m.pbrBaseColorFactor[2], m.pbrBaseColorFactor[3]) * st.baseColorVertexMul;
    if(isTexturePresent(texBaseColor))
      baseColor *= (CORE && MI_PT_CORE_TEX_BATCH) ? (++taps, coreTexFinish(st.tex, cp0, cq0, cq1, texBaseColor, st.tc0, st.tc1, st.texGrad)) : TEX(texBaseColor);
    p.baseColor     = xyz(baseColor);
    p.opacity       = baseColor.w;
    float roughness = m.pbrRoughnessFactor, metallic = m.pbrMetallicFactor;
    if(isTexturePresent(texMetallicRoughness))
    {
      f4 s = TEX(texMetallicRoughness);
      roughness *= s.y;
      metallic *= s.z;
    }
    roughness   = fmaxf(roughness, MICROFACET_MIN_ROUGHNESS);
    p.roughness = mk2(roughness * roughness, roughness * roughness);
    p.metallic  = clampf(metallic, 0.0f, 1.0f);
  }
  p.occlusion = m.occlusionStrength;
  if(isTexturePresent(texOcclusion))
  {
    float occ   = TEX(texOcclusion).x;
    p.occlusion = 1.0f + p.occlusion * (occ - 1.0f);
  }
  p.N  = st.N;
  p.T  = st.T;
  p.B  = st.B;
  p.Ng = st.Ng;
  bool needsTangentUpdate = false;
  if(isTexturePresent(texNormal))
  {
    f3 nv = xyz(TEX(texNormal));
    nv    = nv * 2.0f - mk3(1.0f);
    nv *= mk3(m.normalTextureScale, m.normalTextureScale, 1.0f);
    p.N                = normalize(st.T * nv.x + st.B * nv.y + st.N * nv.z);
    needsTangentUpdate = true;
  }
  p.emissive = mk3(m.emissiveFactor);
  if(isTexturePresent(texEmissive))
    p.emissive *= xyz(TEX(texEmissive));
  p.emissive            = max3(mk3(0.0f), p.emissive);
  if(!SIMPLE)
  {
    p.attenuationColor    = mk3(m.attenuationColor);
    p.attenuationDistance = m.attenuationDistance;
    p.thickness           = m.thicknessFactor;
    if(isTexturePresent(m.thicknessTexture))
      p.thickness *= TEX(m.thicknessTexture).y;
  }
  p.specularColor = mk3(m.specularColorFactor);
  if(isTexturePresent(m.specularColorTexture))
    p.specularColor *= xyz(TEX(m.specularColorTexture));
  p.specular = m.specularFactor;
  if(isTexturePresent(m.specularTexture))
    p.specular *= TEX(m.specularTexture).w;
  float ior1 = 1.0f, ior2 = m.ior;
  if(!SIMPLE && st.isInside && (p.thickness > 0.0f))
  {
    ior1 = ior2;
    ior2 = 1.0f;
  }
  p.ior1         = ior1;
  p.ior2         = ior2;
  bool needsTangentUpdateAniso = false;
  if(!SIMPLE)
  {
  p.transmission = m.transmissionFactor;
  if(isTexturePresent(m.transmissionTexture))
    p.transmission *= TEX(m.transmissionTexture).x;
  f3 ms = mk3(m.multiscatterColorFactor);
  if(ms.x > 0.0f || ms.y > 0.0f || ms.z > 0.0f)
  {
    f3 ssa               = multiToSingleScatterAlbedo(ms);
    f3 attC              = -log3(max3(p.attenuationColor, mk3(0.001f))) / fmaxf(p.attenuationDistance, 0.001f);
    p.scatterCoefficient = attC * ssa;
  }
  p.scatterAnisotropy  = m.scatterAnisotropy;
  p.clearcoat          = m.clearcoatFactor;
  p.clearcoatRoughness = m.clearcoatRoughness;
  p.Nc                 = p.N;
  if(isTexturePresent(m.clearcoatTexture))
    p.clearcoat *= TEX(m.clearcoatTexture).x;
  if(isTexturePresent(m.clearcoatRoughnessTexture))
    p.clearcoatRoughness *= TEX(m.clearcoatRoughnessTexture).y;
  if(isTexturePresent(m.clearcoatNormalTexture))
  {
    f3 nv = xyz(TEX(m.clearcoatNormalTexture));
    nv    = nv * 2.0f - mk3(1.0f);
    p.Nc  = normalize(p.T * nv.x + p.B * nv.y + p.Nc * nv.z);
  }
  p.clearcoatRoughness = fmaxf(p.clearcoatRoughness, 0.001f);
  float iridescence = m.iridescenceFactor, iridescenceThickness = m.iridescenceThicknessMaximum;
  p.iridescenceIor = m.iridescenceIor;
  if(isTexturePresent(m.iridescenceTexture))
    iridescence *= TEX(m.iridescenceTexture).x;
  if(isTexturePresent(m.iridescenceThicknessTexture))
  {
    float t              = TEX(m.iridescenceThicknessTexture).y;
    iridescenceThickness = lerpf(m.iridescenceThicknessMinimum, m.iridescenceThicknessMaximum, t);
  }
  p.iridescence          = (iridescenceThickness > 0.0f) ? iridescence : 0.0f;
  p.iridescenceThickness = iridescenceThickness;
  float anisotropyStrength = m.anisotropyStrength;
  if(anisotropyStrength > 0.0f)
  {
    f2 dir = mk2(1.0f, 0.0f);
    if(isTexturePresent(m.anisotropyTexture))
    {
      f4 a = TEX(m.anisotropyTexture);
      dir  = normalize(mk2(a.x, a.y) * 2.0f - mk2(1.0f, 1.0f));
      anisotropyStrength *= a.z;
    }
    p.roughness.x = lerpf(p.roughness.y, 1.0f, anisotropyStrength * anisotropyStrength);
    float s = m.anisotropyRotation[0], c = m.anisotropyRotation[1];
    dir                = mk2(c * dir.x + s * dir.y, c * dir.y - s * dir.x);
    p.T                = p.T * dir.x + p.B * dir.y;
    needsTangentUpdateAniso = true;
  }
  }
  needsTangentUpdate = needsTangentUpdate || needsTangentUpdateAniso;
  if(needsTangentUpdate)
  {
    p.B         = normalize(cross(p.N, p.T));
    float bsign = signfz(dot(st.B, p.B));
    p.B         = p.B * bsign;
    p.T         = normalize(cross(p.B, p.N) * bsign);
  }
  if(!SIMPLE)
  {
  p.sheenColor = mk3(m.sheenColorFactor);
  if(isTexturePresent(m.sheenColorTexture))
    p.sheenColor *= xyz(TEX(m.sheenColorTexture));
  p.sheenRoughness = m.sheenRoughnessFactor;
  if(isTexturePresent(m.sheenRoughnessTexture))
    p.sheenRoughness *= TEX(m.sheenRoughnessTexture).w;
  p.sheenRoughness            = fmaxf(MICROFACET_MIN_ROUGHNESS, p.sheenRoughness);
  p.dispersion                = m.dispersion;
  p.diffuseTransmissionFactor = m.diffuseTransmissionFactor;
  if(isTexturePresent(m.diffuseTransmissionTexture))
    p.diffuseTransmissionFactor *= TEX(m.diffuseTransmissionTexture).w;
  p.diffuseTransmissionColor = mk3(m.diffuseTransmissionColor);
  if(isTexturePresent(m.diffuseTransmissionColorTexture))
    p.diffuseTransmissionColor *= xyz(TEX(m.diffuseTransmissionColorTexture));
  p.retroreflection = m.retroreflectionFactor;
  if(isTexturePresent(m.retroreflectionTexture))
    p.retroreflection *= TEX(m.retroreflectionTexture).x;
  }
#undef TEX
  return p;
}

// ---- pathtrace_functions.h.slang helpers ------------------------------------------------------------------------------------
PT_DEV f3 safeOffsetRay(f3 p, f3 dir)  // :151-167
{
  const float scaleValue = 256.0f;
  int         ix = int(scaleValue * dir.x), iy = int(scaleValue * dir.y), iz = int(scaleValue * dir.z);
  f3 op = mk3(__int_as_float(__float_as_int(p.x) + ((p.x < 0) ? -ix : ix)), __int_as_float(__float_as_int(p.y) + ((p.y < 0) ? -iy : iy)),
              __int_as_float(__float_as_int(p.z) + ((p.z < 0) ? -iz : iz)));
  const float origin = 1.0f / 32.0f, floatScale = 1.0f / 65536.0f;
  return mk3(fabsf(p.x) < origin ? p.x + floatScale * dir.x : op.x, fabsf(p.y) < origin ? p.y + floatScale * dir.y : op.y,
             fabsf(p.z) < origin ? p.z + floatScale * dir.z : op.z);
}

// getOpacity (pathtrace_functions.h.slang:189-234; 1 for opaque materials) through the per-triangle alpha record: record ->
// texels instead of instance -> material -> primitive -> indices -> uvs -> texture info -> texture -> texels
PT_DEV float getOpacityFast(const DevScene& sc, int triIndex, f3 bary)
{
  const DevAlphaTri rec  = sc.alphaTris[triIndex];
  const uint32_t    flg  = rec.c.z;
  const uint32_t    mode = flg & AT_MODE_MASK;
  if(mode == MI_ALPHA_OPAQUE)
    return 1.0f;
  float alpha = rec.b.z;
  if(flg & AT_HAS_TEXTURE)
  {
    f2          uv = mk2(rec.a.x, rec.a.y) * bary.x + mk2(rec.a.z, rec.a.w) * bary.y + mk2(rec.b.x, rec.b.y) * bary.z;
    const int   w = int(rec.c.y & 0xffffu), h = int(rec.c.y >> 16);
    const int   wrapS = int((flg >> AT_WRAPS_SHIFT) & 3u), wrapT = int((flg >> AT_WRAPT_SHIFT) & 3u);
    const uchar4* texels = sc.texels + rec.c.x;
    float       fx = uv.x * float(w), fy = uv.y * float(h);
    float       ta;
    if(!(flg & AT_LINEAR))
      ta = float(texels[texelIndex(0u, w, wrapCoord(int(floorf(fx)), w, wrapS), wrapCoord(int(floorf(fy)), h, wrapT))].w) * (1.0f / 255.0f);
    else
    {
      fx -= 0.5f;
      fy -= 0.5f;
      float flx = floorf(fx), fly = floorf(fy);
      float tx = fx - flx, ty = fy - fly;
      float a, b, c, d;
      if(sc.texQuads && wrapS != MI_WRAP_MIRRORED_REPEAT && wrapT != MI_WRAP_MIRRORED_REPEAT)
      {
        // the footprint in one gather (DevScene::texQuads; see sampleLevelRef)
        const int   ix = int(flx), iy = int(fly);
        const uint4 q  = sc.texQuads[rec.c.x + texelIndex(0u, w, wrapCoord(ix, w, wrapS), wrapCoord(iy, h, wrapT))];
        const bool  dupX = wrapS == MI_WRAP_CLAMP_TO_EDGE && ix < 0, dupY = wrapT == MI_WRAP_CLAMP_TO_EDGE && iy < 0;
        a = float(q.x >> 24) * (1.0f / 255.0f);
        b = float((dupX ? q.x : q.y) >> 24) * (1.0f / 255.0f);
        c = float((dupY ? q.x : q.z) >> 24) * (1.0f / 255.0f);
        d = float((dupY ? (dupX ? q.x : q.y) : (dupX ? q.z : q.w)) >> 24) * (1.0f / 255.0f);
      }
      else
      {
        int x0, x1, y0, y1;
        wrapCoordPair(int(flx), w, wrapS, x0, x1);
        wrapCoordPair(int(fly), h, wrapT, y0, y1);
        a = float(texels[texelIndex(0u, w, x0, y0)].w) * (1.0f / 255.0f); b = float(texels[texelIndex(0u, w, x1, y0)].w) * (1.0f / 255.0f);
        c = float(texels[texelIndex(0u, w, x0, y1)].w) * (1.0f / 255.0f); d = float(texels[texelIndex(0u, w, x1, y1)].w) * (1.0f / 255.0f);
      }
      ta = (a * (1.0f - tx) + b * tx) * (1.0f - ty) + (c * (1.0f - tx) + d * tx) * ty;
    }
    alpha *= ta;
  }
  if(flg & AT_HAS_COLORS)
  {
    const uint32_t v = rec.c.w;
    alpha *= (float(v & 0xffu) / 255.0f) * bary.x + (float((v >> 8) & 0xffu) / 255.0f) * bary.y + (float((v >> 16) & 0xffu) / 255.0f) * bary.z;
  }
  if(mode == MI_ALPHA_MASK)
    return alpha >= rec.b.w ? 1.0f : 0.0f;
  return alpha;
}
PT_DEV DevShadeTri makeShadeRecord(const DevScene& sc, const DevTri& T)
{
  DevShadeTri             r;
  const uint32_t          rnode = __float_as_uint(T.a.w), prim = __float_as_uint(T.b.w);
  const MiGltfRenderNode& rn    = sc.nodes[rnode];
  const DevPrim           rp    = sc.prims[rn.renderPrimID];
  const u3                ti    = getTriangleIndices(rp, int(prim));
  const uint32_t          base  = uint32_t(rp.verts - sc.geomPool);
  r.v0 = base + 3u * ti.x; r.v1 = base + 3u * ti.y; r.v2 = base + 3u * ti.z;
  r.rnode        = rnode;
  r.renderPrimID = rn.renderPrimID;
  r.materialID   = max(0, rn.materialID);
  r.prim         = prim;
  r.attrs        = (rp.normals ? SHADE_HAS_NORMALS : 0u) | (rp.texCoords0 ? SHADE_HAS_UV0 : 0u) | (rp.tangents ? SHADE_HAS_TANGENTS : 0u)
            | (rp.texCoords1 ? SHADE_HAS_UV1 : 0u) | (rp.colors ? SHADE_HAS_COLORS : 0u);
  return r;
}
// Fills the record of one triangle (run once after the BVH build, for every triangle of the active order).
PT_DEV DevAlphaTri makeAlphaRecord(const DevScene& sc, const DevTri& T)
{
  DevAlphaTri rec;
  rec.a = make_float4(0, 0, 0, 0);
  rec.b = make_float4(0, 0, 1.0f, 0.5f);
  rec.c = make_uint4(0, 0, MI_ALPHA_OPAQUE, 0);
  const int                  rnode = int(__float_as_uint(T.a.w)), prim = int(__float_as_uint(T.b.w));
  const MiGltfRenderNode&    rn    = sc.nodes[rnode];
  const MiGltfShadeMaterial& mat   = sc.materials[max(0, rn.materialID)];
  if(mat.alphaMode == MI_ALPHA_OPAQUE)
    return rec;
  const DevPrim rp   = sc.prims[rn.renderPrimID];
  const u3      ti   = getTriangleIndices(rp, prim);
  uint32_t      flg  = uint32_t(mat.alphaMode) & AT_MODE_MASK;
  const bool    sg   = mat.pbrModel == MI_PBR_SPECULAR_GLOSSINESS;
  const uint16_t slot = sg ? mat.pbrDiffuseTexture : mat.pbrBaseColorTexture;
  rec.b.z            = sg ? mat.pbrDiffuseFactor[3] : mat.pbrBaseColorFactor[3];
  rec.b.w            = mat.alphaCutoff;
  if(isTexturePresent(slot))
  {
    const MiGltfTextureInfo info = sc.texInfos[slot];
    if(info.index >= 0 && info.index < sc.numTextures)
    {
      const DevTexture t  = sc.textures[info.index];
      const float*     tc = info.texCoord == 0 ? rp.texCoords0 : rp.texCoords1;
      if(tc)
      {
        rec.a   = make_float4(tc[2 * size_t(ti.x)], tc[2 * size_t(ti.x) + 1], tc[2 * size_t(ti.y)], tc[2 * size_t(ti.y) + 1]);
        rec.b.x = tc[2 * size_t(ti.z)];
        rec.b.y = tc[2 * size_t(ti.z) + 1];
      }
      rec.c.x = t.levelOffset[0];
      rec.c.y = uint32_t(t.width) | (uint32_t(t.height) << 16);
      flg |= AT_HAS_TEXTURE | (t.magFilter == MI_FILTER_LINEAR ? AT_LINEAR : 0u) | (uint32_t(t.wrapS) << AT_WRAPS_SHIFT) | (uint32_t(t.wrapT) << AT_WRAPT_SHIFT);
    }
  }
  if(rp.colors)
  {
    flg |= AT_HAS_COLORS;
    rec.c.w = (rp.colors[ti.x] >> 24) | ((rp.colors[ti.y] >> 24) << 8) | ((rp.colors[ti.z] >> 24) << 16);
  }
  rec.c.z = flg;
  return rec;
}

// getShadowTransmission, :244-343
PT_DEV f3 getShadowTransmissionBody(const DevScene& sc, int rnode, int triangleID, f3 bary, float hitT, f3 rayDir, bool& isInside, int shadeTri);
// .xyz = transmission, .w != 0: inside after the surface (by value, like sampleLightsCall)
__device__ __noinline__ f4 getShadowTransmissionCall(const DevScene& scIn, int rnode, int triangleID, f3 bary, float hitT, f3 rayDir, bool isInside)
{
  const f3 T = getShadowTransmissionBody(uniformConst(scIn), rnode, triangleID, bary, hitT, rayDir, isInside, -1);
  return mk4(T.x, T.y, T.z, isInside ? 1.0f : 0.0f);
}
PT_DEV f3 getShadowTransmission(const DevScene& sc, int rnode, int triangleID, f3 bary, float hitT, f3 rayDir, bool& isInside)
{
  const f4 r = getShadowTransmissionCall(sc, rnode, triangleID, bary, hitT, rayDir, isInside);
  isInside   = r.w != 0.0f;
  return xyz(r);
}
// the same through the triangle's shade record (`tri`: index in the active structure)
__device__ __noinline__ f4 getShadowTransmissionTriCall(const DevScene& scIn, int rnode, int triangleID, int tri, f3 bary, float hitT, f3 rayDir, bool isInside)
{
  const f3 T = getShadowTransmissionBody(uniformConst(scIn), rnode, triangleID, bary, hitT, rayDir, isInside, tri);
  return mk4(T.x, T.y, T.z, isInside ? 1.0f : 0.0f);
}
PT_DEV f3 getShadowTransmissionTri(const DevScene& sc, int rnode, int triangleID, int tri, f3 bary, float hitT, f3 rayDir, bool& isInside)
{
  const f4 r = getShadowTransmissionTriCall(sc, rnode, triangleID, tri, bary, hitT, rayDir, isInside);
  isInside   = r.w != 0.0f;
  return xyz(r);
}
// SHADE_REC: `shadeTri` >= 0 is the triangle's index in the active structure -- its shade record names the material, the render node and the three
// interleaved vertices (whose first float4 holds the same object-space position as the primitive's position stream), so the chain is record ->
// {material, node, vertices} instead of node -> {material, primitive} -> indices -> positions.  Same values, same arithmetic: the same result bit for bit
// (k_shadow_resolve, round 5); the primitive's pointer table is fetched only for a metallic-roughness TEXTURE.
PT_DEV f3 getShadowTransmissionBody(const DevScene& sc, int rnode, int triangleID, f3 bary, float hitT, f3 rayDir, bool& isInside, int shadeTri)
{
  const MiGltfRenderNode&    rn  = gat(sc.nodes, rnode);
  DevShadeTri                sr{};
  if(shadeTri >= 0)
    sr = gat(sc.shadeTris, shadeTri);
  const MiGltfShadeMaterial& mat = gat(sc.materials, shadeTri >= 0 ? sr.materialID : max(0, rn.materialID));
  float                      tFactor = mat.transmissionFactor;
  if(tFactor <= MIN_TRANSMISSION)
    return mk3(0.0f);
  DevPrim rp{};
  u3      ti{0u, 0u, 0u};
  f3      v0, v1, v2;
  if(shadeTri >= 0)
  {
    const float4 a = gat(sc.geomPool, sr.v0), b = gat(sc.geomPool, sr.v1), c = gat(sc.geomPool, sr.v2);
    v0 = mk3(a.x, a.y, a.z); v1 = mk3(b.x, b.y, b.z); v2 = mk3(c.x, c.y, c.z);
  }
  else
  {
    rp = gat(sc.prims, rn.renderPrimID);
    ti = getTriangleIndices(rp, triangleID);
    v0 = getVertexPosition(rp, ti.x); v1 = getVertexPosition(rp, ti.y); v2 = getVertexPosition(rp, ti.z);
  }
  f3 normal = normalize(cross(v1 - v0, v2 - v0));
  normal    = normalize(mulTransposed(rn.worldToObject, normal));
  float cosTheta = fabsf(dot(rayDir, normal));
  float fresnel  = schlickFresnelIor(mat.ior, cosTheta);
  f3    T        = mk3(mat.pbrBaseColorFactor[0], mat.pbrBaseColorFactor[1], mat.pbrBaseColorFactor[2]) * tFactor;
  T *= (1.0f - fresnel);
  if(mat.thicknessFactor > 0.0f)
  {
    if(isInside)
    {
      f3 absCoeff     = -log3(max3(mk3(mat.attenuationColor), mk3(0.001f))) / fmaxf(mat.attenuationDistance, 0.001f);
      f3 scatterCoeff = absCoeff * multiToSingleScatterAlbedo(mk3(mat.multiscatterColorFactor));
      f3 extinction   = absCoeff + scatterCoeff;
      T *= exp3(extinction * (-hitT));
      float maxScatter = maxComp(scatterCoeff);
      if(maxScatter > 0.001f)
        T *= expf(-(hitT * maxComp(extinction)));
    }
    isInside = !isInside;
  }
  float roughness = mat.pbrRoughnessFactor, metallic = mat.pbrMetallicFactor;
  if(isTexturePresent(mat.pbrMetallicRoughnessTexture))
  {
    if(shadeTri >= 0)
    {
      rp = gat(sc.prims, rn.renderPrimID);
      ti = getTriangleIndices(rp, triangleID);
    }
    const MiGltfTextureInfo info = gat(sc.texInfos, mat.pbrMetallicRoughnessTexture);
    f2                      uv   = getInterpolatedVertexTexCoord(rp, info.texCoord, ti, bary);
    f4                      mr   = sampleTexture(sc, info.index, uv, false, mk2(0, 0), mk2(0, 0));
    roughness *= mr.y;
    metallic *= mr.z;
  }
  float att = (1.0f - metallic);
  att *= lerpf(0.65f, 1.0f, 1.0f - roughness * roughness);
  return T * att;
}

// ---- direct lighting (pathtrace_functions.h.slang:357-492) ------------------------------------------------------------------
struct DirectLight
{
  f3    direction, radianceOverPdf;
  float distance, pdf;
};
PT_DEV void getDirectLightingTechniqueProbabilities(const DevScene& sc, const FrameConsts& fc, float& lightWeight, float& envWeight)
{
  lightWeight = (sc.numLights > 0) ? 0.5f : 0.0f;
  envWeight   = (!hasFlag(fc.frameInfo.flags, MI_SCENE_USE_HDR_ENVIRONMENT) || fc.frameInfo.envIntensity > 0.0f) ? 0.5f : 0.0f;
  float total = lightWeight + envWeight;
  if(total > 0.0f)
  {
    lightWeight /= total;
    envWeight /= total;
  }
}
// (results by value: a reference parameter of a non-inlined function is a pointer into the caller's scratch, written and read
// back through flat memory instructions; nine floats come back in registers)
struct LightSample
{
  DirectLight dl;
  uint32_t    seed;
};
PT_DEV void sampleLightsBody(const DevScene& sc, const FrameConsts& fc, f3 pos, uint32_t& seed, DirectLight& dl);
__device__ __noinline__ LightSample sampleLightsCall(const DevScene& scIn, const FrameConsts& fcIn, f3 pos, uint32_t seed)
{
  LightSample r;
  r.seed = seed;
  sampleLightsBody(uniformConst(scIn), uniformConst(fcIn), pos, r.seed, r.dl);
  return r;
}
PT_DEV void sampleLights(const DevScene& sc, const FrameConsts& fc, f3 pos, uint32_t& seed, DirectLight& dl)  // :379-464
{
  // inlined at its call sites since round 6 (a non-inlined copy until then: the call's argument / result moves and the registers saved around it cost more than
  // the body's second copy once the division expansions were gone -- atrium 748.1 -> 750.9, street 792.3 -> 795.2, helmet 5411 -> 5441 Msamples/s,
  // profiles/r06_shade_walk_ab.txt).  -DMI_PT_CALL_SAMPLE_LIGHTS restores the call.
#ifdef MI_PT_CALL_SAMPLE_LIGHTS
  const LightSample r = sampleLightsCall(sc, fc, pos, seed);
  seed = r.seed;
  dl   = r.dl;
#else
  sampleLightsBody(uniformConst(sc), uniformConst(fc), pos, seed, dl);
#endif
}
PT_DEV void sampleLightsBody(const DevScene& sc, const FrameConsts& fc, f3 pos, uint32_t& seed, DirectLight& dl)
{
  f3 radiance        = mk3(0.0f);
  dl.pdf             = 0.0f;
  dl.distance        = INFINITE_F;
  dl.radianceOverPdf = mk3(0.0f);
  dl.direction       = mk3(0.0f);
  float envPdf       = 0.0f;
  float lightWeight, envWeight;
  getDirectLightingTechniqueProbabilities(sc, fc, lightWeight, envWeight);
  if(lightWeight == 0.0f && envWeight == 0.0f)
    return;
  bool       sampleLight = (rnd(seed) < lightWeight);
  const bool useHdr      = hasFlag(fc.frameInfo.flags, MI_SCENE_USE_HDR_ENVIRONMENT);
  if(sampleLight)
  {
    int         numLights    = sc.numLights;
    float       selectionPdf = 1.0f / float(numLights);
    int         lightIndex   = min(int(rnd(seed) * float(numLights)), numLights - 1);
    MiGltfLight light        = gat(sc.lights, lightIndex);
    float       r1 = rnd(seed), r2 = rnd(seed);
    LightContrib contrib = singleLightContribution(light, pos, mk2(r1, r2));
    dl.direction         = -contrib.incidentVector;
    dl.distance          = contrib.distance;
    radiance             = contrib.intensity / (selectionPdf * lightWeight);
    dl.pdf               = (contrib.pdf == DIRAC) ? DIRAC : selectionPdf * contrib.pdf;
  }
  if(envWeight > 0.0f && dl.pdf != DIRAC)
  {
    if(!useHdr)
    {
      if(!sampleLight)
      {
        float r1 = rnd(seed), r2 = rnd(seed);
        f3    skyRadiance;
        samplePhysicalSky(fc.sky, uniformConst(*fc.skyPre), mk2(r1, r2), dl.direction, envPdf, skyRadiance);
        radiance = skyRadiance / (envPdf * envWeight);
      }
      else
        envPdf = samplePhysicalSkyPDF(fc.sky, uniformConst(*fc.skyPre), skyGamma(uniformConst(*fc.skyPre), dl.direction));
    }
    else
    {
      if(!sampleLight)
      {
        float r1 = rnd(seed), r2 = rnd(seed), r3 = rnd(seed);
        f4    rp = environmentSample(sc, mk3(r1, r2, r3), dl.direction);
        envPdf   = rp.w;
        radiance = xyz(rp) * fc.frameInfo.envIntensity / (envPdf * envWeight);
        dl.direction = rotateAxis(dl.direction, mk3(0, 1, 0), fc.frameInfo.envRotation);
      }
      else
      {
        f3 dir = rotateAxis(dl.direction, mk3(0, 1, 0), -fc.frameInfo.envRotation);
        envPdf = sampleHdr(sc, getSphericalUv(dir)).w;
      }
    }
  }
  float misWeight = 1.0f;
  if(dl.pdf != DIRAC)
  {
    float pdfSum = lightWeight * dl.pdf + envWeight * envPdf;
    if(pdfSum > 0.0f)
      misWeight = (sampleLight ? lightWeight * dl.pdf : envWeight * envPdf) / pdfSum;
    dl.pdf = pdfSum;
  }
  radiance *= misWeight;
  if(!isFinite3(radiance))  // zero-pdf environment texel: treat as "no light"
  {
    radiance = mk3(0.0f);
    dl.pdf   = 0.0f;
  }
  dl.radianceOverPdf = radiance;
}
PT_DEV void sampleEnvironment(const DevScene& sc, const FrameConsts& fc, f3 direction, f3& envColor, float& envPdf)  // :466-481
{
  if(!hasFlag(fc.frameInfo.flags, MI_SCENE_USE_HDR_ENVIRONMENT))
  {
    const SkyPrecomp& pre   = uniformConst(*fc.skyPre);
    const float       gamma = skyGamma(pre, direction);
    envColor = evalPhysicalSky(fc.sky, pre, direction, gamma);
    envPdf   = samplePhysicalSkyPDF(fc.sky, pre, gamma);
  }
  else
  {
    f3 dir   = rotateAxis(direction, mk3(0, 1, 0), -fc.frameInfo.envRotation);
    f4 env   = sampleHdr(sc, getSphericalUv(dir));
    envColor = xyz(env) * fc.frameInfo.envIntensity;
    envPdf   = env.w;
  }
}
PT_DEV float computeEnvHitMisWeight(const DevScene& sc, const FrameConsts& fc, float lastSamplePdf, float envPdf)  // :483-492
{
  if(lastSamplePdf == DIRAC)
    return 1.0f;
  float lw, ew;
  getDirectLightingTechniqueProbabilities(sc, fc, lw, ew);
  return lastSamplePdf / (lastSamplePdf + ew * envPdf);
}
// smoothHDRBlur (backplate only; nvshaders/sample_blur.h.slang is external): 3x3 tent of level-0 taps
PT_DEV f3 smoothHDRBlur(const DevScene& sc, f2 uv, float blur)
{
  f3    sum  = mk3(0.0f);
  float wsum = 0.0f;
  float r    = blur * 0.02f;
  for(int j = -1; j <= 1; ++j)
    for(int i = -1; i <= 1; ++i)
    {
      float w = (2.0f - fabsf(float(i))) * (2.0f - fabsf(float(j)));
      sum += xyz(sampleHdr(sc, mk2(uv.x + float(i) * r, uv.y + float(j) * r * 0.5f))) * w;
      wsum += w;
    }
  return sum / wsum;
}

}  // namespace pt
