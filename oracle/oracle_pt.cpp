/*
 * oracle_pt.cpp — CPU ORACLE for the vk_gltf_renderer path-trace hot path.
 *
 * THIS FILE IS TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build,
 * load or call it.  The product (libmi_pt.so) never links or calls anything in oracle/.
 *
 * It is a scalar, one-thread-per-pixel ("megakernel") restatement of the reference's algorithm, function by function:
 *   processPixel / samplePixel / pathTrace / pathTraceOneBounce   shaders/gltf_pathtrace.slang:87-671
 *   sampleLights, getOpacity, getShadowTransmission, volume, ...   shaders/pathtrace_functions.h.slang:36-991
 *   RayQueryRaytracer::{Trace,TraceLow,TraceShadow}                shaders/raytracer_interface.h.slang:67-188
 *   getHitState                                                    shaders/get_hit.h.slang:59-173
 *   evaluateMaterial / getTexture                                  shaders/gltf_material_eval.h.slang:76-462
 *   vertex accessors                                               shaders/gltf_vertex_access.h.slang:30-150
 * Each function below cites the lines it follows.
 *
 * PARITY UNPINNED: the BSDF, light, sky, HDR-sampling, RNG and shadow-terminator arithmetic of the reference lives
 * in nvpro-samples/nvpro_core2 (branch `main`, unpinned: cmake/FindNvproCore2.cmake:28,84-87), which is NOT in
 * /root/reference, and the reference ships no golden radiance.  Those parts (section "nvshaders restatement") are
 * restated from the published models named at each function; the oracle is therefore pinned only by
 *   (i) analytic known-answer tests (furnace, Lambert plane, RNG bit vectors) in tests/, and
 *  (ii) the host-side material-conversion expectations of the reference's tests/test_material_cache.cpp.
 *
 * Two deliberate, documented deviations from the reference (see DESIGN.md "determinism"):
 *   - the stochastic-alpha draw inside Trace/TraceShadow is rand(hash(seed, renderNode, primitive)) instead of
 *     advancing the path's seed once per candidate, because the reference's candidate order is hardware-defined
 *     (raytracer_interface.h.slang:82-112) and a software BVH must not leak its traversal order into the image;
 *   - shadow-ray candidates are processed in increasing t (the order the reference assumes, :167).
 *
 * Build: see oracle/Makefile (g++ -O2 -ffp-contract=off).
 */
#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "../include/mi_pt.h"
#include "oracle_math.h"
#include "oracle_pt.h"

using namespace orc;

namespace {

//======================================================================================================================
// nvshaders restatement (external to /root/reference — see header)
//======================================================================================================================
constexpr float INFINITE_F  = 1e32f;   // nvshaders/constants.h.slang INFINITE
constexpr float DIRAC       = -1.0f;   // nvshaders/constants.h.slang DIRAC
constexpr float K_PI        = 3.14159265358979323846f;
constexpr float K_TWO_PI    = 6.28318530717958647692f;
constexpr float K_1_OVER_PI = 0.31830988618379067154f;

// nvshaders/random.h.slang: xxhash32 of a uint3 (Collet's xxHash32 specialised to 12 bytes, seed 0 folded in).
uint32_t xxhash32(uint32_t px, uint32_t py, uint32_t pz)
{
  const uint32_t PRIME32_2 = 2246822519U, PRIME32_3 = 3266489917U, PRIME32_4 = 668265263U, PRIME32_5 = 374761393U;
  auto rotl = [](uint32_t v, int r) { return (v << r) | (v >> (32 - r)); };
  uint32_t h32 = pz + PRIME32_5 + px * PRIME32_3;
  h32          = PRIME32_4 * rotl(h32, 17);
  h32 += py * PRIME32_3;
  h32 = PRIME32_4 * rotl(h32, 17);
  h32 = PRIME32_2 * (h32 ^ (h32 >> 15));
  h32 = PRIME32_3 * (h32 ^ (h32 >> 13));
  return h32 ^ (h32 >> 16);
}
// nvshaders/random.h.slang: PCG (same constants as the in-tree shaders/common.h.slang:33-42 hashToColor)
uint32_t pcg(uint32_t& state)
{
  uint32_t prev = state * 747796405u + 2891336453u;
  uint32_t word = ((prev >> ((prev >> 28u) + 4u)) ^ prev) * 277803737u;
  state         = prev;
  return (word >> 22u) ^ word;
}
float u32ToUnitFloat(uint32_t r) { return asfloat(0x3f800000u | (r >> 9)) - 1.0f; }
float rnd(uint32_t& seed) { return u32ToUnitFloat(pcg(seed)); }
// order-independent alpha draw (documented deviation)
float candidateRand(uint32_t seed, int rnode, int prim) { return u32ToUnitFloat(xxhash32(seed, uint32_t(rnode), uint32_t(prim))); }

// nvshaders/functions.h.slang
float2 getSphericalUv(float3 v)
{
  float gamma = std::asin(clampf(-v.y, -1.0f, 1.0f));
  float theta = std::atan2(v.z, v.x);
  return float2(theta * (0.5f * K_1_OVER_PI) + 0.5f, gamma * K_1_OVER_PI + 0.5f);
}
float3 rotate(float3 v, float3 k, float theta)  // Rodrigues
{
  float c = std::cos(theta), s = std::sin(theta);
  return v * c + cross(k, v) * s + k * (dot(k, v) * (1.0f - c));
}
// Duff et al. 2017, "Building an Orthonormal Basis, Revisited"; w = handedness
float4 makeFastTangent(float3 n)
{
  float sign = std::copysign(1.0f, n.z);
  float a    = -1.0f / (sign + n.z);
  float b    = n.x * n.y * a;
  return float4(1.0f + sign * n.x * n.x * a, sign * b, -sign * n.x, 1.0f);
}
// nvshaders/ray_utils.h.slang pointOffset: Hanika 2021, "Hacking the Shadow Terminator"
float3 pointOffset(float3 p, float3 p0, float3 p1, float3 p2, float3 n0, float3 n1, float3 n2, float3 bary)
{
  float3 tmpu = p - p0, tmpv = p - p1, tmpw = p - p2;
  float  dotu = std::fmin(0.0f, dot(tmpu, n0)), dotv = std::fmin(0.0f, dot(tmpv, n1)), dotw = std::fmin(0.0f, dot(tmpw, n2));
  tmpu -= n0 * dotu;
  tmpv -= n1 * dotv;
  tmpw -= n2 * dotw;
  return p + tmpu * bary.x + tmpv * bary.y + tmpw * bary.z;
}

// IEEE binary16 round trip (the reference stores VolumeMedium as float16_t: pathtrace_functions.h.slang:118-123)
float roundToHalf(float f)
{
  uint32_t x    = asuint(f);
  uint32_t sign = x & 0x80000000u;
  uint32_t mag  = x & 0x7fffffffu;
  if(mag >= 0x7f800000u)
    return f;  // inf / nan
  if(mag >= 0x477ff000u)  // rounds to >= 65520 -> inf
    return asfloat(sign | 0x7f800000u);
  if(mag < 0x38800000u)  // half subnormal range: quantum 2^-24
  {
    float a = asfloat(mag);
    float q = std::nearbyint(a * 16777216.0f) * (1.0f / 16777216.0f);
    return asfloat(sign | asuint(q));
  }
  uint32_t rem = mag & 0x1fffu;
  mag &= ~0x1fffu;
  if(rem > 0x1000u || (rem == 0x1000u && (mag & 0x2000u)))
    mag += 0x2000u;
  return asfloat(sign | mag);
}

//----------------------------------------------------------------------------------------------------------------------
// nvshaders/pbr_material_types.h.slang: fields as used by shaders/gltf_material_eval.h.slang:195-453
struct PbrMaterial
{
  float3 baseColor{1, 1, 1};
  float  opacity = 1;
  float2 roughness{1, 1};
  float  metallic  = 1;
  float3 emissive{0, 0, 0};
  float  occlusion = 1;
  float3 N{0, 0, 1}, T{1, 0, 0}, B{0, 1, 0}, Ng{0, 0, 1};
  float  ior1 = 1, ior2 = 1.5f;
  float  specular = 1;
  float3 specularColor{1, 1, 1};
  float  transmission = 0;
  float3 attenuationColor{1, 1, 1};
  float  attenuationDistance = 1;
  float  thickness           = 0;
  float3 scatterCoefficient{0, 0, 0};
  float  scatterAnisotropy  = 0;
  float  clearcoat          = 0;
  float  clearcoatRoughness = 0.01f;
  float3 Nc{0, 0, 1};
  float  iridescence          = 0;
  float  iridescenceIor       = 1.5f;
  float  iridescenceThickness = 0.1f;
  float3 sheenColor{0, 0, 0};
  float  sheenRoughness = 0;
  float  dispersion     = 0;
  float  diffuseTransmissionFactor = 0;
  float3 diffuseTransmissionColor{1, 1, 1};
  float  retroreflection = 0;
};

PbrMaterial defaultPbrMaterial5(float3 baseColor, float metallic, float roughness, float3 N, float3 Ng)
{
  PbrMaterial m;
  m.baseColor = baseColor;
  m.metallic  = metallic;
  m.roughness = float2(roughness * roughness);
  m.N         = N;
  m.Ng        = Ng;
  m.Nc        = N;
  float4 t    = makeFastTangent(N);
  m.T         = t.xyz();
  m.B         = cross(N, m.T);
  return m;
}

// BSDF event flags (nvshaders/bsdf_types.h.slang)
enum : int
{
  BSDF_EVENT_ABSORB       = 0,
  BSDF_EVENT_DIFFUSE      = 1,
  BSDF_EVENT_GLOSSY       = 1 << 1,
  BSDF_EVENT_IMPULSE      = 1 << 2,
  BSDF_EVENT_REFLECTION   = 1 << 3,
  BSDF_EVENT_TRANSMISSION = 1 << 4,
  BSDF_EVENT_DIFFUSE_REFLECTION   = BSDF_EVENT_DIFFUSE | BSDF_EVENT_REFLECTION,
  BSDF_EVENT_DIFFUSE_TRANSMISSION = BSDF_EVENT_DIFFUSE | BSDF_EVENT_TRANSMISSION,
  BSDF_EVENT_GLOSSY_REFLECTION    = BSDF_EVENT_GLOSSY | BSDF_EVENT_REFLECTION,
  BSDF_EVENT_GLOSSY_TRANSMISSION  = BSDF_EVENT_GLOSSY | BSDF_EVENT_TRANSMISSION,
  BSDF_EVENT_IMPULSE_REFLECTION   = BSDF_EVENT_IMPULSE | BSDF_EVENT_REFLECTION,
};
struct BsdfEvaluateData
{
  float3 k1, k2, xi;
  float3 bsdf_diffuse{0, 0, 0}, bsdf_glossy{0, 0, 0};
  float  pdf = 0;
};
struct BsdfSampleData
{
  float3 k1, xi;
  float3 k2{0, 0, 0}, bsdf_over_pdf{0, 0, 0};
  float  pdf        = 0;
  int    event_type = BSDF_EVENT_ABSORB;
};

enum : int
{
  LOBE_DIFFUSE_REFLECTION = 0,
  LOBE_SPECULAR_TRANSMISSION,
  LOBE_SPECULAR_REFLECTION,
  LOBE_METAL_REFLECTION,
  LOBE_SHEEN_REFLECTION,
  LOBE_CLEARCOAT_REFLECTION,
  LOBE_COUNT
};

// --- scalar helpers of the layered GGX model (MDL-SDK libbsdf lineage, as used by nvshaders/bsdf_functions) ---------
float schlickFresnelIor(float ior, float VdotH)
{
  float R0 = sqr((1.0f - ior) / (1.0f + ior));
  return R0 + (1.0f - R0) * std::pow(1.0f - VdotH, 5.0f);
}
float3 mix_rgb(float3 base, float3 layer, float3 factor) { return base * (1.0f - maxComp(factor)) + factor * layer; }
bool   isTIR(float2 ior, float kh)
{
  float b = ior.x / ior.y;
  return 1.0f < (b * b * (1.0f - kh * kh));
}
// Fresnel for an equal mix of polarisations; eta = refracted / reflected ior
float ior_fresnel(float eta, float kh)
{
  float costheta = 1.0f - (1.0f - kh * kh) / (eta * eta);
  if(costheta <= 0.0f)
    return 1.0f;
  costheta   = std::sqrt(costheta);
  float n1t1 = kh, n1t2 = costheta, n2t1 = kh * eta, n2t2 = costheta * eta;
  float r_p = (n1t2 - n2t1) / (n1t2 + n2t1);
  float r_o = (n1t1 - n2t2) / (n1t1 + n2t2);
  return clampf(0.5f * (r_p * r_p + r_o * r_o), 0.0f, 1.0f);
}
float2 fresnel_dielectric(float n_a, float n_b, float cos_a, float cos_b)
{
  float naca = n_a * cos_a, nbcb = n_b * cos_b;
  float r_s  = (naca - nbcb) / (naca + nbcb);
  float nacb = n_a * cos_b, nbca = n_b * cos_a;
  float r_p  = (nbca - nacb) / (nbca + nacb);
  return float2(r_s * r_s, r_p * r_p);
}
// Born & Wolf, Principles of Optics §13.4: squared norms of s/p reflection coefficients + phase shifts
float2 fresnel_conductor(float2& phase_sin, float2& phase_cos, float n_a, float n_b, float k_b, float cos_a, float sin_a_sqd)
{
  float k_b2 = k_b * k_b, n_b2 = n_b * n_b, n_a2 = n_a * n_a;
  float tmp0   = n_b2 - k_b2;
  float half_U = 0.5f * (tmp0 - n_a2 * sin_a_sqd);
  float half_V = std::sqrt(std::fmax(0.0f, half_U * half_U + k_b2 * n_b2));
  float u_b2 = half_U + half_V, v_b2 = half_V - half_U;
  float u_b = std::sqrt(std::fmax(0.0f, u_b2)), v_b = std::sqrt(std::fmax(0.0f, v_b2));
  float tmp1 = tmp0 * cos_a, tmp2 = n_a * u_b, tmp3 = (2.0f * n_b * k_b) * cos_a, tmp4 = n_a * v_b, tmp5 = n_a * cos_a;
  float tmp6 = (2.0f * tmp5) * v_b;
  float tmp7 = (u_b2 + v_b2) - tmp5 * tmp5;
  float tmp8 = (2.0f * tmp5) * ((2.0f * n_b * k_b) * u_b - tmp0 * v_b);
  float tmp9 = sqr((n_b2 + k_b2) * cos_a) - n_a2 * (u_b2 + v_b2);
  float tmp67 = tmp6 * tmp6 + tmp7 * tmp7;
  float inv_x = (0.0f < tmp67) ? 1.0f / std::sqrt(tmp67) : 0.0f;
  float tmp89 = tmp8 * tmp8 + tmp9 * tmp9;
  float inv_y = (0.0f < tmp89) ? 1.0f / std::sqrt(tmp89) : 0.0f;
  phase_cos   = float2(tmp7 * inv_x, tmp9 * inv_y);
  phase_sin   = float2(tmp6 * inv_x, tmp8 * inv_y);
  return float2((sqr(tmp5 - u_b) + v_b2) / (sqr(tmp5 + u_b) + v_b2),
                (sqr(tmp1 - tmp2) + sqr(tmp3 - tmp4)) / (sqr(tmp1 + tmp2) + sqr(tmp3 + tmp4)));
}
// 16-wavelength spectral table, generated by tools/gen_thinfilm_table.py (Wyman et al. 2013 CMF fit -> Rec.709)
const float kThinFilmRgb[16][3] = {
    {1.141170327e-01f, -9.233230420e-02f, 6.243416750e-01f},   {3.891298474e-01f, -5.067071748e-01f, 4.143101779e+00f},
    {5.073899927e-01f, -6.406128719e-01f, 5.882926651e+00f},   {-2.440306838e-01f, -7.686629932e-02f, 4.887233973e+00f},
    {-8.422044385e-01f, 8.145953282e-01f, 1.993870357e+00f},   {-1.600349816e+00f, 2.098418188e+00f, 5.145780610e-01f},
    {-2.124796802e+00f, 3.884327156e+00f, -2.076202664e-01f},  {-1.257502337e+00f, 4.473430149e+00f, -4.918162342e-01f},
    {9.183988254e-01f, 3.804074233e+00f, -5.128916518e-01f},   {3.595737427e+00f, 2.365924612e+00f, -4.060246333e-01f},
    {5.558583049e+00f, 7.311976312e-01f, -2.452682962e-01f},   {5.496764639e+00f, -2.826184233e-01f, -1.104067936e-01f},
    {3.502888886e+00f, -3.898076749e-01f, -4.277698832e-02f},  {1.487011583e+00f, -1.669308389e-01f, -1.796845442e-02f},
    {4.212750925e-01f, -2.408596850e-02f, -8.143819362e-03f},  {7.758770274e-02f, 7.994258193e-03f, -3.135358120e-03f}};
// Thin-film interference reflectance of a dielectric coating on a dielectric base (libbsdf thin_film_factor structure)
float3 thin_film_factor(float coating_thickness, float coating_ior, float base_ior, float incoming_ior, float kh)
{
  coating_thickness = std::fmax(0.0f, coating_thickness);
  float sin0_sqr    = std::fmax(0.0f, 1.0f - kh * kh);
  float eta01       = incoming_ior / coating_ior;
  float sin1_sqr    = eta01 * eta01 * sin0_sqr;
  if(1.0f < sin1_sqr)
    return float3(1.0f);
  float  cos1 = std::sqrt(std::fmax(0.0f, 1.0f - sin1_sqr));
  float2 R01  = fresnel_dielectric(incoming_ior, coating_ior, kh, cos1);
  float2 phi12_sin, phi12_cos;
  float2 R12      = fresnel_conductor(phi12_sin, phi12_cos, coating_ior, base_ior, 0.0f, cos1, sin1_sqr);
  float  tmp      = (4.0f * K_PI) * coating_ior * coating_thickness * cos1;
  float  R01R12_s = std::fmax(0.0f, R01.x * R12.x), r01r12_s = std::sqrt(R01R12_s);
  float  R01R12_p = std::fmax(0.0f, R01.y * R12.y), r01r12_p = std::sqrt(R01R12_p);
  float3 rgb(0.0f);
  float  lambda = 400.0f + 0.5f * 18.75f;
  for(int i = 0; i < 16; ++i)
  {
    float phi = tmp / lambda;
    float ps = std::sin(phi), pc = std::cos(phi);
    float cos_phi_s = pc * phi12_cos.x - ps * phi12_sin.x;
    float tmp_s     = 2.0f * r01r12_s * cos_phi_s;
    float R_s       = (R01.x + R12.x + tmp_s) / (1.0f + R01R12_s + tmp_s);
    float cos_phi_p = pc * phi12_cos.y - ps * phi12_sin.y;
    float tmp_p     = 2.0f * r01r12_p * cos_phi_p;
    float R_p       = (R01.y + R12.y + tmp_p) / (1.0f + R01R12_p + tmp_p);
    float R         = 0.5f * (R_s + R_p);
    rgb += float3(kThinFilmRgb[i][0], kThinFilmRgb[i][1], kThinFilmRgb[i][2]) * R;
    lambda += 18.75f;
  }
  return clamp3(rgb * (1.0f / 16.0f), 0.0f, 1.0f);
}

float3 cosineSampleHemisphere(float r1, float r2)
{
  float r = std::sqrt(r1), phi = K_TWO_PI * r2;
  float3 d;
  d.x = r * std::cos(phi);
  d.y = r * std::sin(phi);
  d.z = std::sqrt(std::fmax(0.0f, 1.0f - d.x * d.x - d.y * d.y));
  return d;
}
// anisotropic GGX on the non-projected hemisphere: D(h) * h.z
float hvd_ggx_eval(float2 invRoughness, float3 h)
{
  float x = h.x * invRoughness.x, y = h.y * invRoughness.y;
  float f = x * x + y * y + h.z * h.z;
  return K_1_OVER_PI * invRoughness.x * invRoughness.y * h.z / (f * f);
}
// Heitz 2017, "A Simpler and Exact Sampling Routine for the GGX Distribution of Visible Normals"
float3 hvd_ggx_sample_vndf(float3 k, float2 roughness, float2 xi)
{
  float3 v  = normalize(float3(k.x * roughness.x, k.y * roughness.y, k.z));
  float3 t1 = (v.z < 0.99999f) ? normalize(cross(v, float3(0, 0, 1))) : float3(1, 0, 0);
  float3 t2 = cross(t1, v);
  float  a  = 1.0f / (1.0f + v.z);
  float  r  = std::sqrt(xi.x);
  float  phi = (xi.y < a) ? xi.y / a * K_PI : K_PI + (xi.y - a) / (1.0f - a) * K_PI;
  float  sp = std::sin(phi), cp = std::cos(phi);
  float  p1 = r * cp;
  float  p2 = r * sp * ((xi.y < a) ? 1.0f : v.z);
  float3 h  = t1 * p1 + t2 * p2 + v * std::sqrt(std::fmax(0.0f, 1.0f - p1 * p1 - p2 * p2));
  h.x *= roughness.x;
  h.y *= roughness.y;
  h.z = std::fmax(0.0f, h.z);
  return normalize(h);
}
float smith_shadow_mask(float3 k, float2 roughness)
{
  float kz2 = k.z * k.z;
  if(kz2 == 0.0f)
    return 0.0f;
  float ax = k.x * roughness.x, ay = k.y * roughness.y;
  float inv_a2 = (ax * ax + ay * ay) / kz2;
  return 2.0f / (1.0f + std::sqrt(1.0f + inv_a2));
}
float ggx_smith_shadow_mask(float& G1, float& G2, float3 k1, float3 k2, float2 roughness)
{
  G1 = smith_shadow_mask(k1, roughness);
  G2 = smith_shadow_mask(k2, roughness);
  return G1 * G2;
}
float3 refractDir(float3 k, float3 n, float b, float nk, bool& tir)
{
  float refraction = b * b * (1.0f - nk * nk);
  tir              = (1.0f <= refraction);
  return tir ? (n * (nk + nk) - k) : normalize(k * (-b) + n * (b * nk - std::sqrt(1.0f - refraction)));
}
float3 compute_half_vector(float3 k1, float3 k2, float3 normal, float2 ior, float nk2, bool transmission, bool thinwalled)
{
  float3 h;
  if(transmission)
  {
    if(thinwalled)
      h = k1 + (normal * (nk2 + nk2) + k2);
    else
    {
      h = k2 * ior.y + k1 * ior.x;
      if(ior.y > ior.x)
        h = -h;
    }
  }
  else
    h = k1 + k2;
  return normalize(h);
}
// Sheen ("Charlie"-style sin^n) half-vector distribution on the non-projected hemisphere
float hvd_sheen_eval(float invRoughness, float nh)
{
  float sinTheta = std::sqrt(std::fmax(0.0f, 1.0f - nh * nh));
  return (invRoughness + 2.0f) * std::pow(sinTheta, invRoughness) * 0.5f * K_1_OVER_PI * nh;
}
float vcavities_mask(float nh, float kh, float nk) { return std::fmin(2.0f * nh * nk / kh, 1.0f); }
float vcavities_shadow_mask(float& G1, float& G2, float nh, float3 k1, float k1h, float3 k2, float k2h)
{
  G1 = vcavities_mask(nh, k1h, k1.z);
  G2 = vcavities_mask(nh, k2h, k2.z);
  return std::fmin(G1, G2);
}
float3 hvd_sheen_sample(float2 xi, float invRoughness)
{
  float phi      = K_TWO_PI * xi.x;
  float sinTheta = std::pow(1.0f - xi.y, 1.0f / (invRoughness + 2.0f));
  float cosTheta = std::sqrt(std::fmax(0.0f, 1.0f - sinTheta * sinTheta));
  return normalize(float3(std::cos(phi) * sinTheta, std::sin(phi) * sinTheta, cosTheta));
}
float3 flipH(float3 h, float3 k, float xi)
{
  float a = h.z * k.z, b = h.x * k.x + h.y * k.y;
  float kh = std::fmax(0.0f, a + b), kh_f = std::fmax(0.0f, a - b);
  float p_flip = kh_f / (kh + kh_f);
  return (xi < p_flip) ? float3(-h.x, -h.y, h.z) : h;
}

float3 absorptionCoefficient(const PbrMaterial& mat)
{
  float d = mat.attenuationDistance;
  return d <= 0.0f ? float3(0.0f) : -log3(mat.attenuationColor) / d;
}
// nvshaders volumeExtinctionCoefficient (call site: pathtrace_functions.h.slang:128)
float3 volumeExtinctionCoefficient(const PbrMaterial& mat) { return absorptionCoefficient(mat) + mat.scatterCoefficient; }

// --- lobe selection --------------------------------------------------------------------------------------------------
struct LobePick
{
  int    lobe;
  float  u;  // rndVal re-stretched to [0,1) inside the chosen lobe's interval
  float3 tint;
};
void computeLobeWeights(const PbrMaterial& mat, float VdotN, float3& tint, float w[LOBE_COUNT])
{
  float frCoat = 0.0f;
  if(mat.clearcoat > 0.0f)
    frCoat = mat.clearcoat * ior_fresnel(1.5f / mat.ior1, VdotN);
  float frDielectric = 0.0f;
  if(mat.specular > 0.0f)
    frDielectric = ior_fresnel(mat.ior2 / mat.ior1, VdotN) * mat.specular;
  if(mat.iridescence > 0.0f)
  {
    float3 frIrid = thin_film_factor(mat.iridescenceThickness, mat.iridescenceIor, mat.ior2, mat.ior1, VdotN);
    frDielectric  = lerp(frDielectric, maxComp(frIrid), mat.iridescence);
    tint          = mix_rgb(tint, mat.specularColor, frIrid * mat.iridescence);
  }
  float sheen = 0.0f;
  if(mat.sheenColor.x != 0.0f || mat.sheenColor.y != 0.0f || mat.sheenColor.z != 0.0f)
  {
    sheen = std::pow(1.0f - std::fabs(VdotN), mat.sheenRoughness);
    sheen = sheen / (sheen + 0.5f);
  }
  float base                    = 1.0f;
  w[LOBE_CLEARCOAT_REFLECTION]  = frCoat;
  base *= 1.0f - frCoat;
  w[LOBE_SHEEN_REFLECTION] = base * sheen;
  base *= 1.0f - sheen;
  w[LOBE_METAL_REFLECTION] = base * mat.metallic;
  base *= 1.0f - mat.metallic;
  w[LOBE_SPECULAR_REFLECTION] = base * frDielectric;
  base *= 1.0f - frDielectric;
  w[LOBE_SPECULAR_TRANSMISSION] = base * mat.transmission;
  w[LOBE_DIFFUSE_REFLECTION]    = base * (1.0f - mat.transmission);
}
LobePick findLobe(const PbrMaterial& mat, float VdotN, float rndVal)
{
  LobePick p;
  p.tint = mat.baseColor;
  float w[LOBE_COUNT];
  computeLobeWeights(mat, VdotN, p.tint, w);
  int   lobe   = LOBE_COUNT;
  float weight = 0.0f, lo = 0.0f;
  while(--lobe > 0)
  {
    lo = weight;
    weight += w[lobe];
    if(rndVal < weight)
      break;
  }
  float hi = weight;
  if(lobe == 0)
  {
    lo = weight;
    hi = 1.0f;
  }
  p.lobe = lobe;
  p.u    = (hi > lo) ? clampf((rndVal - lo) / (hi - lo), 0.0f, 0.99999994f) : 0.0f;
  return p;
}
// split one uniform number into a Bernoulli(p) decision and a fresh uniform number
bool splitRandom(float& u, float p)
{
  if(u < p)
  {
    u = p > 0.0f ? u / p : 0.0f;
    return true;
  }
  u = (1.0f - p) > 0.0f ? (u - p) / (1.0f - p) : 0.0f;
  return false;
}
// KHR_materials_retroreflection (MRM): reflection lobes see the mirrored view vector with probability `retroreflection`
float3 retroView(const PbrMaterial& mat, float3 k1, float3 N, float& u)
{
  if(mat.retroreflection > 0.0f && splitRandom(u, mat.retroreflection))
    return N * (2.0f * dot(N, k1)) - k1;
  return k1;
}
// KHR_materials_dispersion: per-channel IOR, one channel traced per sample (weight 3 on that channel)
float3 applyDispersion(PbrMaterial& mat, float& u)
{
  if(mat.dispersion <= 0.0f)
    return float3(1.0f);
  int   c          = std::min(int(u * 3.0f), 2);
  u                = u * 3.0f - float(c);
  bool  outside    = (mat.ior1 == 1.0f);
  float ior        = outside ? mat.ior2 : mat.ior1;
  float halfSpread = (ior - 1.0f) * 0.025f * mat.dispersion;
  float iorC       = ior + halfSpread * float(c - 1);
  if(outside)
    mat.ior2 = iorC;
  else
    mat.ior1 = iorC;
  return float3(c == 0 ? 3.0f : 0.0f, c == 1 ? 3.0f : 0.0f, c == 2 ? 3.0f : 0.0f);
}

void absorbEval(BsdfEvaluateData& d)
{
  d.bsdf_diffuse = float3(0.0f);
  d.bsdf_glossy  = float3(0.0f);
  d.pdf          = 0.0f;
}
void absorbSample(BsdfSampleData& d)
{
  d.bsdf_over_pdf = float3(0.0f);
  d.pdf           = 0.0f;
  d.event_type    = BSDF_EVENT_ABSORB;
}

// --- lobes -----------------------------------------------------------------------------------------------------------
void brdf_diffuse_eval(BsdfEvaluateData& d, const PbrMaterial& mat, float3 tint)
{
  if(dot(d.k2, mat.Ng) <= 0.0f)
    return absorbEval(d);
  d.pdf          = std::fmax(0.0f, dot(d.k2, mat.N) * K_1_OVER_PI);
  d.bsdf_diffuse = tint * d.pdf;
}
void brdf_diffuse_sample(BsdfSampleData& d, const PbrMaterial& mat, float3 tint)
{
  float3 l        = cosineSampleHemisphere(d.xi.x, d.xi.y);
  d.k2            = normalize(mat.T * l.x + mat.B * l.y + mat.N * l.z);
  d.pdf           = dot(d.k2, mat.N) * K_1_OVER_PI;
  d.bsdf_over_pdf = tint;
  d.event_type    = (0.0f < dot(d.k2, mat.Ng)) ? BSDF_EVENT_DIFFUSE_REFLECTION : BSDF_EVENT_ABSORB;
}
// KHR_materials_diffuse_transmission: Lambert lobe on the far side of the surface
void btdf_diffuse_eval(BsdfEvaluateData& d, const PbrMaterial& mat, float3 tint)
{
  if(dot(d.k2, mat.Ng) >= 0.0f)
    return absorbEval(d);
  d.pdf          = std::fmax(0.0f, -dot(d.k2, mat.N) * K_1_OVER_PI);
  d.bsdf_diffuse = tint * d.pdf;
}
void btdf_diffuse_sample(BsdfSampleData& d, const PbrMaterial& mat, float3 tint)
{
  float3 l        = cosineSampleHemisphere(d.xi.x, d.xi.y);
  d.k2            = normalize(mat.T * l.x + mat.B * l.y - mat.N * l.z);
  d.pdf           = -dot(d.k2, mat.N) * K_1_OVER_PI;
  d.bsdf_over_pdf = tint;
  d.event_type    = (dot(d.k2, mat.Ng) < 0.0f) ? BSDF_EVENT_DIFFUSE_TRANSMISSION : BSDF_EVENT_ABSORB;
}

float3 ggxTint(const PbrMaterial& mat, int lobe, float3 tint, float k1h)
{
  if(lobe == LOBE_METAL_REFLECTION)  // glTF 2.0 Appendix B: metal F = baseColor + (1 - baseColor)(1 - VdotH)^5
    tint = tint + (float3(1.0f) - tint) * std::pow(1.0f - std::fabs(k1h), 5.0f);
  if(mat.iridescence > 0.0f && (lobe == LOBE_SPECULAR_REFLECTION || lobe == LOBE_METAL_REFLECTION))
  {
    float3 factor = thin_film_factor(mat.iridescenceThickness, mat.iridescenceIor, mat.ior2, mat.ior1, k1h);
    if(lobe == LOBE_SPECULAR_REFLECTION)
      tint *= lerp(float3(1.0f), factor, mat.iridescence);
    else
      tint = mix_rgb(tint, mat.specularColor, factor * mat.iridescence);
  }
  return tint;
}
void brdf_ggx_smith_eval(BsdfEvaluateData& d, const PbrMaterial& mat, int lobe, float3 tint)
{
  float nk1 = std::fabs(dot(d.k1, mat.N)), nk2 = std::fabs(dot(d.k2, mat.N));
  if(dot(d.k2, mat.Ng) <= 0.0f)
    return absorbEval(d);
  float3 h   = normalize(d.k1 + d.k2);
  float  nh = dot(mat.N, h), k1h = dot(d.k1, h), k2h = dot(d.k2, h);
  if(nk1 <= 0.0f || nh <= 0.0f || k1h < 0.0f || k2h < 0.0f)
    return absorbEval(d);
  float3 h0 = float3(dot(mat.T, h), dot(mat.B, h), nh);
  d.pdf     = hvd_ggx_eval(float2(1.0f / mat.roughness.x, 1.0f / mat.roughness.y), h0);
  float G1, G2;
  float G12 = ggx_smith_shadow_mask(G1, G2, float3(dot(mat.T, d.k1), dot(mat.B, d.k1), nk1),
                                    float3(dot(mat.T, d.k2), dot(mat.B, d.k2), nk2), mat.roughness);
  d.pdf *= 0.25f / (nk1 * nh);
  float3 bsdf = float3(G12 * d.pdf);
  d.pdf *= G1;
  d.bsdf_glossy = bsdf * ggxTint(mat, lobe, tint, k1h);
}
void brdf_ggx_smith_sample(BsdfSampleData& d, const PbrMaterial& mat, int lobe, float3 tint)
{
  float nk1 = dot(d.k1, mat.N);
  if(nk1 <= 0.0f)
    return absorbSample(d);
  float3 k10 = float3(dot(d.k1, mat.T), dot(d.k1, mat.B), nk1);
  float3 h0  = hvd_ggx_sample_vndf(k10, mat.roughness, float2(d.xi.x, d.xi.y));
  if(std::fabs(h0.z) == 0.0f)
    return absorbSample(d);
  float3 h  = mat.T * h0.x + mat.B * h0.y + mat.N * h0.z;
  float  kh = dot(d.k1, h);
  if(kh <= 0.0f)
    return absorbSample(d);
  d.k2            = h * (2.0f * kh) - d.k1;
  d.bsdf_over_pdf = float3(1.0f);
  d.event_type    = BSDF_EVENT_GLOSSY_REFLECTION;
  if(dot(d.k2, mat.Ng) <= 0.0f)
    return absorbSample(d);
  float nk2 = std::fabs(dot(d.k2, mat.N));
  float G1, G2;
  float G12 = ggx_smith_shadow_mask(G1, G2, k10, float3(dot(d.k2, mat.T), dot(d.k2, mat.B), nk2), mat.roughness);
  if(G12 <= 0.0f)
    return absorbSample(d);
  d.bsdf_over_pdf *= G12 / G1;
  d.pdf = hvd_ggx_eval(float2(1.0f / mat.roughness.x, 1.0f / mat.roughness.y), h0) * G1;
  d.pdf *= 0.25f / (nk1 * h0.z);
  d.bsdf_over_pdf *= ggxTint(mat, lobe, tint, kh);
}
void btdf_ggx_smith_eval(BsdfEvaluateData& d, const PbrMaterial& mat, float3 tint)
{
  bool   thin = (mat.thickness == 0.0f);
  float2 ior(mat.ior1, mat.ior2);
  float  nk1 = std::fabs(dot(d.k1, mat.N)), nk2 = std::fabs(dot(d.k2, mat.N));
  bool   backside = (dot(d.k2, mat.Ng) < 0.0f);
  float3 h        = compute_half_vector(d.k1, d.k2, mat.N, ior, nk2, backside, thin);
  float  nh = dot(mat.N, h), k1h = dot(d.k1, h), k2h = dot(d.k2, h) * (backside ? -1.0f : 1.0f);
  if(nk1 <= 0.0f || nh <= 0.0f || k1h < 0.0f || k2h < 0.0f)
    return absorbEval(d);
  float fr;
  if(!backside)
  {
    if(!isTIR(ior, k1h))
      return absorbEval(d);
    fr = 1.0f;
  }
  else
    fr = 0.0f;
  float3 h0 = float3(dot(mat.T, h), dot(mat.B, h), nh);
  d.pdf     = hvd_ggx_eval(float2(1.0f / mat.roughness.x, 1.0f / mat.roughness.y), h0);
  float G1, G2;
  float G12 = ggx_smith_shadow_mask(G1, G2, float3(dot(mat.T, d.k1), dot(mat.B, d.k1), nk1),
                                    float3(dot(mat.T, d.k2), dot(mat.B, d.k2), nk2), mat.roughness);
  if(!thin && backside)
  {
    float tmp = k1h * ior.x - k2h * ior.y;
    d.pdf *= k1h * k2h * ior.y * ior.y / (nk1 * nh * tmp * tmp);  // Walter et al. 2007 eq. 17 Jacobian
  }
  else
    d.pdf *= 0.25f / (nk1 * nh);
  float  prob = backside ? 1.0f - fr : fr;
  float3 bsdf = float3(prob * G12 * d.pdf);
  d.pdf *= prob * G1;
  d.bsdf_glossy = bsdf * tint;
}
void btdf_ggx_smith_sample(BsdfSampleData& d, const PbrMaterial& mat, float3 tint)
{
  bool   thin = (mat.thickness == 0.0f);
  float2 ior(mat.ior1, mat.ior2);
  float  nk1 = std::fabs(dot(d.k1, mat.N));
  float3 k10 = float3(dot(d.k1, mat.T), dot(d.k1, mat.B), nk1);
  float3 h0  = hvd_ggx_sample_vndf(k10, mat.roughness, float2(d.xi.x, d.xi.y));
  if(std::fabs(h0.z) == 0.0f)
    return absorbSample(d);
  float3 h  = mat.T * h0.x + mat.B * h0.y + mat.N * h0.z;
  float  kh = dot(d.k1, h);
  if(kh <= 0.0f)
    return absorbSample(d);
  bool tir = false;
  if(thin)
  {
    d.k2 = h * (2.0f * kh) - d.k1;
    d.k2 = normalize(d.k2 - mat.N * (2.0f * dot(d.k2, mat.N)));
  }
  else
    d.k2 = refractDir(d.k1, h, ior.x / ior.y, kh, tir);
  d.bsdf_over_pdf = float3(1.0f);
  d.event_type    = tir ? BSDF_EVENT_GLOSSY_REFLECTION : BSDF_EVENT_GLOSSY_TRANSMISSION;
  float gnk2      = dot(d.k2, mat.Ng) * ((d.event_type == BSDF_EVENT_GLOSSY_REFLECTION) ? 1.0f : -1.0f);
  if(gnk2 <= 0.0f)
    return absorbSample(d);
  float nk2 = std::fabs(dot(d.k2, mat.N)), k2h = std::fabs(dot(d.k2, h));
  float G1, G2;
  float G12 = ggx_smith_shadow_mask(G1, G2, k10, float3(dot(d.k2, mat.T), dot(d.k2, mat.B), nk2), mat.roughness);
  if(G12 <= 0.0f)
    return absorbSample(d);
  d.bsdf_over_pdf *= G12 / G1;
  d.pdf = hvd_ggx_eval(float2(1.0f / mat.roughness.x, 1.0f / mat.roughness.y), h0) * G1;
  if(!thin && d.event_type == BSDF_EVENT_GLOSSY_TRANSMISSION)
  {
    float tmp = kh * ior.x - k2h * ior.y;
    if(tmp != 0.0f)
      d.pdf *= kh * k2h * ior.y * ior.y / (nk1 * h0.z * tmp * tmp);
  }
  else
    d.pdf *= 0.25f / (nk1 * h0.z);
  d.bsdf_over_pdf *= tint;
}
void brdf_sheen_eval(BsdfEvaluateData& d, const PbrMaterial& mat)
{
  if(dot(d.k2, mat.Ng) <= 0.0f)
    return absorbEval(d);
  float  nk1 = std::fabs(dot(d.k1, mat.N)), nk2 = std::fabs(dot(d.k2, mat.N));
  float3 h   = normalize(d.k1 + d.k2);
  float  nh = dot(mat.N, h), k1h = dot(d.k1, h), k2h = dot(d.k2, h);
  if(nk1 <= 0.0f || nh <= 0.0f || k1h < 0.0f || k2h < 0.0f)
    return absorbEval(d);
  float invRoughness = 1.0f / (mat.sheenRoughness * mat.sheenRoughness);
  d.pdf              = hvd_sheen_eval(invRoughness, nh);
  float G1, G2;
  float G12 = vcavities_shadow_mask(G1, G2, nh, float3(dot(mat.T, d.k1), dot(mat.B, d.k1), nk1), k1h,
                                    float3(dot(mat.T, d.k2), dot(mat.B, d.k2), nk2), k2h);
  d.pdf *= 0.25f / (nk1 * nh);
  float3 bsdf = float3(G12 * d.pdf);
  d.pdf *= G1;
  d.bsdf_glossy = bsdf * mat.sheenColor;
}
void brdf_sheen_sample(BsdfSampleData& d, const PbrMaterial& mat, float xiFlip)
{
  float nk1 = dot(d.k1, mat.N);
  if(nk1 <= 0.0f)
    return absorbSample(d);
  float3 k10          = float3(dot(d.k1, mat.T), dot(d.k1, mat.B), nk1);
  float  invRoughness = 1.0f / (mat.sheenRoughness * mat.sheenRoughness);
  float3 h0           = flipH(hvd_sheen_sample(float2(d.xi.x, d.xi.y), invRoughness), k10, xiFlip);
  if(std::fabs(h0.z) == 0.0f)
    return absorbSample(d);
  float3 h   = mat.T * h0.x + mat.B * h0.y + mat.N * h0.z;
  float  k1h = dot(d.k1, h);
  if(k1h <= 0.0f)
    return absorbSample(d);
  d.k2            = h * (2.0f * k1h) - d.k1;
  d.bsdf_over_pdf = float3(1.0f);
  d.event_type    = BSDF_EVENT_GLOSSY_REFLECTION;
  if(dot(d.k2, mat.Ng) <= 0.0f)
    return absorbSample(d);
  float nk2 = std::fabs(dot(d.k2, mat.N)), k2h = std::fabs(dot(d.k2, h));
  float G1, G2;
  float G12 = vcavities_shadow_mask(G1, G2, h0.z, k10, k1h, float3(dot(d.k2, mat.T), dot(d.k2, mat.B), nk2), k2h);
  if(G12 <= 0.0f)
    return absorbSample(d);
  d.bsdf_over_pdf *= G12 / G1;
  d.pdf = hvd_sheen_eval(invRoughness, h0.z) * G1;
  d.pdf *= 0.25f / (nk1 * h0.z);
  d.bsdf_over_pdf *= mat.sheenColor;
}

// nvshaders bsdfEvaluate (call site: gltf_pathtrace.slang:333-349): stochastic single-lobe evaluation, lobe picked by xi.z
void bsdfEvaluate(BsdfEvaluateData& d, PbrMaterial mat)
{
  float    VdotN = dot(d.k1, mat.N);
  LobePick pick  = findLobe(mat, VdotN, d.xi.z);
  float    u     = pick.u;
  absorbEval(d);
  switch(pick.lobe)
  {
    case LOBE_DIFFUSE_REFLECTION:
      if(mat.diffuseTransmissionFactor > 0.0f && splitRandom(u, mat.diffuseTransmissionFactor))
        btdf_diffuse_eval(d, mat, mat.diffuseTransmissionColor);
      else
        brdf_diffuse_eval(d, mat, pick.tint);
      break;
    case LOBE_SPECULAR_REFLECTION:
      d.k1 = retroView(mat, d.k1, mat.N, u);
      brdf_ggx_smith_eval(d, mat, LOBE_SPECULAR_REFLECTION, mat.specularColor);
      break;
    case LOBE_SPECULAR_TRANSMISSION: {
      float3 mask = applyDispersion(mat, u);
      btdf_ggx_smith_eval(d, mat, pick.tint * mask);
      break;
    }
    case LOBE_METAL_REFLECTION:
      d.k1 = retroView(mat, d.k1, mat.N, u);
      brdf_ggx_smith_eval(d, mat, LOBE_METAL_REFLECTION, mat.baseColor);
      break;
    case LOBE_CLEARCOAT_REFLECTION:
      mat.roughness   = float2(mat.clearcoatRoughness * mat.clearcoatRoughness);
      mat.N           = mat.Nc;
      mat.iridescence = 0.0f;
      d.k1            = retroView(mat, d.k1, mat.N, u);
      brdf_ggx_smith_eval(d, mat, LOBE_CLEARCOAT_REFLECTION, float3(1.0f));
      break;
    case LOBE_SHEEN_REFLECTION:
      d.k1 = retroView(mat, d.k1, mat.N, u);
      brdf_sheen_eval(d, mat);
      break;
  }
  d.bsdf_diffuse *= mat.occlusion;
}
// nvshaders bsdfSample (call site: gltf_pathtrace.slang:359-368)
void bsdfSample(BsdfSampleData& d, PbrMaterial mat)
{
  float    VdotN = dot(d.k1, mat.N);
  LobePick pick  = findLobe(mat, VdotN, d.xi.z);
  float    u     = pick.u;
  absorbSample(d);
  switch(pick.lobe)
  {
    case LOBE_DIFFUSE_REFLECTION:
      if(mat.diffuseTransmissionFactor > 0.0f && splitRandom(u, mat.diffuseTransmissionFactor))
        btdf_diffuse_sample(d, mat, mat.diffuseTransmissionColor);
      else
        brdf_diffuse_sample(d, mat, pick.tint);
      break;
    case LOBE_SPECULAR_REFLECTION:
      d.k1 = retroView(mat, d.k1, mat.N, u);
      brdf_ggx_smith_sample(d, mat, LOBE_SPECULAR_REFLECTION, mat.specularColor);
      break;
    case LOBE_SPECULAR_TRANSMISSION: {
      float3 mask = applyDispersion(mat, u);
      btdf_ggx_smith_sample(d, mat, pick.tint * mask);
      break;
    }
    case LOBE_METAL_REFLECTION:
      d.k1 = retroView(mat, d.k1, mat.N, u);
      brdf_ggx_smith_sample(d, mat, LOBE_METAL_REFLECTION, mat.baseColor);
      break;
    case LOBE_CLEARCOAT_REFLECTION:
      mat.roughness   = float2(mat.clearcoatRoughness * mat.clearcoatRoughness);
      mat.N           = mat.Nc;
      mat.B           = normalize(cross(mat.N, mat.T));
      mat.T           = cross(mat.B, mat.N);
      mat.iridescence = 0.0f;
      d.k1            = retroView(mat, d.k1, mat.N, u);
      brdf_ggx_smith_sample(d, mat, LOBE_CLEARCOAT_REFLECTION, float3(1.0f));
      break;
    case LOBE_SHEEN_REFLECTION:
      d.k1 = retroView(mat, d.k1, mat.N, u);
      brdf_sheen_sample(d, mat, u);
      break;
  }
  // NaN/Inf guard: a degenerate sample ends the path instead of poisoning the accumulator
  if(!(d.pdf == d.pdf) || !(d.bsdf_over_pdf.x == d.bsdf_over_pdf.x) || !(d.bsdf_over_pdf.y == d.bsdf_over_pdf.y)
     || !(d.bsdf_over_pdf.z == d.bsdf_over_pdf.z) || !(d.k2.x == d.k2.x) || !(d.k2.y == d.k2.y) || !(d.k2.z == d.k2.z))
    absorbSample(d);
}
// nvshaders bsdfSampleSimple (call site: pathtrace_functions.h.slang:537-551, shadow catcher): diffuse + one GGX lobe
void bsdfSampleSimple(BsdfSampleData& d, const PbrMaterial& mat)
{
  float3 tint  = mat.baseColor;
  float  VdotN = dot(d.k1, mat.N);
  float  F     = lerp(schlickFresnelIor(mat.ior2 / mat.ior1, std::fabs(VdotN)), 1.0f, mat.metallic);
  absorbSample(d);
  if(d.xi.z < F)
    brdf_ggx_smith_sample(d, mat, mat.metallic > 0.5f ? LOBE_METAL_REFLECTION : LOBE_SPECULAR_REFLECTION,
                          mat.metallic > 0.5f ? tint : float3(1.0f));
  else
    brdf_diffuse_sample(d, mat, tint * (1.0f - mat.metallic));
}

// nvshaders Henyey-Greenstein (call sites: pathtrace_functions.h.slang:625-627,660)
float henyeyGreensteinPdf(float cosTheta, float g)
{
  float denom = 1.0f + g * g - 2.0f * g * cosTheta;
  return (1.0f - g * g) / (4.0f * K_PI * denom * std::sqrt(std::fmax(denom, 1e-12f)));
}
float3 sampleHenyeyGreenstein(float2 xi, float g, float3 wi)  // wi = propagation direction; returns new propagation direction
{
  float cosTheta;
  if(std::fabs(g) < 1e-3f)
    cosTheta = 1.0f - 2.0f * xi.x;
  else
  {
    float s  = (1.0f - g * g) / (1.0f - g + 2.0f * g * xi.x);
    cosTheta = (1.0f + g * g - s * s) / (2.0f * g);
  }
  cosTheta       = clampf(cosTheta, -1.0f, 1.0f);
  float  sinTheta = std::sqrt(std::fmax(0.0f, 1.0f - cosTheta * cosTheta));
  float  phi      = K_TWO_PI * xi.y;
  float4 t        = makeFastTangent(wi);
  float3 T = t.xyz(), B = cross(wi, T);
  return normalize(T * (sinTheta * std::cos(phi)) + B * (sinTheta * std::sin(phi)) + wi * cosTheta);
}

//----------------------------------------------------------------------------------------------------------------------
// nvshaders/light_contrib.h.slang singleLightContribution (call site: pathtrace_functions.h.slang:406-412).
// KHR_lights_punctual attenuation; a radius / angular size > 0 turns the delta light into a uniformly sampled
// cone with a finite pdf.  `intensity` is radiance-over-pdf at the surface (the caller divides by selection pdfs only).
struct LightContrib
{
  float3 incidentVector{0, 0, 0};  // from the light toward the surface
  float  distance = INFINITE_F;
  float3 intensity{0, 0, 0};
  float  pdf = DIRAC;
};
// Uniform direction inside a cone given 1 - cos(halfAngle); written so that tiny cones (the sun: 1 - cos ~ 1e-5) keep
// full relative precision: sin^2 = (1 - cos)(1 + cos) = s (2 - s).
float3 sampleCone(float2 xi, float oneMinusCosMax, float3 axis)
{
  float  s        = xi.x * oneMinusCosMax;
  float  cosTheta = 1.0f - s;
  float  sinTheta = std::sqrt(std::fmax(0.0f, s * (2.0f - s)));
  float  phi      = K_TWO_PI * xi.y;
  float4 t        = makeFastTangent(axis);
  float3 T = t.xyz(), B = cross(axis, T);
  return normalize(T * (sinTheta * std::cos(phi)) + B * (sinTheta * std::sin(phi)) + axis * cosTheta);
}
LightContrib singleLightContribution(const MiGltfLight& light, float3 pos, float3 normal, float2 xi)
{
  LightContrib c;
  float3       color = float3(light.color) * light.intensity;
  if(light.type == MI_LIGHT_DIRECTIONAL)
  {
    float3 toLight = -normalize(float3(light.direction));
    float  halfAng = 0.5f * light.angularSizeOrInvRange;
    if(halfAng > 0.0f)
    {
      float omc = 2.0f * sqr(std::sin(0.5f * halfAng));  // 1 - cos(halfAng)
      toLight   = sampleCone(xi, omc, toLight);
      c.pdf     = 1.0f / (K_TWO_PI * omc);
    }
    c.incidentVector = -toLight;
    c.distance       = INFINITE_F;
    c.intensity      = color;
  }
  else
  {
    float3 toLight = float3(light.position) - pos;
    float  d       = length(toLight);
    if(d <= 0.0f)
      return c;
    float3 L     = toLight / d;
    float3 axisL = L;
    float  dist  = d;
    if(light.radius > 0.0f)
    {
      float sinMax = std::fmin(light.radius / d, 1.0f);
      float cosMax = std::sqrt(std::fmax(0.0f, 1.0f - sinMax * sinMax));
      float omc    = std::fmax(sinMax * sinMax / (1.0f + cosMax), 1e-12f);  // 1 - cosMax
      L            = sampleCone(xi, omc, axisL);
      c.pdf        = 1.0f / (K_TWO_PI * omc);
      // distance to the sphere surface along L
      float b    = dot(L, toLight);
      float disc = b * b - (d * d - light.radius * light.radius);
      dist       = disc > 0.0f ? std::fmax(b - std::sqrt(disc), 0.0f) : b;
    }
    float atten = 1.0f / (d * d);
    if(light.angularSizeOrInvRange > 0.0f)  // KHR_lights_punctual range window
    {
      float r4 = sqr(sqr(d * light.angularSizeOrInvRange));
      atten *= sqr(clampf(1.0f - r4, 0.0f, 1.0f));
    }
    if(light.type == MI_LIGHT_SPOT)
    {
      float cosOuter = std::cos(light.outerAngle), cosInner = std::cos(light.innerAngle);
      float scale  = 1.0f / std::fmax(0.001f, cosInner - cosOuter);
      float offset = -cosOuter * scale;
      float cd     = dot(normalize(float3(light.direction)), -axisL);
      float a      = clampf(cd * scale + offset, 0.0f, 1.0f);
      atten *= a * a;
    }
    c.incidentVector = -L;
    c.distance       = dist;
    c.intensity      = color * atten;
  }
  (void)normal;
  return c;
}

//----------------------------------------------------------------------------------------------------------------------
// nvshaders/sky_functions.h.slang physical sun & sky (call sites: pathtrace_functions.h.slang:422-429,470-471).
// Restated as: Preetham/Shirley/Smits 1999 Perez luminance + chromaticity, a limb-darkened sun disc with glow,
// ground colour below the horizon; importance sampling = 50/50 mixture of the sun cone and the uniform sphere.
struct SkySamplingResult
{
  float3 direction;
  float  pdf;
  float3 radiance;
};
float perez(float cosTheta, float gamma, float cosGamma, const float c[5])
{
  return (1.0f + c[0] * std::exp(c[1] / std::fmax(cosTheta, 0.01f))) * (1.0f + c[2] * std::exp(c[3] * gamma) + c[4] * cosGamma * cosGamma);
}
float3 skyUp(const MiSkyPhysicalParameters& s) { return s.yIsUp ? float3(0, 1, 0) : float3(0, 0, 1); }
float  skySunAngularRadius(const MiSkyPhysicalParameters& s) { return 0.00465f * std::fmax(s.sunDiskScale, 0.0f) + 1e-6f; }
float  skySunConeAngle(const MiSkyPhysicalParameters& s) { return std::fmin(skySunAngularRadius(s) * 4.0f, 1.5f); }
float  skySunConeOneMinusCos(const MiSkyPhysicalParameters& s) { return 2.0f * sqr(std::sin(0.5f * skySunConeAngle(s))); }
// angle between two unit vectors, well conditioned near 0 (acos is not)
float  angleBetween(float3 a, float3 b) { return std::atan2(length(cross(a, b)), dot(a, b)); }
float3 evalPhysicalSky(const MiSkyPhysicalParameters& s, float3 dir)
{
  if(s.multiplier <= 0.0f)
    return float3(0.0f);
  float3 up     = skyUp(s);
  float3 sunDir = normalize(float3(s.sunDirection));
  float3 scale  = float3(s.rgbUnitConversion) * s.multiplier;
  float  T      = 2.0f + std::fmax(s.haze, 0.0f);
  float  cosS   = clampf(dot(sunDir, up), -1.0f, 1.0f);
  float  thetaS = std::acos(cosS);
  float  cosT   = dot(dir, up) - s.horizonHeight * 0.1f;
  // sun colour/irradiance: 100 klux white sun through a simple optical-depth model
  float  airmass = 1.0f / (std::fmax(cosS, 0.0f) + 0.15f * std::pow(std::fmax(93.885f - thetaS * 57.29578f, 1.0f), -1.253f));
  float3 tau     = float3(0.06f, 0.11f, 0.22f) * (T * 0.5f);
  float3 sunE    = (cosS > -0.05f) ? exp3(-tau * airmass) * 100000.0f : float3(0.0f);
  float3 result;
  if(cosT <= 0.0f)
  {
    // ground: Lambertian, lit by the sun and a flat sky term
    float3 skyE = float3(0.2f, 0.25f, 0.35f) * 20000.0f * std::fmax(cosS, 0.0f) + float3(s.nightColor) * 80000.0f;
    result      = float3(s.groundColor) * (sunE * std::fmax(cosS, 0.0f) + skyE) * K_1_OVER_PI;
    float blur  = std::fmax(s.horizonBlur * 0.1f, 1e-4f);
    float t     = saturate(-cosT / blur);
    if(t < 1.0f)  // blend with the horizon sky colour
    {
      float3 h = float3(0.75f, 0.8f, 0.9f) * 8000.0f * std::fmax(cosS, 0.05f);
      result   = lerp(h, result, t);
    }
  }
  else
  {
    float cosGamma = clampf(dot(dir, sunDir), -1.0f, 1.0f);
    float gamma    = angleBetween(dir, sunDir);
    const float cY[5] = {0.1787f * T - 1.4630f, -0.3554f * T + 0.4275f, -0.0227f * T + 5.3251f, 0.1206f * T - 2.5771f, -0.0670f * T + 0.3703f};
    const float cx[5] = {-0.0193f * T - 0.2592f, -0.0665f * T + 0.0008f, -0.0004f * T + 0.2125f, -0.0641f * T - 0.8989f, -0.0033f * T + 0.0452f};
    const float cy[5] = {-0.0167f * T - 0.2608f, -0.0950f * T + 0.0092f, -0.0079f * T + 0.2102f, -0.0441f * T - 1.6537f, -0.0109f * T + 0.0529f};
    float tS  = std::fmin(thetaS, 1.5f);
    float chi = (4.0f / 9.0f - T / 120.0f) * (K_PI - 2.0f * tS);
    float Yz  = std::fmax((4.0453f * T - 4.9710f) * std::tan(chi) - 0.2155f * T + 2.4192f, 0.0f);  // kcd/m^2
    float t2 = tS * tS, t3 = t2 * tS, T2 = T * T;
    float xz = (0.00166f * t3 - 0.00375f * t2 + 0.00209f * tS) * T2 + (-0.02903f * t3 + 0.06377f * t2 - 0.03202f * tS + 0.00394f) * T
               + (0.11693f * t3 - 0.21196f * t2 + 0.06052f * tS + 0.25886f);
    float yz = (0.00275f * t3 - 0.00610f * t2 + 0.00317f * tS) * T2 + (-0.04214f * t3 + 0.08970f * t2 - 0.04153f * tS + 0.00516f) * T
               + (0.15346f * t3 - 0.26756f * t2 + 0.06670f * tS + 0.26688f);
    float cosTs = std::cos(tS);
    float Y  = Yz * perez(cosT, gamma, cosGamma, cY) / perez(1.0f, tS, cosTs, cY);
    float x  = xz * perez(cosT, gamma, cosGamma, cx) / perez(1.0f, tS, cosTs, cx);
    float y  = yz * perez(cosT, gamma, cosGamma, cy) / perez(1.0f, tS, cosTs, cy);
    Y        = std::fmax(Y, 0.0f) * 1000.0f * saturate((cosS + 0.05f) * 10.0f);  // cd/m^2, fades out at night
    float X = (y > 1e-4f) ? x / y * Y : 0.0f, Z = (y > 1e-4f) ? (1.0f - x - y) / y * Y : 0.0f;
    result  = float3(3.2406f * X - 1.5372f * Y - 0.4986f * Z, -0.9689f * X + 1.8758f * Y + 0.0415f * Z, 0.0557f * X - 0.2040f * Y + 1.0570f * Z);
    result  = max3(result, float3(0.0f)) + float3(s.nightColor) * 80000.0f;
    // sun disc + glow
    float r = skySunAngularRadius(s);
    if(s.sunDiskIntensity > 0.0f && gamma < r * 4.0f)
    {
      float  omega = K_TWO_PI * 2.0f * sqr(std::sin(0.5f * r));
      float3 Lsun  = sunE / omega * s.sunDiskIntensity;
      if(gamma < r)
      {
        float mu = std::sqrt(std::fmax(0.0f, 1.0f - sqr(gamma / r)));
        result += Lsun * (0.4f + 0.6f * mu);  // limb darkening
      }
      else
        result += Lsun * (0.002f * s.sunGlowIntensity) * sqr(1.0f - (gamma - r) / (3.0f * r));
    }
  }
  // red/blue shift and saturation (artistic controls)
  float rb = s.redblueshift;
  result   = result * float3(1.0f + rb, 1.0f, 1.0f - rb);
  float lum = dot(result, float3(0.2126f, 0.7152f, 0.0722f));
  result    = max3(lerp(float3(lum), result, s.saturation), float3(0.0f));
  return result * scale;
}
float skySunWeight(const MiSkyPhysicalParameters& s) { return (s.sunDiskIntensity > 0.0f && s.multiplier > 0.0f) ? 0.5f : 0.0f; }
float samplePhysicalSkyPDF(const MiSkyPhysicalParameters& s, float3 dir)
{
  float wSun   = skySunWeight(s);
  float pdf    = (1.0f - wSun) * (0.25f * K_1_OVER_PI);
  if(wSun > 0.0f && angleBetween(dir, normalize(float3(s.sunDirection))) <= skySunConeAngle(s))
    pdf += wSun / (K_TWO_PI * skySunConeOneMinusCos(s));
  return pdf;
}
SkySamplingResult samplePhysicalSky(const MiSkyPhysicalParameters& s, float2 xi)
{
  SkySamplingResult r;
  float             wSun = skySunWeight(s);
  float             u    = xi.x;
  if(splitRandom(u, wSun))
    r.direction = sampleCone(float2(u, xi.y), skySunConeOneMinusCos(s), normalize(float3(s.sunDirection)));
  else
  {
    float z   = 1.0f - 2.0f * u;
    float rr  = std::sqrt(std::fmax(0.0f, 1.0f - z * z));
    float phi = K_TWO_PI * xi.y;
    r.direction = float3(rr * std::cos(phi), z, rr * std::sin(phi));
  }
  r.pdf      = samplePhysicalSkyPDF(s, r.direction);
  r.radiance = evalPhysicalSky(s, r.direction);
  return r;
}

//======================================================================================================================
// Scene access, textures
//======================================================================================================================
struct Scene
{
  const MiPtSceneDesc*   desc = nullptr;
  const MiPtEnvironment* env  = nullptr;
  MiSceneFrameInfo       frameInfo{};
  MiSkyPhysicalParameters sky{};
  MiPathtraceParams      pc{};
  float                  srgbLut[256];
};

float4 fetchTexel(const Scene& sc, const MiPtTexture& t, int level, int x, int y)
{
  int            w = std::max(1, t.width >> level);
  const uint8_t* p = t.levels[level] + (size_t(y) * size_t(w) + size_t(x)) * 4;
  if(t.srgb)
    return float4(sc.srgbLut[p[0]], sc.srgbLut[p[1]], sc.srgbLut[p[2]], float(p[3]) * (1.0f / 255.0f));
  return float4(float(p[0]) * (1.0f / 255.0f), float(p[1]) * (1.0f / 255.0f), float(p[2]) * (1.0f / 255.0f), float(p[3]) * (1.0f / 255.0f));
}
int wrapCoord(int i, int n, int mode)
{
  if(mode == MI_WRAP_CLAMP_TO_EDGE)
    return std::min(std::max(i, 0), n - 1);
  if(mode == MI_WRAP_MIRRORED_REPEAT)
  {
    int period = 2 * n;
    int m      = i % period;
    if(m < 0)
      m += period;
    return m < n ? m : period - 1 - m;
  }
  int m = i % n;
  return m < 0 ? m + n : m;
}
// One mip level, Vulkan filtering rules: unnormalised coords u*w, linear taps at floor(u*w - 0.5) and +1
float4 sampleLevel(const Scene& sc, const MiPtTexture& t, float2 uv, int level, int filter)
{
  int   w = std::max(1, t.width >> level), h = std::max(1, t.height >> level);
  float fx = uv.x * float(w), fy = uv.y * float(h);
  if(filter == MI_FILTER_NEAREST)
    return fetchTexel(sc, t, level, wrapCoord(int(std::floor(fx)), w, t.wrapS), wrapCoord(int(std::floor(fy)), h, t.wrapT));
  fx -= 0.5f;
  fy -= 0.5f;
  float flx = std::floor(fx), fly = std::floor(fy);
  float tx = fx - flx, ty = fy - fly;
  int   x0 = wrapCoord(int(flx), w, t.wrapS), x1 = wrapCoord(int(flx) + 1, w, t.wrapS);
  int   y0 = wrapCoord(int(fly), h, t.wrapT), y1 = wrapCoord(int(fly) + 1, h, t.wrapT);
  float4 a = fetchTexel(sc, t, level, x0, y0), b = fetchTexel(sc, t, level, x1, y0);
  float4 c = fetchTexel(sc, t, level, x0, y1), d = fetchTexel(sc, t, level, x1, y1);
  return (a * (1.0f - tx) + b * tx) * (1.0f - ty) + (c * (1.0f - tx) + d * tx) * ty;
}
// SampleLevel(uv, 0) and SampleGrad(uv, ddx, ddy) (Vulkan 1.3 §16.5: isotropic LOD = log2 of the larger scaled gradient)
float4 sampleTexture(const Scene& sc, int texIndex, float2 uv, bool useGrad, float2 ddx, float2 ddy)
{
  if(texIndex < 0 || texIndex >= sc.desc->numTextures)
    return float4(1.0f);
  const MiPtTexture& t   = sc.desc->textures[texIndex];
  float              lod = 0.0f;
  if(useGrad)
  {
    float rx = std::sqrt(sqr(ddx.x * float(t.width)) + sqr(ddx.y * float(t.height)));
    float ry = std::sqrt(sqr(ddy.x * float(t.width)) + sqr(ddy.y * float(t.height)));
    float rho = std::fmax(rx, ry);
    lod       = rho > 0.0f ? std::log2(rho) : -126.0f;
  }
  if(lod <= 0.0f)
    return sampleLevel(sc, t, uv, 0, t.magFilter);
  float maxLevel = float(t.numLevels - 1);
  lod            = std::fmin(lod, maxLevel);
  if(t.mipmapMode == MI_FILTER_NEAREST)
    return sampleLevel(sc, t, uv, std::min(int(std::floor(lod + 0.5f)), t.numLevels - 1), t.minFilter);
  int   l0 = int(std::floor(lod)), l1 = std::min(l0 + 1, t.numLevels - 1);
  float f  = lod - float(l0);
  float4 a = sampleLevel(sc, t, uv, l0, t.minFilter);
  if(f == 0.0f || l1 == l0)
    return a;
  float4 b = sampleLevel(sc, t, uv, l1, t.minFilter);
  return a * (1.0f - f) + b * f;
}
// texturesHdr[HDR_IMAGE_INDEX].SampleLevel(uv, 0): bilinear, repeat in both directions
float4 sampleHdr(const Scene& sc, float2 uv)
{
  const MiPtEnvironment* e = sc.env;
  if(!e || !e->rgba)
    return float4(0.0f);
  int   w = e->width, h = e->height;
  float fx = uv.x * float(w) - 0.5f, fy = uv.y * float(h) - 0.5f;
  float flx = std::floor(fx), fly = std::floor(fy);
  float tx = fx - flx, ty = fy - fly;
  int   x0 = wrapCoord(int(flx), w, MI_WRAP_REPEAT), x1 = wrapCoord(int(flx) + 1, w, MI_WRAP_REPEAT);
  int   y0 = wrapCoord(int(fly), h, MI_WRAP_REPEAT), y1 = wrapCoord(int(fly) + 1, h, MI_WRAP_REPEAT);
  auto  px = [&](int x, int y) { return float4(e->rgba + (size_t(y) * size_t(w) + size_t(x)) * 4); };
  return (px(x0, y0) * (1.0f - tx) + px(x1, y0) * tx) * (1.0f - ty) + (px(x0, y1) * (1.0f - tx) + px(x1, y1) * tx) * ty;
}
// nvshaders/hdr_env_sampling.h.slang environmentSample (call site: pathtrace_functions.h.slang:437): alias table,
// then a uniform position inside the chosen texel's spherical rectangle.
float4 environmentSample(const Scene& sc, float3 xi, float3& toLight)
{
  const MiPtEnvironment* e = sc.env;
  uint32_t width = uint32_t(e->width), height = uint32_t(e->height), size = width * height;
  uint32_t idx = std::min(uint32_t(xi.x * float(size)), size - 1);
  MiEnvAccel a = e->accel[idx];
  uint32_t envIdx;
  if(xi.y < a.q)
  {
    envIdx = idx;
    xi.y /= a.q;
  }
  else
  {
    envIdx = a.alias;
    xi.y   = (xi.y - a.q) / (1.0f - a.q);
  }
  uint32_t py = envIdx / width, px = envIdx % width;
  float    u   = (float(px) + xi.y) / float(width);
  float    phi = u * K_TWO_PI - K_PI;
  float    sinPhi = std::sin(phi), cosPhi = std::cos(phi);
  float    stepTheta = K_PI / float(height);
  float    theta0    = float(py) * stepTheta;
  float    cosTheta  = std::cos(theta0) * (1.0f - xi.z) + std::cos(theta0 + stepTheta) * xi.z;
  float    theta     = std::acos(clampf(cosTheta, -1.0f, 1.0f));
  float    sinTheta  = std::sin(theta);
  float    v         = theta * K_1_OVER_PI;
  toLight            = float3(cosPhi * sinTheta, cosTheta, sinPhi * sinTheta);
  return sampleHdr(sc, float2(u, v));
}

//======================================================================================================================
// Geometry: world-space triangle soup + binned-SAH BVH2 (replaces the TLAS/BLAS of src/gltf_scene_rtx.cpp:173-385)
//======================================================================================================================
struct Tri
{
  float3   v0, e1, e2;
  int      rnode, prim;
};
struct Instance  // per render node flags (reference: src/gltf_scene_rtx.cpp:271-295)
{
  bool forceOpaque, cullDisable, flipFacing;
};
struct BvhNode
{
  float3  bmin, bmax;
  int32_t left;   // inner: index of left child (right = left + 1); leaf: first triangle
  int32_t count;  // 0 = inner
};
struct Accel
{
  std::vector<Tri>      tris;
  std::vector<BvhNode>  nodes;
  std::vector<Instance> inst;
};

void buildAccel(const MiPtSceneDesc& d, Accel& A)
{
  A.inst.resize(size_t(d.numRenderNodes));
  for(int n = 0; n < d.numRenderNodes; ++n)
  {
    const MiGltfRenderNode&    rn  = d.renderNodes[n];
    const MiGltfShadeMaterial& mat = d.materials[std::max(0, std::min(rn.materialID, d.numMaterials - 1))];
    Instance&                  I   = A.inst[size_t(n)];
    I.forceOpaque = (mat.transmissionFactor == 0.0f) && (mat.alphaMode == MI_ALPHA_OPAQUE) && (mat.diffuseTransmissionFactor == 0.0f);
    I.cullDisable = (mat.doubleSided == 1) || (mat.thicknessFactor > 0.0f) || (mat.transmissionFactor > 0.0f);
    mat4 M        = loadMat(rn.objectToWorld);
    I.flipFacing  = det3(M) < 0.0f;
    if(d.renderNodeVisible && !d.renderNodeVisible[n])
      continue;
    if(rn.renderPrimID < 0 || rn.renderPrimID >= d.numRenderPrimitives)
      continue;
    const MiPtRenderPrimitive& rp = d.renderPrimitives[rn.renderPrimID];
    if(!rp.positions || !rp.indices)
      continue;
    for(uint32_t t = 0; t < rp.triangleCount; ++t)
    {
      float3 p0 = mulPoint(M, float3(rp.positions + 3 * size_t(rp.indices[3 * t + 0])));
      float3 p1 = mulPoint(M, float3(rp.positions + 3 * size_t(rp.indices[3 * t + 1])));
      float3 p2 = mulPoint(M, float3(rp.positions + 3 * size_t(rp.indices[3 * t + 2])));
      Tri    T;
      T.v0    = p0;
      T.e1    = p1 - p0;
      T.e2    = p2 - p0;
      T.rnode = n;
      T.prim  = int(t);
      A.tris.push_back(T);
    }
  }
  // binned SAH build over centroids
  size_t              n = A.tris.size();
  std::vector<float3> cmin(n), cmax(n), cen(n);
  for(size_t i = 0; i < n; ++i)
  {
    const Tri& T = A.tris[i];
    float3     a = T.v0, b = T.v0 + T.e1, c = T.v0 + T.e2;
    cmin[i] = min3(a, min3(b, c));
    cmax[i] = max3(a, max3(b, c));
    cen[i]  = (cmin[i] + cmax[i]) * 0.5f;
  }
  std::vector<uint32_t> order(n);
  for(size_t i = 0; i < n; ++i)
    order[i] = uint32_t(i);
  A.nodes.clear();
  A.nodes.reserve(2 * n + 1);
  A.nodes.push_back(BvhNode{float3(0.0f), float3(0.0f), 0, 0});
  struct Work
  {
    int    node;
    size_t begin, end;
  };
  std::vector<Work> stack{{0, 0, n}};
  auto area = [](float3 mn, float3 mx) {
    float3 e = mx - mn;
    return 2.0f * (e.x * e.y + e.y * e.z + e.z * e.x);
  };
  while(!stack.empty())
  {
    Work w = stack.back();
    stack.pop_back();
    float3 bmin(FLT_MAX), bmax(-FLT_MAX), kmin(FLT_MAX), kmax(-FLT_MAX);
    for(size_t i = w.begin; i < w.end; ++i)
    {
      bmin = min3(bmin, cmin[order[i]]);
      bmax = max3(bmax, cmax[order[i]]);
      kmin = min3(kmin, cen[order[i]]);
      kmax = max3(kmax, cen[order[i]]);
    }
    BvhNode& node = A.nodes[size_t(w.node)];
    node.bmin     = bmin;
    node.bmax     = bmax;
    size_t cnt    = w.end - w.begin;
    if(cnt <= 4)
    {
      node.left  = int32_t(w.begin);
      node.count = int32_t(cnt);
      continue;
    }
    float3 ext  = kmax - kmin;
    int    axis = ext.x > ext.y ? (ext.x > ext.z ? 0 : 2) : (ext.y > ext.z ? 1 : 2);
    size_t mid  = (w.begin + w.end) / 2;
    if(ext[axis] > 0.0f)
    {
      const int NB = 16;
      float3    bmn[NB], bmx[NB];
      int       bc[NB];
      float     bestCost = FLT_MAX;
      int       bestAxis = axis, bestSplit = -1;
      for(int ax = 0; ax < 3; ++ax)
      {
        if(!(ext[ax] > 0.0f))
          continue;
        for(int b = 0; b < NB; ++b)
        {
          bmn[b] = float3(FLT_MAX);
          bmx[b] = float3(-FLT_MAX);
          bc[b]  = 0;
        }
        float k = float(NB) / ext[ax];
        for(size_t i = w.begin; i < w.end; ++i)
        {
          int b  = std::min(NB - 1, int((cen[order[i]][ax] - kmin[ax]) * k));
          bmn[b] = min3(bmn[b], cmin[order[i]]);
          bmx[b] = max3(bmx[b], cmax[order[i]]);
          bc[b]++;
        }
        float  rightArea[NB];
        float3 rmn(FLT_MAX), rmx(-FLT_MAX);
        for(int b = NB - 1; b > 0; --b)
        {
          rmn          = min3(rmn, bmn[b]);
          rmx          = max3(rmx, bmx[b]);
          rightArea[b] = area(rmn, rmx);
        }
        float3 lmn(FLT_MAX), lmx(-FLT_MAX);
        int    lc = 0;
        for(int b = 0; b < NB - 1; ++b)
        {
          lmn = min3(lmn, bmn[b]);
          lmx = max3(lmx, bmx[b]);
          lc += bc[b];
          int rc = int(cnt) - lc;
          if(lc == 0 || rc == 0)
            continue;
          float cost = area(lmn, lmx) * float(lc) + rightArea[b + 1] * float(rc);
          if(cost < bestCost)
          {
            bestCost  = cost;
            bestAxis  = ax;
            bestSplit = b;
          }
        }
      }
      if(bestSplit >= 0)
      {
        float k  = float(NB) / ext[bestAxis];
        auto  it = std::partition(order.begin() + long(w.begin), order.begin() + long(w.end), [&](uint32_t id) {
          return std::min(NB - 1, int((cen[id][bestAxis] - kmin[bestAxis]) * k)) <= bestSplit;
        });
        mid      = size_t(it - order.begin());
      }
      if(mid == w.begin || mid == w.end)
      {
        mid = (w.begin + w.end) / 2;
        std::nth_element(order.begin() + long(w.begin), order.begin() + long(mid), order.begin() + long(w.end),
                         [&](uint32_t a, uint32_t b) { return cen[a][axis] < cen[b][axis]; });
      }
    }
    int left = int(A.nodes.size());
    A.nodes.push_back(BvhNode{float3(0.0f), float3(0.0f), 0, 0});
    A.nodes.push_back(BvhNode{float3(0.0f), float3(0.0f), 0, 0});
    A.nodes[size_t(w.node)].left  = left;
    A.nodes[size_t(w.node)].count = 0;
    stack.push_back({left, w.begin, mid});
    stack.push_back({left + 1, mid, w.end});
  }
  std::vector<Tri> sorted(n);
  for(size_t i = 0; i < n; ++i)
    sorted[i] = A.tris[order[i]];
  A.tris.swap(sorted);
}

// Moeller-Trumbore with a fixed fmaf evaluation order (the device code uses the same order so t/u/v are bit-identical).
inline float dotFma(float3 a, float3 b) { return std::fmaf(a.z, b.z, std::fmaf(a.y, b.y, a.x * b.x)); }
inline float3 crossFma(float3 a, float3 b)
{
  return float3(std::fmaf(a.y, b.z, -(a.z * b.y)), std::fmaf(a.z, b.x, -(a.x * b.z)), std::fmaf(a.x, b.y, -(a.y * b.x)));
}
struct TriHit
{
  float t, u, v;
  bool  front;  // world-space winding is counter-clockwise seen from the ray origin
};
inline bool intersectTri(const Tri& T, float3 org, float3 dir, TriHit& h)
{
  float3 pvec = crossFma(dir, T.e2);
  float  det  = dotFma(T.e1, pvec);
  if(det == 0.0f)
    return false;
  float  inv  = 1.0f / det;
  float3 tvec = org - T.v0;
  float  u    = dotFma(tvec, pvec) * inv;
  if(u < 0.0f || u > 1.0f)
    return false;
  float3 qvec = crossFma(tvec, T.e1);
  float  v    = dotFma(dir, qvec) * inv;
  if(v < 0.0f || u + v > 1.0f)
    return false;
  h.t     = dotFma(T.e2, qvec) * inv;
  h.u     = u;
  h.v     = v;
  h.front = det > 0.0f;
  return true;
}
inline bool intersectBox(float3 bmin, float3 bmax, float3 org, float3 invDir, float tmax)
{
  float tx0 = (bmin.x - org.x) * invDir.x, tx1 = (bmax.x - org.x) * invDir.x;
  float ty0 = (bmin.y - org.y) * invDir.y, ty1 = (bmax.y - org.y) * invDir.y;
  float tz0 = (bmin.z - org.z) * invDir.z, tz1 = (bmax.z - org.z) * invDir.z;
  float tn = std::fmax(std::fmax(std::fmin(tx0, tx1), std::fmin(ty0, ty1)), std::fmax(std::fmin(tz0, tz1), 0.0f));
  float tf = std::fmin(std::fmin(std::fmax(tx0, tx1), std::fmax(ty0, ty1)), std::fmin(std::fmax(tz0, tz1), tmax));
  // conservative: widen by a few ulps so that no triangle accepted by intersectTri is ever missed
  return tn <= tf * 1.0000004f + 1e-30f;
}
// Visit every triangle whose box test passes; `visit` may shrink tmax by returning a new value.
template <typename F>
inline void traverse(const Accel& A, float3 org, float3 dir, float tmax, F&& visit, uint64_t* nodeCount = nullptr)
{
  if(A.tris.empty())
    return;
  float3 invDir(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
  int    stack[128];
  int    sp    = 0;
  stack[sp++]  = 0;
  while(sp > 0)
  {
    const BvhNode& n = A.nodes[size_t(stack[--sp])];
    if(nodeCount)
      ++*nodeCount;
    if(!intersectBox(n.bmin, n.bmax, org, invDir, tmax))
      continue;
    if(n.count > 0)
    {
      for(int i = 0; i < n.count; ++i)
        tmax = visit(A.tris[size_t(n.left + i)], tmax);
    }
    else if(sp + 2 <= 128)
    {
      stack[sp++] = n.left;
      stack[sp++] = n.left + 1;
    }
  }
}

//======================================================================================================================
// Reference restatement
//======================================================================================================================
struct RayDesc
{
  float3 Origin, Direction;
  float  TMin = 0.0f, TMax = INFINITE_F;
};
struct HitPayload  // raytracer_interface.h.slang:36-47
{
  float  hitT        = 0.0f;
  int    rnodeID     = -1;
  int    rprimID     = -1;
  int    primitiveID = -1;
  float2 bary{0, 0};
};
struct Counters
{
  uint64_t cameraPaths = 0, segments = 0, shadowRays = 0, nodesClosest = 0, trisClosest = 0, nodesShadow = 0, trisShadow = 0, textureTaps = 0, surfaceHits = 0;
};
struct Ctx
{
  const Scene* sc;
  const Accel* accel;
  Counters     cnt;
};

const float MIN_TRANSMISSION                = 0.01f;       // pathtrace_functions.h.slang:36
const float ANTIALIASING_STANDARD_DEVIATION = 0.4246609f;  // :37
const int   RR_MIN_DEPTH                    = 3;           // :38
const float VOLUME_MIN_SCATTER              = 0.001f;      // :39
const float VOLUME_RAND_FLOOR               = 1.0e-10f;    // :40
const int   VOLUME_FREE_BUDGET              = 64;          // :41
const float RR_PCONT_FLOOR                  = 0.001f;      // :42
const float RR_PCONT_CAP                    = 0.95f;       // :43
const float MICROFACET_MIN_ROUGHNESS        = 0.0014142f;  // gltf_material_eval.h.slang:51

inline bool hasFlag(int flags, int f) { return (flags & f) != 0; }

// --- gltf_vertex_access.h.slang:30-150 -------------------------------------------------------------------------------
struct uint3v
{
  uint32_t x, y, z;
};
inline uint3v getTriangleIndices(const MiPtRenderPrimitive& rp, int prim)
{
  return {rp.indices[3 * size_t(prim)], rp.indices[3 * size_t(prim) + 1], rp.indices[3 * size_t(prim) + 2]};
}
inline float3 getVertexPosition(const MiPtRenderPrimitive& rp, uint32_t i) { return float3(rp.positions + 3 * size_t(i)); }
inline float4 unpackUnorm4x8(uint32_t p)
{
  return float4(float((p >> 0) & 0xFF) / 255.0f, float((p >> 8) & 0xFF) / 255.0f, float((p >> 16) & 0xFF) / 255.0f, float((p >> 24) & 0xFF) / 255.0f);
}
inline float2 getInterpolatedVertexTexCoord(const MiPtRenderPrimitive& rp, int channel, uint3v idx, float3 b)
{
  const float* tc = channel == 0 ? rp.texCoords0 : rp.texCoords1;
  if(!tc)
    return float2(0.0f, 0.0f);
  float2 a(tc[2 * size_t(idx.x)], tc[2 * size_t(idx.x) + 1]), bb(tc[2 * size_t(idx.y)], tc[2 * size_t(idx.y) + 1]),
      c(tc[2 * size_t(idx.z)], tc[2 * size_t(idx.z) + 1]);
  return a * b.x + bb * b.y + c * b.z;
}
inline float4 getInterpolatedVertexColor(const MiPtRenderPrimitive& rp, uint3v idx, float3 b)
{
  if(!rp.colors)
    return float4(1.0f);
  return unpackUnorm4x8(rp.colors[idx.x]) * b.x + unpackUnorm4x8(rp.colors[idx.y]) * b.y + unpackUnorm4x8(rp.colors[idx.z]) * b.z;
}

// --- get_hit.h.slang:26-173 ------------------------------------------------------------------------------------------
struct HitState
{
  float3 pos, nrm;
  float4 color;
  float3 geonrm, shadowPos;
  float2 uv[2];
  float3 tangent, bitangent;
  float  texelDensity = 0.0f;
  bool   frontFace    = true;
};
float computeTexelDensity(float3 pos0, float3 pos1, float3 pos2, float2 uv0, float2 uv1, float2 uv2, const mat4& o2w)  // :44-56
{
  float3 we1 = mulVector(o2w, pos1 - pos0), we2 = mulVector(o2w, pos2 - pos0);
  float  wArea = length(cross(we1, we2));
  float2 duv1 = uv1 - uv0, duv2 = uv2 - uv0;
  float  uvArea = std::fabs(duv1.x * duv2.y - duv1.y * duv2.x);
  return std::sqrt(std::fmax(uvArea, 1e-20f) / std::fmax(wArea, 1e-20f));
}
HitState getHitState(const MiPtRenderPrimitive& rp, float3 bary, const mat4& w2o, const mat4& o2w, int triangleID, float3 worldRayDir)  // :59-173
{
  HitState hit;
  uint3v   ti   = getTriangleIndices(rp, triangleID);
  float3   pos0 = getVertexPosition(rp, ti.x), pos1 = getVertexPosition(rp, ti.y), pos2 = getVertexPosition(rp, ti.z);
  float3   position = pos0 * bary.x + pos1 * bary.y + pos2 * bary.z;
  hit.pos           = mulPoint(o2w, position);
  float3 geoNormal  = normalize(cross(pos1 - pos0, pos2 - pos0));
  hit.geonrm        = normalize(mulTransposed(w2o, geoNormal));
  float3 nrm0 = geoNormal, nrm1 = geoNormal, nrm2 = geoNormal, normal = geoNormal;
  if(rp.normals)
  {
    nrm0   = float3(rp.normals + 3 * size_t(ti.x));
    nrm1   = float3(rp.normals + 3 * size_t(ti.y));
    nrm2   = float3(rp.normals + 3 * size_t(ti.z));
    normal = nrm0 * bary.x + nrm1 * bary.y + nrm2 * bary.z;
  }
  hit.nrm        = normalize(mulTransposed(w2o, normal));
  hit.frontFace  = dot(hit.geonrm, worldRayDir) < 0.0f;
  float sideFlip = hit.frontFace ? 1.0f : -1.0f;
  float3 shadowPos = pointOffset(position, pos0, pos1, pos2, nrm0 * sideFlip, nrm1 * sideFlip, nrm2 * sideFlip, bary);
  hit.shadowPos    = mulPoint(o2w, shadowPos);
  hit.uv[0]        = getInterpolatedVertexTexCoord(rp, 0, ti, bary);
  hit.uv[1]        = getInterpolatedVertexTexCoord(rp, 1, ti, bary);
  if(rp.texCoords0)
  {
    float2 uv0(rp.texCoords0[2 * size_t(ti.x)], rp.texCoords0[2 * size_t(ti.x) + 1]);
    float2 uv1(rp.texCoords0[2 * size_t(ti.y)], rp.texCoords0[2 * size_t(ti.y) + 1]);
    float2 uv2(rp.texCoords0[2 * size_t(ti.z)], rp.texCoords0[2 * size_t(ti.z) + 1]);
    hit.texelDensity = computeTexelDensity(pos0, pos1, pos2, uv0, uv1, uv2, o2w);
  }
  else
    hit.texelDensity = 0.0f;
  hit.color = getInterpolatedVertexColor(rp, ti, bary);
  float4 tng[3];
  if(rp.tangents)
  {
    tng[0] = float4(rp.tangents + 4 * size_t(ti.x));
    tng[1] = float4(rp.tangents + 4 * size_t(ti.y));
    tng[2] = float4(rp.tangents + 4 * size_t(ti.z));
  }
  else
  {
    float4 t = makeFastTangent(normal);
    tng[0] = tng[1] = tng[2] = t;
  }
  hit.tangent   = normalize(tng[0].xyz() * bary.x + tng[1].xyz() * bary.y + tng[2].xyz() * bary.z);
  hit.tangent   = mulVector(o2w, hit.tangent);
  hit.tangent   = normalize(hit.tangent - hit.nrm * dot(hit.nrm, hit.tangent));
  hit.bitangent = cross(hit.nrm, hit.tangent) * tng[0].w;
  if(!hit.frontFace)
    hit.geonrm = -hit.geonrm;
  if(dot(hit.geonrm, hit.nrm) < 0.0f)
  {
    hit.nrm       = -hit.nrm;
    hit.tangent   = -hit.tangent;
    hit.bitangent = -hit.bitangent;
  }
  float3 r = reflect(normalize(worldRayDir), hit.nrm);
  if(dot(r, hit.geonrm) < 0.0f)
    hit.nrm = hit.geonrm;
  return hit;
}

// --- gltf_material_eval.h.slang --------------------------------------------------------------------------------------
struct MeshState  // :54-64
{
  float3 N, T, B, Ng;
  float2 tc[2];
  bool   isInside = false;
  float  texGrad  = 0.0f;
  float4 baseColorVertexMul{1, 1, 1, 1};
};
float4 getTexture(Ctx& cx, const MiGltfTextureInfo& ti, const float2 tc[2], float texGrad)  // :76-110
{
  cx.cnt.textureTaps++;
  float2 t  = tc[ti.texCoord];
  const float* U = ti.uvTransform;
  float2 tt(t.x * U[0] + t.y * U[2] + U[4], t.x * U[1] + t.y * U[3] + U[5]);
  if(texGrad > 0.0f)
  {
    float2 ddx(U[0] * texGrad, U[1] * texGrad), ddy(U[2] * texGrad, U[3] * texGrad);
    return sampleTexture(*cx.sc, ti.index, tt, true, ddx, ddy);
  }
  return sampleTexture(*cx.sc, ti.index, tt, false, float2(0, 0), float2(0, 0));
}
inline bool isTexturePresent(uint16_t t) { return t > 0; }  // :115-118
float3 multiToSingleScatterAlbedo(float3 rho)  // :125-129
{
  float3 t = float3(4.09712f) + rho * 4.20863f - sqrt3(float3(9.59217f) + rho * 41.6808f + rho * rho * 17.7126f);
  return float3(1.0f) - t * t;
}
float3 convertSGToMR(float3 diffuseColor, float3 specularColor, float glossiness, float& metallic, float2& roughness)  // :136-161
{
  const float dielectricSpecular = 0.04f;
  float specularIntensity = std::fmax(specularColor.x, std::fmax(specularColor.y, specularColor.z));
  float isMetal           = smoothstep(dielectricSpecular + 0.01f, dielectricSpecular + 0.05f, specularIntensity);
  metallic                = isMetal;
  float3 baseColor;
  if(metallic > 0.0f)
    baseColor = specularColor;
  else
  {
    baseColor = diffuseColor / (1.0f - dielectricSpecular * (1.0f - metallic));
    baseColor = clamp3(baseColor, 0.0f, 1.0f);
  }
  float r   = 1.0f - glossiness;
  roughness = float2(r * r);
  return baseColor;
}
PbrMaterial evaluateMaterial(Ctx& cx, const MiGltfShadeMaterial& m, const MeshState& st)  // :168-457
{
  const MiGltfTextureInfo* ti = cx.sc->desc->textureInfos;
  auto tex = [&](uint16_t slot) { return getTexture(cx, ti[slot], st.tc, st.texGrad); };
  PbrMaterial p;
  if(m.pbrModel == MI_PBR_SPECULAR_GLOSSINESS)
  {
    float4 diffuse    = float4(m.pbrDiffuseFactor) * st.baseColorVertexMul;
    float  glossiness = m.pbrGlossinessFactor;
    float3 specular   = float3(m.pbrSpecularFactor);
    if(isTexturePresent(m.pbrDiffuseTexture))
      diffuse *= tex(m.pbrDiffuseTexture);
    if(isTexturePresent(m.pbrSpecularGlossinessTexture))
    {
      float4 s = tex(m.pbrSpecularGlossinessTexture);
      specular *= s.xyz();
      glossiness *= s.w;
    }
    p.baseColor = convertSGToMR(diffuse.xyz(), specular, glossiness, p.metallic, p.roughness);
    p.opacity   = diffuse.w;
  }
  else
  {
    float4 baseColor = float4(m.pbrBaseColorFactor) * st.baseColorVertexMul;
    if(isTexturePresent(m.pbrBaseColorTexture))
      baseColor *= tex(m.pbrBaseColorTexture);
    p.baseColor     = baseColor.xyz();
    p.opacity       = baseColor.w;
    float roughness = m.pbrRoughnessFactor, metallic = m.pbrMetallicFactor;
    if(isTexturePresent(m.pbrMetallicRoughnessTexture))
    {
      float4 s = tex(m.pbrMetallicRoughnessTexture);
      roughness *= s.y;
      metallic *= s.z;
    }
    roughness   = std::fmax(roughness, MICROFACET_MIN_ROUGHNESS);
    p.roughness = float2(roughness * roughness);
    p.metallic  = clampf(metallic, 0.0f, 1.0f);
  }
  p.occlusion = m.occlusionStrength;
  if(isTexturePresent(m.occlusionTexture))
  {
    float occ   = tex(m.occlusionTexture).x;
    p.occlusion = 1.0f + p.occlusion * (occ - 1.0f);
  }
  p.N  = st.N;
  p.T  = st.T;
  p.B  = st.B;
  p.Ng = st.Ng;
  bool needsTangentUpdate = false;
  if(isTexturePresent(m.normalTexture))
  {
    float3 nv = tex(m.normalTexture).xyz();
    nv        = nv * 2.0f - float3(1.0f);
    nv *= float3(m.normalTextureScale, m.normalTextureScale, 1.0f);
    p.N                = normalize(st.T * nv.x + st.B * nv.y + st.N * nv.z);  // mul(v, float3x3(T,B,N))
    needsTangentUpdate = true;
  }
  p.emissive = float3(m.emissiveFactor);
  if(isTexturePresent(m.emissiveTexture))
    p.emissive *= tex(m.emissiveTexture).xyz();
  p.emissive = max3(float3(0.0f), p.emissive);
  p.attenuationColor    = float3(m.attenuationColor);
  p.attenuationDistance = m.attenuationDistance;
  p.thickness           = m.thicknessFactor;
  if(isTexturePresent(m.thicknessTexture))
    p.thickness *= tex(m.thicknessTexture).y;
  p.specularColor = float3(m.specularColorFactor);
  if(isTexturePresent(m.specularColorTexture))
    p.specularColor *= tex(m.specularColorTexture).xyz();
  p.specular = m.specularFactor;
  if(isTexturePresent(m.specularTexture))
    p.specular *= tex(m.specularTexture).w;
  float ior1 = 1.0f, ior2 = m.ior;
  if(st.isInside && (p.thickness > 0.0f))
  {
    ior1 = ior2;
    ior2 = 1.0f;
  }
  p.ior1         = ior1;
  p.ior2         = ior2;
  p.transmission = m.transmissionFactor;
  if(isTexturePresent(m.transmissionTexture))
    p.transmission *= tex(m.transmissionTexture).x;
  float3 ms(m.multiscatterColorFactor);
  if(ms.x > 0.0f || ms.y > 0.0f || ms.z > 0.0f)
  {
    float3 ssa  = multiToSingleScatterAlbedo(ms);
    float3 attC = -log3(max3(p.attenuationColor, float3(0.001f))) / std::fmax(p.attenuationDistance, 0.001f);
    p.scatterCoefficient = attC * ssa;
  }
  p.scatterAnisotropy  = m.scatterAnisotropy;
  p.clearcoat          = m.clearcoatFactor;
  p.clearcoatRoughness = m.clearcoatRoughness;
  p.Nc                 = p.N;
  if(isTexturePresent(m.clearcoatTexture))
    p.clearcoat *= tex(m.clearcoatTexture).x;
  if(isTexturePresent(m.clearcoatRoughnessTexture))
    p.clearcoatRoughness *= tex(m.clearcoatRoughnessTexture).y;
  if(isTexturePresent(m.clearcoatNormalTexture))
  {
    float3 nv = tex(m.clearcoatNormalTexture).xyz();
    nv        = nv * 2.0f - float3(1.0f);
    p.Nc      = normalize(p.T * nv.x + p.B * nv.y + p.Nc * nv.z);
  }
  p.clearcoatRoughness = std::fmax(p.clearcoatRoughness, 0.001f);
  float iridescence = m.iridescenceFactor, iridescenceThickness = m.iridescenceThicknessMaximum;
  p.iridescenceIor = m.iridescenceIor;
  if(isTexturePresent(m.iridescenceTexture))
    iridescence *= tex(m.iridescenceTexture).x;
  if(isTexturePresent(m.iridescenceThicknessTexture))
  {
    float t              = tex(m.iridescenceThicknessTexture).y;
    iridescenceThickness = lerp(m.iridescenceThicknessMinimum, m.iridescenceThicknessMaximum, t);
  }
  p.iridescence          = (iridescenceThickness > 0.0f) ? iridescence : 0.0f;
  p.iridescenceThickness = iridescenceThickness;
  float anisotropyStrength = m.anisotropyStrength;
  if(anisotropyStrength > 0.0f)
  {
    float2 dir(1.0f, 0.0f);
    if(isTexturePresent(m.anisotropyTexture))
    {
      float4 a = tex(m.anisotropyTexture);
      dir      = normalize(float2(a.x, a.y) * 2.0f - float2(1.0f, 1.0f));
      anisotropyStrength *= a.z;
    }
    p.roughness.x = lerp(p.roughness.y, 1.0f, anisotropyStrength * anisotropyStrength);
    float s = m.anisotropyRotation[0], c = m.anisotropyRotation[1];
    dir                = float2(c * dir.x + s * dir.y, c * dir.y - s * dir.x);
    p.T                = p.T * dir.x + p.B * dir.y;
    needsTangentUpdate = true;
  }
  if(needsTangentUpdate)
  {
    p.B         = normalize(cross(p.N, p.T));
    float bsign = signf(dot(st.B, p.B));
    p.B         = p.B * bsign;
    p.T         = normalize(cross(p.B, p.N) * bsign);
  }
  p.sheenColor = float3(m.sheenColorFactor);
  if(isTexturePresent(m.sheenColorTexture))
    p.sheenColor *= tex(m.sheenColorTexture).xyz();
  p.sheenRoughness = m.sheenRoughnessFactor;
  if(isTexturePresent(m.sheenRoughnessTexture))
    p.sheenRoughness *= tex(m.sheenRoughnessTexture).w;
  p.sheenRoughness = std::fmax(MICROFACET_MIN_ROUGHNESS, p.sheenRoughness);
  p.dispersion     = m.dispersion;
  p.diffuseTransmissionFactor = m.diffuseTransmissionFactor;
  if(isTexturePresent(m.diffuseTransmissionTexture))
    p.diffuseTransmissionFactor *= tex(m.diffuseTransmissionTexture).w;
  p.diffuseTransmissionColor = float3(m.diffuseTransmissionColor);
  if(isTexturePresent(m.diffuseTransmissionColorTexture))
    p.diffuseTransmissionColor *= tex(m.diffuseTransmissionColorTexture).xyz();
  p.retroreflection = m.retroreflectionFactor;
  if(isTexturePresent(m.retroreflectionTexture))
    p.retroreflection *= tex(m.retroreflectionTexture).x;
  return p;
}

// --- pathtrace_functions.h.slang -------------------------------------------------------------------------------------
struct RayCone  // :104-108
{
  float width = 0.0f, spreadAngle = 0.0f;
};
struct VolumeMedium  // :118-123 (fp16 storage)
{
  float3 extinction{0, 0, 0}, scatterCoefficient{0, 0, 0};
  float  scatterAnisotropy = 0.0f;
};
VolumeMedium makeVolumeMedium(const PbrMaterial& p)  // :125-132
{
  VolumeMedium m;
  float3       e = volumeExtinctionCoefficient(p);
  m.extinction         = float3(roundToHalf(e.x), roundToHalf(e.y), roundToHalf(e.z));
  m.scatterCoefficient = float3(roundToHalf(p.scatterCoefficient.x), roundToHalf(p.scatterCoefficient.y), roundToHalf(p.scatterCoefficient.z));
  m.scatterAnisotropy  = roundToHalf(p.scatterAnisotropy);
  return m;
}
bool hasVolumeMedium(const VolumeMedium& m) { return maxComp(m.extinction) > 0.0f || maxComp(m.scatterCoefficient) > 0.0f; }  // :134-140

float3 safeOffsetRay(float3 p, float3 dir)  // :151-167 (Waechter & Binder)
{
  const float scaleValue = 256.0f;
  int         ix = int(scaleValue * dir.x), iy = int(scaleValue * dir.y), iz = int(scaleValue * dir.z);
  float3      op(asfloat(asint(p.x) + ((p.x < 0) ? -ix : ix)), asfloat(asint(p.y) + ((p.y < 0) ? -iy : iy)), asfloat(asint(p.z) + ((p.z < 0) ? -iz : iz)));
  const float origin = 1.0f / 32.0f, floatScale = 1.0f / 65536.0f;
  return float3(std::fabs(p.x) < origin ? p.x + floatScale * dir.x : op.x, std::fabs(p.y) < origin ? p.y + floatScale * dir.y : op.y,
                std::fabs(p.z) < origin ? p.z + floatScale * dir.z : op.z);
}
float rayConeWorldFootprint(RayCone cone, float hitT, float3 N, float3 V)  // :174-178
{
  float w = cone.width + cone.spreadAngle * hitT;
  return w / std::fmax(std::fabs(dot(N, V)), 1e-3f);
}

float getOpacity(Ctx& cx, const MiGltfRenderNode& rn, const MiPtRenderPrimitive& rp, int triangleID, float3 bary)  // :189-234
{
  const MiPtSceneDesc&       d   = *cx.sc->desc;
  const MiGltfShadeMaterial& mat = d.materials[std::max(0, rn.materialID)];
  if(mat.alphaMode == MI_ALPHA_OPAQUE)
    return 1.0f;
  uint3v ti    = getTriangleIndices(rp, triangleID);
  float  alpha = 1.0f;
  auto   tapAlpha = [&](uint16_t slot) {
    const MiGltfTextureInfo& info = d.textureInfos[slot];
    float2                   uv   = getInterpolatedVertexTexCoord(rp, info.texCoord, ti, bary);
    return sampleTexture(*cx.sc, info.index, uv, false, float2(0, 0), float2(0, 0)).w;  // SampleLevel(uv, 0), no uv transform
  };
  if(mat.pbrModel == MI_PBR_SPECULAR_GLOSSINESS)
  {
    alpha = mat.pbrDiffuseFactor[3];
    if(isTexturePresent(mat.pbrDiffuseTexture))
      alpha *= tapAlpha(mat.pbrDiffuseTexture);
  }
  else
  {
    alpha = mat.pbrBaseColorFactor[3];
    if(isTexturePresent(mat.pbrBaseColorTexture))
      alpha *= tapAlpha(mat.pbrBaseColorTexture);
  }
  alpha *= getInterpolatedVertexColor(rp, ti, bary).w;
  if(mat.alphaMode == MI_ALPHA_MASK)
    return alpha >= mat.alphaCutoff ? 1.0f : 0.0f;
  return alpha;
}

float3 getShadowTransmission(Ctx& cx, const MiGltfRenderNode& rn, const MiPtRenderPrimitive& rp, int triangleID, float3 bary, float hitT,
                             const mat4& w2o, float3 rayDir, bool& isInside)  // :244-343
{
  const MiPtSceneDesc&       d   = *cx.sc->desc;
  const MiGltfShadeMaterial& mat = d.materials[std::max(0, rn.materialID)];
  float                      tFactor = mat.transmissionFactor;
  if(tFactor <= MIN_TRANSMISSION)
    return float3(0.0f);
  uint3v ti = getTriangleIndices(rp, triangleID);
  float3 v0 = getVertexPosition(rp, ti.x), v1 = getVertexPosition(rp, ti.y), v2 = getVertexPosition(rp, ti.z);
  float3 normal = normalize(cross(v1 - v0, v2 - v0));
  normal        = normalize(mulTransposed(w2o, normal));
  float  cosTheta = std::fabs(dot(rayDir, normal));
  float  fresnel  = schlickFresnelIor(mat.ior, cosTheta);
  float3 T        = float3(mat.pbrBaseColorFactor) * tFactor;
  T *= (1.0f - fresnel);
  if(mat.thicknessFactor > 0.0f)
  {
    if(isInside)
    {
      float3 absCoeff     = -log3(max3(float3(mat.attenuationColor), float3(0.001f))) / std::fmax(mat.attenuationDistance, 0.001f);
      float3 scatterCoeff = absCoeff * multiToSingleScatterAlbedo(float3(mat.multiscatterColorFactor));
      float3 extinction   = absCoeff + scatterCoeff;
      T *= exp3(extinction * (-hitT));
      float maxScatter = maxComp(scatterCoeff);
      if(maxScatter > 0.001f)
        T *= std::exp(-(hitT * maxComp(extinction)));
    }
    isInside = !isInside;
  }
  float roughness = mat.pbrRoughnessFactor, metallic = mat.pbrMetallicFactor;
  if(isTexturePresent(mat.pbrMetallicRoughnessTexture))
  {
    const MiGltfTextureInfo& info = d.textureInfos[mat.pbrMetallicRoughnessTexture];
    float2                   uv   = getInterpolatedVertexTexCoord(rp, info.texCoord, ti, bary);
    float4                   mr   = sampleTexture(*cx.sc, info.index, uv, false, float2(0, 0), float2(0, 0));
    roughness *= mr.y;
    metallic *= mr.z;
  }
  float att = (1.0f - metallic);
  att *= lerp(0.65f, 1.0f, 1.0f - roughness * roughness);
  return T * att;
}

// --- raytracer_interface.h.slang:67-188 (RayQuery variant) on the software BVH ---------------------------------------
void Trace(Ctx& cx, const RayDesc& ray, HitPayload& payload, uint32_t& seed)  // :69-122
{
  const Accel&         A = *cx.accel;
  const MiPtSceneDesc& d = *cx.sc->desc;
  payload.hitT           = INFINITE_F;
  cx.cnt.segments++;
  float    bestT   = ray.TMax;
  int      bestTri = -1;
  float2   bestUV(0, 0);
  uint64_t bestKey = ~0ull;
  const uint32_t seed0 = seed;
  traverse(A, ray.Origin, ray.Direction, ray.TMax, [&](const Tri& T, float tmax) {
    cx.cnt.trisClosest++;
    TriHit h;
    if(!intersectTri(T, ray.Origin, ray.Direction, h) || !(h.t > ray.TMin))
      return tmax;
    uint64_t key = (uint64_t(uint32_t(T.rnode)) << 32) | uint32_t(T.prim);
    if(!(h.t < bestT || (h.t == bestT && key < bestKey)))
      return tmax;
    const Instance& I = A.inst[size_t(T.rnode)];
    // RAY_FLAG_CULL_BACK_FACING_TRIANGLES unless the instance disables culling; facing is an object-space property
    bool front = h.front != I.flipFacing;
    if(!front && !I.cullDisable)
      return tmax;
    if(!I.forceOpaque)
    {
      const MiGltfRenderNode& rn = d.renderNodes[T.rnode];
      float opacity = getOpacity(cx, rn, d.renderPrimitives[rn.renderPrimID], T.prim, float3(1.0f - h.u - h.v, h.u, h.v));
      if(!(candidateRand(seed0, T.rnode, T.prim) <= opacity))
        return tmax;
    }
    bestT   = h.t;
    bestKey = key;
    bestTri = int(&T - A.tris.data());
    bestUV  = float2(h.u, h.v);
    return bestT;
  }, &cx.cnt.nodesClosest);
  if(bestTri >= 0)
  {
    const Tri& T        = A.tris[size_t(bestTri)];
    payload.hitT        = bestT;
    payload.rnodeID     = T.rnode;
    payload.rprimID     = d.renderNodes[T.rnode].renderPrimID;
    payload.primitiveID = T.prim;
    payload.bary        = bestUV;
  }
}
void TraceLow(Ctx& cx, const RayDesc& ray, HitPayload& payload)  // :124-137 (force opaque, no culling)
{
  const Accel& A = *cx.accel;
  payload.hitT   = INFINITE_F;
  float    bestT = ray.TMax;
  int      best  = -1;
  uint64_t bestKey = ~0ull;
  traverse(A, ray.Origin, ray.Direction, ray.TMax, [&](const Tri& T, float tmax) {
    TriHit h;
    if(!intersectTri(T, ray.Origin, ray.Direction, h) || !(h.t > ray.TMin))
      return tmax;
    uint64_t key = (uint64_t(uint32_t(T.rnode)) << 32) | uint32_t(T.prim);
    if(!(h.t < bestT || (h.t == bestT && key < bestKey)))
      return tmax;
    bestT   = h.t;
    bestKey = key;
    best    = T.rnode;
    return bestT;
  });
  if(best >= 0)
  {
    payload.hitT    = bestT;
    payload.rnodeID = best;
  }
}
float3 TraceShadow(Ctx& cx, const RayDesc& ray, uint32_t& seed, bool initialInside = false)  // :139-187
{
  const Accel&         A = *cx.accel;
  const MiPtSceneDesc& d = *cx.sc->desc;
  cx.cnt.shadowRays++;
  struct Cand
  {
    float  t;
    int    rnode, prim;
    float2 uv;
  };
  std::vector<Cand> cands;
  bool              opaqueHit = false;
  traverse(A, ray.Origin, ray.Direction, ray.TMax, [&](const Tri& T, float tmax) {
    cx.cnt.trisShadow++;
    if(opaqueHit)
      return tmax;
    TriHit h;
    if(!intersectTri(T, ray.Origin, ray.Direction, h) || !(h.t > ray.TMin) || !(h.t < ray.TMax))
      return tmax;
    if(A.inst[size_t(T.rnode)].forceOpaque)  // RAY_FLAG_NONE: no culling; opaque geometry commits
      opaqueHit = true;
    else
      cands.push_back({h.t, T.rnode, T.prim, float2(h.u, h.v)});
    return tmax;
  }, &cx.cnt.nodesShadow);
  if(opaqueHit)
    return float3(0.0f);
  std::sort(cands.begin(), cands.end(), [](const Cand& a, const Cand& b) {
    if(a.t != b.t)
      return a.t < b.t;
    if(a.rnode != b.rnode)
      return a.rnode < b.rnode;
    return a.prim < b.prim;
  });
  float3         total(1.0f);
  bool           isInside = initialInside;
  float          prevHitT = 0.0f;
  const uint32_t seed0    = seed;
  for(const Cand& c : cands)
  {
    const MiGltfRenderNode&    rn   = d.renderNodes[c.rnode];
    const MiPtRenderPrimitive& rp   = d.renderPrimitives[rn.renderPrimID];
    float3                     bary = float3(1.0f - c.uv.x - c.uv.y, c.uv.x, c.uv.y);
    float                      opacity = getOpacity(cx, rn, rp, c.prim, bary);
    float                      r       = candidateRand(seed0, c.rnode, c.prim);
    if(r < opacity)
    {
      float  segment = std::fmax(0.0f, c.t - prevHitT);
      mat4   w2o     = loadMat(rn.worldToObject);
      float3 cur     = getShadowTransmission(cx, rn, rp, c.prim, bary, segment, w2o, ray.Direction, isInside);
      prevHitT       = c.t;
      total *= cur;
      if(maxComp(total) <= MIN_TRANSMISSION)
        return float3(0.0f);
    }
  }
  return total;
}

// --- lights & environment (pathtrace_functions.h.slang:357-492) ------------------------------------------------------
struct DirectLight
{
  float3 direction{0, 0, 0}, radianceOverPdf{0, 0, 0};
  float  distance = INFINITE_F, pdf = 0.0f;
};
void getDirectLightingTechniqueProbabilities(const Scene& sc, float& lightWeight, float& envWeight)  // :357-377
{
  lightWeight = (sc.desc->numLights > 0) ? 0.5f : 0.0f;
  envWeight   = (!hasFlag(sc.frameInfo.flags, MI_SCENE_USE_HDR_ENVIRONMENT) || sc.frameInfo.envIntensity > 0.0f) ? 0.5f : 0.0f;
  float total = lightWeight + envWeight;
  if(total > 0.0f)
  {
    lightWeight /= total;
    envWeight /= total;
  }
}
void sampleLights(Ctx& cx, float3 pos, float3 normal, float3 worldRayDirection, uint32_t& seed, DirectLight& dl, bool isVolumeSample = false)  // :379-464
{
  const Scene& sc = *cx.sc;
  float3 radiance(0.0f);
  dl.pdf             = 0.0f;
  dl.distance        = INFINITE_F;
  dl.radianceOverPdf = float3(0.0f);
  dl.direction       = float3(0.0f);
  float envPdf       = 0.0f;
  float lightWeight, envWeight;
  getDirectLightingTechniqueProbabilities(sc, lightWeight, envWeight);
  if(lightWeight == 0.0f && envWeight == 0.0f)
    return;
  bool sampleLight = (rnd(seed) < lightWeight);
  const bool useHdr = hasFlag(sc.frameInfo.flags, MI_SCENE_USE_HDR_ENVIRONMENT);
  if(sampleLight)
  {
    int   numLights    = sc.desc->numLights;
    float selectionPdf = 1.0f / float(numLights);
    int   lightIndex   = std::min(int(rnd(seed) * float(numLights)), numLights - 1);
    const MiGltfLight& light = sc.desc->lights[lightIndex];
    float3 cullNormal = (isVolumeSample && light.type == MI_LIGHT_DIRECTIONAL) ? -float3(light.direction) : normal;
    float  r1 = rnd(seed), r2 = rnd(seed);
    LightContrib contrib = singleLightContribution(light, pos, cullNormal, float2(r1, r2));
    dl.direction = -contrib.incidentVector;
    dl.distance  = contrib.distance;
    radiance     = contrib.intensity / (selectionPdf * lightWeight);
    dl.pdf       = (contrib.pdf == DIRAC) ? DIRAC : selectionPdf * contrib.pdf;
  }
  if(envWeight > 0.0f && dl.pdf != DIRAC)
  {
    if(!useHdr)
    {
      if(!sampleLight)
      {
        float r1 = rnd(seed), r2 = rnd(seed);
        SkySamplingResult s = samplePhysicalSky(sc.sky, float2(r1, r2));
        dl.direction        = s.direction;
        envPdf              = s.pdf;
        radiance            = s.radiance / (envPdf * envWeight);
      }
      else
        envPdf = samplePhysicalSkyPDF(sc.sky, dl.direction);
    }
    else
    {
      if(!sampleLight)
      {
        float  r1 = rnd(seed), r2 = rnd(seed), r3 = rnd(seed);
        float4 rp = environmentSample(sc, float3(r1, r2, r3), dl.direction);
        envPdf    = rp.w;
        radiance  = rp.xyz() * sc.frameInfo.envIntensity / (envPdf * envWeight);
        dl.direction = rotate(dl.direction, float3(0, 1, 0), sc.frameInfo.envRotation);
      }
      else
      {
        float3 dir = rotate(dl.direction, float3(0, 1, 0), -sc.frameInfo.envRotation);
        envPdf     = sampleHdr(sc, getSphericalUv(dir)).w;
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
  // guard: a zero-pdf environment sample (black texel) yields inf/nan radiance; treat as no light
  if(!(radiance.x == radiance.x) || !(radiance.y == radiance.y) || !(radiance.z == radiance.z) || std::isinf(radiance.x)
     || std::isinf(radiance.y) || std::isinf(radiance.z))
  {
    radiance = float3(0.0f);
    dl.pdf   = 0.0f;
  }
  dl.radianceOverPdf = radiance;
  (void)worldRayDirection;
}
void sampleEnvironment(const Scene& sc, float3 direction, float3& envColor, float& envPdf)  // :466-481
{
  if(!hasFlag(sc.frameInfo.flags, MI_SCENE_USE_HDR_ENVIRONMENT))
  {
    envColor = evalPhysicalSky(sc.sky, direction);
    envPdf   = samplePhysicalSkyPDF(sc.sky, direction);
  }
  else
  {
    float3 dir = rotate(direction, float3(0, 1, 0), -sc.frameInfo.envRotation);
    float4 env = sampleHdr(sc, getSphericalUv(dir));
    envColor   = env.xyz() * sc.frameInfo.envIntensity;
    envPdf     = env.w;
  }
}
float computeEnvHitMisWeight(const Scene& sc, float lastSamplePdf, float envPdf)  // :483-492
{
  if(lastSamplePdf == DIRAC)
    return 1.0f;
  float lw, ew;
  getDirectLightingTechniqueProbabilities(sc, lw, ew);
  return lastSamplePdf / (lastSamplePdf + ew * envPdf);
}

// --- path state (pathtrace_functions.h.slang:822-886) ----------------------------------------------------------------
enum PathStepResult { eOutOfVolume, eVolumeContinue, eEarlyContinue, eBreak };
struct PathTracerState
{
  float3  radiance{0, 0, 0}, throughput{1, 1, 1}, firstHitPos{1e34f, 1e34f, 1e34f};
  float   lastSamplePdf = DIRAC;
  float2  maxRoughness{0, 0};
  bool    solid = true;
  RayCone cone;
  int     surfaceDepth = 0;
  bool    isInside     = false;
  VolumeMedium volumeMedium;
  int     scatterBounces = 0;
  // guides for the a-trous pass (reference: GuideScratch, :84-93; captured at gltf_pathtrace.slang:228-264)
  float3 guideAlbedo{0, 0, 0}, guideNormal{0, 0, 0};
};
struct BounceScratch
{
  bool   nextEventValid = false;
  float3 contribution{0, 0, 0}, shadowRayPos{0, 0, 0}, shadowRayDir{0, 0, 0};
  float  shadowRayDist = 0.0f;
};
struct SampleResult
{
  float4 radiance{0, 0, 0, 0};
  float3 hitPosition{1e34f, 1e34f, 1e34f};
  float3 albedo{0, 0, 0}, normal{0, 0, 0};
};

bool handleShadowCatcher(Ctx& cx, const HitState& hit, const PbrMaterial& pbrMat, RayDesc& ray, float3& radiance, float3& throughput,
                         float& lastSamplePdf, uint32_t& seed)  // :499-554
{
  const Scene& sc = *cx.sc;
  DirectLight  dl;
  sampleLights(cx, hit.pos, pbrMat.N, ray.Direction, seed, dl);
  float3 shadowFactor(1.0f);
  if(dot(dl.direction, hit.nrm) > 0.0f && dl.pdf != 0.0f)
  {
    RayDesc sr;
    sr.Origin    = hit.pos;
    sr.Direction = dl.direction;
    sr.TMin      = 0.0f;
    sr.TMax      = INFINITE_F;
    shadowFactor = TraceShadow(cx, sr, seed);
  }
  float3 envColor;
  float  envPdf;
  sampleEnvironment(sc, ray.Direction, envColor, envPdf);
  if(shadowFactor.x == 1.0f && shadowFactor.y == 1.0f && shadowFactor.z == 1.0f)
  {
    float mis = computeEnvHitMisWeight(sc, lastSamplePdf, envPdf);
    radiance += throughput * mis * envColor;
    return false;
  }
  radiance += envColor * shadowFactor;
  radiance -= envColor * (float3(1.0f) - shadowFactor) * sc.frameInfo.shadowCatcherDarkenAmount;
  BsdfSampleData sd;
  sd.k1 = -ray.Direction;
  float r1 = rnd(seed), r2 = rnd(seed), r3 = rnd(seed);
  sd.xi = float3(r1, r2, r3);
  bsdfSampleSimple(sd, pbrMat);
  if(sd.event_type == BSDF_EVENT_ABSORB)
    return false;
  float3 offsetDir = dot(sd.k2, hit.geonrm) > 0.0f ? hit.geonrm : -hit.geonrm;
  ray.Origin       = safeOffsetRay(hit.pos, offsetDir);
  ray.Direction    = sd.k2;
  throughput *= sd.bsdf_over_pdf;
  lastSamplePdf = sd.pdf;
  return true;
}
bool checkInfinitePlaneIntersection(const Scene& sc, const RayDesc& ray, HitPayload& payload, HitState& hit)  // :556-585
{
  if(!hasFlag(sc.frameInfo.flags, MI_SCENE_USE_INFINITE_PLANE))
    return false;
  float3 normal(0, 1, 0);
  float  planeHeight = sc.frameInfo.infinitePlaneDistance;
  if(ray.Origin.y <= planeHeight)
    return false;
  float Dn = dot(ray.Direction, normal);
  if(std::fabs(Dn) <= 1e-6f)
    return false;
  float On = dot(ray.Origin, normal);
  float t  = (-On + planeHeight) / Dn;
  if(t <= 0.0f || t >= payload.hitT)
    return false;
  payload.hitT  = t;
  hit.pos       = ray.Origin + ray.Direction * payload.hitT;
  hit.shadowPos = hit.pos;
  hit.nrm       = normal;
  hit.geonrm    = normal;
  hit.tangent   = float3(1, 0, 0);
  hit.bitangent = float3(0, 0, 1);
  return true;
}
bool handleVolumeScatter(const VolumeMedium& medium, float hitDistance, RayDesc& ray, float3& throughput, float& lastSamplePdf, uint32_t& seed)  // :605-645
{
  float3 extinction   = medium.extinction;
  float3 scatterCoeff = medium.scatterCoefficient;
  float  maxScatter   = maxComp(scatterCoeff);
  if(maxScatter > VOLUME_MIN_SCATTER)
  {
    float maxExtinction = maxComp(extinction);
    float scatterDist   = -std::log(std::fmax(rnd(seed), VOLUME_RAND_FLOOR)) / maxExtinction;
    if(scatterDist < hitDistance)
    {
      throughput *= float3(1.0f) - (extinction - scatterCoeff) / maxExtinction;
      ray.Origin = ray.Origin + ray.Direction * scatterDist;
      float3 wi  = ray.Direction;
      float  g   = medium.scatterAnisotropy;
      float  r1 = rnd(seed), r2 = rnd(seed);
      ray.Direction = sampleHenyeyGreenstein(float2(r1, r2), g, wi);
      lastSamplePdf = henyeyGreensteinPdf(dot(wi, ray.Direction), g);
      return true;
    }
    throughput *= exp3((float3(maxExtinction) - extinction) * hitDistance);
    return false;
  }
  throughput *= exp3(extinction * (-hitDistance));
  return false;
}
float3 volumeScatterNEE(Ctx& cx, const VolumeMedium& medium, float3 scatterPos, float3 wiBeforeScatter, float3 throughput, uint32_t& seed)  // :651-672
{
  DirectLight dl;
  sampleLights(cx, scatterPos, wiBeforeScatter, wiBeforeScatter, seed, dl, true);
  if(dl.pdf <= 0.0f)  // NB: DIRAC (-1) is <= 0, so delta lights add nothing here — the reference's literal behaviour
    return float3(0.0f);
  float cosTheta = dot(wiBeforeScatter, dl.direction);
  float phasePdf = henyeyGreensteinPdf(cosTheta, medium.scatterAnisotropy);
  float mis      = (dl.pdf == DIRAC) ? 1.0f : dl.pdf / (dl.pdf + phasePdf);
  RayDesc sr;
  sr.Origin    = scatterPos;
  sr.Direction = dl.direction;
  sr.TMin      = 0.0f;
  sr.TMax      = dl.distance;
  float3 shadowFactor = TraceShadow(cx, sr, seed, true);
  return throughput * dl.radianceOverPdf * mis * phasePdf * shadowFactor;
}
PathStepResult processVolumeSegment(Ctx& cx, float hitDistance, RayDesc& ray, PathTracerState& pt, uint32_t& seed)  // :904-939
{
  if(pt.isInside && hasVolumeMedium(pt.volumeMedium))
  {
    float3 wiBefore = ray.Direction, originBefore = ray.Origin;
    if(handleVolumeScatter(pt.volumeMedium, hitDistance, ray, pt.throughput, pt.lastSamplePdf, seed))
    {
      pt.scatterBounces++;
      pt.cone.width += pt.cone.spreadAngle * length(ray.Origin - originBefore);
      pt.radiance += volumeScatterNEE(cx, pt.volumeMedium, ray.Origin, wiBefore, pt.throughput, seed);
      if(pt.scatterBounces >= VOLUME_FREE_BUDGET)
      {
        float rrPcont = std::fmin(maxComp(pt.throughput) + RR_PCONT_FLOOR, RR_PCONT_CAP);
        if(rnd(seed) >= rrPcont)
          return eBreak;
        pt.throughput /= rrPcont;
      }
      return eVolumeContinue;
    }
  }
  return eOutOfVolume;
}
// smoothHDRBlur (nvshaders/sample_blur.h.slang, backplate only): 3x3 tent of level-0 taps scaled by the blur radius
float3 smoothHDRBlur(const Scene& sc, float2 uv, float blur)
{
  float3 sum(0.0f);
  float  wsum = 0.0f;
  float  r    = blur * 0.02f;
  for(int j = -1; j <= 1; ++j)
    for(int i = -1; i <= 1; ++i)
    {
      float w = (2.0f - std::fabs(float(i))) * (2.0f - std::fabs(float(j)));
      sum += sampleHdr(sc, float2(uv.x + float(i) * r, uv.y + float(j) * r * 0.5f)).xyz() * w;
      wsum += w;
    }
  return sum / wsum;
}
bool tryPrimaryMissBackplate(const Scene& sc, const RayDesc& ray, PathTracerState& pt)  // :944-971
{
  pt.solid       = false;
  pt.firstHitPos = ray.Direction;
  if(hasFlag(sc.frameInfo.flags, MI_SCENE_USE_SOLID_BACKGROUND))
  {
    pt.radiance = float3(sc.frameInfo.backgroundColor);
    return true;
  }
  if(hasFlag(sc.frameInfo.flags, MI_SCENE_USE_HDR_ENVIRONMENT) && sc.frameInfo.envBlur > 0.0f)
  {
    float3 dir  = rotate(ray.Direction, float3(0, 1, 0), -sc.frameInfo.envRotation);
    pt.radiance = smoothHDRBlur(sc, getSphericalUv(dir), sc.frameInfo.envBlur) * sc.frameInfo.envIntensity;
    return true;
  }
  return false;
}

// --- gltf_pathtrace.slang:87-430 -------------------------------------------------------------------------------------
PathStepResult pathTraceOneBounce(Ctx& cx, RayDesc& ray, uint32_t& seed, PathTracerState& pt, BounceScratch& bounce)
{
  const Scene&         sc = *cx.sc;
  const MiPtSceneDesc& d  = *sc.desc;
  bounce                  = BounceScratch();
  HitPayload payload;
  Trace(cx, ray, payload, seed);
  HitState                   hit{};
  const MiGltfRenderNode*    renderNode = nullptr;
  const MiPtRenderPrimitive* renderPrim = nullptr;
  if(payload.hitT != INFINITE_F)
  {
    renderNode  = &d.renderNodes[payload.rnodeID];
    renderPrim  = &d.renderPrimitives[payload.rprimID];
    float3 bary = float3(1.0f - payload.bary.x - payload.bary.y, payload.bary.x, payload.bary.y);
    hit = getHitState(*renderPrim, bary, loadMat(renderNode->worldToObject), loadMat(renderNode->objectToWorld), payload.primitiveID, ray.Direction);
  }
  bool firstRay         = (pt.surfaceDepth == 0);
  bool hitInfinitePlane = checkInfinitePlaneIntersection(sc, ray, payload, hit);
  if(payload.hitT != INFINITE_F)
    cx.cnt.surfaceHits++;
  if(payload.hitT == INFINITE_F)  // :129-156
  {
    if(firstRay && tryPrimaryMissBackplate(sc, ray, pt))
      return eBreak;
    float3 envColor;
    float  envPdf;
    sampleEnvironment(sc, ray.Direction, envColor, envPdf);
    float mis = computeEnvHitMisWeight(sc, pt.lastSamplePdf, envPdf);
    pt.radiance += pt.throughput * mis * envColor;
    return eBreak;
  }
  int         materialIndex = -1;
  PbrMaterial pbrMat;
  float       worldFoot = rayConeWorldFootprint(pt.cone, payload.hitT, hit.geonrm, -ray.Direction);
  if(hitInfinitePlane)  // :169-187
  {
    pbrMat = defaultPbrMaterial5(float3(sc.frameInfo.infinitePlaneBaseColor), sc.frameInfo.infinitePlaneMetallic,
                                 sc.frameInfo.infinitePlaneRoughness, hit.nrm, hit.nrm);
    pbrMat.T = hit.tangent;
    pbrMat.B = hit.bitangent;
    if(hasFlag(sc.frameInfo.flags, MI_SCENE_INFINITE_PLANE_SHADOW_CATCHER))
    {
      pt.cone.width = worldFoot;
      bool cont     = handleShadowCatcher(cx, hit, pbrMat, ray, pt.radiance, pt.throughput, pt.lastSamplePdf, seed);
      if(!cont)
        return eBreak;
      return eEarlyContinue;
    }
  }
  else  // :191-216
  {
    materialIndex = std::max(0, renderNode->materialID);
    float     texGrad = worldFoot * hit.texelDensity * sc.pc.texGradScale;
    MeshState mesh;
    mesh.N = hit.nrm;
    mesh.T = hit.tangent;
    mesh.B = hit.bitangent;
    mesh.Ng = hit.geonrm;
    mesh.tc[0] = hit.uv[0];
    mesh.tc[1] = hit.uv[1];
    mesh.isInside = pt.isInside;
    mesh.texGrad  = texGrad;
    mesh.baseColorVertexMul = hit.color;
    pbrMat = evaluateMaterial(cx, d.materials[materialIndex], mesh);
  }
  if(firstRay)  // :228-264
  {
    pt.firstHitPos = hit.pos;
    pt.guideAlbedo = pbrMat.baseColor;
    pt.guideNormal = pbrMat.N;
  }
  pt.maxRoughness  = float2(std::fmax(pbrMat.roughness.x, pt.maxRoughness.x), std::fmax(pbrMat.roughness.y, pt.maxRoughness.y));  // :267-268
  pbrMat.roughness = pt.maxRoughness;
  pt.radiance += pbrMat.emissive * pt.throughput;  // :293
  if(materialIndex >= 0 && d.materials[materialIndex].unlit > 0)  // :298-304
  {
    pt.radiance += pbrMat.baseColor;
    return eBreak;
  }
  PathStepResult volumeStep = processVolumeSegment(cx, payload.hitT, ray, pt, seed);  // :306-308
  if(volumeStep != eOutOfVolume)
    return volumeStep;
  pt.cone.width = worldFoot;  // :313
  DirectLight dl;
  sampleLights(cx, hit.pos, pbrMat.N, ray.Direction, seed, dl);  // :319-320
  bounce.nextEventValid = (dot(dl.direction, hit.nrm) > 0.0f || pbrMat.diffuseTransmissionFactor > 0.0f) && dl.pdf != 0.0f;  // :324-325
  if(bounce.nextEventValid)  // :330-351
  {
    BsdfEvaluateData ev;
    ev.k1 = -ray.Direction;
    ev.k2 = dl.direction;
    float r1 = rnd(seed), r2 = rnd(seed), r3 = rnd(seed);
    ev.xi = float3(r1, r2, r3);
    bsdfEvaluate(ev, pbrMat);
    if(ev.pdf > 0.0f)
    {
      float  mis = (dl.pdf == DIRAC) ? 1.0f : dl.pdf / (dl.pdf + ev.pdf);
      float3 w   = pt.throughput * dl.radianceOverPdf * mis;
      bounce.contribution += w * ev.bsdf_diffuse;
      bounce.contribution += w * ev.bsdf_glossy;
    }
  }
  {  // :357-416
    BsdfSampleData sd;
    sd.k1 = -ray.Direction;
    float r1 = rnd(seed), r2 = rnd(seed), r3 = rnd(seed);
    sd.xi = float3(r1, r2, r3);
    bsdfSample(sd, pbrMat);
    pt.throughput *= sd.bsdf_over_pdf;
    ray.Direction    = sd.k2;
    pt.lastSamplePdf = sd.pdf;
    if(sd.event_type != BSDF_EVENT_ABSORB)
    {
      bool   isTransmission = (sd.event_type & BSDF_EVENT_TRANSMISSION) != 0;
      float3 offsetDir      = dot(ray.Direction, hit.geonrm) > 0.0f ? hit.geonrm : -hit.geonrm;
      ray.Origin            = safeOffsetRay(hit.pos, offsetDir);
      if(isTransmission)
      {
        pt.isInside = !pt.isInside;
        if(pt.isInside)
          pt.volumeMedium = makeVolumeMedium(pbrMat);
      }
    }
    else
      pt.surfaceDepth = sc.pc.maxDepth;
  }
  bool   shadowSideForward = dot(dl.direction, hit.nrm) > 0.0f;  // :421-426
  float3 shadowOffsetDir   = shadowSideForward ? hit.geonrm : -hit.geonrm;
  float3 shadowOffsetBase  = shadowSideForward ? hit.shadowPos : hit.pos;
  bounce.shadowRayPos      = safeOffsetRay(shadowOffsetBase, shadowOffsetDir);
  bounce.shadowRayDir      = dl.direction;
  bounce.shadowRayDist     = dl.distance;
  return eOutOfVolume;
}

SampleResult pathTrace(Ctx& cx, RayDesc ray, uint32_t& seed)  // gltf_pathtrace.slang:441-494
{
  const Scene&    sc = *cx.sc;
  PathTracerState pt;
  pt.cone.spreadAngle = sc.pc.pixelAngle;
  cx.cnt.cameraPaths++;
  int guard = 0;
  while(pt.surfaceDepth < sc.pc.maxDepth && guard++ < 100000)
  {
    ray.Direction = normalize(ray.Direction);
    BounceScratch  bounce;
    PathStepResult step = pathTraceOneBounce(cx, ray, seed, pt, bounce);
    if(step == eBreak)
      break;
    if(step == eVolumeContinue || step == eEarlyContinue)
      continue;
    if(bounce.nextEventValid)
    {
      RayDesc sr;
      sr.Origin    = bounce.shadowRayPos;
      sr.Direction = bounce.shadowRayDir;
      sr.TMin      = 0.0f;
      sr.TMax      = bounce.shadowRayDist;
      float3 shadowFactor = TraceShadow(cx, sr, seed);
      pt.radiance += bounce.contribution * shadowFactor;
    }
    if(pt.surfaceDepth >= RR_MIN_DEPTH)
    {
      float rrPcont = std::fmin(maxComp(pt.throughput) + 0.001f, 0.95f);
      if(rnd(seed) >= rrPcont)
        break;
      pt.throughput /= rrPcont;
    }
    pt.surfaceDepth++;
  }
  SampleResult r;  // packSampleResult, pathtrace_functions.h.slang:973-988
  r.radiance    = float4(pt.radiance, pt.solid ? 1.0f : 0.0f);
  r.hitPosition = pt.firstHitPos;
  r.albedo      = pt.guideAlbedo;
  r.normal      = pt.guideNormal;
  return r;
}

float2 sampleGaussian(float2 u)  // pathtrace_functions.h.slang:784-789
{
  float r     = std::sqrt(-2.0f * std::log(std::fmax(1e-38f, u.x)));
  float theta = 2.0f * K_PI * u.y;
  return float2(r * std::cos(theta), r * std::sin(theta));
}
RayDesc getRay(float2 samplePos, float2 offset, float2 imageSize, const mat4& projInv, const mat4& viewInv, bool ortho)  // :791-811
{
  float2 clip((samplePos.x + offset.x) / imageSize.x * 2.0f - 1.0f, (samplePos.y + offset.y) / imageSize.y * 2.0f - 1.0f);
  float4 view = mulFull(projInv, float4(clip.x, clip.y, -1.0f, 1.0f));
  view        = view / view.w;
  RayDesc ray;
  if(ortho)
  {
    ray.Origin    = mulFull(viewInv, view).xyz();
    ray.Direction = normalize(mulFull(viewInv, float4(0, 0, -1, 0)).xyz());
  }
  else
  {
    ray.Origin    = float3(viewInv.m[12], viewInv.m[13], viewInv.m[14]);
    ray.Direction = normalize(mulFull(viewInv, view).xyz() - ray.Origin);
  }
  ray.TMin = 0.0f;
  ray.TMax = INFINITE_F;
  return ray;
}
SampleResult samplePixel(Ctx& cx, uint32_t& seed, float2 samplePos, float2 jitter, float2 imageSize)  // gltf_pathtrace.slang:502-541
{
  const Scene& sc      = *cx.sc;
  const mat4   projInv = loadMat(sc.frameInfo.projInv), viewInv = loadMat(sc.frameInfo.viewInv);
  const bool   ortho   = hasFlag(sc.frameInfo.flags, MI_SCENE_IS_ORTHOGRAPHIC);
  RayDesc      ray     = getRay(samplePos, jitter, imageSize, projInv, viewInv, ortho);
  if(!ortho)
  {
    float3 focalPoint = ray.Direction * sc.pc.focalDistance;
    float  cam_r1     = rnd(seed) * K_TWO_PI;
    float  cam_r2     = rnd(seed) * sc.pc.aperture;
    float3 cam_right  = float3(viewInv.m[0], viewInv.m[4], viewInv.m[8]);   // mul(viewMatrixI, (1,0,0,0)) = M^T e0
    float3 cam_up     = float3(viewInv.m[1], viewInv.m[5], viewInv.m[9]);
    float3 aperturePos = (cam_right * std::cos(cam_r1) + cam_up * std::sin(cam_r1)) * std::sqrt(cam_r2);
    float3 finalDir    = normalize(focalPoint - aperturePos);
    ray.Origin += aperturePos;
    ray.Direction = finalDir;
  }
  SampleResult r   = pathTrace(cx, ray, seed);
  float        lum = dot(r.radiance.xyz(), float3(1.0f / 3.0f));
  if(lum > sc.pc.fireflyClampThreshold)
    r.radiance *= sc.pc.fireflyClampThreshold / lum;
  return r;
}
uint32_t traceSelectionRay(Ctx& cx, float2 samplePos, float2 imageSize)  // pathtrace_functions.h.slang:813-820
{
  const Scene& sc = *cx.sc;
  RayDesc ray = getRay(samplePos, float2(0.5f), imageSize, loadMat(sc.frameInfo.projInv), loadMat(sc.frameInfo.viewInv),
                       hasFlag(sc.frameInfo.flags, MI_SCENE_IS_ORTHOGRAPHIC));
  HitPayload p;
  TraceLow(cx, ray, p);
  return (p.hitT != INFINITE_F) ? uint32_t(p.rnodeID + 1) : 0u;
}

struct Outputs
{
  float*    accum;      // RGBA32F running mean (eImgRendered)
  uint32_t* selection;  // may be null
  float*    depth;      // may be null
  float*    albedo;     // RGBA32F first-hit albedo guide, may be null
  float*    normal;     // RGBA32F first-hit normal guide, may be null
};
void processPixel(Ctx& cx, int px, int py, int W, int H, const Outputs& out)  // gltf_pathtrace.slang:546-671
{
  const Scene& sc = *cx.sc;
  float2 samplePos{float(px), float(py)}, imageSize{float(W), float(H)};
  uint32_t seed       = xxhash32(uint32_t(px), uint32_t(py), uint32_t(sc.pc.frameCount));
  bool     firstFrame = hasFlag(sc.pc.flags, MI_PT_FIRST_FRAME);
  float2   jitter(0.5f, 0.5f);
  {
    float  r1 = rnd(seed), r2 = rnd(seed);
    float2 g  = sampleGaussian(float2(r1, r2));
    jitter    = jitter + g * ANTIALIASING_STANDARD_DEVIATION;
  }
  SampleResult sr    = samplePixel(cx, seed, samplePos, jitter, imageSize);
  float4       pixel = sr.radiance;
  float3       albedo = sr.albedo, normal = sr.normal;
  for(int s = 1; s < sc.pc.numSamples; ++s)
  {
    float r1 = rnd(seed), r2 = rnd(seed);
    jitter   = float2(r1, r2);
    sr       = samplePixel(cx, seed, samplePos, jitter, imageSize);
    pixel += sr.radiance;
    albedo += sr.albedo;
    normal += sr.normal;
  }
  pixel = pixel / float(sc.pc.numSamples);
  const bool hasSolidHit = sr.radiance.w > 0.0f;
  float4 clipPos  = hasSolidHit ? mulFull(loadMat(sc.frameInfo.viewProjMatrix), float4(sr.hitPosition, 1.0f)) : float4(0, 0, 1, 1);
  float  ndcDepth = hasSolidHit ? (clipPos.z / clipPos.w) : 1.0f;
  size_t idx      = size_t(py) * size_t(W) + size_t(px);
  if(firstFrame)
  {
    uint32_t id = traceSelectionRay(cx, samplePos, imageSize);
    if(out.selection)
      out.selection[idx] = id;
    if(out.depth)
      out.depth[idx] = ndcDepth;
  }
  float* dst = out.accum + idx * 4;
  // guideN.w: second moment of this frame's pixel luminance (our SVGF pass's temporal-variance input; no reference counterpart)
  const float lumP = 0.2126f * pixel.x + 0.7152f * pixel.y + 0.0722f * pixel.z;
  float4 guideA(albedo / float(sc.pc.numSamples), hasSolidHit ? 1.0f : 0.0f), guideN(normal / float(sc.pc.numSamples), lumP * lumP);
  if(firstFrame)
  {
    dst[0] = pixel.x; dst[1] = pixel.y; dst[2] = pixel.z; dst[3] = pixel.w;
  }
  else
  {
    float  after = float(sc.pc.totalSamples + sc.pc.numSamples);
    float4 old(dst);
    float4 v = (old * float(sc.pc.totalSamples) + pixel * float(sc.pc.numSamples)) / after;
    dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
  }
  if(hasFlag(sc.pc.flags, MI_PT_USE_OPTIX_DENOISER) && out.albedo && out.normal)
  {
    float* a = out.albedo + idx * 4;
    float* n = out.normal + idx * 4;
    if(firstFrame)
    {
      a[0] = guideA.x; a[1] = guideA.y; a[2] = guideA.z; a[3] = guideA.w;
      n[0] = guideN.x; n[1] = guideN.y; n[2] = guideN.z; n[3] = guideN.w;
    }
    else
    {
      float after = float(sc.pc.totalSamples + sc.pc.numSamples);
      float wOld = float(sc.pc.totalSamples) / after, wNew = float(sc.pc.numSamples) / after;
      a[0] = a[0] * wOld + guideA.x * wNew; a[1] = a[1] * wOld + guideA.y * wNew; a[2] = a[2] * wOld + guideA.z * wNew; a[3] = a[3] * wOld + guideA.w * wNew;
      n[0] = n[0] * wOld + guideN.x * wNew; n[1] = n[1] * wOld + guideN.y * wNew; n[2] = n[2] * wOld + guideN.z * wNew; n[3] = n[3] * wOld + guideN.w * wNew;
    }
  }
}

}  // namespace

//======================================================================================================================
// C interface
//======================================================================================================================
struct OraclePt
{
  Scene              scene;
  Accel              accel;
  int                width = 0, height = 0;
  std::vector<float> accum, depth, albedo, normal;
  std::vector<uint32_t> selection;
  Counters           counters;
  int                tileRank = 0, tileWorld = 1, tileSize = 64;
};

extern "C" {

int oracle_pt_create(const MiPtSceneDesc* scene, OraclePt** out)
{
  if(!scene || !out)
    return -1;
  OraclePt* o   = new OraclePt();
  o->scene.desc = scene;
  for(int i = 0; i < 256; ++i)
  {
    float c             = float(i) / 255.0f;
    o->scene.srgbLut[i] = c <= 0.04045f ? c / 12.92f : std::pow((c + 0.055f) / 1.055f, 2.4f);
  }
  buildAccel(*scene, o->accel);
  *out = o;
  return 0;
}
void oracle_pt_destroy(OraclePt* o) { delete o; }
int  oracle_pt_set_environment(OraclePt* o, const MiPtEnvironment* env)
{
  o->scene.env = env;
  return 0;
}
int oracle_pt_resize(OraclePt* o, int w, int h)
{
  o->width  = w;
  o->height = h;
  size_t n  = size_t(w) * size_t(h);
  o->accum.assign(n * 4, 0.0f);
  o->albedo.assign(n * 4, 0.0f);
  o->normal.assign(n * 4, 0.0f);
  o->depth.assign(n, 1.0f);
  o->selection.assign(n, 0u);
  return 0;
}
int oracle_pt_set_frame_info(OraclePt* o, const MiSceneFrameInfo* f)
{
  o->scene.frameInfo = *f;
  return 0;
}
int oracle_pt_set_sky(OraclePt* o, const MiSkyPhysicalParameters* s)
{
  o->scene.sky = *s;
  return 0;
}
int oracle_pt_set_tile_partition(OraclePt* o, int rank, int world, int tileSize)
{
  o->tileRank  = rank;
  o->tileWorld = std::max(world, 1);
  o->tileSize  = std::max(tileSize, 1);
  return 0;
}
// Renders one frame (params->numSamples spp per pixel) into the running-mean accumulator, `threads` host threads.
int oracle_pt_render_frame(OraclePt* o, const MiPathtraceParams* params, int threads)
{
  if(!o || !params || o->width <= 0)
    return -1;
  o->scene.pc = *params;
  threads     = std::max(1, threads);
  const int W = o->width, H = o->height;
  std::atomic<int>      nextRow{0};
  std::vector<Counters> perThread(static_cast<size_t>(threads));
  Outputs out{o->accum.data(), o->selection.data(), o->depth.data(), o->albedo.data(), o->normal.data()};
  const int tilesX = (W + o->tileSize - 1) / o->tileSize;
  auto      worker = [&](int tid) {
    Ctx cx{&o->scene, &o->accel, Counters()};
    for(;;)
    {
      int y = nextRow.fetch_add(1);
      if(y >= H)
        break;
      for(int x = 0; x < W; ++x)
      {
        if(o->tileWorld > 1)
        {
          int tile = (y / o->tileSize) * tilesX + (x / o->tileSize);
          if(tile % o->tileWorld != o->tileRank)
            continue;
        }
        processPixel(cx, x, y, W, H, out);
      }
    }
    perThread[size_t(tid)] = cx.cnt;
  };
  std::vector<std::thread> pool;
  for(int t = 1; t < threads; ++t)
    pool.emplace_back(worker, t);
  worker(0);
  for(auto& th : pool)
    th.join();
  for(const Counters& c : perThread)
  {
    o->counters.cameraPaths += c.cameraPaths;
    o->counters.segments += c.segments;
    o->counters.shadowRays += c.shadowRays;
    o->counters.nodesClosest += c.nodesClosest;
    o->counters.trisClosest += c.trisClosest;
    o->counters.nodesShadow += c.nodesShadow;
    o->counters.trisShadow += c.trisShadow;
    o->counters.textureTaps += c.textureTaps;
    o->counters.surfaceHits += c.surfaceHits;
  }
  return 0;
}
const float*    oracle_pt_accum(OraclePt* o) { return o->accum.data(); }
const float*    oracle_pt_depth(OraclePt* o) { return o->depth.data(); }
const uint32_t* oracle_pt_selection(OraclePt* o) { return o->selection.data(); }
const float*    oracle_pt_albedo(OraclePt* o) { return o->albedo.data(); }
const float*    oracle_pt_normal(OraclePt* o) { return o->normal.data(); }
int             oracle_pt_get_stats(OraclePt* o, MiPtStats* s)
{
  memset(s, 0, sizeof(*s));
  s->cameraPaths      = o->counters.cameraPaths;
  s->segments         = o->counters.segments;
  s->shadowRays       = o->counters.shadowRays;
  s->nodesClosest     = o->counters.nodesClosest;
  s->trisClosest      = o->counters.trisClosest;
  s->nodesShadow      = o->counters.nodesShadow;
  s->trisShadow       = o->counters.trisShadow;
  s->textureTaps      = o->counters.textureTaps;
  s->surfaceHits      = o->counters.surfaceHits;
  s->bvhNodeCount     = o->accel.nodes.size();
  s->bvhTriangleCount = o->accel.tris.size();
  s->bvhNodeBytes     = sizeof(BvhNode);
  s->bvhTriangleBytes = sizeof(Tri);
  return 0;
}

// --- known-answer hooks for tests -------------------------------------------------------------------------------------
uint32_t oracle_xxhash32(uint32_t x, uint32_t y, uint32_t z) { return xxhash32(x, y, z); }
float    oracle_rand(uint32_t* seed) { return rnd(*seed); }
void     oracle_sky_eval(const MiSkyPhysicalParameters* s, const float* dir, float* rgb)
{
  float3 r = evalPhysicalSky(*s, normalize(float3(dir)));
  rgb[0] = r.x; rgb[1] = r.y; rgb[2] = r.z;
}
float oracle_sky_pdf(const MiSkyPhysicalParameters* s, const float* dir) { return samplePhysicalSkyPDF(*s, normalize(float3(dir))); }
void  oracle_sky_sample(const MiSkyPhysicalParameters* s, float u, float v, float* dirPdfRgb)
{
  SkySamplingResult r = samplePhysicalSky(*s, float2(u, v));
  dirPdfRgb[0] = r.direction.x; dirPdfRgb[1] = r.direction.y; dirPdfRgb[2] = r.direction.z; dirPdfRgb[3] = r.pdf;
  dirPdfRgb[4] = r.radiance.x; dirPdfRgb[5] = r.radiance.y; dirPdfRgb[6] = r.radiance.z;
}
// Evaluate / sample the layered BSDF of a material described by the flat array `m` (see oracle_pt.h).
static PbrMaterial materialFromArray(const float* m)
{
  PbrMaterial p;
  p.baseColor = float3(m[0], m[1], m[2]);
  p.roughness = float2(m[3], m[4]);
  p.metallic  = m[5];
  p.ior1 = m[6]; p.ior2 = m[7];
  p.specular = m[8];
  p.specularColor = float3(m[9], m[10], m[11]);
  p.transmission = m[12];
  p.thickness = m[13];
  p.clearcoat = m[14]; p.clearcoatRoughness = m[15];
  p.sheenColor = float3(m[16], m[17], m[18]); p.sheenRoughness = m[19];
  p.iridescence = m[20]; p.iridescenceIor = m[21]; p.iridescenceThickness = m[22];
  p.diffuseTransmissionFactor = m[23];
  p.diffuseTransmissionColor = float3(m[24], m[25], m[26]);
  p.dispersion = m[27];
  p.retroreflection = m[28];
  p.N = p.Ng = p.Nc = float3(0, 0, 1);
  p.T = float3(1, 0, 0);
  p.B = float3(0, 1, 0);
  return p;
}
void oracle_bsdf_eval(const float* m, const float* k1, const float* k2, const float* xi, float* out7)
{
  BsdfEvaluateData d;
  d.k1 = float3(k1); d.k2 = float3(k2); d.xi = float3(xi);
  bsdfEvaluate(d, materialFromArray(m));
  out7[0] = d.bsdf_diffuse.x; out7[1] = d.bsdf_diffuse.y; out7[2] = d.bsdf_diffuse.z;
  out7[3] = d.bsdf_glossy.x; out7[4] = d.bsdf_glossy.y; out7[5] = d.bsdf_glossy.z; out7[6] = d.pdf;
}
void oracle_bsdf_sample(const float* m, const float* k1, const float* xi, float* out8)
{
  BsdfSampleData d;
  d.k1 = float3(k1); d.xi = float3(xi);
  bsdfSample(d, materialFromArray(m));
  out8[0] = d.k2.x; out8[1] = d.k2.y; out8[2] = d.k2.z;
  out8[3] = d.bsdf_over_pdf.x; out8[4] = d.bsdf_over_pdf.y; out8[5] = d.bsdf_over_pdf.z; out8[6] = d.pdf;
  out8[7] = float(d.event_type);
}
float oracle_round_to_half(float f) { return roundToHalf(f); }
// ---- hooks for the closed-form pins of tests/test_oracle_pins.py (published formulas, tests/golden/pins_*.json)
float oracle_fresnel_dielectric_unpolarized(float eta, float cosTheta) { return ior_fresnel(eta, cosTheta); }
float oracle_fresnel_schlick(float ior, float cosTheta) { return schlickFresnelIor(ior, cosTheta); }
void  oracle_fresnel_conductor(float n_a, float n_b, float k_b, float cosTheta, float* rs_rp)
{
  float2 ps, pc;
  float2 r = fresnel_conductor(ps, pc, n_a, n_b, k_b, cosTheta, std::fmax(0.0f, 1.0f - cosTheta * cosTheta));
  rs_rp[0] = r.x; rs_rp[1] = r.y;
}
void oracle_thin_film(float thickness, float coatingIor, float baseIor, float incomingIor, float cosTheta, float* rgb)
{
  float3 c = thin_film_factor(thickness, coatingIor, baseIor, incomingIor, cosTheta);
  rgb[0] = c.x; rgb[1] = c.y; rgb[2] = c.z;
}
float oracle_ggx_ndf(float ax, float ay, const float* h) { return hvd_ggx_eval(float2(1.0f / ax, 1.0f / ay), float3(h)); }  // D(h) * cos(theta_h)
float oracle_ggx_g1(float ax, float ay, const float* k) { return smith_shadow_mask(float3(k), float2(ax, ay)); }
void  oracle_ggx_sample_vndf(float ax, float ay, const float* k, float u, float v, float* h)
{
  float3 r = hvd_ggx_sample_vndf(float3(k), float2(ax, ay), float2(u, v));
  h[0] = r.x; h[1] = r.y; h[2] = r.z;
}
float oracle_hg_pdf(float cosTheta, float g) { return henyeyGreensteinPdf(cosTheta, g); }
void  oracle_hg_sample(float u, float v, float g, const float* wi, float* wo)
{
  float3 r = sampleHenyeyGreenstein(float2(u, v), g, normalize(float3(wi)));
  wo[0] = r.x; wo[1] = r.y; wo[2] = r.z;
}
// ---- round 3: the remaining nvshaders pieces that were restated without a pin (tests/test_oracle_pins.py)
float oracle_sheen_ndf(float invRoughness, float nh) { return hvd_sheen_eval(invRoughness, nh); }  // density of h over solid angle
void  oracle_sheen_sample(float u, float v, float invRoughness, float* h)
{
  float3 r = hvd_sheen_sample(float2(u, v), invRoughness);
  h[0] = r.x; h[1] = r.y; h[2] = r.z;
}
float oracle_vcavities_g(float nh, float k1h, float k1z, float k2h, float k2z)
{
  float G1, G2;
  return vcavities_shadow_mask(G1, G2, nh, float3(0, 0, k1z), k1h, float3(0, 0, k2z), k2h);
}
void oracle_lobe_weights(const float* m, float VdotN, float* w6)  // LOBE_* order of computeLobeWeights
{
  PbrMaterial mat = materialFromArray(m);
  float3      tint = mat.baseColor;
  float       w[LOBE_COUNT];
  computeLobeWeights(mat, VdotN, tint, w);
  for(int i = 0; i < LOBE_COUNT && i < 6; ++i)
    w6[i] = w[i];
}
int  oracle_lobe_index(const char* name)
{
  auto is = [&](const char* s) { return std::strcmp(name, s) == 0; };
  if(is("diffuse")) return LOBE_DIFFUSE_REFLECTION;
  if(is("specular_transmission")) return LOBE_SPECULAR_TRANSMISSION;
  if(is("specular")) return LOBE_SPECULAR_REFLECTION;
  if(is("metal")) return LOBE_METAL_REFLECTION;
  if(is("sheen")) return LOBE_SHEEN_REFLECTION;
  if(is("clearcoat")) return LOBE_CLEARCOAT_REFLECTION;
  return -1;
}
void oracle_env_sample(const MiPtEnvironment* env, const float* xi, float* out7)  // direction, rgb, pdf of environmentSample
{
  Scene sc;
  sc.env = env;
  float3 dir;
  float4 r = environmentSample(sc, float3(xi), dir);
  out7[0] = dir.x; out7[1] = dir.y; out7[2] = dir.z; out7[3] = r.x; out7[4] = r.y; out7[5] = r.z; out7[6] = r.w;
}
void oracle_point_offset(const float* p, const float* tri9, const float* nrm9, const float* bary, float* out3)
{
  float3 r = pointOffset(float3(p), float3(tri9), float3(tri9 + 3), float3(tri9 + 6), float3(nrm9), float3(nrm9 + 3), float3(nrm9 + 6), float3(bary));
  out3[0] = r.x; out3[1] = r.y; out3[2] = r.z;
}
float oracle_ray_cone_footprint(float width, float spreadAngle, float hitT, const float* n, const float* v)
{
  RayCone c;
  c.width = width; c.spreadAngle = spreadAngle;
  return rayConeWorldFootprint(c, hitT, float3(n), float3(v));
}
void oracle_light_contribution(const MiGltfLight* light, const float* pos, const float* xi, float* out8)
{
  LightContrib c = singleLightContribution(*light, float3(pos), float3(0, 0, 1), float2(xi[0], xi[1]));
  out8[0] = c.incidentVector.x; out8[1] = c.incidentVector.y; out8[2] = c.incidentVector.z; out8[3] = c.distance;
  out8[4] = c.intensity.x; out8[5] = c.intensity.y; out8[6] = c.intensity.z; out8[7] = c.pdf;
}
}
