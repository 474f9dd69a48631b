// Layered glTF PBR BSDF for the shade kernel: stochastic single-lobe evaluate / sample with MIS-ready pdfs.
//
// Call contract = the reference's use of nvshaders `bsdfEvaluate` / `bsdfSample` (shaders/gltf_pathtrace.slang:330-416):
// k1 = direction to the viewer, k2 = direction to the light / next direction, xi = 3 uniforms, bsdf_* include |N.k2|.
// The bodies live in nvpro_core2 (not vendored in the reference); the model here follows its published structure
// (MDL-SDK libbsdf lineage): lobes {diffuse R, diffuse T, GGX dielectric R, GGX T, GGX metal R, sheen, clearcoat},
// one lobe picked with xi.z by Fresnel/metallic/transmission weights, VNDF sampling (Heitz 2017), Smith G, thin-film
// Fresnel for iridescence, per-channel IOR for dispersion, mirrored view vector for retroreflection.
#pragma once
#include "pt_math.h"

namespace pt {

struct PbrMaterial  // reference: nvshaders/pbr_material_types.h.slang (fields used at gltf_material_eval.h.slang:195-453)
{
  f3    baseColor;
  float opacity;
  f2    roughness;
  float metallic;
  f3    emissive;
  float occlusion;
  f3    N, T, B, Ng;
  float ior1, ior2;
  float specular;
  f3    specularColor;
  float transmission;
  f3    attenuationColor;
  float attenuationDistance;
  float thickness;
  f3    scatterCoefficient;
  float scatterAnisotropy;
  float clearcoat, clearcoatRoughness;
  f3    Nc;
  float iridescence, iridescenceIor, iridescenceThickness;
  f3    sheenColor;
  float sheenRoughness;
  float dispersion;
  float diffuseTransmissionFactor;
  f3    diffuseTransmissionColor;
  float retroreflection;
};

PT_DEV PbrMaterial defaultPbrMaterial()
{
  PbrMaterial m;
  m.baseColor = mk3(1.0f); m.opacity = 1.0f; m.roughness = mk2(1.0f, 1.0f); m.metallic = 1.0f;
  m.emissive = mk3(0.0f); m.occlusion = 1.0f;
  m.N = mk3(0, 0, 1); m.T = mk3(1, 0, 0); m.B = mk3(0, 1, 0); m.Ng = mk3(0, 0, 1);
  m.ior1 = 1.0f; m.ior2 = 1.5f; m.specular = 1.0f; m.specularColor = mk3(1.0f);
  m.transmission = 0.0f; m.attenuationColor = mk3(1.0f); m.attenuationDistance = 1.0f; m.thickness = 0.0f;
  m.scatterCoefficient = mk3(0.0f); m.scatterAnisotropy = 0.0f;
  m.clearcoat = 0.0f; m.clearcoatRoughness = 0.01f; m.Nc = mk3(0, 0, 1);
  m.iridescence = 0.0f; m.iridescenceIor = 1.5f; m.iridescenceThickness = 0.1f;
  m.sheenColor = mk3(0.0f); m.sheenRoughness = 0.0f; m.dispersion = 0.0f;
  m.diffuseTransmissionFactor = 0.0f; m.diffuseTransmissionColor = mk3(1.0f); m.retroreflection = 0.0f;
  return m;
}

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

struct BsdfEval
{
  f3    bsdf;  // diffuse * occlusion + glossy
  float pdf;
};
struct BsdfSample
{
  f3    k2, bsdf_over_pdf;
  float pdf;
  int   event_type;
};

// Orthonormal basis, Duff et al. 2017 (reference call site of makeFastTangent: get_hit.h.slang:139)
PT_DEV f4 makeFastTangent(f3 n)
{
  float sign = copysignf(1.0f, n.z);
  float a    = -1.0f / (sign + n.z);
  float b    = n.x * n.y * a;
  return mk4(1.0f + sign * n.x * n.x * a, sign * b, -sign * n.x, 1.0f);
}

PT_DEV float schlickFresnelIor(float ior, float VdotH)
{
  float R0 = sqr((1.0f - ior) / (1.0f + ior));
  const float m = 1.0f - VdotH, m2 = m * m;
  return R0 + (1.0f - R0) * (m2 * m2 * m);
}
PT_DEV f3 mix_rgb(f3 base, f3 layer, f3 factor) { return base * (1.0f - maxComp(factor)) + factor * layer; }
PT_DEV bool isTIR(f2 ior, float kh)
{
  float b = ior.x / ior.y;
  return 1.0f < (b * b * (1.0f - kh * kh));
}
PT_DEV float ior_fresnel(float eta, float kh)
{
  float costheta = 1.0f - (1.0f - kh * kh) / (eta * eta);
  if(costheta <= 0.0f)
    return 1.0f;
  costheta   = sqrtf(costheta);
  float n1t1 = kh, n1t2 = costheta, n2t1 = kh * eta, n2t2 = costheta * eta;
  float r_p = (n1t2 - n2t1) / (n1t2 + n2t1);
  float r_o = (n1t1 - n2t2) / (n1t1 + n2t2);
  return clampf(0.5f * (r_p * r_p + r_o * r_o), 0.0f, 1.0f);
}
PT_DEV f2 fresnel_dielectric(float n_a, float n_b, float cos_a, float cos_b)
{
  float naca = n_a * cos_a, nbcb = n_b * cos_b;
  float r_s  = (naca - nbcb) / (naca + nbcb);
  float nacb = n_a * cos_b, nbca = n_b * cos_a;
  float r_p  = (nbca - nacb) / (nbca + nacb);
  return mk2(r_s * r_s, r_p * r_p);
}
PT_DEV f2 fresnel_conductor(f2& phase_sin, f2& phase_cos, float n_a, float n_b, float k_b, float cos_a, float sin_a_sqd)
{
  float k_b2 = k_b * k_b, n_b2 = n_b * n_b, n_a2 = n_a * n_a;
  float tmp0   = n_b2 - k_b2;
  float half_U = 0.5f * (tmp0 - n_a2 * sin_a_sqd);
  float half_V = sqrtf(fmaxf(0.0f, half_U * half_U + k_b2 * n_b2));
  float u_b2 = half_U + half_V, v_b2 = half_V - half_U;
  float u_b = sqrtf(fmaxf(0.0f, u_b2)), v_b = sqrtf(fmaxf(0.0f, v_b2));
  float tmp1 = tmp0 * cos_a, tmp2 = n_a * u_b, tmp3 = (2.0f * n_b * k_b) * cos_a, tmp4 = n_a * v_b, tmp5 = n_a * cos_a;
  float tmp6 = (2.0f * tmp5) * v_b;
  float tmp7 = (u_b2 + v_b2) - tmp5 * tmp5;
  float tmp8 = (2.0f * tmp5) * ((2.0f * n_b * k_b) * u_b - tmp0 * v_b);
  float tmp9 = sqr((n_b2 + k_b2) * cos_a) - n_a2 * (u_b2 + v_b2);
  float tmp67 = tmp6 * tmp6 + tmp7 * tmp7;
  float inv_x = (0.0f < tmp67) ? 1.0f / sqrtf(tmp67) : 0.0f;
  float tmp89 = tmp8 * tmp8 + tmp9 * tmp9;
  float inv_y = (0.0f < tmp89) ? 1.0f / sqrtf(tmp89) : 0.0f;
  phase_cos   = mk2(tmp7 * inv_x, tmp9 * inv_y);
  phase_sin   = mk2(tmp6 * inv_x, tmp8 * inv_y);
  return mk2((sqr(tmp5 - u_b) + v_b2) / (sqr(tmp5 + u_b) + v_b2), (sqr(tmp1 - tmp2) + sqr(tmp3 - tmp4)) / (sqr(tmp1 + tmp2) + sqr(tmp3 + tmp4)));
}
// 16-wavelength spectral table, generated by tools/gen_thinfilm_table.py (Wyman et al. 2013 CMF fit -> Rec.709)
__device__ const float kThinFilmRgb[16][3] = {
    {1.141170327e-01f, -9.233230420e-02f, 6.243416750e-01f},   {3.891298474e-01f, -5.067071748e-01f, 4.143101779e+00f},
    {5.073899927e-01f, -6.406128719e-01f, 5.882926651e+00f},   {-2.440306838e-01f, -7.686629932e-02f, 4.887233973e+00f},
    {-8.422044385e-01f, 8.145953282e-01f, 1.993870357e+00f},   {-1.600349816e+00f, 2.098418188e+00f, 5.145780610e-01f},
    {-2.124796802e+00f, 3.884327156e+00f, -2.076202664e-01f},  {-1.257502337e+00f, 4.473430149e+00f, -4.918162342e-01f},
    {9.183988254e-01f, 3.804074233e+00f, -5.128916518e-01f},   {3.595737427e+00f, 2.365924612e+00f, -4.060246333e-01f},
    {5.558583049e+00f, 7.311976312e-01f, -2.452682962e-01f},   {5.496764639e+00f, -2.826184233e-01f, -1.104067936e-01f},
    {3.502888886e+00f, -3.898076749e-01f, -4.277698832e-02f},  {1.487011583e+00f, -1.669308389e-01f, -1.796845442e-02f},
    {4.212750925e-01f, -2.408596850e-02f, -8.143819362e-03f},  {7.758770274e-02f, 7.994258193e-03f, -3.135358120e-03f}};
__device__ __noinline__ f3 thin_film_factor(float coating_thickness, float coating_ior, float base_ior, float incoming_ior, float kh)
{
  coating_thickness = fmaxf(0.0f, coating_thickness);
  float sin0_sqr    = fmaxf(0.0f, 1.0f - kh * kh);
  float eta01       = incoming_ior / coating_ior;
  float sin1_sqr    = eta01 * eta01 * sin0_sqr;
  if(1.0f < sin1_sqr)
    return mk3(1.0f);
  float cos1 = sqrtf(fmaxf(0.0f, 1.0f - sin1_sqr));
  f2    R01  = fresnel_dielectric(incoming_ior, coating_ior, kh, cos1);
  f2    phi12_sin, phi12_cos;
  f2    R12      = fresnel_conductor(phi12_sin, phi12_cos, coating_ior, base_ior, 0.0f, cos1, sin1_sqr);
  float tmp      = (4.0f * K_PI) * coating_ior * coating_thickness * cos1;
  float R01R12_s = fmaxf(0.0f, R01.x * R12.x), r01r12_s = sqrtf(R01R12_s);
  float R01R12_p = fmaxf(0.0f, R01.y * R12.y), r01r12_p = sqrtf(R01R12_p);
  f3    rgb      = mk3(0.0f);
  float lambda   = 400.0f + 0.5f * 18.75f;
  for(int i = 0; i < 16; ++i)
  {
    float phi = tmp / lambda;
    float ps = sinf(phi), pc = cosf(phi);
    float cos_phi_s = pc * phi12_cos.x - ps * phi12_sin.x;
    float tmp_s     = 2.0f * r01r12_s * cos_phi_s;
    float R_s       = (R01.x + R12.x + tmp_s) / (1.0f + R01R12_s + tmp_s);
    float cos_phi_p = pc * phi12_cos.y - ps * phi12_sin.y;
    float tmp_p     = 2.0f * r01r12_p * cos_phi_p;
    float R_p       = (R01.y + R12.y + tmp_p) / (1.0f + R01R12_p + tmp_p);
    float R         = 0.5f * (R_s + R_p);
    rgb += mk3(kThinFilmRgb[i][0], kThinFilmRgb[i][1], kThinFilmRgb[i][2]) * R;
    lambda += 18.75f;
  }
  return clamp3(rgb * (1.0f / 16.0f), 0.0f, 1.0f);
}

PT_DEV f3 cosineSampleHemisphere(float r1, float r2)
{
  float r = sqrtf(r1);  // phi = 2 pi r2
  f3    d;
  d.x = r * cosTurns(r2);
  d.y = r * sinTurns(r2);
  d.z = sqrtf(fmaxf(0.0f, 1.0f - d.x * d.x - d.y * d.y));
  return d;
}
PT_DEV float hvd_ggx_eval(f2 invRoughness, f3 h)
{
  float x = h.x * invRoughness.x, y = h.y * invRoughness.y;
  float f = x * x + y * y + h.z * h.z;
  return K_1_OVER_PI * invRoughness.x * invRoughness.y * h.z / (f * f);
}
PT_DEV f3 hvd_ggx_sample_vndf(f3 k, f2 roughness, f2 xi)
{
  f3    v  = normalize(mk3(k.x * roughness.x, k.y * roughness.y, k.z));
  f3    t1 = (v.z < 0.99999f) ? normalize(cross(v, mk3(0, 0, 1))) : mk3(1, 0, 0);
  f3    t2 = cross(t1, v);
  float a  = 1.0f / (1.0f + v.z);
  float r  = sqrtf(xi.x);
  // phi = (xi.y < a) ? xi.y / a * pi : pi + (xi.y - a) / (1 - a) * pi, kept in revolutions for the hardware sine / cosine
  float turns = (xi.y < a) ? xi.y / a * 0.5f : 0.5f + (xi.y - a) / (1.0f - a) * 0.5f;
  float sp = sinTurns(turns), cp = cosTurns(turns);
  float p1 = r * cp;
  float p2 = r * sp * ((xi.y < a) ? 1.0f : v.z);
  f3    h  = t1 * p1 + t2 * p2 + v * sqrtf(fmaxf(0.0f, 1.0f - p1 * p1 - p2 * p2));
  h.x *= roughness.x;
  h.y *= roughness.y;
  h.z = fmaxf(0.0f, h.z);
  return normalize(h);
}
PT_DEV float smith_shadow_mask(f3 k, f2 roughness)
{
  float kz2 = k.z * k.z;
  if(kz2 == 0.0f)
    return 0.0f;
  float ax = k.x * roughness.x, ay = k.y * roughness.y;
  float inv_a2 = (ax * ax + ay * ay) / kz2;
  return 2.0f / (1.0f + sqrtf(1.0f + inv_a2));
}
PT_DEV float ggx_smith_shadow_mask(float& G1, float& G2, f3 k1, f3 k2, f2 roughness)
{
  G1 = smith_shadow_mask(k1, roughness);
  G2 = smith_shadow_mask(k2, roughness);
  return G1 * G2;
}
PT_DEV f3 refractDir(f3 k, f3 n, float b, float nk, bool& tir)
{
  float refraction = b * b * (1.0f - nk * nk);
  tir              = (1.0f <= refraction);
  return tir ? (n * (nk + nk) - k) : normalize(k * (-b) + n * (b * nk - sqrtf(1.0f - refraction)));
}
PT_DEV f3 compute_half_vector(f3 k1, f3 k2, f3 normal, f2 ior, float nk2, bool transmission, bool thinwalled)
{
  f3 h;
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
PT_DEV float hvd_sheen_eval(float invRoughness, float nh)
{
  float sinTheta = sqrtf(fmaxf(0.0f, 1.0f - nh * nh));
  return (invRoughness + 2.0f) * powf(sinTheta, invRoughness) * 0.5f * K_1_OVER_PI * nh;
}
PT_DEV float vcavities_mask(float nh, float kh, float nk) { return fminf(2.0f * nh * nk / kh, 1.0f); }
PT_DEV float vcavities_shadow_mask(float& G1, float& G2, float nh, f3 k1, float k1h, f3 k2, float k2h)
{
  G1 = vcavities_mask(nh, k1h, k1.z);
  G2 = vcavities_mask(nh, k2h, k2.z);
  return fminf(G1, G2);
}
PT_DEV f3 hvd_sheen_sample(f2 xi, float invRoughness)
{
  float sinTheta = powf(1.0f - xi.y, 1.0f / (invRoughness + 2.0f));
  float cosTheta = sqrtf(fmaxf(0.0f, 1.0f - sinTheta * sinTheta));
  return normalize(mk3(cosTurns(xi.x) * sinTheta, sinTurns(xi.x) * sinTheta, cosTheta));  // phi = 2 pi xi.x
}
PT_DEV f3 flipH(f3 h, f3 k, float xi)
{
  float a = h.z * k.z, b = h.x * k.x + h.y * k.y;
  float kh = fmaxf(0.0f, a + b), kh_f = fmaxf(0.0f, a - b);
  float p_flip = kh_f / (kh + kh_f);
  return (xi < p_flip) ? mk3(-h.x, -h.y, h.z) : h;
}
PT_DEV f3 absorptionCoefficient(const PbrMaterial& mat)
{
  float d = mat.attenuationDistance;
  return d <= 0.0f ? mk3(0.0f) : -log3(mat.attenuationColor) / d;
}
PT_DEV f3 volumeExtinctionCoefficient(const PbrMaterial& mat) { return absorptionCoefficient(mat) + mat.scatterCoefficient; }

// --- lobe selection ------------------------------------------------------------------------------------------------------
struct LobePick
{
  int   lobe;
  float u;
  f3    tint;
};
PT_DEV LobePick findLobe(const PbrMaterial& mat, float VdotN, float rndVal)
{
  LobePick p;
  p.tint       = mat.baseColor;
  float frCoat = 0.0f;
  if(mat.clearcoat > 0.0f)
    frCoat = mat.clearcoat * ior_fresnel(1.5f / mat.ior1, VdotN);
  float frDielectric = 0.0f;
  if(mat.specular > 0.0f)
    frDielectric = ior_fresnel(mat.ior2 / mat.ior1, VdotN) * mat.specular;
  if(mat.iridescence > 0.0f)
  {
    f3 frIrid    = thin_film_factor(mat.iridescenceThickness, mat.iridescenceIor, mat.ior2, mat.ior1, VdotN);
    frDielectric = lerpf(frDielectric, maxComp(frIrid), mat.iridescence);
    p.tint       = mix_rgb(p.tint, mat.specularColor, frIrid * mat.iridescence);
  }
  float sheen = 0.0f;
  if(mat.sheenColor.x != 0.0f || mat.sheenColor.y != 0.0f || mat.sheenColor.z != 0.0f)
  {
    sheen = powf(1.0f - fabsf(VdotN), mat.sheenRoughness);
    sheen = sheen / (sheen + 0.5f);
  }
  float w[LOBE_COUNT];
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
  // walk 5,4,3,2,1 accumulating weights; 0 (diffuse) takes the remainder — same order as the upstream findLobe
  int   lobe   = 0;
  float weight = 0.0f, lo = 0.0f, hi = 1.0f;
  bool  found  = false;
#pragma unroll
  for(int l = LOBE_COUNT - 1; l > 0; --l)
  {
    if(!found)
    {
      lo = weight;
      weight += w[l];
      if(w[l] > 0.0f && rndVal < weight)  // (w == 0 can never be picked: rndVal >= the running weight here)
      {
        found = true;
        lobe  = l;
        hi    = weight;
      }
    }
  }
  if(!found)
  {
    lo = weight;
    hi = 1.0f;
  }
  p.lobe = lobe;
  p.u    = (hi > lo) ? clampf((rndVal - lo) / (hi - lo), 0.0f, 0.99999994f) : 0.0f;
  return p;
}
PT_DEV f3 retroView(const PbrMaterial& mat, f3 k1, f3 N, float& u)
{
  if(mat.retroreflection > 0.0f && splitRandom(u, mat.retroreflection))
    return N * (2.0f * dot(N, k1)) - k1;
  return k1;
}
PT_DEV f3 applyDispersion(PbrMaterial& mat, float& u)
{
  if(mat.dispersion <= 0.0f)
    return mk3(1.0f);
  int c            = min(int(u * 3.0f), 2);
  u                = u * 3.0f - float(c);
  bool  outside    = (mat.ior1 == 1.0f);
  float ior        = outside ? mat.ior2 : mat.ior1;
  float halfSpread = (ior - 1.0f) * 0.025f * mat.dispersion;
  float iorC       = ior + halfSpread * float(c - 1);
  if(outside)
    mat.ior2 = iorC;
  else
    mat.ior1 = iorC;
  return mk3(c == 0 ? 3.0f : 0.0f, c == 1 ? 3.0f : 0.0f, c == 2 ? 3.0f : 0.0f);
}

// --- lobes -----------------------------------------------------------------------------------------------------------------
PT_DEV BsdfEval evalAbsorb() { return BsdfEval{mk3(0.0f), 0.0f}; }
PT_DEV void sampleAbsorb(BsdfSample& d)
{
  d.bsdf_over_pdf = mk3(0.0f);
  d.pdf           = 0.0f;
  d.event_type    = BSDF_EVENT_ABSORB;
}

PT_DEV f3 ggxTint(const PbrMaterial& mat, int lobe, f3 tint, float k1h)
{
  if(lobe == LOBE_METAL_REFLECTION)  // glTF 2.0 Appendix B: metal F = baseColor + (1 - baseColor)(1 - VdotH)^5
    tint = tint + (mk3(1.0f) - tint) * powf(1.0f - fabsf(k1h), 5.0f);
  if(mat.iridescence > 0.0f && (lobe == LOBE_SPECULAR_REFLECTION || lobe == LOBE_METAL_REFLECTION))
  {
    f3 factor = thin_film_factor(mat.iridescenceThickness, mat.iridescenceIor, mat.ior2, mat.ior1, k1h);
    if(lobe == LOBE_SPECULAR_REFLECTION)
      tint *= lerp3(mk3(1.0f), factor, mat.iridescence);
    else
      tint = mix_rgb(tint, mat.specularColor, factor * mat.iridescence);
  }
  return tint;
}
PT_DEV BsdfEval brdf_ggx_smith_eval(f3 k1, f3 k2, const PbrMaterial& mat, f3 N, f2 roughness, int lobe, f3 tint)
{
  float nk1 = fabsf(dot(k1, N)), nk2 = fabsf(dot(k2, N));
  if(dot(k2, mat.Ng) <= 0.0f)
    return evalAbsorb();
  f3    h  = normalize(k1 + k2);
  float nh = dot(N, h), k1h = dot(k1, h), k2h = dot(k2, h);
  if(nk1 <= 0.0f || nh <= 0.0f || k1h < 0.0f || k2h < 0.0f)
    return evalAbsorb();
  f3    h0  = mk3(dot(mat.T, h), dot(mat.B, h), nh);
  float pdf = hvd_ggx_eval(mk2(1.0f / roughness.x, 1.0f / roughness.y), h0);
  float G1, G2;
  float G12 = ggx_smith_shadow_mask(G1, G2, mk3(dot(mat.T, k1), dot(mat.B, k1), nk1), mk3(dot(mat.T, k2), dot(mat.B, k2), nk2), roughness);
  pdf *= 0.25f / (nk1 * nh);
  f3 bsdf = mk3(G12 * pdf);
  pdf *= G1;
  return BsdfEval{bsdf * ggxTint(mat, lobe, tint, k1h), pdf};
}
PT_DEV void brdf_ggx_smith_sample(BsdfSample& d, f3 k1, f2 xi, const PbrMaterial& mat, f3 N, f3 T, f3 B, f2 roughness, int lobe, f3 tint)
{
  float nk1 = dot(k1, N);
  if(nk1 <= 0.0f)
    return sampleAbsorb(d);
  f3 k10 = mk3(dot(k1, T), dot(k1, B), nk1);
  f3 h0  = hvd_ggx_sample_vndf(k10, roughness, xi);
  if(fabsf(h0.z) == 0.0f)
    return sampleAbsorb(d);
  f3    h  = T * h0.x + B * h0.y + N * h0.z;
  float kh = dot(k1, h);
  if(kh <= 0.0f)
    return sampleAbsorb(d);
  d.k2            = h * (2.0f * kh) - k1;
  d.bsdf_over_pdf = mk3(1.0f);
  d.event_type    = BSDF_EVENT_GLOSSY_REFLECTION;
  if(dot(d.k2, mat.Ng) <= 0.0f)
    return sampleAbsorb(d);
  float nk2 = fabsf(dot(d.k2, N));
  float G1, G2;
  float G12 = ggx_smith_shadow_mask(G1, G2, k10, mk3(dot(d.k2, T), dot(d.k2, B), nk2), roughness);
  if(G12 <= 0.0f)
    return sampleAbsorb(d);
  d.bsdf_over_pdf *= G12 / G1;
  d.pdf = hvd_ggx_eval(mk2(1.0f / roughness.x, 1.0f / roughness.y), h0) * G1;
  d.pdf *= 0.25f / (nk1 * h0.z);
  d.bsdf_over_pdf *= ggxTint(mat, lobe, tint, kh);
}
PT_DEV BsdfEval btdf_ggx_smith_eval(f3 k1, f3 k2, const PbrMaterial& mat, f3 tint)
{
  bool  thin = (mat.thickness == 0.0f);
  f2    ior  = mk2(mat.ior1, mat.ior2);
  float nk1 = fabsf(dot(k1, mat.N)), nk2 = fabsf(dot(k2, mat.N));
  bool  backside = (dot(k2, mat.Ng) < 0.0f);
  f3    h        = compute_half_vector(k1, k2, mat.N, ior, nk2, backside, thin);
  float nh = dot(mat.N, h), k1h = dot(k1, h), k2h = dot(k2, h) * (backside ? -1.0f : 1.0f);
  if(nk1 <= 0.0f || nh <= 0.0f || k1h < 0.0f || k2h < 0.0f)
    return evalAbsorb();
  float fr;
  if(!backside)
  {
    if(!isTIR(ior, k1h))
      return evalAbsorb();
    fr = 1.0f;
  }
  else
    fr = 0.0f;
  f3    h0  = mk3(dot(mat.T, h), dot(mat.B, h), nh);
  float pdf = hvd_ggx_eval(mk2(1.0f / mat.roughness.x, 1.0f / mat.roughness.y), h0);
  float G1, G2;
  float G12 = ggx_smith_shadow_mask(G1, G2, mk3(dot(mat.T, k1), dot(mat.B, k1), nk1), mk3(dot(mat.T, k2), dot(mat.B, k2), nk2), mat.roughness);
  if(!thin && backside)
  {
    float tmp = k1h * ior.x - k2h * ior.y;
    pdf *= k1h * k2h * ior.y * ior.y / (nk1 * nh * tmp * tmp);  // Walter et al. 2007 refraction Jacobian
  }
  else
    pdf *= 0.25f / (nk1 * nh);
  float prob = backside ? 1.0f - fr : fr;
  f3    bsdf = mk3(prob * G12 * pdf);
  pdf *= prob * G1;
  return BsdfEval{bsdf * tint, pdf};
}
PT_DEV void btdf_ggx_smith_sample(BsdfSample& d, f3 k1, f2 xi, const PbrMaterial& mat, f3 tint)
{
  bool  thin = (mat.thickness == 0.0f);
  f2    ior  = mk2(mat.ior1, mat.ior2);
  float nk1  = fabsf(dot(k1, mat.N));
  f3    k10  = mk3(dot(k1, mat.T), dot(k1, mat.B), nk1);
  f3    h0   = hvd_ggx_sample_vndf(k10, mat.roughness, xi);
  if(fabsf(h0.z) == 0.0f)
    return sampleAbsorb(d);
  f3    h  = mat.T * h0.x + mat.B * h0.y + mat.N * h0.z;
  float kh = dot(k1, h);
  if(kh <= 0.0f)
    return sampleAbsorb(d);
  bool tir = false;
  if(thin)
  {
    d.k2 = h * (2.0f * kh) - k1;
    d.k2 = normalize(d.k2 - mat.N * (2.0f * dot(d.k2, mat.N)));
  }
  else
    d.k2 = refractDir(k1, h, ior.x / ior.y, kh, tir);
  d.bsdf_over_pdf = mk3(1.0f);
  d.event_type    = tir ? BSDF_EVENT_GLOSSY_REFLECTION : BSDF_EVENT_GLOSSY_TRANSMISSION;
  float gnk2      = dot(d.k2, mat.Ng) * ((d.event_type == BSDF_EVENT_GLOSSY_REFLECTION) ? 1.0f : -1.0f);
  if(gnk2 <= 0.0f)
    return sampleAbsorb(d);
  float nk2 = fabsf(dot(d.k2, mat.N)), k2h = fabsf(dot(d.k2, h));
  float G1, G2;
  float G12 = ggx_smith_shadow_mask(G1, G2, k10, mk3(dot(d.k2, mat.T), dot(d.k2, mat.B), nk2), mat.roughness);
  if(G12 <= 0.0f)
    return sampleAbsorb(d);
  d.bsdf_over_pdf *= G12 / G1;
  d.pdf = hvd_ggx_eval(mk2(1.0f / mat.roughness.x, 1.0f / mat.roughness.y), h0) * G1;
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
PT_DEV BsdfEval brdf_sheen_eval(f3 k1, f3 k2, const PbrMaterial& mat)
{
  if(dot(k2, mat.Ng) <= 0.0f)
    return evalAbsorb();
  float nk1 = fabsf(dot(k1, mat.N)), nk2 = fabsf(dot(k2, mat.N));
  f3    h   = normalize(k1 + k2);
  float nh = dot(mat.N, h), k1h = dot(k1, h), k2h = dot(k2, h);
  if(nk1 <= 0.0f || nh <= 0.0f || k1h < 0.0f || k2h < 0.0f)
    return evalAbsorb();
  float invRoughness = 1.0f / (mat.sheenRoughness * mat.sheenRoughness);
  float pdf          = hvd_sheen_eval(invRoughness, nh);
  float G1, G2;
  float G12 = vcavities_shadow_mask(G1, G2, nh, mk3(dot(mat.T, k1), dot(mat.B, k1), nk1), k1h, mk3(dot(mat.T, k2), dot(mat.B, k2), nk2), k2h);
  pdf *= 0.25f / (nk1 * nh);
  f3 bsdf = mk3(G12 * pdf);
  pdf *= G1;
  return BsdfEval{bsdf * mat.sheenColor, pdf};
}
PT_DEV void brdf_sheen_sample(BsdfSample& d, f3 k1, f2 xi, const PbrMaterial& mat, float xiFlip)
{
  float nk1 = dot(k1, mat.N);
  if(nk1 <= 0.0f)
    return sampleAbsorb(d);
  f3    k10          = mk3(dot(k1, mat.T), dot(k1, mat.B), nk1);
  float invRoughness = 1.0f / (mat.sheenRoughness * mat.sheenRoughness);
  f3    h0           = flipH(hvd_sheen_sample(xi, invRoughness), k10, xiFlip);
  if(fabsf(h0.z) == 0.0f)
    return sampleAbsorb(d);
  f3    h   = mat.T * h0.x + mat.B * h0.y + mat.N * h0.z;
  float k1h = dot(k1, h);
  if(k1h <= 0.0f)
    return sampleAbsorb(d);
  d.k2            = h * (2.0f * k1h) - k1;
  d.bsdf_over_pdf = mk3(1.0f);
  d.event_type    = BSDF_EVENT_GLOSSY_REFLECTION;
  if(dot(d.k2, mat.Ng) <= 0.0f)
    return sampleAbsorb(d);
  float nk2 = fabsf(dot(d.k2, mat.N)), k2h = fabsf(dot(d.k2, h));
  float G1, G2;
  float G12 = vcavities_shadow_mask(G1, G2, h0.z, k10, k1h, mk3(dot(d.k2, mat.T), dot(d.k2, mat.B), nk2), k2h);
  if(G12 <= 0.0f)
    return sampleAbsorb(d);
  d.bsdf_over_pdf *= G12 / G1;
  d.pdf = hvd_sheen_eval(invRoughness, h0.z) * G1;
  d.pdf *= 0.25f / (nk1 * h0.z);
  d.bsdf_over_pdf *= mat.sheenColor;
}

// nvshaders bsdfEvaluate (call site gltf_pathtrace.slang:333-349)
PT_DEV BsdfEval bsdfEvaluate(f3 k1, f3 k2, f3 xi, PbrMaterial mat)
{
  float    VdotN = dot(k1, mat.N);
  LobePick pick  = findLobe(mat, VdotN, xi.z);
  float    u     = pick.u;
  BsdfEval e     = evalAbsorb();
  switch(pick.lobe)
  {
    case LOBE_DIFFUSE_REFLECTION:
      if(mat.diffuseTransmissionFactor > 0.0f && splitRandom(u, mat.diffuseTransmissionFactor))
      {
        if(dot(k2, mat.Ng) < 0.0f)
        {
          e.pdf  = fmaxf(0.0f, -dot(k2, mat.N) * K_1_OVER_PI);
          e.bsdf = mat.diffuseTransmissionColor * e.pdf * mat.occlusion;
        }
      }
      else if(dot(k2, mat.Ng) > 0.0f)
      {
        e.pdf  = fmaxf(0.0f, dot(k2, mat.N) * K_1_OVER_PI);
        e.bsdf = pick.tint * e.pdf * mat.occlusion;
      }
      break;
    case LOBE_SPECULAR_REFLECTION:
      k1 = retroView(mat, k1, mat.N, u);
      e  = brdf_ggx_smith_eval(k1, k2, mat, mat.N, mat.roughness, LOBE_SPECULAR_REFLECTION, mat.specularColor);
      break;
    case LOBE_SPECULAR_TRANSMISSION: {
      f3 mask = applyDispersion(mat, u);
      e       = btdf_ggx_smith_eval(k1, k2, mat, pick.tint * mask);
      break;
    }
    case LOBE_METAL_REFLECTION:
      k1 = retroView(mat, k1, mat.N, u);
      e  = brdf_ggx_smith_eval(k1, k2, mat, mat.N, mat.roughness, LOBE_METAL_REFLECTION, mat.baseColor);
      break;
    case LOBE_CLEARCOAT_REFLECTION:
      mat.iridescence = 0.0f;
      k1              = retroView(mat, k1, mat.Nc, u);
      e = brdf_ggx_smith_eval(k1, k2, mat, mat.Nc, mk2(mat.clearcoatRoughness * mat.clearcoatRoughness, mat.clearcoatRoughness * mat.clearcoatRoughness),
                              LOBE_CLEARCOAT_REFLECTION, mk3(1.0f));
      break;
    case LOBE_SHEEN_REFLECTION:
      k1 = retroView(mat, k1, mat.N, u);
      e  = brdf_sheen_eval(k1, k2, mat);
      break;
  }
  return e;
}
// nvshaders bsdfSample (call site gltf_pathtrace.slang:359-368)
PT_DEV BsdfSample bsdfSample(f3 k1, f3 xi, PbrMaterial mat)
{
  float      VdotN = dot(k1, mat.N);
  LobePick   pick  = findLobe(mat, VdotN, xi.z);
  float      u     = pick.u;
  BsdfSample d;
  d.k2 = mk3(0.0f);
  sampleAbsorb(d);
  f2 xi2 = mk2(xi.x, xi.y);
  switch(pick.lobe)
  {
    case LOBE_DIFFUSE_REFLECTION: {
      bool  trans = mat.diffuseTransmissionFactor > 0.0f && splitRandom(u, mat.diffuseTransmissionFactor);
      f3    l     = cosineSampleHemisphere(xi.x, xi.y);
      float s     = trans ? -1.0f : 1.0f;
      d.k2        = normalize(mat.T * l.x + mat.B * l.y + mat.N * (s * l.z));
      d.pdf       = s * dot(d.k2, mat.N) * K_1_OVER_PI;
      d.bsdf_over_pdf = trans ? mat.diffuseTransmissionColor : pick.tint;
      float g         = s * dot(d.k2, mat.Ng);
      d.event_type    = (0.0f < g) ? (trans ? BSDF_EVENT_DIFFUSE_TRANSMISSION : BSDF_EVENT_DIFFUSE_REFLECTION) : BSDF_EVENT_ABSORB;
      break;
    }
    case LOBE_SPECULAR_REFLECTION:
      k1 = retroView(mat, k1, mat.N, u);
      brdf_ggx_smith_sample(d, k1, xi2, mat, mat.N, mat.T, mat.B, mat.roughness, LOBE_SPECULAR_REFLECTION, mat.specularColor);
      break;
    case LOBE_SPECULAR_TRANSMISSION: {
      f3 mask = applyDispersion(mat, u);
      btdf_ggx_smith_sample(d, k1, xi2, mat, pick.tint * mask);
      break;
    }
    case LOBE_METAL_REFLECTION:
      k1 = retroView(mat, k1, mat.N, u);
      brdf_ggx_smith_sample(d, k1, xi2, mat, mat.N, mat.T, mat.B, mat.roughness, LOBE_METAL_REFLECTION, mat.baseColor);
      break;
    case LOBE_CLEARCOAT_REFLECTION: {
      mat.iridescence = 0.0f;
      f3 Bc           = normalize(cross(mat.Nc, mat.T));
      f3 Tc           = cross(Bc, mat.Nc);
      k1              = retroView(mat, k1, mat.Nc, u);
      float r         = mat.clearcoatRoughness * mat.clearcoatRoughness;
      brdf_ggx_smith_sample(d, k1, xi2, mat, mat.Nc, Tc, Bc, mk2(r, r), LOBE_CLEARCOAT_REFLECTION, mk3(1.0f));
      break;
    }
    case LOBE_SHEEN_REFLECTION:
      k1 = retroView(mat, k1, mat.N, u);
      brdf_sheen_sample(d, k1, xi2, mat, u);
      break;
  }
  // NaN guard: a degenerate sample ends the path instead of poisoning the accumulator
  if(!(d.pdf == d.pdf) || !(d.bsdf_over_pdf.x == d.bsdf_over_pdf.x) || !(d.bsdf_over_pdf.y == d.bsdf_over_pdf.y)
     || !(d.bsdf_over_pdf.z == d.bsdf_over_pdf.z) || !(d.k2.x == d.k2.x) || !(d.k2.y == d.k2.y) || !(d.k2.z == d.k2.z))
    sampleAbsorb(d);
  return d;
}
// nvshaders bsdfSampleSimple (call site pathtrace_functions.h.slang:537-551)
PT_DEV BsdfSample bsdfSampleSimple(f3 k1, f3 xi, const PbrMaterial& mat)
{
  f3         tint  = mat.baseColor;
  float      VdotN = dot(k1, mat.N);
  float      F     = lerpf(schlickFresnelIor(mat.ior2 / mat.ior1, fabsf(VdotN)), 1.0f, mat.metallic);
  BsdfSample d;
  d.k2 = mk3(0.0f);
  sampleAbsorb(d);
  if(xi.z < F)
    brdf_ggx_smith_sample(d, k1, mk2(xi.x, xi.y), mat, mat.N, mat.T, mat.B, mat.roughness,
                          mat.metallic > 0.5f ? LOBE_METAL_REFLECTION : LOBE_SPECULAR_REFLECTION, mat.metallic > 0.5f ? tint : mk3(1.0f));
  else
  {
    f3 l            = cosineSampleHemisphere(xi.x, xi.y);
    d.k2            = normalize(mat.T * l.x + mat.B * l.y + mat.N * l.z);
    d.pdf           = dot(d.k2, mat.N) * K_1_OVER_PI;
    d.bsdf_over_pdf = tint * (1.0f - mat.metallic);
    d.event_type    = (0.0f < dot(d.k2, mat.Ng)) ? BSDF_EVENT_DIFFUSE_REFLECTION : BSDF_EVENT_ABSORB;
  }
  return d;
}

PT_DEV float henyeyGreensteinPdf(float cosTheta, float g)
{
  float denom = 1.0f + g * g - 2.0f * g * cosTheta;
  return (1.0f - g * g) / (4.0f * K_PI * denom * sqrtf(fmaxf(denom, 1e-12f)));
}
PT_DEV f3 sampleHenyeyGreenstein(f2 xi, float g, f3 wi)
{
  float cosTheta;
  if(fabsf(g) < 1e-3f)
    cosTheta = 1.0f - 2.0f * xi.x;
  else
  {
    float s  = (1.0f - g * g) / (1.0f - g + 2.0f * g * xi.x);
    cosTheta = (1.0f + g * g - s * s) / (2.0f * g);
  }
  cosTheta       = clampf(cosTheta, -1.0f, 1.0f);
  float sinTheta = sqrtf(fmaxf(0.0f, 1.0f - cosTheta * cosTheta));
  f3    T        = xyz(makeFastTangent(wi));
  f3    B        = cross(wi, T);
  return normalize(T * (sinTheta * cosTurns(xi.y)) + B * (sinTheta * sinTurns(xi.y)) + wi * cosTheta);  // phi = 2 pi xi.y
}

}  // namespace pt
