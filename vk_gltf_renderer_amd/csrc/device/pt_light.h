// Textures, HDR environment, physical sky and punctual lights on the device.
// Reference consumers: shaders/pathtrace_functions.h.slang:379-492 (sampleLights / sampleEnvironment),
// shaders/gltf_material_eval.h.slang:76-110 (getTexture).  The sampled functions themselves (texture unit,
// nvshaders sky / hdr / light_contrib) are not in the reference tree; CDNA4 has no texture unit, so filtering is
// done in software following the Vulkan sampling rules (Vulkan 1.3 spec §16.5-16.9).
#pragma once
#include "pt_bsdf.h"
#include "pt_scene.h"

namespace pt {

// ---- textures -------------------------------------------------------------------------------------------------------------
// Sampler address modes.  Texture sizes are almost always powers of two, where the modulo is a mask; the general integer
// modulo (some thirty instructions on this hardware, four of them per bilinear fetch) is kept for the other sizes.
PT_DEV int posMod(int i, int n)
{
  if((n & (n - 1)) == 0)
    return i & (n - 1);
  if(i >= -n && i < 2 * n)  // one period either side (a lat-long environment map, uv in [0, 1] +- a texel)
    return i < 0 ? i + n : (i >= n ? i - n : i);
  const int m = i % n;
  return m < 0 ? m + n : m;
}
PT_DEV int wrapCoord(int i, int n, int mode)
{
  if(mode == MI_WRAP_CLAMP_TO_EDGE)
    return min(max(i, 0), n - 1);
  if(mode == MI_WRAP_MIRRORED_REPEAT)
  {
    const int period = 2 * n;
    const int m      = posMod(i, period);
    return m < n ? m : period - 1 - m;
  }
  return posMod(i, n);
}
// wrapCoord(i) and wrapCoord(i + 1) of a bilinear footprint
PT_DEV void wrapCoordPair(int i, int n, int mode, int& c0, int& c1)
{
  c0 = wrapCoord(i, n, mode);
  if(mode == MI_WRAP_REPEAT)
    c1 = c0 + 1 == n ? 0 : c0 + 1;
  else
    c1 = wrapCoord(i + 1, n, mode);
}
// texel index inside the pool: 32-bit with a 24-bit multiply (a level is at most 16384 texels wide, the pool offsets are 32-bit)
PT_DEV uint32_t texelIndex(uint32_t levelOffset, int w, int x, int y) { return levelOffset + __umul24(uint32_t(y), uint32_t(w)) + uint32_t(x); }
// `lut` = the 256-entry sRGB decode table to use (sc.srgbLut, or a copy of it that the caller staged in LDS)
PT_DEV f4 fetchTexel(const DevScene& sc, const float* lut, const DevTexture& t, int level, int w, int x, int y)
{
  uchar4 p = gat(sc.texels, texelIndex(t.levelOffset[level], w, x, y));
  if(t.srgb)
    return mk4(lut[p.x], lut[p.y], lut[p.z], float(p.w) * (1.0f / 255.0f));
  return mk4(float(p.x) * (1.0f / 255.0f), float(p.y) * (1.0f / 255.0f), float(p.z) * (1.0f / 255.0f), float(p.w) * (1.0f / 255.0f));
}
PT_DEV f4 sampleLevel(const DevScene& sc, const float* lut, const DevTexture& t, f2 uv, int level, int filter)
{
  int   w = max(1, int(t.width) >> level), h = max(1, int(t.height) >> level);
  float fx = uv.x * float(w), fy = uv.y * float(h);
  if(filter == MI_FILTER_NEAREST)
    return fetchTexel(sc, lut, t, level, w, wrapCoord(int(floorf(fx)), w, t.wrapS), wrapCoord(int(floorf(fy)), h, t.wrapT));
  fx -= 0.5f;
  fy -= 0.5f;
  float flx = floorf(fx), fly = floorf(fy);
  float tx = fx - flx, ty = fy - fly;
  int   x0, x1, y0, y1;
  wrapCoordPair(int(flx), w, t.wrapS, x0, x1);
  wrapCoordPair(int(fly), h, t.wrapT, y0, y1);
  f4    a = fetchTexel(sc, lut, t, level, w, x0, y0), b = fetchTexel(sc, lut, t, level, w, x1, y0);
  f4    c = fetchTexel(sc, lut, t, level, w, x0, y1), d = fetchTexel(sc, lut, t, level, w, x1, y1);
  return (a * (1.0f - tx) + b * tx) * (1.0f - ty) + (c * (1.0f - tx) + d * tx) * ty;
}
// SampleLevel(uv, 0) when !useGrad, SampleGrad(uv, ddx, ddy) otherwise
PT_DEV f4 sampleTexture(const DevScene& sc, const float* lut, int texIndex, f2 uv, bool useGrad, f2 ddx, f2 ddy)
{
  if(texIndex < 0 || texIndex >= sc.numTextures)
    return mk4(1.0f);
  const DevTexture& t  = gat(sc.textures, texIndex);  // by reference: a by-value copy would put levelOffset[] in scratch
  float             lod = 0.0f;
  if(useGrad)
  {
    float rx  = sqrtf(sqr(ddx.x * float(t.width)) + sqr(ddx.y * float(t.height)));
    float ry  = sqrtf(sqr(ddy.x * float(t.width)) + sqr(ddy.y * float(t.height)));
    float rho = fmaxf(rx, ry);
    lod       = rho > 0.0f ? log2f(rho) : -126.0f;
  }
  if(lod <= 0.0f)
    return sampleLevel(sc, lut, t, uv, 0, t.magFilter);
  float maxLevel = float(int(t.numLevels) - 1);
  lod            = fminf(lod, maxLevel);
  if(t.mipmapMode == MI_FILTER_NEAREST)
    return sampleLevel(sc, lut, t, uv, min(int(floorf(lod + 0.5f)), int(t.numLevels) - 1), t.minFilter);
  int   l0 = int(floorf(lod)), l1 = min(l0 + 1, int(t.numLevels) - 1);
  float f  = lod - float(l0);
  f4    a  = sampleLevel(sc, lut, t, uv, l0, t.minFilter);
  if(f == 0.0f || l1 == l0)
    return a;
  f4 b = sampleLevel(sc, lut, t, uv, l1, t.minFilter);
  return a * (1.0f - f) + b * f;
}

PT_DEV f4 sampleTexture(const DevScene& sc, int texIndex, f2 uv, bool useGrad, f2 ddx, f2 ddy)
{
  return sampleTexture(sc, sc.srgbLut, texIndex, uv, useGrad, ddx, ddy);
}

// ---- HDR environment ------------------------------------------------------------------------------------------------------
PT_DEV f2 getSphericalUv(f3 v)
{
  float gamma = asinf(clampf(-v.y, -1.0f, 1.0f));
  float theta = atan2f(v.z, v.x);
  return mk2(theta * (0.5f * K_1_OVER_PI) + 0.5f, gamma * K_1_OVER_PI + 0.5f);
}
PT_DEV f3 rotateAxis(f3 v, f3 k, float theta)
{
  // theta is a per-frame constant (envRotation): a wave-uniform branch.  With theta = 0 the formula below returns v itself
  // (v * 1 + x * 0 + k * 0), so skipping it changes no result, and it saves the sine / cosine (some hundred vector
  // instructions) that every environment lookup of every lane would otherwise spend on the default setting.
  if(theta == 0.0f)
    return v;
  float c = cosf(theta), s = sinf(theta);
  return v * c + cross(k, v) * s + k * (dot(k, v) * (1.0f - c));
}
PT_DEV f4 sampleHdr(const DevScene& sc, f2 uv)
{
  if(!sc.envPixels)
    return mk4(0.0f);
  int   w = sc.envWidth, h = sc.envHeight;
  float fx = uv.x * float(w) - 0.5f, fy = uv.y * float(h) - 0.5f;
  float flx = floorf(fx), fly = floorf(fy);
  float tx = fx - flx, ty = fy - fly;
  int   x0 = wrapCoord(int(flx), w, MI_WRAP_REPEAT), x1 = wrapCoord(int(flx) + 1, w, MI_WRAP_REPEAT);
  int   y0 = wrapCoord(int(fly), h, MI_WRAP_REPEAT), y1 = wrapCoord(int(fly) + 1, h, MI_WRAP_REPEAT);
  const float4* env = sc.envPixels;
  f4    a = mk4(gat(env, size_t(y0) * w + x0)), b = mk4(gat(env, size_t(y0) * w + x1));
  f4    c = mk4(gat(env, size_t(y1) * w + x0)), d = mk4(gat(env, size_t(y1) * w + x1));
  return (a * (1.0f - tx) + b * tx) * (1.0f - ty) + (c * (1.0f - tx) + d * tx) * ty;
}
// Alias-table importance sampling of the lat-long map (reference call site: pathtrace_functions.h.slang:437)
PT_DEV f4 environmentSample(const DevScene& sc, f3 xi, f3& toLight)
{
  uint32_t   width = uint32_t(sc.envWidth), height = uint32_t(sc.envHeight), size = width * height;
  uint32_t   idx = min(uint32_t(xi.x * float(size)), size - 1);
  MiEnvAccel a   = gat(sc.envAccel, idx);
  uint32_t   envIdx;
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
  uint32_t py, px;  // (the integer division is some forty instructions here; lat-long maps are nearly always 2^n wide)
  if((width & (width - 1u)) == 0u)
  {
    py = envIdx >> uint32_t(__ffs(int(width)) - 1);
    px = envIdx & (width - 1u);
  }
  else
  {
    py = envIdx / width;
    px = envIdx % width;
  }
  float    u   = (float(px) + xi.y) / float(width);
  // phi = u 2 pi - pi, theta0 = py pi / height: both handed to the hardware sine / cosine in revolutions (pt_math.h)
  float    sinPhi = sinTurns(u - 0.5f), cosPhi = cosTurns(u - 0.5f);
  float    stepTurns = 0.5f / float(height);
  float    cosTheta  = cosTurns(float(py) * stepTurns) * (1.0f - xi.z) + cosTurns(float(py + 1u) * stepTurns) * xi.z;
  float    theta     = acosf(clampf(cosTheta, -1.0f, 1.0f));
  float    sinTheta  = sqrtf(fmaxf(0.0f, (1.0f - cosTheta) * (1.0f + cosTheta)));  // sin(acos(c))
  float    v         = theta * K_1_OVER_PI;
  toLight            = mk3(cosPhi * sinTheta, cosTheta, sinPhi * sinTheta);
  return sampleHdr(sc, mk2(u, v));
}

// ---- physical sun & sky -----------------------------------------------------------------------------------------------------
// Preetham/Shirley/Smits 1999 Perez sky + limb-darkened sun disc with glow + Lambertian ground; importance sampling is a
// 50/50 mixture of the sun cone and the uniform sphere.  (nvshaders/sky_functions.h.slang is external to the reference.)
PT_DEV f3    skyUp(const MiSkyPhysicalParameters& s) { return s.yIsUp ? mk3(0, 1, 0) : mk3(0, 0, 1); }
PT_DEV float angleBetween(f3 a, f3 b) { return atan2f(length(cross(a, b)), dot(a, b)); }
PT_DEV float skyPerez(float cosTheta, float gamma, float cosGamma, const float* c)
{
  return (1.0f + c[0] * expf(c[1] / fmaxf(cosTheta, 0.01f))) * (1.0f + c[2] * expf(c[3] * gamma) + c[4] * cosGamma * cosGamma);
}
PT_DEV SkyPrecomp makeSkyPrecomp(const MiSkyPhysicalParameters& s)
{
  SkyPrecomp k{};
  const f3 up = skyUp(s), sunDir = normalize(mk3(s.sunDirection));
  k.sunDir[0] = sunDir.x; k.sunDir[1] = sunDir.y; k.sunDir[2] = sunDir.z;
  const float T = 2.0f + fmaxf(s.haze, 0.0f);
  k.T      = T;
  k.cosS   = clampf(dot(sunDir, up), -1.0f, 1.0f);
  k.thetaS = acosf(k.cosS);
  const float airmass = 1.0f / (fmaxf(k.cosS, 0.0f) + 0.15f * powf(fmaxf(93.885f - k.thetaS * 57.29578f, 1.0f), -1.253f));
  const f3    tau     = mk3(0.06f, 0.11f, 0.22f) * (T * 0.5f);
  const f3    sunE    = (k.cosS > -0.05f) ? exp3(-tau * airmass) * 100000.0f : mk3(0.0f);
  k.sunE[0] = sunE.x; k.sunE[1] = sunE.y; k.sunE[2] = sunE.z;
  k.tS = fminf(k.thetaS, 1.5f);
  const float tS = k.tS, chi = (4.0f / 9.0f - T / 120.0f) * (K_PI - 2.0f * tS);
  k.Yz = fmaxf((4.0453f * T - 4.9710f) * tanf(chi) - 0.2155f * T + 2.4192f, 0.0f);
  const float t2 = tS * tS, t3 = t2 * tS, T2 = T * T;
  k.xz = (0.00166f * t3 - 0.00375f * t2 + 0.00209f * tS) * T2 + (-0.02903f * t3 + 0.06377f * t2 - 0.03202f * tS + 0.00394f) * T
         + (0.11693f * t3 - 0.21196f * t2 + 0.06052f * tS + 0.25886f);
  k.yz = (0.00275f * t3 - 0.00610f * t2 + 0.00317f * tS) * T2 + (-0.04214f * t3 + 0.08970f * t2 - 0.04153f * tS + 0.00516f) * T
         + (0.15346f * t3 - 0.26756f * t2 + 0.06670f * tS + 0.26688f);
  k.cosTs = cosf(tS);
  const float cY[5]  = {0.1787f * T - 1.4630f, -0.3554f * T + 0.4275f, -0.0227f * T + 5.3251f, 0.1206f * T - 2.5771f, -0.0670f * T + 0.3703f};
  const float cX[5]  = {-0.0193f * T - 0.2592f, -0.0665f * T + 0.0008f, -0.0004f * T + 0.2125f, -0.0641f * T - 0.8989f, -0.0033f * T + 0.0452f};
  const float cYy[5] = {-0.0167f * T - 0.2608f, -0.0950f * T + 0.0092f, -0.0079f * T + 0.2102f, -0.0441f * T - 1.6537f, -0.0109f * T + 0.0529f};
  for(int i = 0; i < 5; ++i)
  {
    k.cY[i] = cY[i]; k.cX[i] = cX[i]; k.cYy[i] = cYy[i];
  }
  k.denY  = skyPerez(1.0f, tS, k.cosTs, cY);
  k.denX  = skyPerez(1.0f, tS, k.cosTs, cX);
  k.denYy = skyPerez(1.0f, tS, k.cosTs, cYy);
  k.sunRadius       = 0.00465f * fmaxf(s.sunDiskScale, 0.0f) + 1e-6f;
  k.coneAngle       = fminf(k.sunRadius * 4.0f, 1.5f);
  k.coneOneMinusCos = 2.0f * sqr(sinf(0.5f * k.coneAngle));
  k.omega           = K_TWO_PI * 2.0f * sqr(sinf(0.5f * k.sunRadius));
  return k;
}
// `gamma` = angleBetween(dir, sun direction): the caller has it from the pdf (samplePhysicalSkyPDF) -- one atan2 instead of two
__device__ __noinline__ f3 evalPhysicalSky(const MiSkyPhysicalParameters& sIn, const SkyPrecomp& kIn, f3 dir, float gamma)
{
  const MiSkyPhysicalParameters& s = uniformConst(sIn);
  const SkyPrecomp&              k = uniformConst(kIn);
  if(s.multiplier <= 0.0f)
    return mk3(0.0f);
  f3    up     = skyUp(s);
  f3    sunDir = mk3(k.sunDir);
  f3    scale  = mk3(s.rgbUnitConversion) * s.multiplier;
  float cosS   = k.cosS;
  float cosT   = dot(dir, up) - s.horizonHeight * 0.1f;
  f3    sunE   = mk3(k.sunE);
  f3    result;
  if(cosT <= 0.0f)
  {
    f3 skyE    = mk3(0.2f, 0.25f, 0.35f) * 20000.0f * fmaxf(cosS, 0.0f) + mk3(s.nightColor) * 80000.0f;
    result     = mk3(s.groundColor) * (sunE * fmaxf(cosS, 0.0f) + skyE) * K_1_OVER_PI;
    float blur = fmaxf(s.horizonBlur * 0.1f, 1e-4f);
    float t    = saturatef(-cosT / blur);
    if(t < 1.0f)
    {
      f3 h   = mk3(0.75f, 0.8f, 0.9f) * 8000.0f * fmaxf(cosS, 0.05f);
      result = lerp3(h, result, t);
    }
  }
  else
  {
    float cosGamma = clampf(dot(dir, sunDir), -1.0f, 1.0f);
    float Y = k.Yz * skyPerez(cosT, gamma, cosGamma, k.cY) / k.denY;
    float x = k.xz * skyPerez(cosT, gamma, cosGamma, k.cX) / k.denX;
    float y = k.yz * skyPerez(cosT, gamma, cosGamma, k.cYy) / k.denYy;
    Y       = fmaxf(Y, 0.0f) * 1000.0f * saturatef((cosS + 0.05f) * 10.0f);
    float X = (y > 1e-4f) ? x / y * Y : 0.0f, Z = (y > 1e-4f) ? (1.0f - x - y) / y * Y : 0.0f;
    result  = mk3(3.2406f * X - 1.5372f * Y - 0.4986f * Z, -0.9689f * X + 1.8758f * Y + 0.0415f * Z, 0.0557f * X - 0.2040f * Y + 1.0570f * Z);
    result  = max3(result, mk3(0.0f)) + mk3(s.nightColor) * 80000.0f;
    float r = k.sunRadius;
    if(s.sunDiskIntensity > 0.0f && gamma < r * 4.0f)
    {
      f3 Lsun = sunE / k.omega * s.sunDiskIntensity;
      if(gamma < r)
      {
        float mu = sqrtf(fmaxf(0.0f, 1.0f - sqr(gamma / r)));
        result += Lsun * (0.4f + 0.6f * mu);
      }
      else
        result += Lsun * (0.002f * s.sunGlowIntensity) * sqr(1.0f - (gamma - r) / (3.0f * r));
    }
  }
  float rb  = s.redblueshift;
  result    = result * mk3(1.0f + rb, 1.0f, 1.0f - rb);
  float lum = dot(result, mk3(0.2126f, 0.7152f, 0.0722f));
  result    = max3(lerp3(mk3(lum), result, s.saturation), mk3(0.0f));
  return result * scale;
}
PT_DEV float skySunWeight(const MiSkyPhysicalParameters& s) { return (s.sunDiskIntensity > 0.0f && s.multiplier > 0.0f) ? 0.5f : 0.0f; }
PT_DEV float samplePhysicalSkyPDF(const MiSkyPhysicalParameters& s, const SkyPrecomp& k, float gamma)
{
  float wSun   = skySunWeight(s);
  float pdf    = (1.0f - wSun) * (0.25f * K_1_OVER_PI);
  if(wSun > 0.0f && gamma <= k.coneAngle)
    pdf += wSun / (K_TWO_PI * k.coneOneMinusCos);
  return pdf;
}
PT_DEV float skyGamma(const SkyPrecomp& k, f3 dir) { return angleBetween(dir, mk3(k.sunDir)); }
// Uniform direction inside a cone given 1 - cos(halfAngle); sin^2 = s (2 - s) keeps tiny cones (the sun) well conditioned.
PT_DEV f3 sampleCone(f2 xi, float oneMinusCosMax, f3 axis)
{
  float s        = xi.x * oneMinusCosMax;
  float cosTheta = 1.0f - s;
  float sinTheta = sqrtf(fmaxf(0.0f, s * (2.0f - s)));
  f3    T        = xyz(makeFastTangent(axis));
  f3    B        = cross(axis, T);
  return normalize(T * (sinTheta * cosTurns(xi.y)) + B * (sinTheta * sinTurns(xi.y)) + axis * cosTheta);  // phi = 2 pi xi.y
}
PT_DEV void samplePhysicalSky(const MiSkyPhysicalParameters& s, const SkyPrecomp& k, f2 xi, f3& direction, float& pdf, f3& radiance)
{
  float wSun = skySunWeight(s);
  float u    = xi.x;
  if(splitRandom(u, wSun))
    direction = sampleCone(mk2(u, xi.y), k.coneOneMinusCos, mk3(k.sunDir));
  else
  {
    float z   = 1.0f - 2.0f * u;
    float rr  = sqrtf(fmaxf(0.0f, 1.0f - z * z));
    direction = mk3(rr * cosTurns(xi.y), z, rr * sinTurns(xi.y));  // phi = 2 pi xi.y
  }
  const float gamma = skyGamma(k, direction);
  pdf      = samplePhysicalSkyPDF(s, k, gamma);
  radiance = evalPhysicalSky(s, k, direction, gamma);
}

// ---- punctual lights (KHR_lights_punctual; reference call site pathtrace_functions.h.slang:406-412) -------------------------
struct LightContrib
{
  f3    incidentVector;
  float distance;
  f3    intensity;
  float pdf;
};
PT_DEV LightContrib singleLightContribution(const MiGltfLight& light, f3 pos, f2 xi)
{
  LightContrib c;
  c.incidentVector = mk3(0.0f);
  c.distance       = INFINITE_F;
  c.intensity      = mk3(0.0f);
  c.pdf            = DIRAC;
  f3 color         = mk3(light.color) * light.intensity;
  if(light.type == MI_LIGHT_DIRECTIONAL)
  {
    f3    toLight = -normalize(mk3(light.direction));
    float halfAng = 0.5f * light.angularSizeOrInvRange;
    if(halfAng > 0.0f)
    {
      float omc = 2.0f * sqr(sinf(0.5f * halfAng));
      toLight   = sampleCone(xi, omc, toLight);
      c.pdf     = 1.0f / (K_TWO_PI * omc);
    }
    c.incidentVector = -toLight;
    c.distance       = INFINITE_F;
    c.intensity      = color;
  }
  else
  {
    f3    toLight = mk3(light.position) - pos;
    float d       = length(toLight);
    if(d <= 0.0f)
      return c;
    f3    L     = toLight / d;
    f3    axisL = L;
    float dist  = d;
    if(light.radius > 0.0f)
    {
      float sinMax = fminf(light.radius / d, 1.0f);
      float cosMax = sqrtf(fmaxf(0.0f, 1.0f - sinMax * sinMax));
      float omc    = fmaxf(sinMax * sinMax / (1.0f + cosMax), 1e-12f);
      L            = sampleCone(xi, omc, axisL);
      c.pdf        = 1.0f / (K_TWO_PI * omc);
      float b      = dot(L, toLight);
      float disc   = b * b - (d * d - light.radius * light.radius);
      dist         = disc > 0.0f ? fmaxf(b - sqrtf(disc), 0.0f) : b;
    }
    float atten = 1.0f / (d * d);
    if(light.angularSizeOrInvRange > 0.0f)
    {
      float r4 = sqr(sqr(d * light.angularSizeOrInvRange));
      atten *= sqr(clampf(1.0f - r4, 0.0f, 1.0f));
    }
    if(light.type == MI_LIGHT_SPOT)
    {
      float cosOuter = cosf(light.outerAngle), cosInner = cosf(light.innerAngle);
      float scale  = 1.0f / fmaxf(0.001f, cosInner - cosOuter);
      float offset = -cosOuter * scale;
      float cd     = dot(normalize(mk3(light.direction)), -axisL);
      float a      = clampf(cd * scale + offset, 0.0f, 1.0f);
      atten *= a * a;
    }
    c.incidentVector = -L;
    c.distance       = dist;
    c.intensity      = color * atten;
  }
  return c;
}

}  // namespace pt
