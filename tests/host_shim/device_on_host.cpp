// TEST INFRASTRUCTURE ONLY.  The device headers of the shade kernel (BSDF, lights, sky), compiled for the host through the
// stand-in <hip/hip_runtime.h> of this directory and exported with the SAME hook signatures the oracle offers
// (oracle/oracle_pt.h: oracle_bsdf_eval / _sample, oracle_sky_*), so that tests/test_device_headers_on_host.py can diff the two
// restatements lobe by lobe on the CPU.  This library is built by the test session only and is never loaded by the product.
#include "pt_shading.h"
#include "pt_packet.h"

using namespace pt;

namespace {
PbrMaterial materialFromArray(const float* m)  // layout: oracle/oracle_pt.h
{
  PbrMaterial p = defaultPbrMaterial();
  p.baseColor = mk3(m[0], m[1], m[2]);
  p.roughness = mk2(m[3], m[4]);
  p.metallic  = m[5];
  p.ior1 = m[6]; p.ior2 = m[7];
  p.specular = m[8];
  p.specularColor = mk3(m[9], m[10], m[11]);
  p.transmission = m[12];
  p.thickness = m[13];
  p.clearcoat = m[14]; p.clearcoatRoughness = m[15];
  p.sheenColor = mk3(m[16], m[17], m[18]); p.sheenRoughness = m[19];
  p.iridescence = m[20]; p.iridescenceIor = m[21]; p.iridescenceThickness = m[22];
  p.diffuseTransmissionFactor = m[23];
  p.diffuseTransmissionColor = mk3(m[24], m[25], m[26]);
  p.dispersion = m[27];
  p.retroreflection = m[28];
  p.N = p.Ng = p.Nc = mk3(0, 0, 1);
  p.T = mk3(1, 0, 0);
  p.B = mk3(0, 1, 0);
  return p;
}
}  // namespace

extern "C" {
// FrameConsts::slotsMagic / slotsShift (pt_scene.h: divideMagic) as the camera-ray generation uses them: mulhi(n, magic) >> shift
__attribute__((visibility("default"))) uint32_t dev_divide_by_magic(uint32_t n, uint32_t d)
{
  uint32_t magic, shift;
  divideMagic(d, magic, shift);
  return uint32_t((uint64_t(n) * magic) >> 32) >> shift;
}
__attribute__((visibility("default"))) void dev_bsdf_eval(const float* m, const float* k1, const float* k2, const float* xi, float* out4)
{
  BsdfEval e = bsdfEvaluate(mk3(k1), mk3(k2), mk3(xi), materialFromArray(m));
  out4[0] = e.bsdf.x; out4[1] = e.bsdf.y; out4[2] = e.bsdf.z; out4[3] = e.pdf;
}
__attribute__((visibility("default"))) void dev_bsdf_sample(const float* m, const float* k1, const float* xi, float* out8)
{
  BsdfSample d = bsdfSample(mk3(k1), mk3(xi), materialFromArray(m));
  out8[0] = d.k2.x; out8[1] = d.k2.y; out8[2] = d.k2.z;
  out8[3] = d.bsdf_over_pdf.x; out8[4] = d.bsdf_over_pdf.y; out8[5] = d.bsdf_over_pdf.z; out8[6] = d.pdf;
  out8[7] = float(d.event_type);
}
__attribute__((visibility("default"))) void dev_sky_eval(const MiSkyPhysicalParameters* s, const float* dir, float* rgb)
{
  const SkyPrecomp k = makeSkyPrecomp(*s);
  const f3 d = mk3(dir);
  const f3 c = evalPhysicalSky(*s, k, d, skyGamma(k, d));
  rgb[0] = c.x; rgb[1] = c.y; rgb[2] = c.z;
}
__attribute__((visibility("default"))) float dev_sky_pdf(const MiSkyPhysicalParameters* s, const float* dir)
{
  const SkyPrecomp k = makeSkyPrecomp(*s);
  return samplePhysicalSkyPDF(*s, k, skyGamma(k, mk3(dir)));
}
__attribute__((visibility("default"))) void dev_sky_sample(const MiSkyPhysicalParameters* s, float u, float v, float* dirPdfRgb)
{
  const SkyPrecomp k = makeSkyPrecomp(*s);
  f3 dir, rad; float pdf;
  samplePhysicalSky(*s, k, mk2(u, v), dir, pdf, rad);
  dirPdfRgb[0] = dir.x; dirPdfRgb[1] = dir.y; dirPdfRgb[2] = dir.z; dirPdfRgb[3] = pdf;
  dirPdfRgb[4] = rad.x; dirPdfRgb[5] = rad.y; dirPdfRgb[6] = rad.z;
}
// light: 16 floats = MiGltfLight; out: incidentVector[3] distance intensity[3] pdf
__attribute__((visibility("default"))) void dev_light_contribution(const MiGltfLight* light, const float* pos, const float* xi, float* out8)
{
  LightContrib c = singleLightContribution(*light, mk3(pos), mk2(xi[0], xi[1]));
  out8[0] = c.incidentVector.x; out8[1] = c.incidentVector.y; out8[2] = c.incidentVector.z; out8[3] = c.distance;
  out8[4] = c.intensity.x; out8[5] = c.intensity.y; out8[6] = c.intensity.z; out8[7] = c.pdf;
}

// The packet walk's interval test (pt_packet.h) next to the per-ray test it stands in for (pt_bvh8.h: bvh8TestChildrenPlanes, whose
// arithmetic is restated here -- that header is not host-compilable), on one node and one packet of up to 64 rays that all point
// into one octant.  exactAny: children some ray's own test enters; interval: children the interval test enters.  Returns 0 when
// the rays do not share an octant.
__attribute__((visibility("default"))) int dev_packet_masks(const float* P, const float* s, const float* planes, int nrays, const float* org, const float* dir,
                                                            const float* tmax, uint32_t* exactAny, uint32_t* interval)
{
  float        idir[64][3];
  PacketBounds B;
  uint32_t     negMask = 0;
  float        tmaxPacket = -INFINITY;
  for(int a = 0; a < 3; ++a)
  {
    B.omin[a] = B.imin[a] = INFINITY;
    B.omax[a] = B.imax[a] = -INFINITY;
  }
  for(int r = 0; r < nrays; ++r)
  {
    for(int a = 0; a < 3; ++a)
    {
      const float d = dir[3 * r + a], eps = 1e-30f;  // makeRaySetup, pt_bvh.h
      idir[r][a]    = 1.0f / (std::fabs(d) < eps ? std::copysign(eps, d) : d);
      B.omin[a] = std::min(B.omin[a], org[3 * r + a]); B.omax[a] = std::max(B.omax[a], org[3 * r + a]);
      B.imin[a] = std::min(B.imin[a], idir[r][a]);     B.imax[a] = std::max(B.imax[a], idir[r][a]);
      const uint32_t neg = idir[r][a] < 0.0f ? 1u : 0u;
      if(r == 0)
        negMask |= neg << a;
      else if(((negMask >> a) & 1u) != neg)
        return 0;
    }
    tmaxPacket = std::max(tmaxPacket, tmax[r]);
  }
  // ---- per ray, as bvh8TestChildrenPlanes does it
  uint32_t any = 0;
  for(int r = 0; r < nrays; ++r)
    for(int c = 0; c < 8; ++c)
    {
      float tn = 0.0f, tf = tmax[r];
      for(int a = 0; a < 3; ++a)
      {
        const bool  neg = (negMask >> a) & 1u;
        const float Pa = P[a] - org[3 * r + a], k = 4.76837158e-7f;
        const float A  = s[a] * idir[r][a];
        // slabOffsets (pt_bvh8.h, round 4): the pad goes onto the plane TIME, the same expression whatever the direction's sign
        const float t0 = Pa * idir[r][a];
        const float e  = k * std::fmaf(255.0f, std::fabs(A), std::fabs(t0));
        const float Bn = t0 - e, Bf = t0 + e;
        const float qn = planes[(2 * a + (neg ? 1 : 0)) * 8 + c], qf = planes[(2 * a + (neg ? 0 : 1)) * 8 + c];
        tn = std::max(tn, std::fmaf(qn, A, Bn));
        tf = std::min(tf, std::fmaf(qf, A, Bf));
      }
      if(!std::signbit(tf - tn))
        any |= 1u << c;
    }
  // ---- the interval test, lane by lane
  uint32_t iv = 0;
  for(int c = 0; c < 8; ++c)
  {
    float tn = -INFINITY, tf = INFINITY;
    for(int pl = 0; pl < 8; ++pl)
    {
      const PacketLane L = makePacketLane(uint32_t(c * 8 + pl), negMask, B);
      if(!L.live)
        continue;
      const float t = packetPlaneTime(L, planes[L.planeOffset], P[L.axis], s[L.axis]);
      if(L.entry) tn = std::max(tn, t); else tf = std::min(tf, t);
    }
    if(packetChildHit(tn, tf, tmaxPacket))
      iv |= 1u << c;
  }
  *exactAny = any;
  *interval = iv;
  return 1;
}
}
