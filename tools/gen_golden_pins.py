#!/usr/bin/env python3
"""Writes tests/golden/pins_closed_forms.json: known answers for the nvshaders-side functions of the oracle that do NOT come from
the oracle's (or the device's) own code.  Every block is an independent float64 / complex restatement of a PUBLISHED formula, with
its source, evaluated on a grid; tests/test_oracle_pins.py holds the oracle's float32 hooks against them.

Why: the reference's BSDF / sky / RNG bodies live in un-vendored nvpro_core2 and the reference holds no radiance fixtures
(SURVEY.md §0, §8c), so the oracle cannot be pinned against the Vulkan renderer here.  These pins at least tie every physical
building block to the literature instead of to the oracle itself.  SURVEY Appendix B symbols covered are listed per block.

Sources
  fresnel_dielectric   Born & Wolf, Principles of Optics, §1.5.2 (Fresnel formulae); unpolarised R = (Rs + Rp) / 2     [schlickFresnel / ior_fresnel inside bsdfEvaluate, bsdfSample]
  fresnel_schlick      Schlick 1994, "An Inexpensive BRDF Model for Physically-based Rendering", eq. 15               [schlickFresnel: pathtrace_functions.h.slang:283,285,744]
  fresnel_conductor    Born & Wolf §14.2 with the complex index N = n + ik (evaluated here in complex arithmetic)      [thin-film base layer]
  thin_film            Airy summation of a single film, Born & Wolf §7.6.1 eq. (7.6.8): r = (r01 + r12 e^{i d}) / (1 + r01 r12 e^{i d}),
                       d = 4 pi n1 h cos(theta1) / lambda, per polarisation, complex arithmetic; spectral integration over the 16
                       wavelengths of tools/gen_thinfilm_table.py (Wyman, Sloan, Shirley, JCGT 2013 colour-matching fit)    [KHR_materials_iridescence inside bsdfEvaluate / bsdfSample]
  ggx                  Heitz 2014 "Understanding the Masking-Shadowing Function", eq. 72 (Lambda), 85-86 (anisotropic D);
                       Heitz 2018 "Sampling the GGX Distribution of Visible Normals", eq. 1-3 (D_v, its reflected pdf, weight G2/G1)  [bsdfEvaluate / bsdfSample glossy lobes]
  henyey_greenstein    Henyey & Greenstein 1941; normalised phase function p = (1 - g^2) / (4 pi (1 + g^2 - 2 g cos)^1.5)   [henyeyGreensteinPdf, sampleHenyeyGreenstein]
  preetham             Preetham, Shirley, Smits 1999, "A Practical Analytic Model for Daylight", Appendix A.2 (Perez coefficients,
                       zenith luminance / chromaticity); CIE xyY -> XYZ -> linear sRGB (IEC 61966-2-1)                        [evalPhysicalSky above the horizon, outside the sun glow]
  xxhash32             Collet, xxHash specification (XXH32), via the `xxhash` Python package: the 3-word hash of nvshaders
                       equals XXH32 of the 8 bytes (x, y) with seed z - 8                                                   [xxhash32: gltf_pathtrace.slang:560]
  pcg                  Jarzynski & Olano 2020, "Hash Functions for GPU Rendering" (JCGT 9(3)), listing `pcg`                   [rand]
  sheen                Conty Estevez & Kulla 2017, "Production Friendly Microfacet Sheen BRDF", eq. 2: D(h) = (2 + n) sin^n(theta_h) / (2 pi)
                       (n = 1 / alpha; the MDL-lineage lobe restated here uses n = 1 / roughness^2), a density over the solid angle of h
                       once multiplied by cos(theta_h); V-cavities masking of Torrance & Sparrow 1967, G = min(1, 2 (n.h)(n.k1)/(k1.h), 2 (n.h)(n.k2)/(k2.h))   [KHR_materials_sheen inside bsdfEvaluate / bsdfSample]
  clearcoat            KHR_materials_clearcoat (Khronos glTF extension text): a dielectric layer of IOR 1.5 (F0 = 0.04) on top, reflected with
                       probability clearcoat x Fresnel(1.5, cos), the layers below attenuated by the rest                          [lobe weights of bsdfEvaluate / bsdfSample]
  point_offset         Hanika 2021, "Hacking the Shadow Terminator" (Ray Tracing Gems II, ch. 4), listing 4-1                        [pointOffset: get_hit.h.slang:124-131 call site]
  ray_cone             Akenine-Moller et al. 2019, "Texture Level of Detail Strategies for Real-Time Ray Tracing" (Ray Tracing Gems, ch. 20),
                       eq. 29-30: cone width w = w0 + gamma t, footprint on the surface w / |n . d|                                  [rayConeWorldFootprint: pathtrace_functions.h.slang:174-178]
  hdr_importance       definition of importance sampling a lat-long map by max(r, g, b) x texel solid angle (Pharr, Jakob, Humphreys, PBRT 3rd ed.
                       section 14.2.4 for the distribution, Vose 1991 for the alias method): checked in tests/test_oracle_pins.py directly against
                       numpy float64 on assets/std_env.hdr -- no fixture needed, the asset is in the tree                              [nvvk::HdrIbl::loadEnvironment, environmentSample]
"""
import json
import math
import os
import struct

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rng = np.random.default_rng(20260925)


def fresnel_dielectric(eta, c):
    s2 = (1.0 - c * c) / (eta * eta)
    if s2 >= 1.0:
        return 1.0
    ct = math.sqrt(1.0 - s2)
    rs = (c - eta * ct) / (c + eta * ct)
    rp = (eta * c - ct) / (eta * c + ct)
    return 0.5 * (rs * rs + rp * rp)


def fresnel_conductor(n_a, n_b, k_b, c):
    N = complex(n_b, k_b)
    s = math.sqrt(max(0.0, 1.0 - c * c))
    ct = np.sqrt(1.0 - (n_a * s / N) ** 2)
    rs = (n_a * c - N * ct) / (n_a * c + N * ct)
    rp = (N * c - n_a * ct) / (N * c + n_a * ct)
    return abs(rs) ** 2, abs(rp) ** 2


def cmf_rows():
    def g(lam, mu, s1, s2):
        s = s1 if lam < mu else s2
        return math.exp(-0.5 * ((lam - mu) / s) ** 2)

    def cmf(lam):
        x = 1.056 * g(lam, 599.8, 37.9, 31.0) + 0.362 * g(lam, 442.0, 16.0, 26.7) - 0.065 * g(lam, 501.1, 20.4, 26.2)
        y = 0.821 * g(lam, 568.8, 46.9, 40.5) + 0.286 * g(lam, 530.9, 16.3, 31.1)
        z = 1.217 * g(lam, 437.0, 11.8, 36.0) + 0.681 * g(lam, 459.0, 26.0, 13.8)
        return np.array([x, y, z])

    M = np.array([[3.2406, -1.5372, -0.4986], [-0.9689, 1.8758, 0.0415], [0.0557, -0.2040, 1.0570]])
    lams = [400.0 + (i + 0.5) * 18.75 for i in range(16)]
    rgb = np.array([M @ cmf(lam) for lam in lams])
    return lams, rgb / rgb.mean(axis=0)


def thin_film(thickness, n1, n2, n0, c0):
    """Unpolarised reflectance of a film (index n1, thickness in nm) on a substrate n2 seen from n0 at cos(theta0) = c0 -> linear RGB."""
    s0 = 1.0 - c0 * c0
    s1 = (n0 / n1) ** 2 * s0
    if s1 > 1.0:
        return [1.0, 1.0, 1.0]
    c1 = math.sqrt(1.0 - s1)
    s2 = (n1 / n2) ** 2 * s1
    c2 = np.sqrt(complex(1.0 - s2))  # evanescent in the substrate when s2 > 1: total reflection at the lower interface
    r01s, r01p = (n0 * c0 - n1 * c1) / (n0 * c0 + n1 * c1), (n1 * c0 - n0 * c1) / (n1 * c0 + n0 * c1)
    r12s, r12p = (n1 * c1 - n2 * c2) / (n1 * c1 + n2 * c2), (n2 * c1 - n1 * c2) / (n2 * c1 + n1 * c2)
    lams, rows = cmf_rows()
    out = np.zeros(3)
    for lam, row in zip(lams, rows):
        e = np.exp(1j * 4.0 * math.pi * n1 * thickness * c1 / lam)
        Rs = abs((r01s + r12s * e) / (1.0 + r01s * r12s * e)) ** 2
        Rp = abs((r01p + r12p * e) / (1.0 + r01p * r12p * e)) ** 2
        out += row * 0.5 * (Rs + Rp)
    return np.clip(out / 16.0, 0.0, 1.0).tolist()


def ggx_D(ax, ay, h):  # Heitz 2014 eq. 85-86
    return 1.0 / (math.pi * ax * ay * ((h[0] / ax) ** 2 + (h[1] / ay) ** 2 + h[2] ** 2) ** 2)


def ggx_G1(ax, ay, v):  # Heitz 2014 eq. 72 + 43
    lam = 0.5 * (-1.0 + math.sqrt(1.0 + ((ax * v[0]) ** 2 + (ay * v[1]) ** 2) / (v[2] ** 2)))
    return 1.0 / (1.0 + lam)


def unit(v):
    v = np.asarray(v, np.float64)
    return (v / np.linalg.norm(v)).tolist()


def preetham_rgb(T, theta_s, cos_t, gamma):
    """Luminance in kcd/m^2 and chromaticity of the Preetham sky for turbidity T, sun zenith angle theta_s, view cos(zenith) and
    angle to the sun gamma -> linear sRGB with Y in kcd/m^2."""
    A = {"Y": [0.1787 * T - 1.4630, -0.3554 * T + 0.4275, -0.0227 * T + 5.3251, 0.1206 * T - 2.5771, -0.0670 * T + 0.3703],
         "x": [-0.0193 * T - 0.2592, -0.0665 * T + 0.0008, -0.0004 * T + 0.2125, -0.0641 * T - 0.8989, -0.0033 * T + 0.0452],
         "y": [-0.0167 * T - 0.2608, -0.0950 * T + 0.0092, -0.0079 * T + 0.2102, -0.0441 * T - 1.6537, -0.0109 * T + 0.0529]}

    def F(c, ct, g):
        return (1.0 + c[0] * math.exp(c[1] / ct)) * (1.0 + c[2] * math.exp(c[3] * g) + c[4] * math.cos(g) ** 2)

    chi = (4.0 / 9.0 - T / 120.0) * (math.pi - 2.0 * theta_s)
    Yz = (4.0453 * T - 4.9710) * math.tan(chi) - 0.2155 * T + 2.4192
    t = np.array([theta_s ** 3, theta_s ** 2, theta_s, 1.0])
    TT = np.array([T * T, T, 1.0])
    xz = TT @ np.array([[0.00166, -0.00375, 0.00209, 0.0], [-0.02903, 0.06377, -0.03202, 0.00394], [0.11693, -0.21196, 0.06052, 0.25886]]) @ t
    yz = TT @ np.array([[0.00275, -0.00610, 0.00317, 0.0], [-0.04214, 0.08970, -0.04153, 0.00516], [0.15346, -0.26756, 0.06670, 0.26688]]) @ t
    Y = Yz * F(A["Y"], cos_t, gamma) / F(A["Y"], 1.0, theta_s)
    x = xz * F(A["x"], cos_t, gamma) / F(A["x"], 1.0, theta_s)
    y = yz * F(A["y"], cos_t, gamma) / F(A["y"], 1.0, theta_s)
    X, Z = x / y * Y, (1.0 - x - y) / y * Y
    M = np.array([[3.2406, -1.5372, -0.4986], [-0.9689, 1.8758, 0.0415], [0.0557, -0.2040, 1.0570]])
    return (M @ np.array([X, Y, Z])).tolist()


def pcg_hash(state):  # Jarzynski & Olano 2020
    state = (state * 747796405 + 2891336453) & 0xffffffff
    word = (((state >> ((state >> 28) + 4)) ^ state) * 277803737) & 0xffffffff
    return state, ((word >> 22) ^ word) & 0xffffffff


def main():
    out = {"_doc": __doc__}
    out["fresnel_dielectric"] = [{"eta": eta, "cos": c, "R": fresnel_dielectric(eta, c)}
                                 for eta in (1.33, 1.5, 2.0, 1.0 / 1.5, 1.0 / 1.33) for c in (1.0, 0.95, 0.8, 0.7071067811865476, 0.5547, 0.3, 0.1, 0.02)]
    # textbook spot values: normal incidence on glass R = ((n - 1) / (n + 1))^2 = 0.04; Brewster angle of n = 1.5: Rp = 0 -> R = Rs / 2
    out["fresnel_dielectric"].append({"eta": 1.5, "cos": math.cos(math.atan(1.5)), "R": fresnel_dielectric(1.5, math.cos(math.atan(1.5))), "note": "Brewster"})
    out["fresnel_schlick"] = [{"ior": n, "cos": c, "R": ((1 - n) / (1 + n)) ** 2 + (1 - ((1 - n) / (1 + n)) ** 2) * (1 - c) ** 5} for n in (1.33, 1.5, 2.4) for c in (1.0, 0.8, 0.5, 0.2, 0.0)]
    out["fresnel_conductor"] = []
    for n_a, n_b, k_b in ((1.0, 0.2, 3.0), (1.0, 1.5, 0.0), (1.3, 2.0, 1.0), (1.5, 0.05, 4.2), (1.0, 1.8, 0.0)):
        for c in (1.0, 0.9, 0.6, 0.3, 0.05):
            rs, rp = fresnel_conductor(n_a, n_b, k_b, c)
            out["fresnel_conductor"].append({"n_a": n_a, "n_b": n_b, "k_b": k_b, "cos": c, "Rs": rs, "Rp": rp})
    out["thin_film"] = []
    for thick in (0.0, 50.0, 120.0, 250.0, 400.0, 650.0, 900.0, 1200.0):
        for n1, n2, n0 in ((1.3, 1.5, 1.0), (1.8, 1.5, 1.0), (2.2, 1.33, 1.0), (1.3, 1.0, 1.5)):
            for c0 in (1.0, 0.8, 0.45, 0.15):
                # The real-arithmetic form the oracle restates (MDL-SDK libbsdf lineage) keeps the SQUARED Fresnel terms of the upper
                # interface and the phase of the lower one only: it is the Airy formula exactly as long as r01 has the sign of an
                # interface into a denser medium below its Brewster angle (r01_s <= 0, r01_p >= 0).  Beyond the upper interface's
                # Brewster angle (r01_p < 0), or for a film rarer than the medium above it, that sign is lost and the affected
                # polarisation interferes with the wrong sign: flagged here, bounded loosely in the test, documented in DESIGN.md §6.
                s1 = (n0 / n1) ** 2 * (1 - c0 * c0)
                band, near = False, False
                if s1 <= 1.0:
                    c1 = math.sqrt(1 - s1)
                    c2 = np.sqrt(complex(1 - (n1 / n2) ** 2 * s1))
                    r01p, r12p = (n1 * c0 - n0 * c1) / (n1 * c0 + n0 * c1), ((n2 * c1 - n1 * c2) / (n2 * c1 + n1 * c2))
                    r01s = (n0 * c0 - n1 * c1) / (n0 * c0 + n1 * c1)
                    band = bool(r01p < 0 or r01s > 0)
                    near = bool(abs(r12p) < 0.01)  # at the lower interface's Brewster angle the float32 phase of a vanishing r12_p is rounding noise
                out["thin_film"].append({"thickness": thick, "coating_ior": n1, "base_ior": n2, "incoming_ior": n0, "cos": c0, "rgb": thin_film(thick, n1, n2, n0, c0),
                                         "p_sign_band": band, "near_brewster": near})
    out["ggx"] = []
    for ax, ay in ((0.5, 0.5), (0.1, 0.1), (0.8, 0.15), (0.05, 0.6), (1.0, 1.0)):
        for _ in range(12):
            v = rng.normal(size=3); v[2] = abs(v[2]) + 0.05; v = unit(v)
            h = rng.normal(size=3); h[2] = abs(h[2]) + 0.2; h = unit(h)
            vh = float(np.dot(v, h))
            D = ggx_D(ax, ay, h)
            entry = {"ax": ax, "ay": ay, "v": v, "h": h, "D_cos": D * h[2], "G1_v": ggx_G1(ax, ay, v)}
            if vh > 0:
                l = (2.0 * vh * np.array(h) - np.array(v)).tolist()
                entry.update({"l": l, "vndf_reflected_pdf": ggx_G1(ax, ay, v) * vh * D / v[2] / (4.0 * vh),
                              "G1_l": ggx_G1(ax, ay, [l[0], l[1], abs(l[2])]) if abs(l[2]) > 1e-6 else 0.0})
            out["ggx"].append(entry)
    out["henyey_greenstein"] = [{"g": g, "cos": c, "pdf": (1 - g * g) / (4 * math.pi * (1 + g * g - 2 * g * c) ** 1.5)}
                                for g in (-0.8, -0.3, 0.0, 0.3, 0.7, 0.95) for c in (-1.0, -0.5, 0.0, 0.4, 0.9, 1.0)]
    out["preetham"] = []
    for T in (2.0, 2.1, 3.5, 6.0, 9.0):
        for theta_s in (0.2, 0.9553166181245093, 1.3):
            for cos_t, gamma in ((1.0, theta_s), (0.7, 0.6), (0.3, 1.4), (0.1, 2.4), (0.02, 0.9)):
                out["preetham"].append({"T": T, "theta_s": theta_s, "cos_theta": cos_t, "gamma": gamma, "rgb_kcd": preetham_rgb(T, theta_s, cos_t, gamma)})
    import xxhash
    out["xxhash32"] = []
    for _ in range(64):
        x, y, z = (int(v) for v in rng.integers(0, 1 << 32, 3, dtype=np.uint64))
        out["xxhash32"].append({"x": x, "y": y, "z": z, "h": xxhash.xxh32(struct.pack("<II", x, y), seed=(z - 8) & 0xffffffff).intdigest()})
    out["xxhash32"].append({"x": 0, "y": 0, "z": 8, "h": xxhash.xxh32(struct.pack("<II", 0, 0), seed=0).intdigest()})
    out["pcg"] = []
    for s0 in (0, 1, 0xdeadbeef, 0xffffffff, 123456789):
        s, seq = s0, []
        for _ in range(4):
            s, o = pcg_hash(s)
            seq.append(o)
        out["pcg"].append({"seed": s0, "outputs": seq, "final_state": s})
    # sheen: density of h over solid angle D(h) cos(theta_h), D = (n + 2) sin^n / (2 pi); V-cavities G
    out["sheen"] = [{"n": n, "cos_h": c, "pdf_h": (n + 2.0) * (1.0 - c * c) ** (0.5 * n) / (2.0 * math.pi) * c} for n in (1.0, 4.0, 11.11, 100.0, 400.0) for c in (0.02, 0.2, 0.5, 0.8, 0.98)]
    out["vcavities"] = []
    for _ in range(40):
        k1 = rng.normal(size=3); k1[2] = abs(k1[2]) + 0.05; k1 = np.array(unit(k1))
        k2 = rng.normal(size=3); k2[2] = abs(k2[2]) + 0.05; k2 = np.array(unit(k2))
        h = (k1 + k2) / np.linalg.norm(k1 + k2)
        out["vcavities"].append({"nh": float(h[2]), "k1h": float(k1 @ h), "k1z": float(k1[2]), "k2h": float(k2 @ h), "k2z": float(k2[2]),
                                 "G": float(min(1.0, 2 * h[2] * k1[2] / (k1 @ h), 2 * h[2] * k2[2] / (k2 @ h)))})
    # clearcoat lobe probability: clearcoat x unpolarised Fresnel of an IOR-1.5 layer seen from a medium of IOR ior1
    out["clearcoat_weight"] = [{"clearcoat": cc, "ior1": n1, "cos": c, "w": cc * fresnel_dielectric(1.5 / n1, c)} for cc in (1.0, 0.6, 0.25) for n1 in (1.0, 1.33) for c in (1.0, 0.8, 0.5, 0.2, 0.05)]
    # Hanika's shadow-terminator offset, float64, on random triangles with perturbed vertex normals
    out["point_offset"] = []
    for _ in range(48):
        tri = rng.normal(size=(3, 3))
        ng = np.cross(tri[1] - tri[0], tri[2] - tri[0]); ng /= np.linalg.norm(ng)
        nrm = np.array([unit(ng + 0.6 * rng.normal(size=3)) for _ in range(3)])
        b = rng.dirichlet((1, 1, 1))
        P = b @ tri
        t = [P - tri[i] for i in range(3)]
        t = [t[i] - min(0.0, float(t[i] @ nrm[i])) * nrm[i] for i in range(3)]
        out["point_offset"].append({"tri": tri.reshape(-1).tolist(), "nrm": nrm.reshape(-1).tolist(), "bary": b.tolist(), "p": P.tolist(),
                                    "offset_p": (P + b[0] * t[0] + b[1] * t[1] + b[2] * t[2]).tolist()})
    out["ray_cone"] = [{"width": w0, "spread": g, "t": t, "cos": c, "footprint": (w0 + g * t) / max(abs(c), 1e-3)} for w0 in (0.0, 0.01, 0.3) for g in (0.0, 5e-4, 2e-3) for t in (0.1, 3.0, 50.0)
                       for c in (1.0, -0.6, 0.2, 1e-5)]
    path = os.path.join(ROOT, "tests", "golden", "pins_closed_forms.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path, {k: len(v) for k, v in out.items() if isinstance(v, list)})


if __name__ == "__main__":
    main()
