"""The oracle's nvshaders-side building blocks against PUBLISHED closed forms (tests/golden/pins_closed_forms.json, written by
tools/gen_golden_pins.py from independent float64 / complex-arithmetic restatements of the cited papers; sources in the file).
The reference holds no radiance fixtures and its BSDF / sky / RNG bodies are not vendored, so these are the pins that do not come
from the oracle's own code; DESIGN.md §6 lists which SURVEY Appendix B symbol each one covers.  The device side of the same
functions is tied to the oracle by tests/test_device_headers_on_host.py (CPU) and the GPU parity tests."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import oracle_lib
from vk_gltf_renderer_amd import pathtracer as ptmod

F = C.c_float
GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pins_closed_forms.json")))


def f3(v):
    return (F * 3)(*v)


def test_fresnel_dielectric_schlick_conductor(built):
    O = oracle_lib.lib()
    for e in GOLD["fresnel_dielectric"]:
        assert O.oracle_fresnel_dielectric_unpolarized(e["eta"], e["cos"]) == pytest.approx(e["R"], rel=2e-5, abs=2e-7), e
    assert O.oracle_fresnel_dielectric_unpolarized(1.5, 1.0) == pytest.approx(0.04, rel=1e-6)  # glass at normal incidence
    for e in GOLD["fresnel_schlick"]:
        assert O.oracle_fresnel_schlick(e["ior"], e["cos"]) == pytest.approx(e["R"], rel=2e-5, abs=1e-7), e
    out = (F * 2)()
    for e in GOLD["fresnel_conductor"]:
        O.oracle_fresnel_conductor(e["n_a"], e["n_b"], e["k_b"], e["cos"], out)
        assert out[0] == pytest.approx(e["Rs"], rel=5e-5, abs=1e-6) and out[1] == pytest.approx(e["Rp"], rel=5e-5, abs=1e-6), (e, out[:])


def test_thin_film_interference_matches_the_airy_formula(built):
    """KHR_materials_iridescence: the real-arithmetic phase form of the oracle against the complex Airy summation."""
    O = oracle_lib.lib()
    rgb = (F * 3)()
    worst, band_worst, exact_cases = 0.0, 0.0, 0
    for e in GOLD["thin_film"]:
        O.oracle_thin_film(e["thickness"], e["coating_ior"], e["base_ior"], e["incoming_ior"], e["cos"], rgb)
        err = float(np.abs(np.array(rgb[:]) - np.array(e["rgb"])).max())
        if e["p_sign_band"]:
            # Known, documented deviation of the restated model: it keeps |r01|^2 and the phase of r12 only, which is the
            # Airy formula as long as r01 has the sign of an interface into a denser medium below Brewster; outside that range the
            # interference term of the affected polarisation has the wrong sign.
            band_worst = max(band_worst, err)
            assert err < 0.12, (e, rgb[:])
        else:
            if not e["near_brewster"]:
                worst = max(worst, err)
                exact_cases += 1
            if e["near_brewster"]:  # the phase of a vanishing r12_p is float32 rounding noise: bounded by 2 |r01_p r12_p| <= 0.01 in R_p
                assert err < 5e-3, (e, rgb[:])
                continue
            assert np.allclose(rgb[:], e["rgb"], rtol=3e-4, atol=3e-5), (e, rgb[:])
    assert worst < 3e-4 and exact_cases >= 40, (worst, band_worst, exact_cases)
    print('thin film: exact cases', exact_cases, 'worst', worst, '| sign-band worst', band_worst)
    # a film of zero thickness is no film: plain Fresnel reflectance of the bare interface, the same in every channel
    O.oracle_thin_film(0.0, 1.3, 1.5, 1.0, 1.0, rgb)
    assert np.allclose(rgb[:], 0.04, rtol=1e-3)


def test_ggx_distribution_masking_and_vndf_pdf(built):
    """Heitz 2014 / 2018 closed forms: D(h) cos(theta_h), Smith G1, and -- through the BSDF hooks -- the reflected VNDF pdf
    G1(v) D(h) / (4 v.z) and the sampling weight G1(l) of a white metal."""
    O = oracle_lib.lib()
    mat = np.zeros(29, np.float32)
    mat[0:3] = 1.0; mat[5] = 1.0; mat[6], mat[7], mat[8] = 1.0, 1.5, 1.0; mat[9:12] = 1.0; mat[15] = 0.01; mat[21], mat[22] = 1.5, 100.0; mat[24:27] = 1.0
    ev = (F * 7)()
    checked = 0
    for e in GOLD["ggx"]:
        ax, ay = e["ax"], e["ay"]
        assert O.oracle_ggx_ndf(ax, ay, f3(e["h"])) == pytest.approx(e["D_cos"], rel=3e-5), e
        assert O.oracle_ggx_g1(ax, ay, f3(e["v"])) == pytest.approx(e["G1_v"], rel=3e-5), e
        if "l" in e and e["l"][2] > 1e-3:
            m = mat.copy()
            m[3], m[4] = ax, ay
            O.oracle_bsdf_eval((F * 29)(*m), f3(e["v"]), f3(e["l"]), f3((0.3, 0.6, 0.5)), ev)
            assert ev[6] == pytest.approx(e["vndf_reflected_pdf"], rel=1e-4), (e, ev[:])
            # white metal at F = 1: bsdf * cos / pdf = G2 / G1(v) = G1(l) (separable Smith)
            assert (ev[3] / ev[6]) == pytest.approx(e["G1_l"], rel=1e-4), (e, ev[:])
            checked += 1
    assert checked > 20


def test_ggx_vndf_samples_follow_the_visible_normal_distribution(built):
    """Heitz 2018: the half vectors drawn for a view direction v have density D_v(h) = G1(v) max(0, v.h) D(h) / v.z.  Checked by
    comparing sample means of three test functions with their quadrature over D_v."""
    O = oracle_lib.lib()
    rng = np.random.default_rng(5)
    for ax, ay, v in ((0.4, 0.4, (0.6, 0.0, 0.8)), (0.7, 0.2, (0.5, 0.5, 0.7071)), (0.15, 0.5, (0.9, -0.3, 0.316))):
        v = np.array(v) / np.linalg.norm(v)
        n = 60000
        hs = np.zeros((n, 3), np.float32)
        h = (F * 3)()
        for i, (a, b) in enumerate(rng.random((n, 2))):
            O.oracle_ggx_sample_vndf(ax, ay, f3(v), a, b, h)
            hs[i] = h[:]
        # quadrature of D_v over the hemisphere of h
        th, ph = np.meshgrid((np.arange(400) + 0.5) / 400 * (np.pi / 2), (np.arange(800) + 0.5) / 800 * 2 * np.pi, indexing="ij")
        H = np.stack([np.sin(th) * np.cos(ph), np.sin(th) * np.sin(ph), np.cos(th)], -1)
        D = 1.0 / (np.pi * ax * ay * ((H[..., 0] / ax) ** 2 + (H[..., 1] / ay) ** 2 + H[..., 2] ** 2) ** 2)
        lam = 0.5 * (-1 + np.sqrt(1 + ((ax * v[0]) ** 2 + (ay * v[1]) ** 2) / v[2] ** 2))
        Dv = np.maximum(H @ v, 0) * D / (1 + lam) / v[2]
        w = Dv * np.sin(th) * (np.pi / 2 / 400) * (2 * np.pi / 800)
        assert w.sum() == pytest.approx(1.0, abs=5e-3)  # D_v is normalised (Heitz 2018 eq. 2)
        for fn in (lambda X: X[..., 2], lambda X: X[..., 0], lambda X: X[..., 1] ** 2):
            assert fn(hs).mean() == pytest.approx((fn(H) * w).sum() / w.sum(), abs=6e-3)


def test_henyey_greenstein_pdf_and_sampling(built):
    O = oracle_lib.lib()
    for e in GOLD["henyey_greenstein"]:
        if abs(1 + e["g"] ** 2 - 2 * e["g"] * e["cos"]) < 1e-4:
            continue
        # (float32 cancellation in 1 + g^2 - 2 g cos near the forward peak of a strongly anisotropic medium)
        assert O.oracle_hg_pdf(e["cos"], e["g"]) == pytest.approx(e["pdf"], rel=2e-4 if abs(e["g"]) > 0.9 else 3e-5), e
    rng = np.random.default_rng(2)
    wi, wo = np.array([0.3, -0.5, 0.81]), (F * 3)()
    wi /= np.linalg.norm(wi)
    for g in (-0.6, 0.0, 0.35, 0.85):
        cs = []
        for a, b in rng.random((20000, 2)):
            O.oracle_hg_sample(a, b, g, f3(wi), wo)
            assert np.linalg.norm(wo[:]) == pytest.approx(1.0, abs=1e-5)
            cs.append(np.dot(wo[:], wi))
        assert np.mean(cs) == pytest.approx(g, abs=0.012)  # the mean cosine of the Henyey-Greenstein phase function is g
        # normalisation over the sphere
        c = (np.arange(4000) + 0.5) / 4000 * 2 - 1
        assert sum(O.oracle_hg_pdf(float(x), g) for x in c) * 2 * np.pi * (2 / 4000) == pytest.approx(1.0, abs=5e-3)


def test_sky_matches_the_published_preetham_model(built):
    """evalPhysicalSky above the horizon and away from the sun's glow, with the artistic controls neutral, is the Preetham / Perez
    daylight model with turbidity T = 2 + haze (Y in kcd/m^2 x 1000 x rgbUnitConversion x multiplier)."""
    O = oracle_lib.lib()
    rgb = (F * 3)()
    n = 0
    for e in GOLD["preetham"]:
        sky = ptmod.default_sky()
        sky.haze, sky.redblueshift, sky.saturation, sky.horizonHeight = e["T"] - 2.0, 0.0, 1.0, 0.0
        sky.nightColor[:] = [0, 0, 0]
        sky.sunDiskIntensity = 0.0  # no disc, no glow: the sky dome alone
        ts = e["theta_s"]
        sky.sunDirection[:] = [np.sin(ts), np.cos(ts), 0.0]
        # a direction with the requested zenith cosine and angle to the sun: solve for the azimuth
        ct, g = e["cos_theta"], e["gamma"]
        st = np.sqrt(1 - ct * ct)
        cphi = (np.cos(g) - ct * np.cos(ts)) / (st * np.sin(ts)) if st * np.sin(ts) > 1e-9 else 2.0
        if abs(cphi) > 1.0:
            if not (st < 1e-9 and abs(g - ts) < 1e-9):
                continue  # this (cos, gamma) pair does not exist on the sphere for this sun position
            d = [0.0, 1.0, 0.0]
        else:
            d = [st * cphi, ct, st * np.sqrt(1 - cphi * cphi)]
        O.oracle_sky_eval(C.byref(sky), f3(d), rgb)
        scale = 1000.0 * sky.rgbUnitConversion[0] * sky.multiplier * min(1.0, max(0.0, (np.cos(ts) + 0.05) * 10.0))
        expect = np.maximum(np.array(e["rgb_kcd"]), 0.0) * scale
        assert np.allclose(rgb[:], expect, rtol=2e-3, atol=1e-6 * scale), (e, rgb[:], expect.tolist())
        n += 1
    assert n >= 40


def test_rng_is_xxh32_and_the_jcgt_pcg_hash(built):
    O = oracle_lib.lib()
    for e in GOLD["xxhash32"]:
        assert O.oracle_xxhash32(e["x"], e["y"], e["z"]) == e["h"], e
    for e in GOLD["pcg"]:
        s = C.c_uint32(e["seed"])
        for o in e["outputs"]:
            u = O.oracle_rand(C.byref(s))
            assert u == np.float32(np.uint32((0x3f800000 | (o >> 9))).view(np.float32) - np.float32(1.0))
        assert s.value == e["final_state"]
