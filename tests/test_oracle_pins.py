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


def test_sheen_lobe_shape_and_vcavities_masking(built):
    """KHR_materials_sheen: the sin^n half-vector density (Conty Estevez & Kulla 2017 eq. 2, as a density over the solid angle of h)
    on a grid, its normalisation by float64 quadrature, its sampler against the analytic mean E[sin(theta_h)] = (n + 2) / (n + 3),
    and the V-cavities masking term (Torrance & Sparrow 1967)."""
    O = oracle_lib.lib()
    for e in GOLD["sheen"]:
        assert O.oracle_sheen_ndf(e["n"], e["cos_h"]) == pytest.approx(e["pdf_h"], rel=5e-5, abs=1e-30), e
    for n in (1.0, 11.11, 100.0):
        c = (np.arange(20000) + 0.5) / 20000  # cos(theta_h) in (0, 1): d(omega) = 2 pi d(cos)
        assert sum(O.oracle_sheen_ndf(n, float(x)) for x in c[::20]) * 2 * np.pi * (20 / 20000) == pytest.approx(1.0, abs=4e-3)
    rng = np.random.default_rng(11)
    h = (F * 3)()
    for n in (2.0, 25.0, 400.0):
        s = []
        for a, b in rng.random((20000, 2)):
            O.oracle_sheen_sample(a, b, n, h)
            assert np.linalg.norm(h[:]) == pytest.approx(1.0, abs=1e-5) and h[2] >= 0.0
            s.append(np.hypot(h[0], h[1]))
        assert np.mean(s) == pytest.approx((n + 2.0) / (n + 3.0), abs=4e-3)
    for e in GOLD["vcavities"]:
        assert O.oracle_vcavities_g(e["nh"], e["k1h"], e["k1z"], e["k2h"], e["k2z"]) == pytest.approx(e["G"], rel=2e-5), e


def test_clearcoat_lobe_probability_is_the_fresnel_term_of_an_ior_1_5_layer(built):
    """KHR_materials_clearcoat: the coat reflects with probability clearcoat x F(1.5 / ior1, cos) and everything below shares the
    rest (the lobe weights sum to one)."""
    O = oracle_lib.lib()
    coat = O.oracle_lobe_index(b"clearcoat")
    assert coat >= 0
    w = (F * 6)()
    for e in GOLD["clearcoat_weight"]:
        m = np.zeros(29, np.float32)
        m[0:3] = 0.8; m[3] = m[4] = 0.3; m[5] = 0.2; m[6], m[7], m[8] = e["ior1"], 1.5, 1.0; m[9:12] = 1.0
        m[14], m[15] = e["clearcoat"], 0.1; m[21], m[22] = 1.5, 100.0; m[24:27] = 1.0
        O.oracle_lobe_weights((F * 29)(*m), e["cos"], w)
        assert w[coat] == pytest.approx(e["w"], rel=3e-5, abs=1e-7), (e, w[:])
        assert sum(w[:]) == pytest.approx(1.0, abs=1e-5)
        assert all(x >= 0.0 for x in w[:])


def test_shadow_terminator_offset_and_ray_cone_footprint(built):
    """pointOffset against Hanika's listing in float64, and its two defining properties: a triangle whose vertex normals are its
    geometric normal is not moved, and on a convex tessellated sphere the offset point leaves the facet outwards and stays below the
    highest of the three vertex tangent planes it blends (a convex combination of projections onto them).
    rayConeWorldFootprint against the ray-cone formulas of Akenine-Moller et al. 2019."""
    O = oracle_lib.lib()
    out = (F * 3)()
    for e in GOLD["point_offset"]:
        O.oracle_point_offset(f3(e["p"]), (F * 9)(*e["tri"]), (F * 9)(*e["nrm"]), f3(e["bary"]), out)
        assert np.allclose(out[:], e["offset_p"], rtol=2e-5, atol=2e-6), (e, out[:])
    rng = np.random.default_rng(3)
    for _ in range(32):
        tri = rng.normal(size=(3, 3))
        ng = np.cross(tri[1] - tri[0], tri[2] - tri[0]); ng /= np.linalg.norm(ng)
        b = rng.dirichlet((1, 1, 1))
        P = b @ tri
        O.oracle_point_offset(f3(P), (F * 9)(*tri.reshape(-1)), (F * 9)(*np.tile(ng, 3)), f3(b), out)
        assert np.allclose(out[:], P, atol=2e-6)
        c = rng.normal(size=3)
        v = rng.normal(size=(3, 3)) * 0.2 + c / np.linalg.norm(c)
        v /= np.linalg.norm(v, axis=1, keepdims=True)  # three nearby points of the unit sphere, normals = positions
        P = b @ v
        O.oracle_point_offset(f3(P), (F * 9)(*v.reshape(-1)), (F * 9)(*v.reshape(-1)), f3(b), out)
        r = np.linalg.norm(out[:])
        Ph = P / np.linalg.norm(P)
        assert np.linalg.norm(P) - 1e-6 <= r <= 1.0 / min(float(Ph @ v[i]) for i in range(3)) + 1e-5
    for e in GOLD["ray_cone"]:
        n, v = [0.0, 0.0, 1.0], [float(np.sqrt(max(0.0, 1 - e["cos"] ** 2))), 0.0, e["cos"]]
        assert O.oracle_ray_cone_footprint(e["width"], e["spread"], e["t"], f3(n), f3(v)) == pytest.approx(e["footprint"], rel=3e-5), e


def test_hdr_importance_table_and_sampler_against_numpy(built, assets):
    """nvvk::HdrIbl's job (src/renderer.cpp:1982-2017) as libmi_host does it, on the reference's own std_env.hdr: (1) the pdf stored
    in alpha is max(r, g, b) / integral of max(r, g, b) over the sphere (float64 numpy, exact texel solid angles); (2) the alias
    table selects texel i with probability importance_i / total (Vose's construction, checked by summing the table's mass per
    texel); (3) the oracle's environmentSample lands in coarse bins of the map with the frequencies the numpy distribution
    predicts, inside the texel the table chose, and returns the map's bilinear (rgb, pdf) there."""
    O = oracle_lib.lib()
    hdr = ptmod.HdrEnvironment(path=os.path.join(assets, "std_env.hdr"))
    env = hdr.env.contents if hasattr(hdr.env, "contents") else hdr.env
    w, h = env.width, env.height
    rgba = np.ctypeslib.as_array(env.rgba, shape=(h, w, 4)).astype(np.float64)
    m = rgba[..., :3].max(axis=-1)
    theta = np.arange(h + 1) * np.pi / h
    area = (np.cos(theta[:-1]) - np.cos(theta[1:]))[:, None] * (2 * np.pi / w)  # solid angle of the texels of a row
    imp = m * area
    total = imp.sum()
    assert env.integral == pytest.approx(total, rel=1e-5)
    assert np.allclose(rgba[..., 3], m / total, rtol=2e-5, atol=1e-12)
    assert (rgba[..., 3] * area).sum() == pytest.approx(1.0, rel=1e-5)  # a density over the sphere
    acc = np.ctypeslib.as_array(C.cast(env.accel, C.POINTER(C.c_uint32)), shape=(h * w, 2))
    alias, q = acc[:, 0].astype(np.int64), acc[:, 1].copy().view(np.float32).astype(np.float64)
    assert (q >= 0).all() and (q <= 1.0 + 1e-6).all() and (alias < h * w).all()
    mass = q.copy()
    np.add.at(mass, alias, 1.0 - q)
    p_table, p_true = mass / (h * w), imp.reshape(-1) / total
    assert np.abs(p_table - p_true).max() < 2e-6 * p_true.max() + 1e-9 and np.abs(p_table - p_true).sum() < 2e-4
    # the sampler: bins of 8 x 4 texel blocks ... coarse enough for 200k samples
    rng = np.random.default_rng(9)
    n = 200000
    bx, by = next(d for d in (16, 15, 12, 10, 8, 5, 4, 2, 1) if w % d == 0), next(d for d in (8, 10, 6, 5, 4, 2, 1) if h % d == 0)
    counts = np.zeros((by, bx))
    out = (F * 7)()
    xi = rng.random((n, 3)).astype(np.float32)
    for k in range(n):
        O.oracle_env_sample(hdr.env, f3(xi[k]), out)
        d = np.array(out[:3], np.float64)
        assert abs(np.linalg.norm(d) - 1.0) < 1e-4
        u = (np.arctan2(d[2], d[0]) + np.pi) / (2 * np.pi)
        v = np.arccos(np.clip(d[1], -1, 1)) / np.pi
        counts[min(by - 1, int(v * by)), min(bx - 1, int(u * bx))] += 1
        if k < 2000:  # the returned (rgb, pdf) is the map's bilinear value at the sampled direction
            fx, fy = u * w - 0.5, v * h - 0.5
            x0, y0 = int(np.floor(fx)), int(np.floor(fy))
            tx, ty = fx - x0, fy - y0
            px = lambda x, y: rgba[min(max(y, 0), h - 1), x % w]
            ref = (px(x0, y0) * (1 - tx) + px(x0 + 1, y0) * tx) * (1 - ty) + (px(x0, y0 + 1) * (1 - tx) + px(x0 + 1, y0 + 1) * tx) * ty
            assert np.allclose(out[3:7], ref, rtol=2e-2, atol=1e-3 * ref.max()), (k, out[:], ref)
    expect = p_true.reshape(by, h // by, bx, w // bx).sum(axis=(1, 3)) * n
    sigma = np.sqrt(np.maximum(expect, 1.0))
    assert (np.abs(counts - expect) < 5 * sigma + 3).all(), (counts - expect) / sigma
