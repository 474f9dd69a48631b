"""Known-answer tests that pin the CPU oracle independently of the GPU (the reference ships no golden
radiance, so the oracle supplies analytic KATs): RNG bit vectors, binary16 rounding, BSDF energy/pdf checks, sky pdf
normalisation, furnace and Lambert-plane renders with closed-form answers."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import oracle_lib
import parity_util as pu
from vk_gltf_renderer_amd import _capi as capi
from vk_gltf_renderer_amd import pathtracer as ptmod
from vk_gltf_renderer_amd import scenegen

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
F = C.c_float


def _xxhash32_py(x, y, z):
    M = 0xFFFFFFFF
    P2, P3, P4, P5 = 2246822519, 3266489917, 668265263, 374761393
    rotl = lambda v, r: ((v << r) | (v >> (32 - r))) & M
    h = (z + P5 + x * P3) & M
    h = (P4 * rotl(h, 17)) & M
    h = (h + y * P3) & M
    h = (P4 * rotl(h, 17)) & M
    h = (P2 * (h ^ (h >> 15))) & M
    h = (P3 * (h ^ (h >> 13))) & M
    return h ^ (h >> 16)


def _pcg_py(state):
    M = 0xFFFFFFFF
    prev = (state * 747796405 + 2891336453) & M
    word = (((prev >> ((prev >> 28) + 4)) ^ prev) * 277803737) & M
    return prev, ((word >> 22) ^ word) & M


def test_rng_bit_vectors(built):
    """xxhash32 / pcg / rand against an independent pure-Python implementation and the committed golden vectors."""
    O = oracle_lib.lib()
    golden = json.load(open(os.path.join(GOLDEN, "rng_vectors.json")))
    for x, y, z, h in golden["xxhash32"]:
        assert O.oracle_xxhash32(x, y, z) == h == _xxhash32_py(x, y, z)
    for seed0, expected_bits in golden["rand_stream"]:
        seed = C.c_uint32(seed0)
        state = seed0
        for bits in expected_bits:
            v = np.float32(O.oracle_rand(C.byref(seed)))
            state, word = _pcg_py(state)
            ref = np.uint32(0x3F800000 | (word >> 9)).view(np.float32) - np.float32(1.0)
            assert v == ref and v.view(np.uint32) == bits
            assert 0.0 <= v < 1.0 and seed.value == state


def test_round_to_half(built):
    O = oracle_lib.lib()
    rng = np.random.default_rng(0)
    vals = np.concatenate([rng.normal(size=2000) * 10.0 ** rng.integers(-9, 6, 2000), [0.0, 65504.0, 65519.9, 65520.0, 1e-8, 6e-8, 3e-5, -2.5]]).astype(np.float32)
    with np.errstate(over="ignore"):
        ref = vals.astype(np.float16).astype(np.float32)
    got = np.array([O.oracle_round_to_half(float(v)) for v in vals], dtype=np.float32)
    assert (got.view(np.uint32) == ref.view(np.uint32)).all()


def _mat(**kw):
    """29-float material array of oracle_bsdf_* (see oracle/oracle_pt.h)."""
    d = dict(baseColor=(1, 1, 1), roughness=(0.25, 0.25), metallic=0.0, ior1=1.0, ior2=1.5, specular=1.0, specularColor=(1, 1, 1), transmission=0.0,
             thickness=0.0, clearcoat=0.0, clearcoatRoughness=0.01, sheenColor=(0, 0, 0), sheenRoughness=0.0, iridescence=0.0, iridescenceIor=1.3,
             iridescenceThickness=400.0, diffuseTransmissionFactor=0.0, diffuseTransmissionColor=(1, 1, 1), dispersion=0.0, retroreflection=0.0)
    d.update(kw)
    flat = [*d["baseColor"], *d["roughness"], d["metallic"], d["ior1"], d["ior2"], d["specular"], *d["specularColor"], d["transmission"], d["thickness"],
            d["clearcoat"], d["clearcoatRoughness"], *d["sheenColor"], d["sheenRoughness"], d["iridescence"], d["iridescenceIor"], d["iridescenceThickness"],
            d["diffuseTransmissionFactor"], *d["diffuseTransmissionColor"], d["dispersion"], d["retroreflection"]]
    assert len(flat) == 29
    return (F * 29)(*flat)


def _sample_many(m, k1, n, seed=1):
    O = oracle_lib.lib()
    rng = np.random.default_rng(seed)
    out = (F * 8)()
    k1c = (F * 3)(*k1)
    res = np.empty((n, 8), np.float32)
    xis = rng.random((n, 3)).astype(np.float32)
    for i in range(n):
        O.oracle_bsdf_sample(m, k1c, (F * 3)(*xis[i]), out)
        res[i] = out[:]
    return res


@pytest.mark.parametrize("mat", [
    dict(baseColor=(1, 1, 1), metallic=0.0, specular=0.0),                       # pure Lambert
    dict(baseColor=(1, 1, 1), metallic=1.0, roughness=(0.3, 0.3)),               # white metal
    dict(baseColor=(1, 1, 1), metallic=0.0, roughness=(0.2, 0.2)),               # dielectric + diffuse
    dict(baseColor=(1, 1, 1), transmission=1.0, roughness=(0.1, 0.1), thickness=1.0),  # rough glass
    dict(baseColor=(1, 1, 1), clearcoat=1.0, roughness=(0.5, 0.5)),
    dict(baseColor=(1, 1, 1), sheenColor=(1, 1, 1), sheenRoughness=0.5),
])
def test_bsdf_white_furnace(built, mat):
    """Energy conservation: with every colour = 1 the sampling weight (bsdf*cos/pdf) never exceeds 1 and its mean stays
    within (0.5, 1] — i.e. the layered lobes lose at most the single-scattering masking energy, never gain any."""
    theta = np.radians(50.0)
    k1 = (np.sin(theta), 0.0, np.cos(theta))
    res = _sample_many(_mat(**mat), k1, 4000)
    w = res[:, 3:6]
    assert np.isfinite(res).all()
    assert w.max() <= 1.0 + 1e-4
    assert 0.5 < w.mean() <= 1.0 + 1e-6


def test_bsdf_sample_eval_consistency(built):
    """For the directions bsdfSample produces, bsdfEvaluate with the same lobe choice returns bsdf/pdf == bsdf_over_pdf
    and the same pdf (the MIS weights of gltf_pathtrace.slang:344 rely on it)."""
    O = oracle_lib.lib()
    rng = np.random.default_rng(5)
    m = _mat(baseColor=(0.8, 0.5, 0.3), metallic=0.3, roughness=(0.35, 0.2), clearcoat=0.4, clearcoatRoughness=0.2)
    k1 = np.array([0.4, 0.2, 0.89], np.float32)
    k1 /= np.linalg.norm(k1)
    s, e = (F * 8)(), (F * 7)()
    checked = 0
    for _ in range(3000):
        xi = rng.random(3).astype(np.float32)
        O.oracle_bsdf_sample(m, (F * 3)(*k1), (F * 3)(*xi), s)
        if int(s[7]) == 0:
            continue
        O.oracle_bsdf_eval(m, (F * 3)(*k1), (F * 3)(s[0], s[1], s[2]), (F * 3)(*xi), e)
        pdf_s, pdf_e = s[6], e[6]
        if pdf_e <= 0:
            continue
        assert pdf_e == pytest.approx(pdf_s, rel=2e-3)
        ratio = (np.array(e[0:3]) + np.array(e[3:6])) / pdf_e
        assert ratio == pytest.approx(np.array(s[3:6]), rel=3e-3, abs=1e-5)
        checked += 1
    assert checked > 2000


def test_bsdf_pdf_normalised(built):
    """Monte-Carlo estimate of the integral of the evaluated pdf over the sphere is <= 1 (== 1 minus absorbed samples)."""
    O = oracle_lib.lib()
    rng = np.random.default_rng(7)
    m = _mat(baseColor=(0.9, 0.9, 0.9), metallic=0.0, roughness=(0.3, 0.3))
    k1 = (F * 3)(0.5, 0.0, 0.8660254)
    n, acc = 20000, 0.0
    e = (F * 7)()
    for _ in range(n):
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        O.oracle_bsdf_eval(m, k1, (F * 3)(*d), (F * 3)(*rng.random(3)), e)
        acc += e[6]
    integral = acc / n * 4 * np.pi
    assert 0.85 < integral < 1.05


def test_sky_pdf_and_sampling(built):
    O = oracle_lib.lib()
    sky = ptmod.default_sky()
    rng = np.random.default_rng(11)
    out = (F * 7)()
    # sampled direction is unit, its pdf equals samplePhysicalSkyPDF(direction), radiance equals evalPhysicalSky(direction)
    n_sun = 0
    for _ in range(500):
        u, v = rng.random(2)
        O.oracle_sky_sample(C.byref(sky), u, v, out)
        d = np.array(out[0:3])
        assert np.linalg.norm(d) == pytest.approx(1.0, abs=1e-5)
        dd = (F * 3)(*d)
        assert O.oracle_sky_pdf(C.byref(sky), dd) == pytest.approx(out[3], rel=1e-5)
        rgb = (F * 3)()
        O.oracle_sky_eval(C.byref(sky), dd, rgb)
        assert np.array(rgb[:]) == pytest.approx(np.array(out[4:7]), rel=1e-4, abs=1e-7)
        n_sun += out[3] > 1.0
    assert 180 < n_sun < 320  # ~50 % of the samples go to the sun cone
    # the uniform-sphere part of the pdf integrates to 1 - wSun = 0.5
    dirs = rng.normal(size=(4000, 3))
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    p = np.array([O.oracle_sky_pdf(C.byref(sky), (F * 3)(*d)) for d in dirs])
    assert np.median(p) * 4 * np.pi == pytest.approx(0.5, rel=1e-4)


def _uniform_env(value, w=64, h=32):
    return np.full((h, w, 3), value, np.float32)


def test_empty_scene_shows_environment(built, tmp_path):
    """A camera ray that misses everything returns the environment radiance exactly (lastSamplePdf = DIRAC -> MIS weight 1;
    gltf_pathtrace.slang:129-156)."""
    b = scenegen.GlbBuilder()
    b.material({})
    pos = np.array([[100, 100, 100], [101, 100, 100], [100, 101, 100]], np.float32)  # far away from the view
    b.node(mesh=b.mesh([b.primitive(pos, np.array([0, 1, 2]), material=0)]))
    b.camera_node((0, 0, 3), (0, 0, 0))
    s = pu.Setup(b.save(str(tmp_path / "e.glb")), 32, 24, hdr_pixels=_uniform_env(0.7))
    o = pu.render_oracle(s, 2, threads=2)
    assert np.allclose(o["accum"][..., :3], 0.7, rtol=1e-6) and (o["accum"][..., 3] == 0).all()
    assert (o["selection"] == 0).all() and (o["depth"] == 1.0).all()


def test_white_furnace_sphere(built, tmp_path):
    """A convex, purely Lambertian white object inside a uniform environment is invisible: radiance == L everywhere
    (NEE + BSDF-sampled environment hits combined by MIS must sum to exactly one bounce of energy)."""
    path = scenegen.scene_sphere(str(tmp_path / "s.glb"), scenegen.lambert_material((1, 1, 1)), 48, 24)
    s = pu.Setup(path, 48, 48, hdr_pixels=_uniform_env(0.5), max_depth=4, spp_per_frame=8,
                 params_edit=lambda p: setattr(p, "fireflyClampThreshold", 1e9))
    o = pu.render_oracle(s, 24)  # 192 spp
    img, hit = o["accum"][..., :3], o["accum"][..., 3] > 0.5
    assert hit.mean() > 0.15
    assert img[hit].mean() == pytest.approx(0.5, rel=0.01)
    assert np.abs(img[hit].reshape(-1, 3).mean(0) - 0.5).max() < 0.01
    assert (o["selection"][o["accum"][..., 3] == 1.0] == 1).all()  # every sample hit -> the centre ray hits render node 0


def test_lambert_plane_directional_light(built, tmp_path):
    """Closed form: L = albedo / pi * E * cos(theta) for a Lambert plane under a delta directional light, black
    environment (one-sample NEE with pdf = DIRAC; pathtrace_functions.h.slang:396-415)."""
    ang = np.radians(40.0)
    light = {"def": {"type": "directional", "intensity": 3.0, "color": [1.0, 0.5, 0.25]},
             # node -Z must point along the light's travel direction: rotate -Z (0,0,-1) to (sin a, -cos a, 0)... use a quaternion about x then z
             "node": {"rotation": _quat_from_to((0, 0, -1), (np.sin(ang), -np.cos(ang), 0.0))}}
    path = scenegen.scene_plane_with_light(str(tmp_path / "p.glb"), albedo=0.5, light=light)
    s = pu.Setup(path, 32, 32, hdr_pixels=_uniform_env(0.0), max_depth=2, spp_per_frame=16)
    o = pu.render_oracle(s, 16)
    expect = 0.5 / np.pi * 3.0 * np.cos(ang) * np.array([1.0, 0.5, 0.25])
    got = o["accum"][8:24, 8:24, :3].reshape(-1, 3).mean(0)
    assert got == pytest.approx(expect, rel=0.02)


def _quat_from_to(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    a, b = a / np.linalg.norm(a), b / np.linalg.norm(b)
    c = np.cross(a, b)
    w = 1.0 + a @ b
    q = np.array([*c, w])
    q /= np.linalg.norm(q)
    return [float(v) for v in q]


def test_tile_partition_sums_to_full_frame(built, assets):
    """Tiles owned by different ranks are disjoint and their union is the 1-rank image, bit for bit."""
    s = pu.Setup(os.path.join(assets, "Box.glb"), 96, 80, max_depth=3)
    full = pu.render_oracle(s, 2)["accum"]
    parts = [pu.render_oracle(s, 2, tile=(r, 3, 16))["accum"] for r in range(3)]
    assert (np.sum(parts, axis=0) == full).all()
    assert all((p != 0).any() for p in parts)
