"""CPU tier: the DEVICE headers of the shade kernel (csrc/device/pt_bsdf.h, pt_light.h) compiled for the host through
tests/host_shim and diffed against the oracle, lobe by lobe, on seeded random inputs.

Why: the oracle and the device code are two restatements of the same nvshaders functions; a logic slip in one lobe (clearcoat,
sheen, iridescence, anisotropy, diffuse transmission, retroreflection, dispersion ...) would otherwise only show up on the GPU
box.  Both sides here are built with -ffp-contract=off and the same libm, so agreement is expected to a few ulps; the GPU parity
tests (tests/test_gpu_parity.py, test_gpu_lobes.py) remain the parity tests proper.  The shim library is test infrastructure:
it is built here, into the pytest temp directory, and never shipped or loaded by the product (which has no CPU path).
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle_lib
from vk_gltf_renderer_amd import _capi as capi
from vk_gltf_renderer_amd import pathtracer as ptmod

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = C.c_float


@pytest.fixture(scope="module")
def dev(built, tmp_path_factory):
    out = str(tmp_path_factory.mktemp("host_shim") / "libdevice_on_host.so")
    shim = os.path.join(ROOT, "tests", "host_shim")
    subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-shared", "-I" + shim, "-I" + os.path.join(ROOT, "include"),
                    "-I" + os.path.join(ROOT, "vk_gltf_renderer_amd", "csrc", "device"), "-o", out, os.path.join(shim, "device_on_host.cpp")], check=True)
    L = C.CDLL(out)
    P = C.POINTER
    L.dev_divide_by_magic.argtypes, L.dev_divide_by_magic.restype = [C.c_uint32, C.c_uint32], C.c_uint32
    L.dev_bsdf_eval.argtypes = [P(F)] * 5
    L.dev_bsdf_sample.argtypes = [P(F)] * 4
    L.dev_sky_eval.argtypes = [P(capi.MiSkyPhysicalParameters), P(F), P(F)]
    L.dev_sky_pdf.argtypes, L.dev_sky_pdf.restype = [P(capi.MiSkyPhysicalParameters), P(F)], F
    L.dev_sky_sample.argtypes = [P(capi.MiSkyPhysicalParameters), F, F, P(F)]
    L.dev_light_contribution.argtypes = [P(capi.MiGltfLight), P(F), P(F), P(F)]
    return L


def _mat(rng, **over):
    """Flat material of oracle/oracle_pt.h (29 floats), random but valid; `over` pins the lobe under test."""
    d = dict(baseColor=rng.uniform(0.05, 1.0, 3), roughness=rng.uniform(0.02, 1.0, 2), metallic=0.0, ior1=1.0, ior2=rng.uniform(1.1, 2.0), specular=1.0,
             specularColor=(1, 1, 1), transmission=0.0, thickness=0.0, clearcoat=0.0, clearcoatRoughness=0.01, sheenColor=(0, 0, 0), sheenRoughness=0.0,
             iridescence=0.0, iridescenceIor=1.5, iridescenceThickness=100.0, diffuseTransmissionFactor=0.0, diffuseTransmissionColor=(1, 1, 1), dispersion=0.0,
             retroreflection=0.0)
    d.update(over)
    flat = []
    for k in ("baseColor", "roughness", "metallic", "ior1", "ior2", "specular", "specularColor", "transmission", "thickness", "clearcoat", "clearcoatRoughness",
              "sheenColor", "sheenRoughness", "iridescence", "iridescenceIor", "iridescenceThickness", "diffuseTransmissionFactor", "diffuseTransmissionColor",
              "dispersion", "retroreflection"):
        flat.extend(np.atleast_1d(np.asarray(d[k], np.float64)).tolist())
    assert len(flat) == 29
    return (F * 29)(*flat)


LOBES = {
    "diffuse": lambda r: dict(specular=0.0),
    "dielectric": lambda r: dict(),
    "metal": lambda r: dict(metallic=1.0),
    "metal_mix": lambda r: dict(metallic=r.uniform(0.2, 0.8)),
    "anisotropic_metal": lambda r: dict(metallic=1.0, roughness=(r.uniform(0.3, 1.0), r.uniform(0.02, 0.2))),
    "specular_ext": lambda r: dict(specular=r.uniform(0.1, 1.0), specularColor=r.uniform(0.1, 1.0, 3)),
    "clearcoat": lambda r: dict(clearcoat=r.uniform(0.3, 1.0), clearcoatRoughness=r.uniform(0.001, 0.6), metallic=r.uniform(0, 1)),
    "sheen": lambda r: dict(sheenColor=r.uniform(0.1, 1.0, 3), sheenRoughness=r.uniform(0.0015, 1.0)),
    "iridescence_dielectric": lambda r: dict(iridescence=r.uniform(0.2, 1.0), iridescenceIor=r.uniform(1.1, 2.2), iridescenceThickness=r.uniform(50, 1200),
                                             specularColor=r.uniform(0.3, 1.0, 3)),
    "iridescence_metal": lambda r: dict(iridescence=r.uniform(0.2, 1.0), iridescenceIor=r.uniform(1.1, 2.2), iridescenceThickness=r.uniform(50, 1200), metallic=1.0),
    "transmission_thin": lambda r: dict(transmission=r.uniform(0.3, 1.0), thickness=0.0),
    "transmission_volume": lambda r: dict(transmission=1.0, thickness=1.0),
    "transmission_inside": lambda r: dict(transmission=1.0, thickness=1.0, ior1=1.5, ior2=1.0),
    "dispersion": lambda r: dict(transmission=1.0, thickness=1.0, dispersion=r.uniform(1.0, 20.0)),
    "dispersion_inside": lambda r: dict(transmission=1.0, thickness=1.0, dispersion=r.uniform(1.0, 20.0), ior1=1.6, ior2=1.0),
    "diffuse_transmission": lambda r: dict(diffuseTransmissionFactor=r.uniform(0.2, 1.0), diffuseTransmissionColor=r.uniform(0.1, 1.0, 3)),
    "retroreflection": lambda r: dict(retroreflection=r.uniform(0.2, 1.0), metallic=r.uniform(0, 1)),
    "retro_coat_sheen": lambda r: dict(retroreflection=r.uniform(0.2, 1.0), clearcoat=0.7, clearcoatRoughness=0.2, sheenColor=(0.5, 0.5, 0.5), sheenRoughness=0.5),
    "everything": lambda r: dict(metallic=r.uniform(0, 1), clearcoat=r.uniform(0, 1), clearcoatRoughness=r.uniform(0.01, 0.5), sheenColor=r.uniform(0, 1, 3),
                                 sheenRoughness=r.uniform(0.01, 1), iridescence=r.uniform(0, 1), iridescenceThickness=r.uniform(100, 800), transmission=r.uniform(0, 1),
                                 thickness=float(r.integers(0, 2)), diffuseTransmissionFactor=r.uniform(0, 1), retroreflection=r.uniform(0, 1), dispersion=r.uniform(0, 10)),
}


def _unit(rng, upper=False):
    d = rng.normal(size=3)
    d /= np.linalg.norm(d)
    if upper:
        d[2] = abs(d[2])
    return (F * 3)(*d)


@pytest.mark.parametrize("lobe", sorted(LOBES))
def test_bsdf_lobes_device_headers_match_oracle(dev, lobe):
    O = oracle_lib.lib()
    rng = np.random.default_rng(abs(hash(lobe)) % (1 << 31) if False else sum(map(ord, lobe)))
    so, sd, eo, ed = (F * 8)(), (F * 8)(), (F * 7)(), (F * 4)()
    events = set()
    nonzero_eval = 0
    for _ in range(1500):
        m = _mat(rng, **LOBES[lobe](rng))
        k1 = _unit(rng, upper=rng.random() < 0.9)  # mostly the front side; some views from behind (absorb paths)
        xi = (F * 3)(*rng.random(3))
        O.oracle_bsdf_sample(m, k1, xi, so)
        dev.dev_bsdf_sample(m, k1, xi, sd)
        a, b = np.array(so[:]), np.array(sd[:])
        assert int(a[7]) == int(b[7]), (lobe, a, b)
        events.add(int(a[7]))
        if int(a[7]) != 0:
            # the device headers take sines / cosines of 2 pi u in revolutions (pt_math.h sinTurns: the shim evaluates sin(2 pi t)), the
            # oracle rounds the angle first: 1e-7 of an angle, which grazing half vectors amplify to some 5e-5 of a weight
            assert np.allclose(a[:7], b[:7], rtol=1e-4, atol=1e-6), (lobe, a, b)
        k2 = _unit(rng) if rng.random() < 0.5 else (F * 3)(a[0], a[1], a[2])  # random direction, or the sampled one (non-zero eval)
        O.oracle_bsdf_eval(m, k1, k2, xi, eo)
        dev.dev_bsdf_eval(m, k1, k2, xi, ed)
        eo_ = np.array(eo[:])
        # the device returns diffuse * occlusion + glossy in one vector (occlusion = 1 in this hook)
        assert np.allclose(eo_[0:3] + eo_[3:6], np.array(ed[0:3]), rtol=2e-5, atol=1e-7), (lobe, eo_, ed[:])
        assert np.isclose(eo_[6], ed[3], rtol=2e-5, atol=1e-7), (lobe, eo_, ed[:])
        nonzero_eval += eo_[6] > 0
    assert len(events - {0}) >= 1 and nonzero_eval > 100, (lobe, events, nonzero_eval)


def test_sky_device_headers_match_oracle(dev):
    O = oracle_lib.lib()
    rng = np.random.default_rng(3)
    for trial in range(6):
        sky = ptmod.default_sky()
        if trial:
            sky.haze, sky.redblueshift, sky.saturation = rng.uniform(0, 8), rng.uniform(-0.5, 0.5), rng.uniform(0.2, 1.5)
            sky.horizonHeight, sky.horizonBlur, sky.sunDiskScale, sky.sunGlowIntensity = rng.uniform(-0.2, 0.2), rng.uniform(0.05, 1), rng.uniform(0.5, 4), rng.uniform(0, 2)
            d = rng.normal(size=3); d[1] = abs(d[1]) + 0.05; d /= np.linalg.norm(d)
            sky.sunDirection[:] = d.tolist()
            sky.yIsUp = int(trial % 2)
            if not sky.yIsUp:
                sky.sunDirection[:] = [d[0], d[2], d[1]]
        a, b = (F * 3)(), (F * 3)()
        sa, sb = (F * 7)(), (F * 7)()
        for _ in range(400):
            dr = _unit(rng)
            O.oracle_sky_eval(C.byref(sky), dr, a); dev.dev_sky_eval(C.byref(sky), dr, b)
            assert np.allclose(a[:], b[:], rtol=3e-5, atol=1e-7), (trial, a[:], b[:])
            assert np.isclose(O.oracle_sky_pdf(C.byref(sky), dr), dev.dev_sky_pdf(C.byref(sky), dr), rtol=3e-5)
            u, v = rng.random(2)
            O.oracle_sky_sample(C.byref(sky), u, v, sa); dev.dev_sky_sample(C.byref(sky), u, v, sb)
            assert np.allclose(sa[:], sb[:], rtol=1e-4, atol=1e-6), (trial, sa[:], sb[:])


def test_lights_device_headers_match_oracle(dev):
    """singleLightContribution for directional (delta and with an angular size), point (delta, sphere, range window) and spot cones."""
    O = oracle_lib.lib()
    rng = np.random.default_rng(9)
    a, b = (F * 8)(), (F * 8)()
    kinds = 0
    for _ in range(3000):
        L = capi.MiGltfLight()
        L.type = int(rng.choice([capi.MI_LIGHT_DIRECTIONAL, capi.MI_LIGHT_POINT, capi.MI_LIGHT_SPOT]))
        d = rng.normal(size=3); d /= np.linalg.norm(d)
        L.direction[:] = d.tolist()
        L.position[:] = rng.uniform(-3, 3, 3).tolist()
        L.color[:] = rng.uniform(0.1, 1, 3).tolist()
        L.intensity = rng.uniform(0.5, 50)
        L.radius = float(rng.choice([0.0, rng.uniform(0.05, 1.5)]))
        if L.type == capi.MI_LIGHT_DIRECTIONAL:
            L.angularSizeOrInvRange = float(rng.choice([0.0, rng.uniform(1e-4, 0.3)]))
        else:
            L.angularSizeOrInvRange = float(rng.choice([0.0, 1.0 / rng.uniform(2.0, 10.0)]))
        L.innerAngle = rng.uniform(0.0, 0.6)
        L.outerAngle = L.innerAngle + rng.uniform(0.0, 0.6)
        pos = (F * 3)(*rng.uniform(-4, 4, 3))
        xi = (F * 2)(*rng.random(2))
        O.oracle_light_contribution(C.byref(L), pos, xi, a)
        dev.dev_light_contribution(C.byref(L), pos, xi, b)
        assert np.allclose(a[:], b[:], rtol=3e-5, atol=1e-7), (L.type, a[:], b[:])
        kinds |= 1 << L.type
    assert kinds == (1 << capi.MI_LIGHT_DIRECTIONAL) | (1 << capi.MI_LIGHT_POINT) | (1 << capi.MI_LIGHT_SPOT)


def test_slot_division_by_multiply_high_is_exact(dev):
    """k_trace_primary / k_generate turn `slot / numSlots` into mulhi(slot, magic) >> shift (FrameConsts::slotsMagic, pt_scene.h:
    divideMagic): exact for every slot below 2^31 -- checked at the multiples of the divisor and their neighbours, at the ends of the
    range and on random numbers, for the slot counts real resolutions give, powers of two, and awkward divisors."""
    rng = np.random.default_rng(11)
    divisors = [2, 3, 256, 1024, 2304, 8192, 921600, 2073600, 2088960, 8294400, 8355840, 1 << 20, (1 << 20) + 1024, 123456789, (1 << 31) - 1]
    divisors += [int(d) for d in rng.integers(2, 1 << 24, 40)]
    for d in divisors:
        ns = {0, 1, d - 1, d, d + 1, (1 << 31) - 1, (1 << 31) - 2}
        top = ((1 << 31) - 1) // d
        for q in {1, 2, 3, top // 2, top - 1, top} | {int(q) for q in rng.integers(0, top + 1, 50)}:
            for n in (q * d - 1, q * d, q * d + 1):
                if 0 <= n < (1 << 31):
                    ns.add(n)
        ns |= {int(n) for n in rng.integers(0, 1 << 31, 200)}
        for n in ns:
            assert dev.dev_divide_by_magic(n, d) == n // d, (n, d)
