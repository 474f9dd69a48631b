"""The device tonemapper (replaces GltfRenderer::tonemap -> nvshaders::Tonemapper, reference src/renderer.cpp:992-1056,
method set src/renderer.cpp:173-179) against the PUBLISHED form of each operator written in numpy float64: Hejl / Burgess-Dawson
filmic, Hable "Uncharted 2" (W = 11.2, exposure bias 2), S. Hill's ACES fit, B. Wrensch's minimal AgX, Khronos PBR Neutral, IEC
61966-2-1 sRGB.  8-bit outputs may differ by one code where the float32 device arithmetic rounds the other way."""
import os

import numpy as np
import pytest

import parity_util as pu
from vk_gltf_renderer_amd import _capi as capi
from vk_gltf_renderer_amd import pathtracer as ptmod

pytestmark = pytest.mark.gpu


def _srgb(x):
    x = np.asarray(x, np.float64)
    return np.where(x > 0.0031308, 1.055 * np.power(np.maximum(x, 1e-30), 1 / 2.4) - 0.055, 12.92 * x)


def _filmic(c):
    t = np.maximum(0.0, c - 0.004)
    return (t * (6.2 * t + 0.5)) / (t * (6.2 * t + 1.7) + 0.06)


def _hable(x):
    a, b, c, d, e, f = 0.15, 0.50, 0.10, 0.20, 0.02, 0.30
    return ((x * (a * x + c * b) + d * e) / (x * (a * x + b) + d * f)) - e / f


def _uncharted(c):
    return _srgb(_hable(c * 2.0) / _hable(11.2))


def _aces(c):
    m_in = np.array([[0.59719, 0.35458, 0.04823], [0.07600, 0.90834, 0.01566], [0.02840, 0.13383, 0.83777]])
    m_out = np.array([[1.60475, -0.53108, -0.07367], [-0.10208, 1.10813, -0.00605], [-0.00327, -0.07276, 1.07602]])
    v = c @ m_in.T
    v = (v * (v + 0.0245786) - 0.000090537) / (v * (0.983729 * v + 0.4329510) + 0.238081)
    return _srgb(v @ m_out.T)


def _agx(c):
    # column-major GLSL mat3 constants of the published minimal AgX, written here as row-major numpy matrices (transposed)
    inset = np.array([[0.842479062253094, 0.0784335999999992, 0.0792237451477643], [0.0423282422610123, 0.878468636469772, 0.0791661274605434],
                      [0.0423756549057051, 0.0784336, 0.879142973793104]])
    outset = np.array([[1.19687900512017, -0.0980208811401368, -0.0990297440797205], [-0.0528968517574562, 1.15190312990417, -0.0989611768448433],
                       [-0.0529716355144438, -0.0980434501171241, 1.15107367264116]])
    lo, hi = -12.47393, 4.026069
    v = c @ inset.T
    with np.errstate(divide="ignore", invalid="ignore"):
        x = (np.clip(np.where(v > 0, np.log2(np.maximum(v, 1e-300)), lo), lo, hi) - lo) / (hi - lo)
    x2, x4 = x * x, x * x * x * x
    v = 15.5 * x4 * x2 - 40.14 * x4 * x + 31.96 * x4 - 6.868 * x2 * x + 0.4298 * x2 + 0.1191 * x - 0.00232
    return v @ outset.T


def _khronos(c):
    start, desat = 0.8 - 0.04, 0.15
    x = c.min(-1, keepdims=True)
    c = c - np.where(x < 0.08, x - 6.25 * x * x, 0.04)
    peak = c.max(-1, keepdims=True)
    d = 1.0 - start
    new_peak = 1.0 - d * d / (peak + d - start)
    comp = c * (new_peak / np.maximum(peak, 1e-30))
    g = 1.0 - 1.0 / (desat * (peak - new_peak) + 1.0)
    comp = comp + (new_peak - comp) * g
    return _srgb(np.where(peak < start, c, comp))


OPERATORS = {"filmic": _filmic, "uncharted": _uncharted, "clip": lambda c: _srgb(np.maximum(c, 0.0)), "aces": _aces, "agx": _agx, "khronos_pbr": _khronos}


def _reference(img, method, exposure=1.0, brightness=1.0, contrast=1.0, saturation=1.0, vignette=0.0):
    H, W, _ = img.shape
    c = OPERATORS[method](img[..., :3].astype(np.float64) * exposure)
    c = np.clip(0.5 + (c - 0.5) * contrast, 0.0, 1.0) ** (1.0 / brightness)
    luma = (c * np.array([0.299, 0.587, 0.114])).sum(-1, keepdims=True)
    c = luma + (c - luma) * saturation
    u = ((np.arange(W) + 0.5) / W - 0.5) * 2.0
    v = ((np.arange(H) + 0.5) / H - 0.5) * 2.0
    c = c * (1.0 - (u[None, :] ** 2 + v[:, None] ** 2) * vignette)[..., None]
    out = np.empty((H, W, 4), np.float64)
    out[..., :3] = c
    out[..., 3] = img[..., 3]
    return np.clip(out, 0.0, 1.0) * 255.0


def _tracer_with_image(assets, img):
    """A PathTracer whose accumulator holds `img` (mi_pt_write_accum)."""
    H, W, _ = img.shape
    scene = ptmod.Scene(os.path.join(assets, "Box.glb"))
    t = ptmod.PathTracer(scene)
    t.resize(W, H)
    t.write_accum(img)
    return t, None


@pytest.mark.parametrize("method", capi.TONEMAP_METHODS)
def test_operators_match_published_forms(built, assets, method):
    rng = np.random.default_rng(7)
    H, W = 96, 160
    img = np.empty((H, W, 4), np.float32)
    img[..., :3] = np.exp2(rng.uniform(-14.0, 5.0, (H, W, 3))).astype(np.float32)  # 19 stops, every channel on its own
    img[0, :16, :3] = 0.0  # black, and a few exact greys / saturated primaries
    img[1, :16, :3] = np.linspace(0.0, 4.0, 16, dtype=np.float32)[:, None]
    img[2, :3, :3] = np.eye(3, dtype=np.float32) * 3.0
    img[..., 3] = rng.uniform(0.0, 1.0, (H, W)).astype(np.float32)
    t, keep = _tracer_with_image(assets, img)
    for kw in (dict(), dict(exposure=2.5, brightness=1.3, contrast=0.8, saturation=1.4, vignette=0.35)):
        got = t.tonemap(method=method, **kw).astype(np.float64)
        want = _reference(img, method, **kw)
        diff = np.abs(got - np.floor(want + 0.5))
        # float32 vs float64 may land on the other side of a rounding boundary: allow one code, and only near a boundary
        assert diff.max() <= 1.0, (method, kw, diff.max())
        near = np.abs((want % 1.0) - 0.5) < 2e-2
        assert (diff[~near] == 0).all(), (method, kw, int((diff[~near] != 0).sum()))
        assert (diff != 0).mean() < 5e-3
    t.close()
    del keep


def test_inactive_passes_through_and_bad_arguments(built, assets):
    rng = np.random.default_rng(3)
    img = rng.uniform(-0.2, 1.4, (32, 48, 4)).astype(np.float32)
    t, keep = _tracer_with_image(assets, img)
    got = t.tonemap(isActive=0)
    assert (got == np.floor(np.clip(img.astype(np.float64), 0, 1) * 255.0 + 0.5)).all()
    with pytest.raises(ptmod.MiError):
        t.tonemap(method=9)
    with pytest.raises(ptmod.MiError):
        t.tonemap(brightness=0.0)
    with pytest.raises(ptmod.MiError):
        t.tonemap(source=1)  # no denoise result yet
    t.close()
    del keep


def test_auto_exposure_meters_the_geometric_mean(built, assets):
    """autoExposure (the reference's default, src/resources.hpp:212): exposure scaled by key 0.18 / geometric-mean luminance, metered
    through a 256-bin log2 histogram over [evMin, evMax]; a uniformly brighter image therefore tonemaps to the same picture, and the
    exposure eases towards its target with 1 - exp(-dt * speed)."""
    rng = np.random.default_rng(11)
    H, W = 64, 64
    base = np.exp2(rng.uniform(-6.0, 2.0, (H, W, 1))).astype(np.float32) * np.ones((1, 1, 3), np.float32)
    img = np.concatenate([base, np.ones((H, W, 1), np.float32)], -1)
    tm = capi.MiTonemapperData()
    capi.pt_lib().mi_pt_default_tonemapper(tm, 1)
    assert (tm.method, tm.isActive, tm.exposure, tm.brightness, tm.contrast, tm.saturation, tm.vignette, tm.autoExposure) == (0, 1, 1.0, 1.0, 1.0, 1.0, 0.0, 1)
    t, keep = _tracer_with_image(assets, img)
    a = t.tonemap(tm, dt_seconds=-1.0).astype(np.int32)
    # the metered exposure: bins are (evMax - evMin) / 256 = 1/8 stop wide
    lum = (img[..., :3].astype(np.float64) * np.array([0.2126, 0.7152, 0.0722])).sum(-1)
    expo = 0.18 / np.exp2(np.log2(lum).mean())
    want = _reference(img, "filmic", exposure=expo)
    assert np.abs(a - np.floor(want + 0.5))[..., :3].max() <= 6  # histogram quantisation: a 1/16-stop error at most
    t.close()
    del keep
    t2, keep2 = _tracer_with_image(assets, img * np.array([8.0, 8.0, 8.0, 1.0], np.float32))
    b = t2.tonemap(tm, dt_seconds=-1.0).astype(np.int32)
    assert np.abs(a - b).max() <= 1  # 3 stops brighter = exactly 24 bins up: the same picture
    # easing: one step of dt = ln 2 / speed moves the exposure half way to the new target
    t2.write_accum(img * np.array([2.0, 2.0, 2.0, 1.0], np.float32))
    c = t2.tonemap(tm, dt_seconds=float(np.log(2.0)) / tm.autoExposureSpeed).astype(np.int32)
    want_c = _reference(img * 2.0, "filmic", exposure=(expo / 8.0 + (expo / 2.0 - expo / 8.0) * 0.5))
    assert np.abs(c - np.floor(want_c + 0.5))[..., :3].max() <= 6
    t2.close()
    del keep2
