"""The a-trous denoiser (SURVEY §8f.1, I/O contract of the reference's OptiX adapter: src/optix_denoiser.hpp:128-153) against
a numpy restatement of the same filter on the same inputs, plus the properties a denoiser must have."""
import os

import numpy as np
import pytest

import parity_util as pu
from vk_gltf_renderer_amd import _capi as capi
from vk_gltf_renderer_amd import pathtracer as ptmod

pytestmark = pytest.mark.gpu


def _atrous_numpy(img, albedo, normal, iterations, sigma_color, sigma_normal, sigma_albedo):
    """Dammertz et al. 2010 edge-avoiding a-trous, B3-spline taps, weights = colour x albedo x normal^sigma, geometry is
    only filtered with geometry (albedo.w = first-hit fraction) — the definition csrc/device/denoise.hip implements."""
    H, W, _ = img.shape
    kern = np.array([1 / 16, 1 / 4, 3 / 8, 1 / 4, 1 / 16], np.float64)
    cur = img.astype(np.float64)
    a, n = albedo.astype(np.float64), normal.astype(np.float64)
    solid = a[..., 3] > 0.5
    inv_c, inv_a = 1.0 / max(sigma_color ** 2, 1e-8), 1.0 / max(sigma_albedo ** 2, 1e-8)
    for it in range(iterations):
        step = 1 << it
        acc = np.zeros((H, W, 3))
        wsum = np.zeros((H, W))
        for dy in range(-2, 3):
            for dx in range(-2, 3):
                oy, ox = dy * step, dx * step
                ys, xs = np.arange(H) + oy, np.arange(W) + ox
                vy, vx = (ys >= 0) & (ys < H), (xs >= 0) & (xs < W)
                valid = vy[:, None] & vx[None, :]
                qy, qx = np.clip(ys, 0, H - 1), np.clip(xs, 0, W - 1)
                qc, qa, qn, qs = cur[qy][:, qx], a[qy][:, qx], n[qy][:, qx], solid[qy][:, qx]
                valid &= qs == solid
                w_c = np.exp(-((qc[..., :3] - cur[..., :3]) ** 2).sum(-1) * inv_c)
                w_a = np.exp(-((qa[..., :3] - a[..., :3]) ** 2).sum(-1) * inv_a)
                nd = np.maximum(0.0, (qn[..., :3] * n[..., :3]).sum(-1))
                w_n = np.where(solid, nd ** sigma_normal, 1.0)
                w = kern[dy + 2] * kern[dx + 2] * w_c * w_a * w_n * valid
                acc += qc[..., :3] * w[..., None]
                wsum += w
        out = cur.copy()
        ok = wsum > 0
        out[ok, :3] = acc[ok] / wsum[ok, None]
        cur = out
    return cur


def test_atrous_matches_numpy_and_reduces_noise(built, assets):
    hdr = os.path.join(assets, "std_env.hdr")
    s = pu.Setup(os.path.join(assets, "shader_ball.gltf"), 160, 120, max_depth=5, hdr_path=hdr,
                 params_edit=lambda p: setattr(p, "flags", p.flags | capi.MI_PT_USE_OPTIX_DENOISER))
    tracer = ptmod.PathTracer(s.scene)
    try:
        tracer.set_environment(s.hdr)
        tracer.resize(s.width, s.height)
        tracer.set_frame_info(s.frame_info)
        tracer.set_sky(s.sky)
        total = 0
        for f in range(4):
            p = s.frame_params(f, total)
            tracer.render_frame(p)
            total += p.numSamples
        noisy = tracer.read_accum()
        albedo, normal = tracer.read_guides()
        den = tracer.denoise(iterations=4, sigma_color=3.0, sigma_normal=64.0, sigma_albedo=0.2)
        # a converged reference of the same view
        for f in range(4, 260):
            p = s.frame_params(f, total)
            tracer.render_frame(p)
            total += p.numSamples
        ref = tracer.read_accum()
    finally:
        tracer.close()
    # guides: albedo.w is the first-hit fraction; normals are means of unit vectors over the samples that hit
    hit = albedo[..., 3] > 0.5
    assert 0.05 < hit.mean() < 0.95
    nlen = np.linalg.norm(normal[hit][:, :3], axis=1)
    assert nlen.max() <= 1.0 + 1e-3 and nlen.mean() > 0.85
    want = _atrous_numpy(noisy, albedo, normal, 4, 3.0, 64.0, 0.2)
    # the device uses the fast exp/pow intrinsics: compare with a tolerance relative to the image scale
    scale = np.abs(want[..., :3]).mean()
    assert np.abs(den[..., :3] - want[..., :3]).max() <= 2e-2 * scale + 1e-3 * np.abs(want[..., :3]).max()
    assert np.array_equal(den[..., 3], noisy[..., 3])
    # and it denoises: closer to the converged image than the 4-spp input on the geometry
    err_noisy = np.sqrt(((noisy[hit][:, :3] - ref[hit][:, :3]) ** 2).mean())
    err_den = np.sqrt(((den[hit][:, :3] - ref[hit][:, :3]) ** 2).mean())
    assert err_den < 0.85 * err_noisy, (err_den, err_noisy)
