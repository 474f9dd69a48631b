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


def _svgf_numpy(color, albedo, normal, depth, frames, iterations, sigma_l, sigma_n, sigma_z):
    """The definition csrc/device/denoise.hip implements (Schied et al. 2017 without reprojection), in float64."""
    H, W, _ = color.shape
    lumw = np.array([0.2126, 0.7152, 0.0722])
    c, a, n = color.astype(np.float64), albedo.astype(np.float64), normal.astype(np.float64)
    solid = a[..., 3] > 0.5
    dem = np.where(solid[..., None], np.maximum(a[..., :3], 0.02), 1.0)
    il = c[..., :3] / dem
    if frames >= 4:
        l = c[..., :3] @ lumw
        var = np.maximum(0.0, n[..., 3] - l * l) / frames / (dem @ lumw) ** 2
    else:
        li = il @ lumw
        s1, s2, sw = np.zeros((H, W)), np.zeros((H, W)), np.zeros((H, W))
        for dy in range(-3, 4):
            for dx in range(-3, 4):
                ys, xs = np.arange(H) + dy, np.arange(W) + dx
                valid = ((ys >= 0) & (ys < H))[:, None] & ((xs >= 0) & (xs < W))[None, :]
                qy, qx = np.clip(ys, 0, H - 1), np.clip(xs, 0, W - 1)
                valid &= solid[qy][:, qx] == solid
                nd = np.maximum(0.0, (n[qy][:, qx][..., :3] * n[..., :3]).sum(-1))
                w = np.where(solid, (nd > 0.9).astype(np.float64), 1.0) * valid
                lq = li[qy][:, qx]
                s1 += w * lq
                s2 += w * lq * lq
                sw += w
        m = np.where(sw > 0, s1 / np.maximum(sw, 1e-30), 0.0)
        var = np.where(sw > 0, np.maximum(0.0, s2 / np.maximum(sw, 1e-30) - m * m), 0.0)
    cur = np.concatenate([il, var[..., None]], -1)
    zk = 1.0 / np.maximum(1.0 - depth.astype(np.float64), 1e-7)
    zx = np.empty_like(zk)
    zx[:, :-1] = zk[:, 1:]
    zx[:, -1] = zk[:, -2] if W > 1 else zk[:, -1]
    zy = np.empty_like(zk)
    zy[:-1] = zk[1:]
    zy[-1] = zk[-2] if H > 1 else zk[-1]
    gz = np.maximum(np.abs(zx - zk), np.abs(zy - zk))
    kern = {0: 3 / 8, 1: 1 / 4, 2: 1 / 16}
    gauss = {0: 0.5, 1: 0.25}
    for it in range(iterations):
        step = 1 << it
        gv, gw = np.zeros((H, W)), np.zeros((H, W))
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                ys, xs = np.arange(H) + dy, np.arange(W) + dx
                valid = ((ys >= 0) & (ys < H))[:, None] & ((xs >= 0) & (xs < W))[None, :]
                qy, qx = np.clip(ys, 0, H - 1), np.clip(xs, 0, W - 1)
                w = gauss[abs(dx)] * gauss[abs(dy)] * valid
                gv += w * cur[qy][:, qx][..., 3]
                gw += w
        sdev = np.sqrt(np.maximum(gv / gw, 0.0))
        lc = cur[..., :3] @ lumw
        h0 = kern[0] * kern[0]
        acc = cur[..., :3] * h0
        accv = cur[..., 3] * h0 * h0
        sw = np.full((H, W), h0)
        for dy in range(-2, 3):
            for dx in range(-2, 3):
                if dx == 0 and dy == 0:
                    continue
                ys, xs = np.arange(H) + dy * step, np.arange(W) + dx * step
                valid = ((ys >= 0) & (ys < H))[:, None] & ((xs >= 0) & (xs < W))[None, :]
                qy, qx = np.clip(ys, 0, H - 1), np.clip(xs, 0, W - 1)
                valid &= solid[qy][:, qx] == solid
                qc = cur[qy][:, qx]
                w = np.exp(-np.abs(qc[..., :3] @ lumw - lc) / (sigma_l * sdev + 1e-6))
                nd = np.maximum(0.0, (n[qy][:, qx][..., :3] * n[..., :3]).sum(-1))
                dist = step * np.sqrt(dx * dx + dy * dy)
                wg = nd ** sigma_n * np.exp(-np.abs(zk[qy][:, qx] - zk) / (sigma_z * gz * dist + 1e-6 * zk))
                w = np.where(solid, w * wg, w) * valid
                h = kern[abs(dx)] * kern[abs(dy)] * w
                acc += qc[..., :3] * h[..., None]
                accv += qc[..., 3] * h * h
                sw += h
        cur = np.concatenate([acc / sw[..., None], (accv / (sw * sw))[..., None]], -1)
    out = np.empty((H, W, 4))
    out[..., :3] = cur[..., :3] * dem
    out[..., 3] = c[..., 3]
    return out


@pytest.mark.parametrize("frames", [2, 16])
def test_svgf_matches_numpy_and_beats_plain_atrous(built, assets, frames):
    """Variance-guided pass: numpy parity on the same inputs (temporal variance at 16 frames, the spatial fallback at 2), the second
    moment really is E[l^2] of the per-frame luminance, and on equal inputs it ends closer to the converged image than the 4-spp input."""
    hdr = os.path.join(assets, "std_env.hdr")
    s = pu.Setup(os.path.join(assets, "shader_ball.gltf"), 160, 120, max_depth=5, hdr_path=hdr,
                 params_edit=lambda p: setattr(p, "flags", p.flags | capi.MI_PT_USE_OPTIX_DENOISER))
    tracer = ptmod.PathTracer(s.scene)
    try:
        tracer.set_environment(s.hdr)
        tracer.resize(s.width, s.height)
        tracer.set_frame_info(s.frame_info)
        tracer.set_sky(s.sky)
        total, lum2 = 0, np.zeros((s.height, s.width))
        for f in range(frames):
            p = s.frame_params(f, total)
            tracer.render_frame(p)
            total += p.numSamples
            if frames <= 16:  # per-frame pixel value from consecutive running means (float64: exact enough for a 1e-3 check)
                cur = tracer.read_accum().astype(np.float64)
                frame_px = cur * (f + 1) - prev * f if f else cur
                prev = cur
                lum2 += (frame_px[..., :3] @ np.array([0.2126, 0.7152, 0.0722])) ** 2
        noisy = tracer.read_accum()
        albedo, normal = tracer.read_guides()
        depth = tracer.read_depth()
        den = tracer.denoise_svgf(iterations=4, sigma_luminance=4.0, sigma_normal=64.0, sigma_depth=1.0)
        den_dev = tracer.tonemap(source=1, method="clip")  # the denoised image is routed to the tonemapper like the OptiX output
        for f in range(frames, 300):
            p = s.frame_params(f, total)
            tracer.render_frame(p)
            total += p.numSamples
        ref = tracer.read_accum()
    finally:
        tracer.close()
    m2 = lum2 / frames
    assert np.abs(normal[..., 3] - m2).max() <= 2e-3 * max(1.0, m2.max())
    want = _svgf_numpy(noisy, albedo, normal, depth, frames, 4, 4.0, 64.0, 1.0)
    scale = np.abs(want[..., :3]).mean()
    err = np.abs(den[..., :3] - want[..., :3])
    assert np.quantile(err, 0.999) <= 2e-3 * scale and err.max() <= 5e-2 * np.abs(want[..., :3]).max(), (np.quantile(err, 0.999), err.max(), scale)
    assert np.array_equal(den[..., 3], noisy[..., 3])
    srgb = np.where(den[..., :3] > 0.0031308, 1.055 * np.maximum(den[..., :3], 1e-30) ** (1 / 2.4) - 0.055, 12.92 * den[..., :3])
    assert np.abs(den_dev[..., :3].astype(np.float64) - np.clip(srgb, 0, 1) * 255.0).max() <= 0.51
    hit = albedo[..., 3] > 0.5
    err_noisy = np.sqrt(((noisy[hit][:, :3] - ref[hit][:, :3]) ** 2).mean())
    err_den = np.sqrt(((den[hit][:, :3] - ref[hit][:, :3]) ** 2).mean())
    assert err_den < 0.8 * err_noisy, (err_den, err_noisy)
