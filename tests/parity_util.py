"""Shared helpers of the parity tests: render the same frames with the CPU oracle and with the HIP path tracer."""
import ctypes as C
import os

import numpy as np

import oracle_lib
from vk_gltf_renderer_amd import _capi as capi
from vk_gltf_renderer_amd import pathtracer as ptmod


class Setup:
    """Scene + camera + frame constants shared by both renderers."""

    def __init__(self, scene_path, width, height, hdr_path=None, hdr_pixels=None, max_depth=5, spp_per_frame=1, camera=None,
                 flags=0, frame_info_edit=None, params_edit=None, sky_edit=None, alpha_cut=0):
        self.scene = ptmod.Scene(scene_path)
        self.alpha_cut_dropped = self.scene.cut_alpha(alpha_cut) if alpha_cut > 0 else 0  # (load-time bake, mi_scene_cut_alpha)
        self.width, self.height = width, height
        cam = camera if camera is not None else self.scene.camera(0)
        self.frame_info, pixel_angle, focal = ptmod.camera_frame_info(cam, width, height)
        self.hdr = None
        if hdr_path is not None or hdr_pixels is not None:
            self.hdr = ptmod.HdrEnvironment(path=hdr_path, pixels=hdr_pixels)
            self.frame_info.flags |= capi.MI_SCENE_USE_HDR_ENVIRONMENT
        self.frame_info.flags |= flags
        if frame_info_edit:
            frame_info_edit(self.frame_info)
        self.sky = ptmod.default_sky()
        if sky_edit:
            sky_edit(self.sky)
        self.params = ptmod.default_params()
        self.params.maxDepth = max_depth
        self.params.numSamples = spp_per_frame
        self.params.pixelAngle = pixel_angle
        self.params.focalDistance = focal
        if params_edit:
            params_edit(self.params)

    def frame_params(self, frame, total):
        p = capi.MiPathtraceParams()
        C.memmove(C.byref(p), C.byref(self.params), C.sizeof(p))
        p.frameCount = frame
        p.totalSamples = total
        p.flags = (self.params.flags & ~capi.MI_PT_FIRST_FRAME) | (capi.MI_PT_FIRST_FRAME if frame == 0 else 0)
        return p


def render_oracle(setup, frames, threads=None, tile=None):
    O = oracle_lib.lib()
    o = C.c_void_p()
    assert O.oracle_pt_create(setup.scene.desc, C.byref(o)) == 0
    try:
        if setup.hdr is not None:
            O.oracle_pt_set_environment(o, setup.hdr.env)
        O.oracle_pt_resize(o, setup.width, setup.height)
        O.oracle_pt_set_frame_info(o, C.byref(setup.frame_info))
        O.oracle_pt_set_sky(o, C.byref(setup.sky))
        if tile is not None:
            O.oracle_pt_set_tile_partition(o, *tile)
        total = 0
        nthreads = threads or os.cpu_count() or 1
        for f in range(frames):
            p = setup.frame_params(f, total)
            assert O.oracle_pt_render_frame(o, C.byref(p), nthreads) == 0
            total += p.numSamples
        shape = (setup.height, setup.width)
        out = {
            "accum": np.ctypeslib.as_array(O.oracle_pt_accum(o), shape=shape + (4,)).copy(),
            "depth": np.ctypeslib.as_array(O.oracle_pt_depth(o), shape=shape).copy(),
            "selection": np.ctypeslib.as_array(O.oracle_pt_selection(o), shape=shape).copy(),
            "albedo": np.ctypeslib.as_array(O.oracle_pt_albedo(o), shape=shape + (4,)).copy(),
            "normal": np.ctypeslib.as_array(O.oracle_pt_normal(o), shape=shape + (4,)).copy(),
        }
        st = capi.MiPtStats()
        O.oracle_pt_get_stats(o, C.byref(st))
        out["stats"] = {n: int(getattr(st, n)) for n, _ in st._fields_}
        return out
    finally:
        O.oracle_pt_destroy(o)


def render_gpu(setup, frames, collect_counters=True, tile=None, bvh=0, in_flight=1):
    tracer = ptmod.PathTracer(setup.scene, collect_counters=collect_counters, bvh=bvh)
    try:
        if setup.hdr is not None:
            tracer.set_environment(setup.hdr)
        if tile is not None:
            tracer.set_tile_partition(*tile)
        tracer.resize(setup.width, setup.height)
        tracer.set_frame_info(setup.frame_info)
        tracer.set_sky(setup.sky)
        total, f = 0, 0
        while f < frames:
            batch = min(in_flight, frames - f)
            p = setup.frame_params(f, total)
            if batch == 1:
                tracer.render_frame(p)
            else:
                tracer.render_frames(p, batch)
            total += p.numSamples * batch
            f += batch
        out = {"accum": tracer.read_accum(), "depth": tracer.read_depth(), "selection": tracer.read_selection(), "stats": tracer.stats()}
        return out
    finally:
        tracer.close()


def compare_images(a, b, rel_floor=1e-3):
    """Returns dict of parity metrics between two RGBA float images (a = oracle, b = device)."""
    a3, b3 = a[..., :3].astype(np.float64), b[..., :3].astype(np.float64)
    diff = b3 - a3
    rel_l2 = float(np.sqrt((diff ** 2).sum()) / max(np.sqrt((a3 ** 2).sum()), 1e-30))
    scale = np.maximum(np.abs(a3), np.abs(a3).mean() * rel_floor + 1e-12)
    per_px = np.abs(diff / scale).max(axis=-1)
    return {
        "rel_l2": rel_l2,
        "max_abs": float(np.abs(diff).max()),
        "frac_exact": float((a[..., :3] == b[..., :3]).all(axis=-1).mean()),
        "frac_within_1e-4": float((per_px <= 1e-4).mean()),
        "frac_within_1e-2": float((per_px <= 1e-2).mean()),
        "alpha_max_abs": float(np.abs(a[..., 3].astype(np.float64) - b[..., 3]).max()),
        "mean_rel_bias": float(diff.sum() / max(a3.sum(), 1e-30)),
    }
