"""Thin Python handles over the C-ABI (include/mi_pt.h, include/mi_host.h) for tests, bench.py and tooling.

Mirrors the reference's host objects for the path-trace mode only:
  Scene            ~ nvvkgltf::Scene::load + SceneVk tables          (src/gltf_scene.cpp:298, src/gltf_scene_vk.cpp:218)
  HdrEnvironment   ~ nvvk::HdrIbl::loadEnvironment                   (src/renderer.cpp:1982-2017)
  PathTracer       ~ class PathTracer : BaseRenderer                 (src/renderer_pathtracer.cpp:500-614)
  HeadlessRenderer ~ the GltfRenderer::onRender slice that drives it (src/renderer.cpp:588-742, :1959-1977)
Everything numerical happens inside libmi_pt.so (HIP); there is no Python or CPU fallback for rendering.
"""
import ctypes as C

import numpy as np

from . import _capi as capi


class MiError(RuntimeError):
    pass


def _check_host(rc):
    if rc != 0:
        raise MiError(f"libmi_host: rc={rc}: {capi.host_lib().mi_host_last_error().decode()}")


def _check_pt(rc):
    if rc != 0:
        raise MiError(f"libmi_pt: rc={rc}: {capi.pt_lib().mi_pt_last_error().decode()}")


class Scene:
    def __init__(self, path):
        self._h = capi.host_lib()
        self._p = C.c_void_p()
        _check_host(self._h.mi_scene_load(str(path).encode(), C.byref(self._p)))
        self.path = str(path)

    @property
    def desc(self):
        return self._h.mi_scene_desc(self._p)

    @property
    def num_triangles(self):
        return int(self._h.mi_scene_num_triangles(self._p))

    @property
    def num_cameras(self):
        return int(self._h.mi_scene_num_cameras(self._p))

    def camera(self, index=0):
        cam = capi.MiCamera()
        _check_host(self._h.mi_scene_camera(self._p, index, C.byref(cam)))
        return cam

    def recompute_tangents(self, force_creation=True, mikktspace=True):
        """recomputeTangents of the reference (src/gltf_create_tangent.hpp:40); returns the vertices added by the MikkTSpace splitting.
        The scene's desc changes: create PathTracers after this call."""
        n = self._h.mi_scene_recompute_tangents(self._p, int(force_creation), int(mikktspace))
        if n < 0:
            _check_host(n)
        return n

    def cut_alpha(self, subdivisions=8):
        """Load-time bake for alpha-MASK geometry (mi_scene_cut_alpha): drops the parts of alpha-tested triangles on which the test
        cannot pass.  Returns the number of (sub-)triangles dropped.  The scene's desc changes: create PathTracers after this call."""
        n = self._h.mi_scene_cut_alpha(self._p, int(subdivisions))
        if n < 0:
            _check_host(int(n))
        return int(n)

    @property
    def num_animations(self):
        return int(self._h.mi_scene_num_animations(self._p))

    def animation_info(self, index=0):
        """(name, start, end) of a clip (AnimationInfo of the reference, src/gltf_scene.hpp:159-189)."""
        a, b, name = C.c_float(), C.c_float(), C.create_string_buffer(256)
        _check_host(self._h.mi_scene_animation_info(self._p, index, C.byref(a), C.byref(b), name, 256))
        return name.value.decode(), a.value, b.value

    def update_animation(self, index, time):
        """Poses the scene at `time`: the render-node matrices and light placements of `desc` change in place.  True when something
        moved; follow with PathTracer.update_from_scene(scene)."""
        r = self._h.mi_scene_update_animation(self._p, index, float(time))
        if r < 0:
            _check_host(r)
        return bool(r)

    def bounds(self):
        lo, hi = (C.c_float * 3)(), (C.c_float * 3)()
        self._h.mi_scene_bounds(self._p, lo, hi)
        return np.array(lo[:]), np.array(hi[:])

    def close(self):
        if self._p:
            self._h.mi_scene_destroy(self._p)
            self._p = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HdrEnvironment:
    def __init__(self, path=None, pixels=None):
        self._h = capi.host_lib()
        self._p = C.c_void_p()
        if path is not None:
            _check_host(self._h.mi_hdr_load(str(path).encode(), C.byref(self._p)))
        else:
            px = np.ascontiguousarray(pixels, dtype=np.float32)
            assert px.ndim == 3 and px.shape[2] == 3
            _check_host(self._h.mi_hdr_from_pixels(px.shape[1], px.shape[0], px.ctypes.data_as(C.POINTER(C.c_float)), C.byref(self._p)))

    @property
    def env(self):
        return self._h.mi_hdr_env(self._p)

    def close(self):
        if self._p:
            self._h.mi_hdr_destroy(self._p)
            self._p = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def default_sky():
    sky = capi.MiSkyPhysicalParameters()
    capi.host_lib().mi_default_sky(C.byref(sky))
    return sky


def default_params():
    p = capi.MiPathtraceParams()
    capi.host_lib().mi_default_params(C.byref(p))
    return p


def camera_frame_info(cam, width, height):
    """Returns (MiSceneFrameInfo, pixelAngle, focalDistance) — reference: src/renderer.cpp:675-705."""
    fi, pa, fd = capi.MiSceneFrameInfo(), C.c_float(), C.c_float()
    capi.host_lib().mi_camera_frame_info(C.byref(cam), width, height, C.byref(fi), C.byref(pa), C.byref(fd))
    return fi, pa.value, fd.value


class PathTracer:
    """The HIP path tracer instance (libmi_pt.so)."""

    def __init__(self, scene, device=0, collect_counters=False, bvh=0):
        self._l = capi.pt_lib()
        self._p = C.c_void_p()
        self._scene = scene  # keep host tables alive for the duration of mi_pt_create only (they are copied)
        opts = capi.MiPtCreateOptions()
        opts.device = device
        opts.collectCounters = 1 if collect_counters else 0
        opts.bvhBuilder = bvh  # 0: 8-wide compressed BVH (default), 1: plain BVH2
        _check_pt(self._l.mi_pt_create(scene.desc, C.byref(opts), C.byref(self._p)))
        self.width = self.height = 0

    def update_render_nodes(self, render_nodes, count, visible=None):
        """New transforms / materials / visibility for the instances: rebuilds the acceleration structure on the device."""
        _check_pt(self._l.mi_pt_update_render_nodes(self._p, render_nodes, count, visible))

    def update_lights(self, lights, count):
        """New placement / colour / cone of the lights (same count as at creation)."""
        _check_pt(self._l.mi_pt_update_lights(self._p, lights, count))

    def update_from_scene(self, scene):
        """After Scene.update_animation: hands the scene's render-node and light tables to the device again."""
        d = scene.desc.contents
        self.update_render_nodes(d.renderNodes, d.numRenderNodes, d.renderNodeVisible)
        self.update_lights(d.lights, d.numLights)

    def set_environment(self, hdr):
        _check_pt(self._l.mi_pt_set_environment(self._p, hdr.env if hdr is not None else None))

    def resize(self, width, height):
        _check_pt(self._l.mi_pt_resize(self._p, width, height))
        self.width, self.height = width, height

    def set_frame_info(self, fi):
        _check_pt(self._l.mi_pt_set_frame_info(self._p, C.byref(fi)))

    def set_sky(self, sky):
        _check_pt(self._l.mi_pt_set_sky(self._p, C.byref(sky)))

    def set_tile_partition(self, rank, world, tile_size=64):
        _check_pt(self._l.mi_pt_set_tile_partition(self._p, rank, world, tile_size))

    def bind_accum(self, device_ptr):
        _check_pt(self._l.mi_pt_bind_accum(self._p, C.c_void_p(device_ptr)))

    def bind_guides(self, albedo_ptr=0, normal_ptr=0, depth_ptr=0):
        """Denoiser guide / depth images in caller-owned device memory (0 = the internal image)."""
        _check_pt(self._l.mi_pt_bind_guides(self._p, C.c_void_p(albedo_ptr or None), C.c_void_p(normal_ptr or None), C.c_void_p(depth_ptr or None)))

    def set_frame_queue(self, depth):
        """render_frame calls are held back and issued `depth` at a time as one batch (mi_pt_set_frame_queue); 1 = at once."""
        _check_pt(self._l.mi_pt_set_frame_queue(self._p, int(depth)))

    def render_frame(self, params, stream=None):
        _check_pt(self._l.mi_pt_render_frame(self._p, C.byref(params), C.c_void_p(stream or 0)))

    def render_frames(self, params, num_frames, stream=None):
        """num_frames frames in flight; bit-identical to num_frames successive render_frame calls (include/mi_pt.h)."""
        _check_pt(self._l.mi_pt_render_frames(self._p, C.byref(params), num_frames, C.c_void_p(stream or 0)))

    def synchronize(self):
        _check_pt(self._l.mi_pt_synchronize(self._p))

    def read_accum(self):
        out = np.empty((self.height, self.width, 4), dtype=np.float32)
        _check_pt(self._l.mi_pt_read_accum(self._p, out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def write_accum(self, img):
        img = np.ascontiguousarray(img, dtype=np.float32)
        assert img.shape == (self.height, self.width, 4)
        _check_pt(self._l.mi_pt_write_accum(self._p, img.ctypes.data_as(C.POINTER(C.c_float))))

    def read_guides(self):
        """(albedo RGBA, normal RGBA) guide layers of the denoiser (valid when MI_PT_USE_OPTIX_DENOISER was set)."""
        a = np.empty((self.height, self.width, 4), dtype=np.float32)
        n = np.empty((self.height, self.width, 4), dtype=np.float32)
        _check_pt(self._l.mi_pt_read_guides(self._p, a.ctypes.data_as(C.POINTER(C.c_float)), n.ctypes.data_as(C.POINTER(C.c_float))))
        return a, n

    def read_selection(self):
        out = np.empty((self.height, self.width), dtype=np.uint32)
        _check_pt(self._l.mi_pt_read_selection(self._p, out.ctypes.data_as(C.POINTER(C.c_uint32))))
        return out

    def read_depth(self):
        out = np.empty((self.height, self.width), dtype=np.float32)
        _check_pt(self._l.mi_pt_read_depth(self._p, out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def denoise(self, iterations=5, sigma_color=0.6, sigma_normal=64.0, sigma_albedo=0.2):
        out = np.empty((self.height, self.width, 4), dtype=np.float32)
        _check_pt(self._l.mi_pt_denoise(self._p, iterations, sigma_color, sigma_normal, sigma_albedo,
                                        out.ctypes.data_as(C.POINTER(C.c_float)), None))
        return out

    def denoise_svgf(self, iterations=5, sigma_luminance=4.0, sigma_normal=128.0, sigma_depth=1.0, read=True, stream=None):
        """Variance-guided denoise (mi_pt_denoise_svgf); read=False leaves the result on the device (tonemap(source=1) picks it up)."""
        out = np.empty((self.height, self.width, 4), dtype=np.float32) if read else None
        _check_pt(self._l.mi_pt_denoise_svgf(self._p, iterations, sigma_luminance, sigma_normal, sigma_depth,
                                             out.ctypes.data_as(C.POINTER(C.c_float)) if read else None, C.c_void_p(stream or 0)))
        return out

    def tonemap(self, tm=None, source=0, dt_seconds=-1.0, **fields):
        """HDR -> display RGBA8 (H, W, 4 uint8) on the device; tm = MiTonemapperData (default: the reference's defaults with auto exposure
        off), fields override members (method may be a name from capi.TONEMAP_METHODS); source 1 = the last denoise() result."""
        if tm is None:
            tm = capi.MiTonemapperData()
            self._l.mi_pt_default_tonemapper(C.byref(tm), 0)
        for k, v in fields.items():
            setattr(tm, k, capi.TONEMAP_METHODS.index(v) if (k == "method" and isinstance(v, str)) else v)
        out = np.empty((self.height, self.width, 4), dtype=np.uint8)
        _check_pt(self._l.mi_pt_tonemap(self._p, C.byref(tm), source, dt_seconds, out.ctypes.data_as(C.POINTER(C.c_uint8)), None))
        return out

    def stats(self):
        st = capi.MiPtStats()
        _check_pt(self._l.mi_pt_get_stats(self._p, C.byref(st)))
        return {n: int(getattr(st, n)) for n, _ in st._fields_}

    def memory(self):
        """mi_pt_get_memory: bytes of the scene (geometry, textures, acceleration structure), of the renderer (path state, queues, images) and of the device."""
        m = capi.MiPtMemory()
        _check_pt(self._l.mi_pt_get_memory(self._p, C.byref(m)))
        return {n: int(getattr(m, n)) for n, _ in m._fields_}

    def reset_stats(self):
        _check_pt(self._l.mi_pt_reset_stats(self._p))

    def enable_timing(self, on=True):
        _check_pt(self._l.mi_pt_enable_timing(self._p, 1 if on else 0))

    def frame_timing(self):
        t = capi.MiPtFrameTiming()
        _check_pt(self._l.mi_pt_get_frame_timing(self._p, C.byref(t)))
        return {n: getattr(t, n) for n, _ in t._fields_}

    def close(self):
        if self._p:
            self._l.mi_pt_destroy(self._p)
            self._p = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HeadlessRenderer:
    """Frame loop of the reference's headless path-trace mode: frameCount starts at -1 and is pre-incremented
    (src/renderer.cpp:1939-1977), frame 0 carries ePtFirstFrame and resets the sample counter
    (src/renderer_pathtracer.cpp:1502-1505, :1546-1549), totalSamples grows by numSamples per frame (:1401)."""

    def __init__(self, tracer, params):
        self.tracer = tracer
        self.params = params
        self.frame_count = -1
        self.total_samples = 0

    def reset_frame(self):
        self.frame_count = -1

    def render(self, frames=1, stream=None, in_flight=1):
        """Advance `frames` frames; in_flight > 1 issues them in batches that share the wavefront launches."""
        done = 0
        while done < frames:
            batch = min(max(1, in_flight), frames - done)
            self.frame_count += 1
            if self.frame_count == 0:
                self.total_samples = 0
            p = self.params
            p.frameCount = self.frame_count
            p.totalSamples = self.total_samples
            p.flags = (p.flags & ~capi.MI_PT_FIRST_FRAME) | (capi.MI_PT_FIRST_FRAME if self.frame_count == 0 else 0)
            if batch == 1:
                self.tracer.render_frame(p, stream)
            else:
                self.tracer.render_frames(p, batch, stream)
            self.frame_count += batch - 1
            self.total_samples += p.numSamples * batch
            done += batch
