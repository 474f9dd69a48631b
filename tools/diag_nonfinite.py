"""Finds the frame and pixel of a non-finite value in a workload's accumulator: renders the frames in batches that each start a NEW accumulation, bisects the first
batch that shows a NaN / Inf down to one frame, prints the pixels, and renders that frame with the CPU oracle for comparison.
usage: python tools/diag_nonfinite.py <workload> [frames] [first frame]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import bench
from vk_gltf_renderer_amd import pathtracer as ptmod, _capi as capi

name = sys.argv[1]
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 512
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
w = bench.WORKLOADS[name]
W, H = w["width"], w["height"]
scene = ptmod.Scene(bench.scene_path(name, 0))
if bench.ALPHA_CUT_DEFAULT > 0:
    scene.cut_alpha(bench.ALPHA_CUT_DEFAULT)
hdr = ptmod.HdrEnvironment(path=os.path.join(bench.ROOT, "assets", "std_env.hdr")) if w["hdr"] else None
fi, pixel_angle, focal = ptmod.camera_frame_info(scene.camera(0), W, H)
if hdr is not None:
    fi.flags |= capi.MI_SCENE_USE_HDR_ENVIRONMENT
t = ptmod.PathTracer(scene)
if hdr is not None:
    t.set_environment(hdr)
t.resize(W, H); t.set_frame_info(fi); t.set_sky(ptmod.default_sky())


def render(f0, n):
    p = ptmod.default_params()
    p.maxDepth, p.numSamples, p.pixelAngle, p.focalDistance = w["depth"], 1, pixel_angle, focal
    p.frameCount, p.totalSamples, p.flags = f0, 0, capi.MI_PT_FIRST_FRAME  # a new accumulation of frames f0 .. f0 + n - 1 (the FIRST flag only resets; seeds come from frameCount)
    if n == 1:
        t.render_frame(p)
    else:
        t.render_frames(p, n)
    return t.read_accum()


B = 64
bad = None
for f0 in range(first, first + frames, B):
    img = render(f0, B)
    nf = ~np.isfinite(img).all(-1)
    if nf.any():
        bad = (f0, B)
        print("non-finite pixels in frames", f0, "..", f0 + B - 1, ":", np.argwhere(nf)[:8].tolist(), flush=True)
        break
if bad is None:
    print("no non-finite value in frames", first, "..", first + frames - 1)
    sys.exit(0)
f0, n = bad
while n > 1:
    h = n // 2
    if (~np.isfinite(render(f0, h)).all(-1)).any():
        n = h
    else:
        f0, n = f0 + h, n - h
img = render(f0, 1)
nf = np.argwhere(~np.isfinite(img).all(-1))
print("frame", f0, "pixels (y, x):", nf[:8].tolist(), "values", [img[y, x].tolist() for y, x in nf[:4]])
# the same frame through the oracle (IEEE), the pixel's tile only
import parity_util as pu, oracle_lib, ctypes as C
oracle_lib.use_native()
s = pu.Setup(scene.path, W, H, hdr_path=os.path.join(bench.ROOT, "assets", "std_env.hdr") if w["hdr"] else None, max_depth=w["depth"])
O = oracle_lib.lib(); o = C.c_void_p()
O.oracle_pt_create(s.scene.desc, C.byref(o))
if s.hdr is not None:
    O.oracle_pt_set_environment(o, s.hdr.env)
O.oracle_pt_resize(o, W, H); O.oracle_pt_set_frame_info(o, C.byref(s.frame_info)); O.oracle_pt_set_sky(o, C.byref(s.sky))
y, x = nf[0]
tx = (W + 63) // 64
tile = (y // 64) * tx + x // 64
O.oracle_pt_set_tile_partition(o, int(tile), 1 << 20, 64)
p = s.frame_params(f0, 0); p.flags |= capi.MI_PT_FIRST_FRAME
O.oracle_pt_render_frame(o, C.byref(p), 8)
oi = np.ctypeslib.as_array(O.oracle_pt_accum(o), shape=(H, W, 4))
print("oracle at that pixel:", oi[y, x].tolist(), " GPU neighbours:", img[y, max(0, x - 1)].tolist(), img[y, min(W - 1, x + 1)].tolist())
