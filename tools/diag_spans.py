"""Per-launch spans WITH traversal counters for one batch of a bench workload (MI_PT_TRACE_SPANS=1 + collectCounters).
usage: MI_PT_TRACE_SPANS=1 python tools/diag_spans.py <workload> [frames]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from vk_gltf_renderer_amd import _capi as capi, pathtracer as ptmod
name = sys.argv[1]; frames = int(sys.argv[2]) if len(sys.argv) > 2 else 32
w = bench.WORKLOADS[name]
scene = ptmod.Scene(bench.scene_path(name, 0))
hdr = ptmod.HdrEnvironment(path=os.path.join(ROOT, "assets", "std_env.hdr")) if w["hdr"] else None
fi, pa, focal = ptmod.camera_frame_info(scene.camera(0), w["width"], w["height"])
if hdr is not None:
    fi.flags |= capi.MI_SCENE_USE_HDR_ENVIRONMENT
p = ptmod.default_params(); p.maxDepth, p.numSamples, p.pixelAngle, p.focalDistance = w["depth"], 1, pa, focal
t = ptmod.PathTracer(scene, collect_counters=os.environ.get("MI_PT_TRACE_SPANS") is not None)
if hdr is not None:
    t.set_environment(hdr)
t.resize(w["width"], w["height"]); t.set_frame_info(fi); t.set_sky(ptmod.default_sky())
ptmod.HeadlessRenderer(t, p).render(frames, in_flight=frames)
t.synchronize()
