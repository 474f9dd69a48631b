"""Frame-by-frame throughput with and without the hipGraph replay of small batches (MI_PT_GRAPH), per-launch timing off.
usage: python tools/graph_ab.py [workload] [frames]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from vk_gltf_renderer_amd import _capi as capi, pathtracer as ptmod
name = sys.argv[1] if len(sys.argv) > 1 else "helmet"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 256
w = bench.WORKLOADS[name]
W, H = w["width"], w["height"]
scene = ptmod.Scene(bench.scene_path(name, 0))
scene.cut_alpha(bench.ALPHA_CUT_DEFAULT)
hdr = ptmod.HdrEnvironment(path=os.path.join(ROOT, "assets", "std_env.hdr")) if w["hdr"] else None
fi, pa, focal = ptmod.camera_frame_info(scene.camera(0), W, H)
if hdr is not None:
    fi.flags |= capi.MI_SCENE_USE_HDR_ENVIRONMENT
p = ptmod.default_params(); p.maxDepth, p.numSamples, p.pixelAngle, p.focalDistance = w["depth"], 1, pa, focal
for F in (1, 2, 4):
    for graph in ("8", "0"):
        os.environ["MI_PT_GRAPH"] = graph
        t = ptmod.PathTracer(scene)
        if hdr is not None:
            t.set_environment(hdr)
        t.resize(W, H); t.set_frame_info(fi); t.set_sky(ptmod.default_sky())
        r = ptmod.HeadlessRenderer(t, p)
        r.render(8 * F, in_flight=F); t.synchronize(); r.reset_frame()
        t0 = time.perf_counter()
        r.render(frames, in_flight=F); t.synchronize()
        dt = time.perf_counter() - t0
        img = t.read_accum()
        print(f"RESULT {name} in-flight {F} graph {graph}: {W * H * frames / dt / 1e6:8.1f} Msamples/s, {dt / frames * 1e3:.3f} ms/frame, checksum {float(img.sum()):.6f}", flush=True)
        t.close()
