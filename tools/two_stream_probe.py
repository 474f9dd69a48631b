#!/usr/bin/env python3
"""Would interleaving two half-size batches on two streams beat one full-size batch?  (A probe, not a product path: two MiPt instances on one GPU,
each with F/2 frames in flight on its own stream, against one instance with F in flight -- the same path-state memory.  Kernels of the two streams
can fill each other's drain phases and the nearly empty late-bounce launches; against that, every launch is half as long.)

usage: python tools/two_stream_probe.py [workload=atrium] [F=64] [frames=256] [steps=3]
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from vk_gltf_renderer_amd import _capi as capi, pathtracer as ptmod

def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "atrium"
    F = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    frames = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    w = bench.WORKLOADS[workload]
    W, H = w["width"], w["height"]
    scene = ptmod.Scene(bench.scene_path(workload, 0))
    scene.cut_alpha(4)
    cam = scene.camera(0)
    fi, pixel_angle, focal = ptmod.camera_frame_info(cam, W, H)
    hdr = ptmod.HdrEnvironment(path=os.path.join(ROOT, "assets", "std_env.hdr")) if w["hdr"] else None
    if hdr is not None:
        fi.flags |= capi.MI_SCENE_USE_HDR_ENVIRONMENT
    params = ptmod.default_params()
    params.maxDepth, params.numSamples, params.pixelAngle, params.focalDistance = w["depth"], 1, pixel_angle, focal

    def make():
        t = ptmod.PathTracer(scene, device=0)
        if hdr is not None:
            t.set_environment(hdr)
        t.resize(W, H); t.set_frame_info(fi); t.set_sky(ptmod.default_sky())
        return t

    def run(n_inst, in_flight):
        tracers = [make() for _ in range(n_inst)]
        streams = [torch.cuda.Stream() for _ in range(n_inst)]
        runners = [ptmod.HeadlessRenderer(t, params) for t in tracers]
        per = frames // n_inst
        def step():
            # batches alternate between the instances so that both streams always have work queued
            for b in range(0, per, in_flight):
                for r, s in zip(runners, streams):
                    r.render(min(in_flight, per - b), s.cuda_stream, in_flight=in_flight)
        step(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        for t in tracers:
            t.close()
        return W * H * frames * steps / dt / 1e6
    for n_inst, f in ((1, F), (2, F // 2), (1, F), (2, F // 2), (2, F)):
        v = run(n_inst, f)
        print(f"PROBE {workload}: {n_inst} instance(s) x {f} frames in flight: {v:.1f} Msamples/s", flush=True)

if __name__ == "__main__":
    main()
