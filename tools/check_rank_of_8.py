"""What ONE rank of an 8-GPU `bench.py` run executes (minus the RCCL reduce): tile partition 0/8 .. 7/8, 256 frames in flight,
on a single GPU.  Checks that the per-rank step runs, how long it takes next to the 1-GPU step, and that the eight partial
accumulators sum to the unpartitioned image bit for bit."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from vk_gltf_renderer_amd import pathtracer as ptmod, _capi as capi

w = bench.WORKLOADS["helmet"]
W, H = 1920, 1080
scene = ptmod.Scene(bench.scene_path("helmet", 0))
hdr = ptmod.HdrEnvironment(path=os.path.join(bench.ROOT, "assets", "std_env.hdr"))
fi, pixel_angle, focal = ptmod.camera_frame_info(scene.camera(0), W, H)
fi.flags |= capi.MI_SCENE_USE_HDR_ENVIRONMENT
p = ptmod.default_params()
p.maxDepth, p.numSamples, p.pixelAngle, p.focalDistance = w["depth"], 1, pixel_angle, focal


def run(rank, world, frames, in_flight):
    t = ptmod.PathTracer(scene, device=0)
    t.set_environment(hdr); t.set_tile_partition(rank, world, 64); t.resize(W, H); t.set_frame_info(fi); t.set_sky(ptmod.default_sky())
    acc = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    t.bind_accum(acc.data_ptr())
    r = ptmod.HeadlessRenderer(t, p)
    r.render(frames, torch.cuda.current_stream().cuda_stream, in_flight=in_flight)  # warm-up
    torch.cuda.synchronize(); r.reset_frame()
    t0 = time.perf_counter()
    r.render(frames, torch.cuda.current_stream().cuda_stream, in_flight=in_flight)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    img = acc.cpu().numpy(); t.close()
    return img, dt

full, dt1 = run(0, 1, 32, 32)
print(f"1 GPU : 32 frames in flight, whole image      {dt1*1e3:7.2f} ms/step")
part, dt8 = run(0, 8, 256, 256)
print(f"rank 0 of 8: 256 frames in flight, 1/8 tiles   {dt8*1e3:7.2f} ms/step -> ideal 8-GPU value {W*H*256/dt8/1e6:.0f} Msamples/s")
small = [run(r, 8, 8, 8)[0] for r in range(8)]
ref = run(0, 1, 8, 8)[0]
print("sum of the 8 partial accumulators == unpartitioned image:", bool((np.sum(small, axis=0) == ref).all()))
times = [run(r, 8, 256, 256)[1] * 1e3 for r in range(8)]
print("per-rank step times (ms):", " ".join(f"{t:.2f}" for t in times), f"-> max {max(times):.2f}, mean {np.mean(times):.2f}")
for ts in (32, 16):
    times = []
    for r in range(8):
        t = ptmod.PathTracer(scene, device=0)
        t.set_environment(hdr); t.set_tile_partition(r, 8, ts); t.resize(W, H); t.set_frame_info(fi); t.set_sky(ptmod.default_sky())
        acc = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda"); t.bind_accum(acc.data_ptr())
        rr = ptmod.HeadlessRenderer(t, p)
        rr.render(256, torch.cuda.current_stream().cuda_stream, in_flight=256); torch.cuda.synchronize(); rr.reset_frame()
        t0 = time.perf_counter(); rr.render(256, torch.cuda.current_stream().cuda_stream, in_flight=256); torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) * 1e3); t.close()
    print(f"tile {ts}: per-rank (ms):", " ".join(f"{t:.2f}" for t in times), f"-> max {max(times):.2f}, mean {np.mean(times):.2f}")
