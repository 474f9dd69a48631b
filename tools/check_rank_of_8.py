"""What ONE rank of an N-GPU `bench.py` run executes (minus the RCCL reduce), timed on a single GPU: tile partition r/N, the scene
with bench.py's load-time alpha cut, bench.py's frames in flight for the scaling mode (bench.step_shape: strong = the configuration's own step, all of its frames
in flight, 1/N of their path slots per GPU; weak = in_flight x N).  Prints the per-rank step times (best of three: a single timing showed a 7 % outlier in round 5
that two more runs of the same rank do not reproduce) for a few tile sizes next to the 1-GPU step, and checks that the partial accumulators sum to the
unpartitioned image bit for bit.   usage: tools/check_rank_of_8.py [workload] [world] [strong|weak]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from vk_gltf_renderer_amd import pathtracer as ptmod, _capi as capi

name = sys.argv[1] if len(sys.argv) > 1 else "helmet"
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
w = bench.WORKLOADS[name]
W, H = w["width"], w["height"]
scene = ptmod.Scene(bench.scene_path(name, 0))
if bench.ALPHA_CUT_DEFAULT > 0:
    scene.cut_alpha(bench.ALPHA_CUT_DEFAULT)
mode = sys.argv[3] if len(sys.argv) > 3 else "strong"
F1, _ = bench.step_shape(mode, 1, w.get("in_flight", bench.IN_FLIGHT_DEFAULT), w.get("frames_per_step", bench.FRAMES_PER_STEP_DEFAULT), W, H)
FN, _ = bench.step_shape(mode, world, w.get("in_flight", bench.IN_FLIGHT_DEFAULT), w.get("frames_per_step", bench.FRAMES_PER_STEP_DEFAULT), W, H)
hdr = ptmod.HdrEnvironment(path=os.path.join(bench.ROOT, "assets", "std_env.hdr")) if w["hdr"] else None
fi, pixel_angle, focal = ptmod.camera_frame_info(scene.camera(0), W, H)
if hdr is not None:
    fi.flags |= capi.MI_SCENE_USE_HDR_ENVIRONMENT
p = ptmod.default_params()
p.maxDepth, p.numSamples, p.pixelAngle, p.focalDistance = w["depth"], 1, pixel_angle, focal


def run(rank, nranks, frames, tile=32, timed=True):
    t = ptmod.PathTracer(scene, device=0)
    if hdr is not None:
        t.set_environment(hdr)
    t.set_tile_partition(rank, nranks, tile); t.resize(W, H); t.set_frame_info(fi); t.set_sky(ptmod.default_sky())
    acc = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    t.bind_accum(acc.data_ptr())
    r = ptmod.HeadlessRenderer(t, p)
    stream = torch.cuda.current_stream().cuda_stream
    r.render(frames, stream, in_flight=frames)
    torch.cuda.synchronize()
    dt = 0.0
    if timed:
        dt = 1e30
        for _ in range(3):
            r.reset_frame()
            t0 = time.perf_counter()
            r.render(frames, stream, in_flight=frames)
            torch.cuda.synchronize()
            dt = min(dt, time.perf_counter() - t0)
    img = acc.cpu().numpy(); t.close()
    return img, dt * 1e3

one = run(0, 1, F1)[1] / F1
print(f"{name} {W}x{H}, --scaling {mode}: 1 GPU, {F1} frames in flight, whole image: {one:.3f} ms/frame = {W * H / one / 1e3:.1f} Msamples/s; one rank of {world}: {FN} frames in flight "
      f"over 1/{world} of the pixels = {FN / world:.0f} frames' worth of path slots")
ref = run(0, 1, 4, timed=False)[0]
print(f"sum of the {world} partial accumulators == unpartitioned image:", bool((np.sum([run(r, world, 4, timed=False)[0] for r in range(world)], axis=0) == ref).all()))
for tile in (64, 32, 16):
    times = [run(r, world, FN, tile)[1] * world / FN for r in range(world)]  # ms per frame-equivalent of work (a rank renders 1/world of each frame)
    print(f"tile {tile}: rank ms per 1-GPU-frame of work:", " ".join(f"{t:.3f}" for t in times), f"-> max {max(times):.3f}: {world} GPUs before the reduce = "
          f"{W * H / max(times) / 1e3 * world:.0f} Msamples/s = {one / max(times) * world:.2f} x the 1-GPU rate (efficiency {one / max(times):.3f})")
