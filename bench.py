#!/usr/bin/env python3
"""bench.py — throughput of the path-trace hot path on N MI355X GPUs of one node.

A "step" is one pass of the hot path over one batch of synthetic input: a batch of `--in-flight` (default 32) consecutive
frames per GPU, each ptSamples = 1 sample per pixel like the reference's headless run `--frames K --ptSamples 1`
(docs/benchmarking.md:16-23), issued through mi_pt_render_frames so that the frames share every wavefront launch
(bit-identical to rendering them one after the other; --in-flight 1 gives exactly that), of the workload BASELINE.json's
metric is quoted on and that fits one GPU: configs[1], DamagedHelmet-class + std_env.hdr, 1920x1080, depth 8 (the asset
itself is not available offline; vk_gltf_renderer_amd.scenegen writes a seeded stand-in of the same class as a .glb).
Metric = the reference's throughput_MSps (src/benchmarking.cpp:272-279): W*H*spp / wall_s / 1e6 with spp = all samples
of the timed region, inputs resident in HBM before the timed region.

N > 1: the image is split into interleaved 32x32 tiles (--tile; tile % N == rank), every rank renders its tiles with no data-path
collective, and ONE RCCL reduce(sum) of the RGBA32F accumulator over xGMI closes the frame set (inside the timed
region).  The batch is in_flight * N frames, i.e. the work per GPU is fixed as N grows -> "weak" scaling.

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

WORKLOADS = {
    # name: (BASELINE config, generator kwargs, width, height, maxDepth, env)
    "helmet": dict(config="configs[1]: DamagedHelmet-class + std_env.hdr, 1920x1080, depth 8", gen="scene_helmet_class",
                   kw=dict(seed=1234, tess=272, tex_size=2048), width=1920, height=1080, depth=8, hdr=True),
    "atrium": dict(config="configs[2]: Sponza-class, 1920x1080, depth 12, NEE+MIS (directional light + sky)", gen="scene_atrium_class",
                   kw=dict(seed=4321, detail=0.8, tex_size=512), width=1920, height=1080, depth=12, hdr=False),
    "street": dict(config="configs[3]: BistroExterior-class (instanced street, ~2.8 M triangles, ~1000 render nodes, 130 materials), 3840x2160, depth 8",
                   gen="scene_street_class", kw=dict(seed=777, detail=1.27, tex_size=256), width=3840, height=2160, depth=8, hdr=False),
    "glass": dict(config="configs[4]: TransmissionTest-class, 1920x1080, depth 24", gen="scene_glass_class",
                  kw=dict(seed=99, tess=96), width=1920, height=1080, depth=24, hdr=True),
    "box": dict(config="configs[0]: resources/Box.glb, 256x256, depth 4", gen=None, kw={}, width=256, height=256, depth=4, hdr=True),
}
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def scene_path(name, rank):
    w = WORKLOADS[name]
    if w["gen"] is None:
        return os.path.join(ROOT, "assets", "Box.glb")
    from vk_gltf_renderer_amd import scenegen
    d = os.path.join(tempfile.gettempdir(), "mi_pt_scenes")
    os.makedirs(d, exist_ok=True)
    tag = "_".join(f"{k}{v}" for k, v in sorted(w["kw"].items()))
    path = os.path.join(d, f"{name}_{tag}.glb")
    if not os.path.exists(path):
        tmp = f"{path}.{os.getpid()}.{rank}.tmp"
        getattr(scenegen, w["gen"])(tmp, **w["kw"])
        os.replace(tmp, path)
    return path


def algorithmic_bytes(stats, kernel):
    """SURVEY §8(d) per-launch algorithmic bytes from exported counters (no cache effects, by definition)."""
    s_node, s_tri = stats["bvhNodeBytes"], stats["bvhTriangleBytes"]
    if kernel == "trace_closest":
        # per ray: ray read (origin+tmax, dir+cone: 32 B) + hit write (16 B) + nodes * S_node + triangles * S_tri
        return stats["segments"] * (32 + 16) + stats["nodesClosest"] * s_node + stats["trisClosest"] * s_tri
    if kernel == "trace_shadow":
        return stats["shadowRays"] * (3 * 16 + 2 * 16) + stats["nodesShadow"] * s_node + stats["trisShadow"] * s_tri
    if kernel == "shade":
        # hit + ray + throughput + radiance + misc read (5 x 16) and write-back of ray/throughput/radiance/misc (5 x 16),
        # hit attribute gather 192 B, instance+primitive+material records 136+56+288, shadow record 48, 48 B per texture tap
        return stats["segments"] * (80 + 80 + 192 + 480 + 48) + stats["textureTaps"] * 48
    raise KeyError(kernel)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="helmet", choices=sorted(WORKLOADS))
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--tile", type=int, default=32,
                    help="edge of the interleaved tiles the image is dealt out in (tile %% world == rank); 32 balances the 8 ranks of the "
                         "helmet workload to 4 %% (64: 19 %%, tools/check_rank_of_8.py)")
    ap.add_argument("--bvh", type=int, default=0, help="bit0: 0 = 8-wide compressed BVH (default), 1 = plain BVH2; bit1: 0 = PLOC topology (default), 1 = LBVH")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--in-flight", type=int, default=32,
                    help="frames in flight per GPU and step (mi_pt_render_frames, bit-identical to sequential frames); a step renders in_flight * n_gpus frames")
    args = ap.parse_args()

    import torch
    from vk_gltf_renderer_amd import _capi as capi
    from vk_gltf_renderer_amd import pathtracer as ptmod

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    # (BENCH_SHARE_GPU=1 + BENCH_DIST_BACKEND=gloo: all ranks on one device over gloo -- a functional check of the N > 1 code on
    #  a single-GPU box, not a measurement)
    if os.environ.get("BENCH_SHARE_GPU") == "1":
        local_rank = 0
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    w = WORKLOADS[args.workload]
    W, H = args.width or w["width"], args.height or w["height"]
    scene = ptmod.Scene(scene_path(args.workload, rank))
    hdr = ptmod.HdrEnvironment(path=os.path.join(ROOT, "assets", "std_env.hdr")) if w["hdr"] else None
    cam = scene.camera(0)
    frame_info, pixel_angle, focal = ptmod.camera_frame_info(cam, W, H)
    if hdr is not None:
        frame_info.flags |= capi.MI_SCENE_USE_HDR_ENVIRONMENT
    params = ptmod.default_params()
    params.maxDepth, params.numSamples, params.pixelAngle, params.focalDistance = w["depth"], 1, pixel_angle, focal

    def make_tracer(counters):
        t = ptmod.PathTracer(scene, device=local_rank, collect_counters=counters, bvh=args.bvh)
        if hdr is not None:
            t.set_environment(hdr)
        t.set_tile_partition(rank, world, args.tile)
        t.resize(W, H)
        t.set_frame_info(frame_info)
        t.set_sky(ptmod.default_sky())
        return t

    tracer = make_tracer(False)
    accum = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    tracer.bind_accum(accum.data_ptr())
    stream = torch.cuda.current_stream()
    runner = ptmod.HeadlessRenderer(tracer, params)

    def sync_all():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # a step = one batch of F frames (1 spp each) sharing every wavefront launch.  Every rank owns 1/world of the tiles of each
    # frame, so the batch grows with the world size to keep the rays in flight per GPU constant (weak scaling).
    F = min(256, max(1, args.in_flight) * world)
    runner.render(args.warmup * F, stream.cuda_stream, in_flight=F)
    if dist is not None:  # warm the RCCL path too
        dist.reduce(accum.clone(), dst=0)
    sync_all()
    runner.reset_frame()
    tracer.enable_timing(True)
    sync_all()
    t0 = time.perf_counter()
    runner.render(args.steps * F, stream.cuda_stream, in_flight=F)
    if dist is not None:
        dist.reduce(accum, dst=0, op=dist.ReduceOp.SUM)  # disjoint tiles: sum == gather
    sync_all()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    timing = tracer.frame_timing()
    tracer.enable_timing(False)
    samples = float(W) * float(H) * float(args.steps) * F
    value = samples / elapsed / 1e6

    result = None
    if rank == 0:
        img = accum.cpu().numpy()
        assert np.isfinite(img).all()
        # counter pass (deterministic: same frames -> same counts) for the algorithmic-bytes model
        ctr = make_tracer(True)
        ctr_runner = ptmod.HeadlessRenderer(ctr, params)
        n_ctr = min(args.steps * F, 4)
        ctr_runner.render(n_ctr)
        stats = ctr.stats()
        ctr.close()
        per_frame = {k: (v / n_ctr if k not in ("bvhNodeCount", "bvhTriangleCount", "bvhNodeBytes", "bvhTriangleBytes") else v) for k, v in stats.items()}
        kernels = {"trace_closest": ("traceClosestMs", "traceClosestLaunches"), "shade": ("shadeMs", "shadeLaunches"),
                   "trace_shadow": ("traceShadowMs", "traceShadowLaunches")}
        dominant = max(kernels, key=lambda k: timing[kernels[k][0]])
        ms_key, n_key = kernels[dominant]
        launches = max(timing[n_key], 1)
        avg_launch_ms = timing[ms_key] / launches
        # counters were taken on rank 0's tiles; bytes per launch = bytes per frame / launches per frame
        launches_per_frame = launches / (args.steps * F)
        bytes_per_launch = algorithmic_bytes(per_frame, dominant) / launches_per_frame
        achieved = bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9
        traffic = None
        pmc_file = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc_file):
            try:
                pmc = json.load(open(pmc_file))
                if pmc.get("workload") == args.workload and pmc.get("frames_in_flight") == F and pmc.get("resolution") == [W, H] and world == 1:
                    traffic = pmc.get("bench_kernel_traffic", {}).get(dominant)
            except Exception:
                traffic = None
        result = {
            "metric": "Msamples/s (and ms/frame @ fixed spp) 1080p & 4K, 1/2/4/8 MI355X", "value": round(value, 3), "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": w["config"] + " (seeded synthetic stand-in)" if w["gen"] else w["config"], "scene_triangles": scene.num_triangles,
                       "resolution": [W, H], "spp_per_step": F, "frames_in_flight": F, "max_depth": w["depth"], "tile": args.tile,
                       "parallelism": f"tiles{world}" if world > 1 else "single"},
            "roofline": {"bound": "hbm", "kernel": dominant, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "avg_launch_ms": round(avg_launch_ms, 5),
                         # measured HBM bytes (profiles/pmc_latest.json) over the same launch time: what actually crossed the memory
                         # interface; `achieved` counts the records the algorithm touches whether or not a cache served them
                         "traffic_GBps": (round(traffic / (avg_launch_ms * 1e-3) / 1e9, 1) if traffic else None),
                         "algorithmic_bytes_per_launch": round(bytes_per_launch),
                         "per_frame": {k: round(per_frame[k], 1) for k in ("segments", "shadowRays", "nodesClosest", "trisClosest", "nodesShadow", "trisShadow", "textureTaps")},
                         "kernel_ms_per_frame": {k: round(timing[v[0]] / (args.steps * F), 4) for k, v in kernels.items()},
                         "frame_ms_device": round(timing["totalMs"] / (args.steps * F), 4)},
        }
        if not args.no_cpu_baseline and world == 1:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import parity_util as pu  # the oracle is used here only as the timed CPU baseline
            setup = pu.Setup(scene.path, W, H, hdr_path=os.path.join(ROOT, "assets", "std_env.hdr") if w["hdr"] else None, max_depth=w["depth"])
            import oracle_lib
            O = oracle_lib.lib()
            o = C.c_void_p()
            O.oracle_pt_create(setup.scene.desc, C.byref(o))
            if setup.hdr is not None:
                O.oracle_pt_set_environment(o, setup.hdr.env)
            cores = os.cpu_count() or 1
            # bounded sample: every 16th 64x64 tile of the same frames (same scene bytes, seeds, depth), then more tiles if fast
            O.oracle_pt_resize(o, W, H)
            O.oracle_pt_set_frame_info(o, C.byref(setup.frame_info))
            O.oracle_pt_set_sky(o, C.byref(setup.sky))
            tiles_total = ((W + 63) // 64) * ((H + 63) // 64)
            done_px, frames_done, t_cpu0 = 0, 0, time.perf_counter()
            part = 16
            O.oracle_pt_set_tile_partition(o, 0, part, 64)
            owned = sum(1 for t in range(tiles_total) if t % part == 0)
            # pixels in owned tiles (edge tiles are partial)
            tx = (W + 63) // 64
            px_owned = sum(min(64, W - (t % tx) * 64) * min(64, H - (t // tx) * 64) for t in range(tiles_total) if t % part == 0)
            while time.perf_counter() - t_cpu0 < args.cpu_seconds and frames_done < 4096:
                p = setup.frame_params(frames_done, frames_done)
                O.oracle_pt_render_frame(o, C.byref(p), cores)
                frames_done += 1
                done_px += px_owned
            t_cpu = time.perf_counter() - t_cpu0
            O.oracle_pt_destroy(o)
            result["cpu_baseline"] = {"value": round(done_px / t_cpu / 1e6, 4), "unit": "Msamples/s", "cores": cores, "kind": "port",
                                      "sample": f"{frames_done} frame(s) x {owned}/{tiles_total} tiles (every {part}th 64x64 tile) of the same workload, {t_cpu:.1f} s"}
        print(json.dumps(result), flush=True)
    tracer.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
