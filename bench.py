#!/usr/bin/env python3
"""bench.py — throughput of the path-trace hot path on N MI355X GPUs of one node.

A "step" is one pass of the hot path over one batch of synthetic input: `--frames-per-step` (default 256) consecutive frames
per GPU, each ptSamples = 1 sample per pixel like the reference's headless run `--frames K --ptSamples 1`
(docs/benchmarking.md:16-23).  The default workload is the one the north-star target is stated on and that fits one GPU:
configs[2], Sponza-class atrium, 1920x1080, depth 12, NEE + MIS (the asset itself is not available offline;
vk_gltf_renderer_amd.scenegen writes a seeded stand-in of the same class as a .glb).  The frames of a step are issued through
mi_pt_render_frames in groups of `--in-flight` (default 128) that share every wavefront launch (bit-identical to rendering them
one after the other; --in-flight 1 gives exactly that).  Metric = the reference's throughput_MSps (src/benchmarking.cpp:272-279):
W*H*spp / wall_s / 1e6 with spp = all samples of the timed region, inputs resident in HBM before the timed region.

N > 1: the image is split into interleaved 32x32 tiles (--tile; tile % N == rank), every rank renders its tiles with no data-path
collective, and ONE RCCL reduce(sum) of the RGBA32F accumulator over xGMI closes every step (inside the timed region).
--scaling strong (default): a step is the configuration's own sample count whatever N is (256 frames for configs[2]; the frames of a
step are in flight together, a GPU holds 1/N of their path slots) -- total work fixed, the north star's 1 -> 8 claim.
--scaling weak: a step renders frames_per_step * N frames, in_flight * N in flight -- work per GPU fixed (step_shape()).
`python bench.py --gpus N` without a launcher starts its own N ranks (torch.distributed.run on 127.0.0.1) and refuses to run when
fewer than N devices are visible.

Rank 0 prints ONE JSON line.  Besides the driver's fields it carries
  north_star    the Sponza-class value again with the per-GPU rate the 8-GPU target needs;
  roofline      the dominant kernel against the roof that bounds it, and under "kernels" the same for every kernel of the step.
                "hbm" kernels: frac = bytes that CROSS THE MEMORY INTERFACE per launch (rocprofv3 FETCH_SIZE / WRITE_SIZE passes of this
                command, calibrated, committed as profiles/pmc_latest_<workload>.json) / the launch time measured in THIS run / 8 TB/s --
                it cannot pass 1; SURVEY 8(d)'s algorithmic bytes stand beside it as `algorithmic_frac` (which can: caches serve most of
                them).  "valu" kernels (the BVH walks, bound by vector-instruction issue): frac = USEFUL vector-lane operations (slab and
                triangle tests at 64 lanes, per-kernel instruction counts of the compiled code) per launch / launch time / 78.6 Tlaneop/s, with
                the measured issue fraction (`issue_frac`: vector instructions x 4 cycles / SIMD cycles, from the committed SQ counter pass)
                beside it.  LABNOTES.md section 4 states both models;
  also          the other configurations of BASELINE.json measured right after the headline on the same GPU, each with its own per-kernel
                table: "helmet" (configs[1]), "helmet_4k" (the same at 3840x2160), "street" (configs[3], 3840x2160) and "glass_denoise"
                (configs[4] with its a-trous pass), the ones with a different scene with their own cpu_baseline + parity leg;
  cpu_baseline  the CPU oracle timed on the host cores on a bounded sample of the same frames;
  parity        the GPU accumulator against the oracle's on exactly those sample tiles (same frames, same seeds) after the configuration's
                OWN sample count (256 spp for configs[2]), with the figure at a quarter and half of it (`by_spp`) so that the
                1/sqrt(spp) fall of the difference is visible.
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

# Frames in flight per GPU and frames per step: 64 -> 96 -> 128 frames measured 3809 / 3896 / 3927 Msamples/s on the helmet and 484 / 489 / 492
# on the atrium (round 3; a batch of 128 1080p frames is 2.65e8 path slots, ~80 GB of the 288)
# Round 6: 64 frames in flight by default (33.8 GB of path state at 1080p instead of 67.5) -- 97.6 % of the 128-frame rate on the atrium (728.6 / 746.4 / 758.5 Msamples/s at
# 64 / 128 / 256, profiles/r06_in_flight.txt); the 128-frame figure stands beside the headline as also.atrium_f128.  The glass workload keeps 256 (528 / 546 / 563 at 128 / 192 / 256)
IN_FLIGHT_DEFAULT, FRAMES_PER_STEP_DEFAULT = 64, 256
ALPHA_CUT_DEFAULT = 4  # measured (adaptive cut): atrium 462 -> 483 / 479 / 494 and street 455 -> 510 / 507 / 490 Msamples/s at 4 / 8 / 16
WORKLOADS = {
    # name: BASELINE config, generator kwargs, width, height, maxDepth, env, spp = the configuration's OWN sample count (the parity leg's)
    "helmet": dict(config="configs[1]: DamagedHelmet-class + std_env.hdr, 1920x1080, 64 spp, depth 8", gen="scene_helmet_class",
                   kw=dict(seed=1234, tess=272, tex_size=2048), width=1920, height=1080, depth=8, hdr=True, spp=64),
    "atrium": dict(config="configs[2]: Sponza-class, 1920x1080, 256 spp, depth 12, NEE+MIS (directional light + sky)", gen="scene_atrium_class",
                   kw=dict(seed=4321, detail=0.8, tex_size=512), width=1920, height=1080, depth=12, hdr=False, spp=256),
    "atrium_f128": dict(config="configs[2] with 128 frames in flight: Sponza-class, 1920x1080, 256 spp, depth 12, NEE+MIS", gen="scene_atrium_class",
                        kw=dict(seed=4321, detail=0.8, tex_size=512), width=1920, height=1080, depth=12, hdr=False, spp=256, in_flight=128),
    "street": dict(config="configs[3]: BistroExterior-class (instanced street, ~2.8 M triangles, ~1000 render nodes, 130 materials), 3840x2160, 64 spp, depth 8",
                   gen="scene_street_class", kw=dict(seed=777, detail=1.27, tex_size=256), width=3840, height=2160, depth=8, hdr=False, spp=64),
    "glass": dict(config="configs[4]: TransmissionTest-class sphere grid + textured glass slabs + DragonDispersion-class blob (869 k triangles, dispersion + volume), "
                         "1920x1080, 512 spp, depth 24", gen="scene_glass_class",
                  kw=dict(seed=99, tess=96, dragon=932), width=1920, height=1080, depth=24, hdr=True, in_flight=256, frames_per_step=512, spp=512),
    # the sphere grid alone (rounds 1-5's configs[4] stand-in, 166 k triangles, no texture): the unit-test size, kept for comparison
    "glass_grid": dict(config="configs[4] (sphere grid only): TransmissionTest-class, 1920x1080, 512 spp, depth 24", gen="scene_glass_class",
                       kw=dict(seed=99, tess=96), width=1920, height=1080, depth=24, hdr=True, in_flight=256, frames_per_step=512, spp=512),
    # SURVEY 8(d) "triangle sizes log-uniform": the same hall / street with hall-sized wall triangles, 16:1 strips, long thin beams and cables next to
    # finely tessellated detail (scenegen sliver=True) -- what a BVH builder meets on the real assets
    "atrium_sliver": dict(config="configs[2] with log-uniform triangle sizes (edges 2 mm .. 38 m, long thin beams): Sponza-class, 1920x1080, 256 spp, depth 12, NEE+MIS",
                          gen="scene_atrium_class", kw=dict(seed=4321, detail=0.8, tex_size=512, sliver=True), width=1920, height=1080, depth=12, hdr=False, spp=256),
    "street_sliver": dict(config="configs[3] with 16:1 facade strips, 240-m road strips and overhead cables: BistroExterior-class, 3840x2160, 64 spp, depth 8",
                          gen="scene_street_class", kw=dict(seed=777, detail=1.27, tex_size=256, sliver=True), width=3840, height=2160, depth=8, hdr=False, spp=64),
    "box": dict(config="configs[0]: resources/Box.glb, 256x256, 16 spp, depth 4", gen=None, kw={}, width=256, height=256, depth=4, hdr=True, spp=16),
}
# lines of the default run's "also": name -> (workload, width, height, denoise, parity leg)
ALSO_LINES = {
    "helmet": ("helmet", 0, 0, False, True),
    "helmet_4k": ("helmet", 3840, 2160, False, False),  # the same scene as "helmet": no second parity leg
    "atrium": ("atrium", 0, 0, False, True),
    "atrium_f128": ("atrium_f128", 0, 0, False, False),  # the headline configuration with 128 frames in flight (rounds 3-5's default)
    "street": ("street", 0, 0, False, True),
    "glass": ("glass", 0, 0, False, True),
    "glass_denoise": ("glass", 0, 0, True, True),
    "glass_grid": ("glass_grid", 0, 0, False, False),
    "glass_grid_denoise": ("glass_grid", 0, 0, True, False),
    "atrium_sliver": ("atrium_sliver", 0, 0, False, True),
    "street_sliver": ("street_sliver", 0, 0, False, False),  # (parity leg on request: --also street_sliver_parity)
    "street_sliver_parity": ("street_sliver", 0, 0, False, True),
}
ALSO_DEFAULT = "helmet,helmet_4k,street,glass_denoise,atrium_sliver,street_sliver,atrium_f128"
NORTH_STAR = {"workload": "atrium", "target": ">= 2 Gsamples/s on Sponza 1080p at 8 x MI355X (BASELINE.json north_star)", "needs_per_gpu_Msamples_s": 250.0}
# /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy); 256 CUs x 4 SIMD-32 x 2.4 GHz = 78.6 T
# vector-lane operations per second (x2 flops per fma = the 157.3 TFLOP/s fp32 vector peak)
HBM_PEAK_GBS = 8000.0
VALU_PEAK_TLANEOPS = 256 * 4 * 32 * 2.4e9 / 1e12
# Vector-ALU instructions of the inner operations of the BVH walks, counted in the compiled code (tools/count_valu.sh): one 8-wide
# node visit of the per-lane walk (decode + 8 slab tests + child order), one triangle test (Moeller-Trumbore + candidate update),
# and one node of the packet walk's INTERVAL test (k_trace_primary with pixel-major slots: lane = child x 8 + plane tests one plane
# of the node against the packet's interval ray -- all 64 lanes of ~50 instructions work on ONE node, pt_packet.h)
VALU_PER_NODE, VALU_PER_TRI, VALU_PER_PACKET_NODE = 196, 56, 50  # (node visit: 235 until round 4's leaf word, 205 until its sign-free offsets)
# SURVEY §8(d) algorithmic bytes, per unit: ray + hit record, path state read + write, hit-attribute gather, instance +
# primitive + material records, shadow-ray record, one texture tap, pixel accumulate
B_RAYHIT, B_STATE, B_ATTR, B_RECORDS, B_SHADOW, B_TAP, B_PIXEL = 60, 192, 192, 480, 76, 48, 32


def scene_path(name, rank):
    w = WORKLOADS[name]
    if w["gen"] is None:
        return os.path.join(ROOT, "assets", "Box.glb")
    from vk_gltf_renderer_amd import scenegen
    d = os.path.join(tempfile.gettempdir(), "mi_pt_scenes")
    os.makedirs(d, exist_ok=True)
    tag = "_".join(f"{k}{v}" for k, v in sorted(w["kw"].items()))
    path = os.path.join(d, f"{w['gen']}_{tag}.glb")  # (keyed by generator + arguments: workloads that differ only in how they are run share the file)
    if not os.path.exists(path):
        tmp = f"{path}.{os.getpid()}.{rank}.tmp"
        getattr(scenegen, w["gen"])(tmp, **w["kw"])
        os.replace(tmp, path)
    return path


# configurations of the default run whose counter passes (profiles/pmc_latest_<workload>.json) have not been collected yet for the current kernels / scenes: their
# lines print `traffic: null`.  tests/test_bench_contract.py lets exactly these pass without a file.
PMC_PENDING = set()


def load_pmc(workload, F, W, H):
    """The committed counter passes of this configuration (profiles/pmc_latest_<workload>[_4k].json, written by tools/make_pmc_latest.py
    from separate rocprofv3 --pmc runs of this command), or None when no pass matches workload / resolution / frames in flight.
    Never measured inside this run: counter collection serialises the kernels."""
    tag = workload + ("_4k" if (workload == "helmet" and W == 3840) else "")
    for f in (os.path.join(ROOT, "profiles", f"pmc_latest_{tag}.json"),):
        try:
            pmc = json.load(open(f))
        except Exception:
            continue
        if pmc.get("workload") == workload and pmc.get("frames_in_flight") == F and pmc.get("resolution") == [W, H]:
            pmc["_file"] = f"profiles/{os.path.basename(f)}"
            return pmc
    return None


def kernel_table(all_b, first_b, timing, frames, in_flight=0, pmc=None, guides=False):
    """Per-kernel roofline entries.  all_b / first_b: counters PER FRAME of the whole path loop and of bounce 0 alone (a second
    counter pass with maxDepth = 1); timing: MiPtFrameTiming totals over `frames` frames; pmc: load_pmc() of this configuration.
    "hbm" rows: achieved = bytes across the memory interface per launch (pmc) / THIS run's launch time, frac = achieved / 8 TB/s; the
    SURVEY 8(d) bytes (hit/miss aware: a segment that leaves the scene carries its ray and path state only) / launch time are
    `algorithmic_GBps`, `algorithmic_frac` -- a figure that can pass 1 because caches serve most of those bytes.
    "valu" rows: achieved = useful vector-lane operations per launch / launch time, frac = achieved / 78.6 Tlaneop/s; `issue_frac`
    (pmc) = vector instructions x 4 cycles / (1024 SIMDs x launch time x 2.4 GHz), `active_lanes` = lanes per vector instruction."""
    rest = {k: all_b[k] - first_b.get(k, 0) for k in all_b}
    fused = timing["tracePrimaryLaunches"] > 0
    interval = in_flight >= 64 and in_flight % 64 == 0  # pixel-major slots: k_trace_primary tests nodes against the packet's interval ray
    pk = (pmc or {}).get("kernels", {})

    def shade_bytes(c, shaded):
        return shaded * (B_RAYHIT + B_STATE) + c["surfaceHits"] * (B_ATTR + B_RECORDS + B_SHADOW) + c["textureTaps"] * B_TAP

    def walk_ops(nodes, tris):
        return nodes * VALU_PER_NODE + tris * VALU_PER_TRI

    def packet_ops(nodes, tris):  # every lane of the wave works on every record the wave fetches
        return 64 * (nodes * (VALU_PER_PACKET_NODE if interval else VALU_PER_NODE) + tris * VALU_PER_TRI)

    rows = {}

    def add(name, ms, launches, bound, work_per_frame, note):
        if launches <= 0 or ms <= 0:
            return
        per_launch = work_per_frame * frames / launches
        avg_ms = ms / launches
        p = pk.get(name)
        row = {"bound": bound}
        if bound == "hbm":
            alg = per_launch / (avg_ms * 1e-3) / 1e9
            traffic = p["hbm_bytes_per_launch"] if p else None
            achieved = traffic / (avg_ms * 1e-3) / 1e9 if traffic else None
            row.update({"achieved": round(achieved, 1) if achieved else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(achieved / HBM_PEAK_GBS, 4) if achieved else None, "traffic": traffic,
                        "algorithmic_GBps": round(alg, 1), "algorithmic_frac": round(alg / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": round(per_launch)})
        else:
            achieved = per_launch / (avg_ms * 1e-3) / 1e12
            row.update({"achieved": round(achieved, 3), "peak": round(VALU_PEAK_TLANEOPS, 2), "unit": "Tlaneop/s", "frac": round(achieved / VALU_PEAK_TLANEOPS, 4),
                        "traffic": p["hbm_bytes_per_launch"] if p else None, "useful_laneops_per_launch": round(per_launch)})
        if p:
            for k in ("issue_frac", "issue_frac_bounds", "issue_cycles_per_instruction", "active_lanes", "l2_hit_rate", "wave_time_waiting"):
                if p.get(k) is not None:
                    row[k] = p[k]
        row.update({"avg_launch_ms": round(avg_ms, 5), "launches": launches, "ms_per_frame": round(ms / frames, 5), "model": note})
        rows[name] = row

    if fused:
        add("trace_primary", timing["tracePrimaryMs"], timing["tracePrimaryLaunches"], "valu", packet_ops(all_b["nodesPrimary"], all_b["trisPrimary"]),
            f"64 x (nodesPrimary x {VALU_PER_PACKET_NODE if interval else VALU_PER_NODE} + trisPrimary x {VALU_PER_TRI}) lane-ops (packet walk: "
            + ("interval node test, one plane per lane" if interval else "every lane tests every child") + ")")
        add("shade_first", timing["shadeFirstMs"], timing["shadeFirstLaunches"], "hbm", shade_bytes(first_b, first_b["surfaceHits"]),
            "algorithmic: bounce 0: surfaceHits x (60 + 192 + 192 + 480 + 76) + textureTaps x 48 B (camera rays that leave the scene end in k_trace_primary)")
        add("trace_closest", timing["traceClosestMs"] - timing["tracePrimaryMs"], timing["traceClosestLaunches"] - timing["tracePrimaryLaunches"], "valu",
            walk_ops(all_b["nodesClosest"], all_b["trisClosest"]), f"nodesClosest x {VALU_PER_NODE} + trisClosest x {VALU_PER_TRI} lane-ops (bounces >= 1)")
        add("shade", timing["shadeMs"] - timing["shadeFirstMs"], timing["shadeLaunches"] - timing["shadeFirstLaunches"], "hbm", shade_bytes(rest, rest["segments"]),
            "algorithmic: bounces >= 1: segments x (60 + 192) + surfaceHits x (192 + 480 + 76) + textureTaps x 48 B")
    else:
        add("trace_closest", timing["traceClosestMs"], timing["traceClosestLaunches"], "valu", walk_ops(all_b["nodesClosest"], all_b["trisClosest"]),
            f"nodesClosest x {VALU_PER_NODE} + trisClosest x {VALU_PER_TRI} lane-ops")
        add("shade", timing["shadeMs"], timing["shadeLaunches"], "hbm", shade_bytes(all_b, all_b["segments"]),
            "algorithmic: segments x (60 + 192) + surfaceHits x (192 + 480 + 76) + textureTaps x 48 B")
    add("trace_shadow", timing["traceShadowMs"], timing["traceShadowLaunches"], "valu", walk_ops(all_b["nodesShadow"], all_b["trisShadow"]),
        f"nodesShadow x {VALU_PER_NODE} + trisShadow x {VALU_PER_TRI} lane-ops")
    if timing["accumulateMs"] > 0 and in_flight > 0:
        # k_finish_sample (SURVEY 8(d) "pixel accumulate"): one 16-B path record per pixel and frame, the accumulator once per launch
        # ... and with the denoiser guides captured (--denoise): the path's albedo and normal sums (2 x 16 B) and the two guide images beside the accumulator
        per_path = (48.0 + 96.0 / in_flight) if guides else (16.0 + 32.0 / in_flight)
        add("finish_sample", timing["accumulateMs"], max(1, round(frames / in_flight)), "hbm", all_b["cameraPaths"] * per_path,
            "algorithmic: cameraPaths x (16 B path record + 32 B of accumulator per launch" + (" + 32 B of guide sums + 64 B of guide images per launch)" if guides else ")"))
    return rows


def overlap_note(kernels, ms_per_frame):
    """Sum of the per-kernel device times over the frame time.  ~1 when the launches of a frame run one after the other; clearly above 1 when
    the library runs a bounce's shadow stage on a second stream next to the following closest-hit walk (small batches, and every batch of a
    volume-scatter scene: LABNOTES.md section 2) -- per-kernel times and fractions are then those of kernels that SHARE the device."""
    ratio = sum(k["ms_per_frame"] for k in kernels.values()) / max(ms_per_frame, 1e-9)
    out = {"kernel_time_sum_over_frame_time": round(ratio, 3)}
    if ratio > 1.1:
        out["note"] = ("two streams in the timed run: the shadow stage of a bounce runs next to the following bounce's closest-hit walk (MI_PT_OVERLAP), so the "
                       "per-kernel times of that run overlap")
    return out


def single_stream_timing(make_tracer, params, frames_step, F, steps=2, denoise=False):
    """Per-launch times of a configuration whose timed run overlaps kernels on two streams: the same frames once more on ONE stream
    (MI_PT_OVERLAP=0, read at mi_pt_create), so that the kernel table prices every kernel with the device to itself.  The throughput
    of the line stays that of the shipped two-stream schedule."""
    from vk_gltf_renderer_amd import pathtracer as ptmod
    old = os.environ.get("MI_PT_OVERLAP")
    os.environ["MI_PT_OVERLAP"] = "0"
    try:
        t = make_tracer()
    finally:
        if old is None:
            del os.environ["MI_PT_OVERLAP"]
        else:
            os.environ["MI_PT_OVERLAP"] = old
    r = ptmod.HeadlessRenderer(t, params)
    r.render(F, in_flight=F)
    t.synchronize()
    r.reset_frame()
    t.enable_timing(True)
    for _ in range(steps):
        r.render(frames_step, in_flight=F)
        if denoise:
            t.denoise_svgf(iterations=5, read=False)
    t.synchronize()
    timing = t.frame_timing()
    t.close()
    return timing, steps * frames_step


def roofline_of(kernels, pmc):
    """The `roofline` object: the kernel with the largest share of the step, from the table above."""
    dominant = max(kernels, key=lambda k: kernels[k]["avg_launch_ms"] * kernels[k]["launches"])
    roof = dict(kernels[dominant], kernel=dominant)
    if pmc is not None:
        roof["traffic_source"] = (f"{pmc['_file']} ({pmc.get('command', 'rocprofv3 --pmc passes')}; round {pmc.get('round')}; FETCH_SIZE calibrated per access class, "
                                  f"{pmc.get('fetch_size_calibration', '')}): counters of a separate run of this configuration, launch time of THIS run")
        if roof.get("traffic"):
            roof["traffic_GBps"] = round(roof["traffic"] / (roof["avg_launch_ms"] * 1e-3) / 1e9, 1)
    else:
        roof["traffic_source"] = None
    return roof


# path slots (frames in flight x pixels) per GPU: ~0.3 KB of path state and queue entries per slot, ~160 GB of the 288 -- enough for 64
# 4K frames: a multiple of 64 frames in flight lays the path slots out pixel major (a wave = 64 samples of one pixel, LABNOTES.md section 2)
SLOT_BUDGET = 5.4e8


def frames_in_flight(wanted, width, height, world=1):
    """Frames in flight within the slot budget; rounded down to a multiple of 64 when at least 64 fit (pixel-major path slots)."""
    f = max(1, min(wanted, int(SLOT_BUDGET * world // (width * height))))
    return f - f % 64 if f >= 64 else f


def step_shape(scaling, world, in_flight, frames_per_step, W, H, exact=False):
    """(frames in flight, frames per step) of an N-rank run.  Every rank owns 1/world of the pixels of each frame, so F frames in flight are
    F * W * H / world path slots on each GPU.
    weak:   frames per step and frames in flight grow with the world size -- path slots per GPU constant (frames_per_step * world, in_flight * world).
    strong: a step stays the configuration's own frames_per_step (256 spp for configs[2]) whatever the world size; its frames are in flight together up
            to in_flight * world, so at N = 8 a GPU holds 256 x W*H/8 = 32 W*H path slots -- the per-GPU rate is the one of a 32-frame batch
            (DESIGN section 5 quotes it), which is what a fixed total job costs.  At world = 1 both modes are the same run."""
    if scaling == "strong":
        frames_step = max(1, frames_per_step)
        F = min(1024, max(1, in_flight) * world, frames_step)
    else:
        frames_step = max(1, frames_per_step) * world
        F = min(1024, max(1, in_flight) * world)
    # ... within the budget of path slots per GPU (SLOT_BUDGET): 4K frames run 64 in flight
    F = frames_in_flight(F, W, H, world) if not exact else max(1, min(F, int(SLOT_BUDGET * world // (W * H))))
    return F, frames_step


def alpha_cut_note(subdivisions, triangles_loaded, dropped):
    if subdivisions <= 0:
        return None
    return {"subdivisions": subdivisions, "triangles_loaded": triangles_loaded, "sub_triangles_dropped": dropped,
            "note": "GEOMETRY BAKED AT LOAD: alpha-MASK triangles are re-tessellated and the pieces on which the alpha test cannot pass are removed "
                    "(mi_scene_cut_alpha, the counterpart of the reference's opacity micro-map bake); the parity leg renders the scene AS LOADED (uncut) with the CPU oracle"}


def cpu_baseline_and_parity(scene, w, W, H, F, cpu_seconds, make_gpu_tracer, gpu_params, parity_spp=0, tile_step=16):
    """The CPU oracle on the host cores over a bounded sample of the workload -- every 16th 64x64 tile of the same frames at 1080p, every
    64th at 4K: about 32 tiles spread over the image (same scene bytes, seeds, depth) -- and the GPU accumulator of exactly those frames compared with the oracle's on exactly those tiles.
    Frames = the configuration's own sample count (`parity_spp`, default WORKLOADS[..]["spp"]: 256 for configs[2]); the difference is
    also taken at a quarter and at half of it.  The CPU baseline figure is the oracle's rate over the first `cpu_seconds` of that run.
    The oracle is used here only as the timed CPU baseline and as the checker of the GPU image (it renders the scene AS LOADED:
    no alpha cut)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import parity_util as pu
    import oracle_lib
    from vk_gltf_renderer_amd import pathtracer as ptmod
    if oracle_lib._lib is None:
        oracle_lib.use_native()  # the timed baseline: -O3 -march=native, built on this box
    setup = pu.Setup(scene.path, W, H, hdr_path=w.get("hdr_path", os.path.join(ROOT, "assets", "std_env.hdr")) if w["hdr"] else None, max_depth=w["depth"])
    O = oracle_lib.lib()
    o = C.c_void_p()
    O.oracle_pt_create(setup.scene.desc, C.byref(o))
    if setup.hdr is not None:
        O.oracle_pt_set_environment(o, setup.hdr.env)
    cores = os.cpu_count() or 1
    O.oracle_pt_resize(o, W, H)
    O.oracle_pt_set_frame_info(o, C.byref(setup.frame_info))
    O.oracle_pt_set_sky(o, C.byref(setup.sky))
    tx, ty = (W + 63) // 64, (H + 63) // 64
    tiles_total = tx * ty
    # ~32 tiles whatever the resolution (4K: every 64th tile), so that the leg's CPU time follows spp, not pixels; the `also` lines take every second of those
    # (tile_step 32: ~16 tiles -- their five CPU legs were two thirds of the default run's wall time)
    part = tile_step * max(1, round(tiles_total / 510))
    O.oracle_pt_set_tile_partition(o, 0, part, 64)
    owned = [t for t in range(tiles_total) if t % part == 0]
    px_owned = sum(min(64, W - (t % tx) * 64) * min(64, H - (t // tx) * 64) for t in owned)
    mask = np.zeros((H, W), bool)
    for t in owned:
        mask[(t // tx) * 64:(t // tx) * 64 + 64, (t % tx) * 64:(t % tx) * 64 + 64] = True
    spp = int(parity_spp or w.get("spp", 64))
    marks = sorted({max(1, spp // 4), max(1, spp // 2), spp})
    cpu_imgs, timed = {}, None
    t_cpu0 = time.perf_counter()
    for f in range(spp):
        p = setup.frame_params(f, f)
        O.oracle_pt_render_frame(o, C.byref(p), cores)
        if timed is None and (time.perf_counter() - t_cpu0 >= cpu_seconds or f + 1 == spp):
            timed = (f + 1, time.perf_counter() - t_cpu0)
        if f + 1 in marks:
            cpu_imgs[f + 1] = np.ctypeslib.as_array(O.oracle_pt_accum(o), shape=(H, W, 4))[mask].copy()
    t_cpu = time.perf_counter() - t_cpu0
    O.oracle_pt_destroy(o)
    cpu = {"value": round(timed[0] * px_owned / timed[1] / 1e6, 4), "unit": "Msamples/s", "cores": cores, "kind": "port", "flags": oracle_lib.FLAGS,
           "sample": f"the first {timed[0]} frame(s) x {len(owned)}/{tiles_total} tiles (every {part}th 64x64 tile) of the same workload, {timed[1]:.1f} s "
                     f"(the parity leg went on to {spp} frames: {t_cpu:.1f} s)"}
    # parity at the FULL configuration: the same frames on the GPU, compared on the tiles the oracle rendered
    chk = make_gpu_tracer()
    runner, done, by_spp = ptmod.HeadlessRenderer(chk, gpu_params), 0, {}
    for m_ in marks:
        runner.render(m_ - done, in_flight=min(F, m_ - done))
        done = m_
        m = pu.compare_images(cpu_imgs[m_][None], chk.read_accum()[mask][None])
        by_spp[str(m_)] = {"rel_l2": float(f"{m['rel_l2']:.3e}"), "frac_within_1e-2": round(m["frac_within_1e-2"], 5), "frac_within_1e-4": round(m["frac_within_1e-4"], 5),
                           "frac_exact": round(m["frac_exact"], 5), "mean_rel_bias": float(f"{m['mean_rel_bias']:.2e}")}
    chk.close()
    parity = dict(by_spp[str(spp)])
    parity.update({"spp": spp, "frames": spp, "by_spp": by_spp, "tiles": len(owned), "pixels": int(mask.sum()), "resolution": [W, H], "tolerance_rel_l2": 1e-3,
                   "within_tolerance": bool(parity["rel_l2"] <= 1e-3),
                   "reference": "CPU oracle (oracle/oracle_pt.cpp), same scene bytes (as loaded: no alpha cut), seeds and frame indices"})
    return cpu, parity


def uncut_value(args, w, W, H, F, frames_step, device, hdr, frame_info, params, steps):
    """Msamples/s of the headline configuration on the scene as loaded (alpha cut off): one warm-up batch, `steps` timed steps."""
    import torch
    from vk_gltf_renderer_amd import pathtracer as ptmod
    scene = ptmod.Scene(args.scenefile or scene_path(args.workload, 0))
    t = ptmod.PathTracer(scene, device=device, collect_counters=False, bvh=args.bvh)
    if hdr is not None:
        t.set_environment(hdr)
    t.resize(W, H)
    t.set_frame_info(frame_info)
    t.set_sky(ptmod.default_sky())
    r = ptmod.HeadlessRenderer(t, params)
    r.render(F, in_flight=F)
    t.synchronize()
    r.reset_frame()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        r.render(frames_step, in_flight=F)
    t.synchronize()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    t.close()
    return {"value": round(float(W) * H * steps * frames_step / elapsed / 1e6, 3), "unit": "Msamples/s", "scene_triangles": scene.num_triangles, "steps": steps,
            "note": "the same timed region on the geometry as loaded (--alpha-cut 0)"}


def secondary_line(name, args, device, width=0, height=0, steps=5, parity=True, denoise=False):
    """Throughput + per-kernel roofline table of another workload (or the same one at another resolution) on this GPU: the same
    step definition as the headline (frames in flight x 2 frames per step, `steps` steps timed after one warm-up batch; with `denoise`
    one variance-guided a-trous pass closes every step inside the timed region, configs[4]), and with `parity` the same CPU-oracle
    baseline + full-size parity leg as the headline."""
    import torch
    from vk_gltf_renderer_amd import _capi as capi
    from vk_gltf_renderer_amd import pathtracer as ptmod
    w = WORKLOADS[name]
    W, H = width or w["width"], height or w["height"]
    scene = ptmod.Scene(scene_path(name, 0))
    triangles_loaded = scene.num_triangles
    dropped = scene.cut_alpha(args.alpha_cut) if args.alpha_cut > 0 else 0
    hdr = ptmod.HdrEnvironment(path=os.path.join(ROOT, "assets", "std_env.hdr")) if w["hdr"] else None
    frame_info, pixel_angle, focal = ptmod.camera_frame_info(scene.camera(0), W, H)
    if hdr is not None:
        frame_info.flags |= capi.MI_SCENE_USE_HDR_ENVIRONMENT

    def params(depth, guides=False):
        p = ptmod.default_params()
        p.maxDepth, p.numSamples, p.pixelAngle, p.focalDistance = depth, 1, pixel_angle, focal
        if guides:
            p.flags |= capi.MI_PT_USE_OPTIX_DENOISER
        return p

    def tracer(counters):
        t = ptmod.PathTracer(scene, device=device, collect_counters=counters, bvh=args.bvh)
        if hdr is not None:
            t.set_environment(hdr)
        t.resize(W, H)
        t.set_frame_info(frame_info)
        t.set_sky(ptmod.default_sky())
        return t

    F = frames_in_flight(w.get("in_flight", IN_FLIGHT_DEFAULT), W, H)
    frames_step = 2 * F
    t = tracer(False)
    r = ptmod.HeadlessRenderer(t, params(w["depth"], denoise))

    def step():
        r.render(frames_step, in_flight=F)
        if denoise:
            t.denoise_svgf(iterations=5, read=False)

    r.render(F, in_flight=F)  # warm-up
    if denoise:
        t.denoise_svgf(iterations=5, read=False)
    t.synchronize()
    r.reset_frame()
    t.enable_timing(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    t.synchronize()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    timing = t.frame_timing()
    mem = t.memory()
    t.close()

    def counter_pass(depth):
        c = tracer(True)
        ptmod.HeadlessRenderer(c, params(depth)).render(4)
        st = c.stats()
        c.close()
        return {k: (v if k.startswith("bvh") else v / 4) for k, v in st.items()}

    per_frame, first = counter_pass(w["depth"]), counter_pass(1)
    frames = steps * frames_step
    pmc = load_pmc(name, F, W, H)
    kernels = kernel_table(per_frame, first, timing, frames, F, pmc, guides=denoise)
    streams = overlap_note(kernels, elapsed / frames * 1e3)
    if "note" in streams:  # two streams in the timed run: the kernel table from the same frames on one stream
        timing1, frames1 = single_stream_timing(lambda: tracer(False), params(w["depth"], denoise), frames_step, F, denoise=denoise)
        kernels = kernel_table(per_frame, first, timing1, frames1, F, pmc, guides=denoise)
        streams["kernel_table"] = f"per-launch times of {frames1} frames of the same configuration on ONE stream (MI_PT_OVERLAP=0); `value` is the two-stream run"
    keys = ("cameraPaths", "segments", "surfaceHits", "shadowRays", "nodesPrimary", "trisPrimary", "nodesClosest", "trisClosest", "nodesShadow", "trisShadow", "textureTaps")
    line = {"value": round(float(W) * H * frames / elapsed / 1e6, 3), "unit": "Msamples/s", "ms_per_frame": round(elapsed / frames * 1e3, 5),
            "config": {"workload": w["config"].replace("1920x1080", f"{W}x{H}") + " (seeded synthetic stand-in)", "scene_triangles": scene.num_triangles,
                       "alpha_cut": alpha_cut_note(args.alpha_cut, triangles_loaded, dropped), "bvh_reinsertion_passes": int(os.environ.get("MI_PT_REINSERT", "16") or 0),
                       "resolution": [W, H],
                       "frames_in_flight": F, "max_depth": w["depth"], "frames_timed": frames, "steps": steps,
                       "denoise": ("variance-guided a-trous (mi_pt_denoise_svgf, 5 iterations) once per step, inside the timed region" if denoise else None)},
            "timed_region_s": round(elapsed, 3),
            "device_memory_GB": {"scene": round(mem["sceneBytes"] / 1e9, 3), "path_state_queues_images": round(mem["rendererBytes"] / 1e9, 3)},
            "bytes_per_path_slot": round(mem["pathStateBytes"] / max(1, mem["pathSlots"]), 1),
            "roofline": roofline_of(kernels, pmc), "kernels": kernels, "streams": streams,
            "per_frame": {k: round(per_frame[k], 1) for k in keys},
            "node_visits_per_secondary_ray": round(per_frame["nodesClosest"] / max(1.0, per_frame["segments"] - per_frame["cameraPaths"]), 2),
            "triangle_tests_per_secondary_ray": round(per_frame["trisClosest"] / max(1.0, per_frame["segments"] - per_frame["cameraPaths"]), 2)}
    if parity:
        line["cpu_baseline"], line["parity"] = cpu_baseline_and_parity(scene, w, W, H, F, min(args.cpu_seconds, 8.0), lambda: tracer(False), params(w["depth"]), args.parity_spp, tile_step=32)
    return line


ROOFLINE_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "hbm_frac", "algorithmic_frac", "issue_frac", "active_lanes", "l2_hit_rate",
                 "avg_launch_ms", "launches", "traffic")
FINAL_LINE_LIMIT = 6144  # bytes: the driver keeps the tail of stdout; round 4's single 33 KB line could not be parsed from it


def compact_roofline(roof, full=True):
    """The numbers of a `roofline` object (no prose).  `hbm_frac` = counter bytes across the memory interface per launch / launch time / 8 TB/s for
    every kernel, whatever roof `frac` is taken against (for an "hbm" kernel the two are the same figure)."""
    if not roof or "error" in roof:
        return roof
    r = dict(roof)
    if r.get("traffic") and r.get("avg_launch_ms"):
        r["hbm_frac"] = round(r["traffic"] / (r["avg_launch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    keys = ROOFLINE_KEYS if full else ("kernel", "bound", "frac", "hbm_frac", "active_lanes")
    return {k: r[k] for k in keys if r.get(k) is not None}


def compact_line(result):
    """The LAST stdout line of a run: the driver's contract fields, the dominant kernel's roofline as numbers, the CPU baseline, the parity
    figure, and one short record per other configuration -- the shape of the reference's own one-line summary record
    (src/benchmarking.cpp:281-303).  Kernel tables, per-frame counters and every prose field stay in the full record (bench_full.json, stderr)."""
    c = result["config"]
    out = {k: result[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    out["config"] = {k: c[k] for k in ("workload", "resolution", "spp_per_step", "frames_in_flight", "path_slots_per_gpu_in_frames", "max_depth", "tile", "parallelism", "world_size_reported_by_backend",
                                         "devices_visible", "reduce", "denoise", "library") if c.get(k) is not None}
    out["ms_per_frame"] = result["ms_per_frame"]
    out["roofline"] = compact_roofline(result.get("roofline"))

    def cpu(b):
        if not b or "error" in b:
            return b
        o = {k: b[k] for k in ("value", "unit", "cores", "kind", "flags") if k in b}
        o["sample"] = b.get("sample", "")[:120]
        return o

    def par(p):
        return p if (not p or "error" in p) else {k: p[k] for k in ("rel_l2", "spp", "tolerance_rel_l2", "within_tolerance", "pixels") if k in p}

    if "cpu_baseline" in result:
        out["cpu_baseline"], out["parity"] = cpu(result["cpu_baseline"]), par(result.get("parity"))
    for k in ("node_visits_per_secondary_ray", "bytes_per_path_slot"):
        if k in result:
            out[k] = result[k]
    if isinstance(result.get("value_uncut_geometry"), dict):
        out["value_uncut_geometry"] = result["value_uncut_geometry"].get("value", result["value_uncut_geometry"].get("error", "")[:80])
    if "device_memory_GB" in result:
        out["device_memory_GB"] = result["device_memory_GB"]
    if "north_star" in result:
        out["north_star"] = {k: result["north_star"][k] for k in ("value", "n_gpus", "value_per_gpu", "needs_per_gpu_Msamples_s", "frac_of_needed_per_gpu")}
    if "also" in result:
        out["also"] = {}
        for name, ln in result["also"].items():
            if "error" in ln:
                out["also"][name] = {"error": ln["error"][:160]}
                continue
            e = {"value": ln["value"], "ms_per_frame": ln["ms_per_frame"], "resolution": ln["config"]["resolution"], "roofline": compact_roofline(ln.get("roofline"), full=False)}
            if "bytes_per_path_slot" in ln:
                e["bytes_per_path_slot"], e["path_state_GB"] = ln["bytes_per_path_slot"], ln["device_memory_GB"]["path_state_queues_images"]
            if ln.get("cpu_baseline") and "error" not in ln["cpu_baseline"]:
                e["cpu_baseline"] = ln["cpu_baseline"]["value"]
            if ln.get("parity") and "error" not in ln["parity"]:
                e["parity_rel_l2"], e["parity_spp"] = ln["parity"]["rel_l2"], ln["parity"]["spp"]
            elif ln.get("parity"):
                e["parity_error"] = ln["parity"]["error"][:120]
            out["also"][name] = e
    out["full_record"] = "bench_full.json"
    return out


def emit(result):
    """The full record -> bench_full.json (BENCH_FULL overrides the path) and stderr; the compact line -> stdout, last."""
    full = json.dumps(result)
    path = os.environ.get("BENCH_FULL", os.path.join(ROOT, "bench_full.json"))
    try:
        with open(path, "w") as f:
            f.write(full + "\n")
    except OSError as e:
        print(f"bench.py: could not write {path}: {e}", file=sys.stderr)
    print(full, file=sys.stderr, flush=True)
    line = json.dumps(compact_line(result))
    assert len(line) < FINAL_LINE_LIMIT, len(line)
    print(line, flush=True)


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N ranks of this command under torch.distributed.run (one process per GPU,
    rendezvous on 127.0.0.1) and become that launcher.  Refuses when fewer than N devices are visible -- unless BENCH_SHARE_GPU=1
    (all ranks on device 0 over gloo: a functional check of the N > 1 code on a single-GPU box, not a measurement)."""
    import socket
    import torch
    share = os.environ.get("BENCH_SHARE_GPU") == "1"
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < n and not share:
        raise SystemExit(f"bench.py --gpus {n}: only {have} GPU(s) visible; refusing to report an {n}-GPU number from fewer devices "
                         f"(BENCH_SHARE_GPU=1 BENCH_DIST_BACKEND=gloo runs the ranks on one device as a functional check)")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "8"))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execvpe(cmd[0], cmd, env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default=NORTH_STAR["workload"], choices=sorted(WORKLOADS),
                    help="default: atrium = configs[2], the Sponza-class workload the north-star target is stated on")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--tile", type=int, default=32,
                    help="edge of the interleaved tiles the image is dealt out in (tile %% world == rank); 32 balances the 8 ranks of the "
                         "helmet workload to 4 %% (64: 19 %%, tools/check_rank_of_8.py)")
    ap.add_argument("--bvh", type=int, default=0, help="bit0: 0 = 8-wide compressed BVH (default), 1 = plain BVH2; bit1: 0 = PLOC topology (default), 1 = LBVH")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="the CPU baseline figure is the oracle's rate over the first this many seconds of the parity leg")
    ap.add_argument("--parity-spp", type=int, default=0,
                    help="frames (1 spp each) of the parity leg; 0 = the configuration's own sample count (atrium 256, helmet 64, street 64, glass 512)")
    ap.add_argument("--also", default=None,
                    help="comma-separated lines measured after the headline on a single GPU and reported under \"also\" (" + ", ".join(sorted(ALSO_LINES)) + "): by default `"
                         + ALSO_DEFAULT + "` next to the atrium -- every other configuration of BASELINE.json, the ones with their own scene with a CPU baseline and parity leg; "
                         "`--also none` switches it off")
    ap.add_argument("--denoise", action="store_true",
                    help="configs[4]'s denoise pass: the guide layers are captured with every frame and one variance-guided a-trous pass (mi_pt_denoise_svgf, "
                         "5 iterations) closes every step inside the timed region -- on rank 0, after the reduce, when N > 1")
    ap.add_argument("--alpha-cut", type=int, default=ALPHA_CUT_DEFAULT,
                    help="load-time bake for alpha-MASK geometry (mi_scene_cut_alpha: the counterpart of the reference's opacity micro-map bake): "
                         "subdivisions per triangle edge, 0 = off.  The parity leg renders the UNCUT scene with the CPU oracle")
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong",
                    help="N > 1.  strong (default): a step is the configuration's OWN sample count (frames_per_step frames, 256 for configs[2]) whatever N is -- every rank "
                         "renders its 1/N of the pixels of those frames, all of them in flight up to in_flight * N (N = 8: 256 frames x W*H/8 = 32 W*H path slots per GPU): total "
                         "work fixed, the north star's '>= 6x 1 -> 8 on Sponza 1080p'.  weak: frames_per_step * N frames per step, in_flight * N in flight -- path slots per GPU "
                         "constant.  At N = 1 the two are the same run")
    ap.add_argument("--scenefile", default=None,
                    help="a .gltf / .glb of your own (e.g. the real Sponza) instead of the seeded stand-in of --workload: same timed region, counter passes, CPU baseline and "
                         "parity leg; resolution / depth / spp default to those of --workload (override: --width --height --depth --parity-spp); the scene's first camera")
    ap.add_argument("--hdrfile", default=None, help="Radiance .hdr environment for --scenefile (default: std_env.hdr if --workload uses one, else the physical sky)")
    ap.add_argument("--depth", type=int, default=0, help="maxDepth override (default: the workload's)")
    ap.add_argument("--no-uncut", action="store_true", help="skip the second timed run on the geometry AS LOADED (no alpha cut) that the default single-GPU run reports as `value_uncut_geometry`")
    ap.add_argument("--frame-queue", type=int, default=0,
                    help="issue the frames through mi_pt_render_frame ONE CALL PER FRAME -- the reference's onRender cadence, INTEGRATION.md's stub -- with mi_pt_set_frame_queue(N): the library "
                         "holds the calls back and issues N frames at a time (N replaces --in-flight; same image).  N = 1: one frame per set of launches, no batching at all")
    ap.add_argument("--exact-in-flight", action="store_true", help="do not round the frames in flight down to a multiple of 64 (A/B of the slot layouts)")
    ap.add_argument("--frames-per-step", type=int, default=0,
                    help="frames (1 spp each) per GPU and step (default 256); a step renders frames_per_step * n_gpus frames")
    ap.add_argument("--in-flight", type=int, default=0,
                    help="frames in flight per GPU (mi_pt_render_frames, bit-identical to sequential frames): the frames of a step are issued in "
                         "groups of in_flight * n_gpus (capped at 1024).  Default 128 (helmet 3809 / 3927 Msamples/s at 64 / 128; 256 on the glass workload, whose "
                         "volume random walks leave a long tail of ~100 nearly empty bounce iterations per batch: 238 / 469 / 559 / 605 Msamples/s at 32 / 128 / 128 "
                         "(round 3) / 256 frames)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus)  # does not return: this process becomes the launcher of N ranks of the same command

    import torch
    from vk_gltf_renderer_amd import _capi as capi
    from vk_gltf_renderer_amd import pathtracer as ptmod

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
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
        assert dist.get_world_size() == world

    w = dict(WORKLOADS[args.workload])
    if args.scenefile:  # a supplied asset through the same legs: the workload entry only lends its resolution / depth / spp defaults
        w.update(config=f"--scenefile {os.path.basename(args.scenefile)} ({w['config'].split(':')[0]} settings)", gen="file", hdr=bool(args.hdrfile) or w["hdr"])
    if args.depth > 0:
        w["depth"] = args.depth
    args.in_flight = args.frame_queue or args.in_flight or w.get("in_flight", IN_FLIGHT_DEFAULT)
    args.frames_per_step = args.frames_per_step or w.get("frames_per_step", FRAMES_PER_STEP_DEFAULT)
    W, H = args.width or w["width"], args.height or w["height"]
    hdr_path = args.hdrfile or os.path.join(ROOT, "assets", "std_env.hdr")
    w["hdr_path"] = hdr_path
    scene = ptmod.Scene(args.scenefile or scene_path(args.workload, rank))
    triangles_loaded = scene.num_triangles
    alpha_cut_dropped = scene.cut_alpha(args.alpha_cut) if args.alpha_cut > 0 else 0
    hdr = ptmod.HdrEnvironment(path=hdr_path) if w["hdr"] else None
    cam = scene.camera(0)
    frame_info, pixel_angle, focal = ptmod.camera_frame_info(cam, W, H)
    if hdr is not None:
        frame_info.flags |= capi.MI_SCENE_USE_HDR_ENVIRONMENT
    params = ptmod.default_params()
    params.maxDepth, params.numSamples, params.pixelAngle, params.focalDistance = w["depth"], 1, pixel_angle, focal
    if args.denoise:
        params.flags |= capi.MI_PT_USE_OPTIX_DENOISER

    def make_tracer(counters, partition=True):
        t = ptmod.PathTracer(scene, device=local_rank, collect_counters=counters, bvh=args.bvh)
        if hdr is not None:
            t.set_environment(hdr)
        if partition:
            t.set_tile_partition(rank, world, args.tile)
        t.resize(W, H)
        t.set_frame_info(frame_info)
        t.set_sky(ptmod.default_sky())
        return t

    t_create0 = time.perf_counter()
    tracer = make_tracer(False)
    tracer.synchronize()
    create_s = time.perf_counter() - t_create0
    # accumulator (+ with --denoise the albedo / normal guides and the frame-0 depth) in ONE caller-owned allocation, so that a
    # multi-GPU step closes with one reduce of one buffer
    px = H * W
    frame_buf = torch.zeros(px * (13 if args.denoise else 4), dtype=torch.float32, device="cuda")
    reduced_buf = torch.zeros_like(frame_buf) if dist is not None else None

    def views(buf):
        v = {"accum": buf[:px * 4].view(H, W, 4)}
        if args.denoise:
            v.update(albedo=buf[px * 4:px * 8].view(H, W, 4), normal=buf[px * 8:px * 12].view(H, W, 4), depth=buf[px * 12:px * 13].view(H, W))
        return v

    def bind(v):
        tracer.bind_accum(v["accum"].data_ptr())
        if args.denoise:
            tracer.bind_guides(v["albedo"].data_ptr(), v["normal"].data_ptr(), v["depth"].data_ptr())

    local, total = views(frame_buf), (views(reduced_buf) if dist is not None else None)
    accum, reduced = local["accum"], (total["accum"] if total is not None else None)
    bind(local)
    stream = torch.cuda.current_stream()
    runner = ptmod.HeadlessRenderer(tracer, params)

    def sync_all():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    F, frames_step = step_shape(args.scaling, world, args.in_flight, args.frames_per_step, W, H, args.exact_in_flight)
    # (a rank owns numSlots = its tiles' pixels, ~W*H/world: F frames in flight are F*W*H/world path slots on this GPU)
    assert F * float(W) * float(H) / world <= SLOT_BUDGET * 1.02, "path slots per GPU beyond the budget"

    if args.frame_queue:  # one mi_pt_render_frame call per frame; the library batches them (exact-in-flight: the queue depth is the caller's figure)
        tracer.set_frame_queue(F)

    def step():
        runner.render(frames_step, stream.cuda_stream, in_flight=1 if args.frame_queue else F)
        if dist is not None:  # one reduce of the accumulator (+ guides) per step: disjoint tiles, so sum == gather
            reduced_buf.copy_(frame_buf)
            dist.reduce(reduced_buf, dst=0, op=dist.ReduceOp.SUM)
        if args.denoise and rank == 0:
            if dist is not None:  # denoise the reduced frame: point the library at it for the pass
                bind(total)
            tracer.denoise_svgf(iterations=5, read=False, stream=stream.cuda_stream)
            if dist is not None:
                bind(local)

    for _ in range(args.warmup):
        step()
    if dist is not None and args.warmup == 0:  # warm the RCCL path
        dist.reduce(frame_buf.clone(), dst=0)
    sync_all()
    runner.reset_frame()
    tracer.enable_timing(True)
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    timing = tracer.frame_timing()
    tracer.enable_timing(False)
    mem = tracer.memory()
    frames_timed = args.steps * frames_step
    samples = float(W) * float(H) * float(frames_timed)
    value = samples / elapsed / 1e6

    result = None
    if rank == 0:
        img = (reduced if dist is not None else accum).cpu().numpy()
        assert np.isfinite(img).all()
        if os.environ.get("BENCH_DUMP"):  # test hook: the final frame (and its denoised version) of this run
            dump = {"accum": img}
            if args.denoise:
                bind(total if dist is not None else local)
                dump["denoised"] = tracer.denoise_svgf(iterations=5, read=True)
                bind(local)
            np.savez(os.environ["BENCH_DUMP"], **dump)
        static = ("bvhNodeCount", "bvhTriangleCount", "bvhNodeBytes", "bvhTriangleBytes")

        def counter_pass(depth):
            # deterministic: same frames -> same counts; rank 0's tiles; 4 sequential frames
            ctr = make_tracer(True)
            p = ptmod.default_params()
            p.maxDepth, p.numSamples, p.pixelAngle, p.focalDistance = depth, 1, pixel_angle, focal
            n = min(frames_timed, 4)
            ptmod.HeadlessRenderer(ctr, p).render(n)
            st = ctr.stats()
            ctr.close()
            return {k: (v if k in static else v / n) for k, v in st.items()}

        per_frame = counter_pass(w["depth"])
        first = counter_pass(1) if w["depth"] >= 1 else dict(per_frame)  # bounce 0 alone: paths end after their first shade
        # the timed frames belong to this rank's tiles: timing and counters are both rank 0's
        pmc = load_pmc(args.workload, F, W, H) if world == 1 else None  # (the committed counter passes are single-GPU runs)
        kernels = kernel_table(per_frame, first, timing, frames_timed, F, pmc, guides=args.denoise)
        roof = roofline_of(kernels, pmc)
        keys = ("cameraPaths", "segments", "surfaceHits", "shadowRays", "nodesPrimary", "trisPrimary", "nodesClosest", "trisClosest", "nodesShadow", "trisShadow", "textureTaps")
        assert world == args.gpus
        result = {
            "metric": "Msamples/s (and ms/frame @ fixed spp) 1080p & 4K, 1/2/4/8 MI355X", "value": round(value, 3), "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": w["config"] + " (seeded synthetic stand-in)" if w["gen"] else w["config"], "scene_triangles": scene.num_triangles,
                       "alpha_cut": alpha_cut_note(args.alpha_cut, triangles_loaded, alpha_cut_dropped),
                       "bvh_reinsertion_passes": int(os.environ.get("MI_PT_REINSERT", "16") or 0),  # (read by the library at mi_pt_create; 0 = the builder's tree as clustered)
                       "resolution": [W, H], "spp_per_step": frames_step, "frames_in_flight": F, "frame_queue": (F if args.frame_queue else None), "path_slots_per_gpu_in_frames": round(F / world, 2), "max_depth": w["depth"], "tile": args.tile,
                       "parallelism": f"tiles{world}" if world > 1 else "single", "world_size_reported_by_backend": (dist.get_world_size() if dist is not None else 1),
                       "devices_visible": torch.cuda.device_count(),
                       "reduce": ((f"one RCCL reduce(sum) of {frame_buf.numel() * 4 / 1e6:.1f} MB (RGBA32F accumulator" + (" + albedo / normal guides + depth" if args.denoise else "") + ") per step") if dist is not None else None),
                       "denoise": ("variance-guided a-trous (mi_pt_denoise_svgf, 5 iterations) once per step, inside the timed region" if args.denoise else None),
                       "library": capi.pt_lib().mi_pt_version().decode()},
            "ms_per_frame": round(elapsed / frames_timed * 1e3, 5),
            "timed_region_s": round(elapsed, 3), "scene_build_s": round(create_s, 3),
            "roofline": roof,
            "kernels": kernels,
            "streams": overlap_note(kernels, elapsed / frames_timed * 1e3),
            "per_frame": {k: round(per_frame[k], 1) for k in keys},
            "per_frame_bounce0": {k: round(first[k], 1) for k in keys},
            "frame_ms_device": round(timing["totalMs"] / frames_timed, 4),
            "device_memory_GB": {"scene": round(mem["sceneBytes"] / 1e9, 3), "path_state_queues_images": round(mem["rendererBytes"] / 1e9, 3)},
            "bytes_per_path_slot": round(mem["pathStateBytes"] / max(1, mem["pathSlots"]), 1),
            "node_visits_per_secondary_ray": round(per_frame["nodesClosest"] / max(1.0, per_frame["segments"] - per_frame["cameraPaths"]), 2),
            "triangle_tests_per_secondary_ray": round(per_frame["trisClosest"] / max(1.0, per_frame["segments"] - per_frame["cameraPaths"]), 2),
        }
        if args.workload == NORTH_STAR["workload"] and not (args.width or args.height):
            result["north_star"] = dict(NORTH_STAR, workload=result["config"]["workload"], value=result["value"], unit="Msamples/s", n_gpus=world,
                                        value_per_gpu=round(value / world, 3), frac_of_needed_per_gpu=round(value / world / NORTH_STAR["needs_per_gpu_Msamples_s"], 3))
        if world == 1 and args.alpha_cut > 0 and alpha_cut_dropped > 0 and not args.no_uncut:
            # The same timed region on the geometry AS LOADED (no load-time alpha cut: a preprocessing step the reference does not perform on an asset without
            # opacity micro-maps) -- reported next to `value`, which is measured on the cut geometry
            tracer.close()
            try:
                result["value_uncut_geometry"] = uncut_value(args, w, W, H, F, frames_step, local_rank, hdr, frame_info, params, min(args.steps, 5))
            except Exception as e:  # noqa: BLE001
                result["value_uncut_geometry"] = {"error": f"{type(e).__name__}: {e}"}
        if not args.no_cpu_baseline and world == 1:
            tracer.close()  # (its path state is not needed any more; the parity leg's tracer may want as much again)
            try:
                result["cpu_baseline"], result["parity"] = cpu_baseline_and_parity(scene, w, W, H, F, args.cpu_seconds, lambda: make_tracer(False, partition=False), params,
                                                                                   args.parity_spp)
            except Exception as e:  # noqa: BLE001  (the measured line stands even if the checker's leg cannot run)
                result["cpu_baseline"] = result["parity"] = {"error": f"{type(e).__name__}: {e}"}
        # Next to the default line (the Sponza-class atrium, configs[2]), on the same GPU: every other configuration of BASELINE.json --
        # helmet (configs[1]) and the same at 3840x2160, street (configs[3], 4K), glass + denoise (configs[4]) -- each scene with its own CPU
        # baseline + full-size parity leg.
        default_run = args.workload == NORTH_STAR["workload"] and not (args.width or args.height or args.denoise)
        also = args.also if args.also is not None else (ALSO_DEFAULT if default_run else "none")
        if also != "none" and world == 1:
            tracer.close()  # (its ~80 GB of path state are not needed any more)
            result["also"] = {}
            for name in also.split(","):
                wl, aw, ah, den, par = ALSO_LINES[name]
                try:  # (a secondary line that fails -- e.g. out of device memory next to another tenant -- must not take the headline with it)
                    result["also"][name] = secondary_line(wl, args, local_rank, width=aw, height=ah, steps=5, parity=par and not args.no_cpu_baseline, denoise=den)
                except Exception as e:  # noqa: BLE001
                    result["also"][name] = {"error": f"{type(e).__name__}: {e}"}
                    import gc
                    gc.collect()
                    torch.cuda.empty_cache()
        emit(result)
    tracer.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
