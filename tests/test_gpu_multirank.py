"""N > 1 path with the HIP renderer itself: two processes share the one GPU of the test box (torch.distributed gloo on the host
side, the data path is the same tile partition + reduce(sum) of the RGBA32F accumulator that bench.py issues over RCCL).  The
reduced frame must equal the single-rank frame bit for bit: the partition only decides who renders a pixel, never its value.
Also runs bench.py --gpus 2 end to end in the same shared-GPU mode (rendezvous, per-step reduce, max-over-ranks timing, JSON line)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["MI_ROOT"]); sys.path.insert(0, os.path.join(os.environ["MI_ROOT"], "tests"))
import parity_util as pu
from vk_gltf_renderer_amd import scenegen
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
path = os.path.join(os.environ["MI_TMP"], f"zoo_{rank}.glb")
scenegen.scene_material_zoo(path, "clearcoat")   # same seeded bytes on every rank
s = pu.Setup(path, 200, 136, max_depth=5, hdr_path=os.path.join(os.environ["MI_ROOT"], "assets", "std_env.hdr"))
part = pu.render_gpu(s, 5, collect_counters=False, tile=(rank, world, 32), in_flight=3)["accum"]
t = torch.from_numpy(part.copy())
dist.barrier()
dist.reduce(t, dst=0, op=dist.ReduceOp.SUM)
if rank == 0:
    full = pu.render_gpu(s, 5, collect_counters=False)["accum"]
    assert (part != 0).any() and not (part == full).all()
    assert (t.numpy() == full).all(), "reduced tiles differ from the single-rank frame"
    np.save(os.environ["MI_OUT"], t.numpy())
dist.barrier()
# the same with 64 frames in flight: pixel-major path slots over the rank's own tiles, camera-ray packets of one pixel's samples
# (interval node test) -- the layout bench.py --gpus N runs with
part64 = pu.render_gpu(s, 64, collect_counters=False, tile=(rank, world, 32), in_flight=64)["accum"]
t64 = torch.from_numpy(part64.copy())
dist.barrier()
dist.reduce(t64, dst=0, op=dist.ReduceOp.SUM)
if rank == 0:
    full64 = pu.render_gpu(s, 64, collect_counters=False, in_flight=64)["accum"]
    assert (t64.numpy() == full64).all(), "64 frames in flight: reduced tiles differ from the single-rank frame"
dist.barrier()
dist.destroy_process_group()
'''


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_ranks_of_the_hip_renderer_reduce_to_the_single_rank_frame(built, tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    out = tmp_path / "reduced.npy"
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MI_ROOT=ROOT, MI_OUT=str(out), MI_TMP=str(tmp_path),
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    img = np.load(out)
    assert img.shape == (136, 200, 4) and np.isfinite(img).all() and img[..., :3].max() > 0


@pytest.mark.parametrize("scaling", ["strong", "weak", None])
def test_bench_two_ranks_on_one_gpu(built, scaling):
    """bench.py's N = 2 code path, started the way the driver may start it -- plain `python bench.py --gpus 2`, no launcher: bench.py
    spawns its own ranks (functional check; the numbers of a shared GPU mean nothing).  Strong scaling (the default): a step stays the
    configuration's 8 frames, all of them in flight (4 x 2 ranks); weak: 8 x 2 frames per step."""
    env = dict(os.environ, BENCH_SHARE_GPU="1", BENCH_DIST_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--workload", "box", "--frames-per-step", "8", "--in-flight", "4"]
    if scaling:
        cmd += ["--scaling", scaling]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 2 and j["config"]["world_size_reported_by_backend"] == 2 and j["value"] > 0
    assert j["scaling"] == (scaling or "strong")
    assert j["config"]["spp_per_step"] == (16 if scaling == "weak" else 8) and j["config"]["frames_in_flight"] == 8
    assert j["config"]["path_slots_per_gpu_in_frames"] == 4.0
    assert j["config"]["reduce"].startswith("one RCCL reduce")


def test_bench_refuses_more_gpus_than_visible(built):
    """`bench.py --gpus N` must not print an N-GPU line from fewer devices (round-3 review: it silently ran one)."""
    import torch
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "BENCH_SHARE_GPU")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0", "--workload", "box"],
                       env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode != 0 and "refusing" in (r.stdout + r.stderr)
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_bench_denoise_two_ranks_equals_one_rank(built, tmp_path):
    """configs[4]'s denoise pass in bench.py: with 2 ranks the guides and the depth travel with the accumulator in the one reduce and
    rank 0 denoises the reduced frame -- the same accumulator and the same denoised image, bit for bit, as the single-rank run of
    the same frames (tiles are disjoint: sum == gather)."""
    outs = []
    for world in (1, 2):
        port = _free_port()
        out = tmp_path / f"dump{world}.npz"
        env = dict(os.environ, BENCH_SHARE_GPU="1", BENCH_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), BENCH_DUMP=str(out))
        # (weak: 4 frames x 2 ranks per step; the strong-mode default renders the same 8 frames from `--frames-per-step 8` at either world size)
        args = [os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "1", "--warmup", "0", "--workload", "box", "--frames-per-step", str(8 // world),
                "--in-flight", str(4 // world), "--denoise", "--no-cpu-baseline", "--scaling", "weak"]
        cmd = [sys.executable] + args if world == 1 else [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                                                          "--master-port", str(port)] + args
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        assert j["config"]["denoise"].startswith("variance-guided") and j["config"]["spp_per_step"] == 8
        outs.append(np.load(out))
    assert np.array_equal(outs[0]["accum"], outs[1]["accum"])
    assert np.array_equal(outs[0]["denoised"], outs[1]["denoised"])
    assert np.abs(outs[0]["denoised"][..., :3] - outs[0]["accum"][..., :3]).max() > 0


def test_bench_strong_scaling_two_ranks_equals_one_rank(built, tmp_path):
    """--scaling strong: the SAME 8 frames whether one rank renders them or two ranks render half the tiles each -- the reduced accumulator is the single-rank
    one bit for bit (tiles are disjoint, seeds are a function of pixel and frame), and the step's sample count does not grow with the world size."""
    outs = []
    for world in (1, 2):
        port = _free_port()
        out = tmp_path / f"strong{world}.npz"
        env = dict(os.environ, BENCH_SHARE_GPU="1", BENCH_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), BENCH_DUMP=str(out))
        args = [os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "1", "--warmup", "0", "--workload", "box", "--frames-per-step", "8", "--in-flight", "4",
                "--no-cpu-baseline", "--scaling", "strong"]
        cmd = [sys.executable] + args if world == 1 else [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                                                          "--master-port", str(port)] + args
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        assert j["scaling"] == "strong" and j["config"]["spp_per_step"] == 8 and j["config"]["frames_in_flight"] == 4 * world
        outs.append(np.load(out))
    assert np.array_equal(outs[0]["accum"], outs[1]["accum"])
