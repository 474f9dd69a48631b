"""N > 1 path on CPU: two processes, torch.distributed gloo, interleaved tile partition + ONE reduce(sum) of the accumulator
(the exact pattern bench.py uses over RCCL), with the CPU oracle standing in for the per-rank renderer.  The reduced frame
must equal the single-process frame bit for bit (disjoint tiles: sum == gather)."""
import os
import socket
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["MI_ROOT"]); sys.path.insert(0, os.path.join(os.environ["MI_ROOT"], "tests"))
import parity_util as pu
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
s = pu.Setup(os.path.join(os.environ["MI_ROOT"], "assets", "Box.glb"), 96, 64, max_depth=3,
             hdr_path=os.path.join(os.environ["MI_ROOT"], "assets", "std_env.hdr"))
part = pu.render_oracle(s, 2, threads=2, tile=(rank, world, 16))["accum"]
t = torch.from_numpy(part.copy())
dist.barrier()
dist.reduce(t, dst=0, op=dist.ReduceOp.SUM)
if rank == 0:
    full = pu.render_oracle(s, 2, threads=2)["accum"]
    assert (part != 0).any() and not (part == full).all()
    assert (t.numpy() == full).all(), "reduced tiles differ from the single-process frame"
    np.save(os.environ["MI_OUT"], t.numpy())
dist.barrier()
dist.destroy_process_group()
'''


def test_two_rank_tile_farm_with_gloo_reduce(built, tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    out = tmp_path / "reduced.npy"
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MI_ROOT=ROOT, MI_OUT=str(out),
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    img = np.load(out)
    assert img.shape == (64, 96, 4) and np.isfinite(img).all() and img[..., :3].max() > 0
