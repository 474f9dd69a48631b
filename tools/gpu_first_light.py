#!/usr/bin/env python3
"""First-light check on a GPU box: render in-tree scenes with the oracle and the HIP tracer, print parity metrics."""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import parity_util as pu

out_dir = os.path.join(ROOT, "gpurun_out")
os.makedirs(out_dir, exist_ok=True)
cases = [
    ("box_sky", dict(scene_path=os.path.join(ROOT, "assets/Box.glb"), width=256, height=256, max_depth=4), 16),
    ("box_hdr", dict(scene_path=os.path.join(ROOT, "assets/Box.glb"), width=256, height=256, max_depth=4, hdr_path=os.path.join(ROOT, "assets/std_env.hdr")), 16),
    ("ball_hdr", dict(scene_path=os.path.join(ROOT, "assets/shader_ball.gltf"), width=320, height=240, max_depth=5, hdr_path=os.path.join(ROOT, "assets/std_env.hdr")), 4),
]
for name, kw, frames in cases:
    s = pu.Setup(**kw)
    t0 = time.time(); g = pu.render_gpu(s, frames); t1 = time.time()
    o = pu.render_oracle(s, frames); t2 = time.time()
    m = pu.compare_images(o["accum"], g["accum"])
    m["gpu_s"] = t1 - t0; m["cpu_s"] = t2 - t1
    m["sel_equal"] = float((o["selection"] == g["selection"]).mean())
    m["depth_max_abs"] = float(np.abs(o["depth"] - g["depth"]).max())
    m["oracle_stats"] = {k: o["stats"][k] for k in ("cameraPaths", "segments", "shadowRays", "textureTaps")}
    m["gpu_stats"] = {k: g["stats"][k] for k in ("cameraPaths", "segments", "shadowRays", "textureTaps", "bvhNodeCount", "bvhTriangleCount")}
    print(name, json.dumps(m))
    np.save(os.path.join(out_dir, f"{name}_gpu.npy"), g["accum"]); np.save(os.path.join(out_dir, f"{name}_cpu.npy"), o["accum"])
