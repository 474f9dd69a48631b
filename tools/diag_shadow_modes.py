"""Diagnostics: the glass-class test scene through the oracle and through the three transmissive-shadow paths (see
tests/test_gpu_parity.py::test_transmissive_shadow_paths_agree_bit_for_bit); prints counters and image differences."""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import parity_util as pu
    s = pu.Setup(sys.argv[2], 160, 96, max_depth=12, hdr_path=os.path.join(ROOT, "assets", "std_env.hdr"))
    g = pu.render_gpu(s, 3)
    np.save(sys.argv[3], g["accum"])
    print({k: g["stats"][k] for k in ("segments", "surfaceHits", "shadowRays", "nodesShadow", "trisShadow")})
    sys.exit(0)
import parity_util as pu
from vk_gltf_renderer_amd import scenegen
path = scenegen.scene_glass_class("/tmp/diag_glass.glb", seed=3, tess=24)
s = pu.Setup(path, 160, 96, max_depth=12, hdr_path=os.path.join(ROOT, "assets", "std_env.hdr"))
o = pu.render_oracle(s, 3)
print("oracle", {k: o["stats"][k] for k in ("segments", "surfaceHits", "shadowRays")})
imgs = {}
for pool in ("", "0", "200"):
    env = dict(os.environ)
    if pool:
        env["MI_PT_DIAG_CAND_POOL"] = pool
    out = f"/tmp/diag_pool{pool or 'default'}.npy"
    r = subprocess.run([sys.executable, __file__, "child", path, out], env=env, capture_output=True, text=True)
    print("pool", pool or "default", r.stdout.strip().splitlines()[-1] if r.returncode == 0 else r.stderr[-800:])
    if r.returncode == 0:
        imgs[pool] = np.load(out)
        print("   vs oracle", pu.compare_images(o["accum"], imgs[pool]))
for a in imgs:
    for b in imgs:
        if a < b:
            d = np.abs(imgs[a] - imgs[b])
            print(f"pool {a or 'default'} vs {b}: max abs diff {d.max():.3e}, differing pixels {(d.max(-1) > 0).sum()}")
