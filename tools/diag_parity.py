"""Diagnostics (GPU box): render a named parity case with the oracle and the HIP tracer and dump both images + a summary of where
they differ into gpurun_out/ (npz), for offline analysis.  Usage: python tools/diag_parity.py <case> [frames] [spp_per_frame]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_util as pu  # noqa: E402
from vk_gltf_renderer_amd import scenegen  # noqa: E402

HDR = os.path.join(ROOT, "assets", "std_env.hdr")
case = sys.argv[1]
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 6
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 1
tmp = "/tmp/diag"
os.makedirs(tmp, exist_ok=True)
if case == "dof":
    path = scenegen.scene_material_zoo(tmp + "/cc.glb", "clearcoat")
    def dof(p):
        p.aperture, p.focalDistance = 0.12, 6.2
    s = pu.Setup(path, 160, 120, max_depth=5, hdr_path=HDR, params_edit=dof, spp_per_frame=spp)
elif case == "nodof":
    path = scenegen.scene_material_zoo(tmp + "/cc.glb", "clearcoat")
    s = pu.Setup(path, 160, 120, max_depth=5, hdr_path=HDR, spp_per_frame=spp)
elif case == "ortho":
    path = scenegen.scene_material_zoo(tmp + "/ortho.glb", "specular", camera="ortho")
    s = pu.Setup(path, 160, 120, max_depth=5, hdr_path=HDR, spp_per_frame=spp)
elif case == "glass":
    path = scenegen.scene_glass_class(tmp + "/glass.glb", seed=3, tess=16)
    s = pu.Setup(path, 96, 64, max_depth=12, hdr_path=HDR, spp_per_frame=spp)
else:
    raise SystemExit("unknown case")
o, g = pu.render_oracle(s, frames), pu.render_gpu(s, frames)
m = pu.compare_images(o["accum"], g["accum"])
print(case, frames, spp, m)
print("stats oracle", {k: o["stats"][k] for k in ("segments", "shadowRays", "textureTaps")}, "gpu", {k: g["stats"][k] for k in ("segments", "shadowRays", "textureTaps")})
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", f"diag_{case}_{frames}x{spp}.npz"), o=o["accum"], g=g["accum"], od=o["depth"], gd=g["depth"])
