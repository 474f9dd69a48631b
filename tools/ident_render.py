"""Renders three small scenes with the library MI_PT_LIB selects and saves the accumulators: two runs with different builds of
libmi_pt.so must give equal files when the builds differ in data movement only (tools/run_r03_*.sh A/B calls).
usage: python tools/ident_render.py <tag> [out_dir]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_util as pu  # noqa: E402
from vk_gltf_renderer_amd import scenegen  # noqa: E402

tag = sys.argv[1]
out = sys.argv[2] if len(sys.argv) > 2 else "/tmp/ident"
os.makedirs(out, exist_ok=True)
hdr = os.path.join(ROOT, "assets", "std_env.hdr")
paths = [scenegen.scene_helmet_class(os.path.join(out, "helmet.glb"), seed=7, tess=48, tex_size=256),
         scenegen.scene_material_zoo(os.path.join(out, "zoo.glb"), "texture_transform", tess=24),
         scenegen.scene_atrium_class(os.path.join(out, "atrium.glb"), seed=5, detail=0.2, tex_size=64),
         scenegen.scene_glass_class(os.path.join(out, "glass.glb"), seed=3, tess=16)]
for k, p in enumerate(paths):
    s = pu.Setup(p, 320, 192, max_depth=6, hdr_path=hdr if k != 2 else None)
    for F in (1, 64):
        g = pu.render_gpu(s, 64 if F == 64 else 3, in_flight=F)
        np.save(os.path.join(out, f"{tag}_{k}_{F}.npy"), g["accum"])
print("rendered", tag)
