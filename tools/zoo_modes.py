"""Renders one material-zoo group in the two modes test_material_zoo compares (sequential frames on the 8-wide BVH; 3 frames in
flight on the BVH2) with the library MI_PT_LIB selects and saves both accumulators.  usage: python tools/zoo_modes.py <tag> <group>"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_util as pu  # noqa: E402
from vk_gltf_renderer_amd import scenegen  # noqa: E402

tag, group = sys.argv[1], sys.argv[2]
os.makedirs("/tmp/zoo", exist_ok=True)
hdr = os.path.join(ROOT, "assets", "std_env.hdr")
path = scenegen.scene_material_zoo(f"/tmp/zoo/{group}.glb", group)
s = pu.Setup(path, 192, 144, max_depth=7, hdr_path=hdr)
a = pu.render_gpu(s, 6)["accum"]
a2 = pu.render_gpu(s, 6)["accum"]
b = pu.render_gpu(s, 6, in_flight=3, bvh=1, collect_counters=False)["accum"]
c = pu.render_gpu(s, 6, in_flight=3, bvh=0, collect_counters=False)["accum"]
d = pu.render_gpu(s, 6, in_flight=1, bvh=1, collect_counters=False)["accum"]
np.save(f"/tmp/zoo/{tag}_{group}_a.npy", a)
def cmp(n, x, y):
    df = np.abs(x - y)
    print(f"ZOO {tag} {group} {n}: equal {bool(np.array_equal(x, y))} differing pixels {int((df.max(axis=-1) > 0).sum())} max abs {float(df.max()):.3g}")
cmp("seq8 vs seq8 again", a, a2)
cmp("seq8 vs f3-bvh2", a, b)
cmp("seq8 vs f3-bvh8", a, c)
cmp("seq8 vs seq-bvh2", a, d)
