"""Flattens a bench workload (after the same alpha cut bench.py applies) to world-space triangles for tools/lab/bvh_lab.cpp.
usage: python tools/lab/dump_tris.py <workload> <out.bin> [alpha_cut]
File: int32 n, then n x (9 float32 world-space vertices), then n x uint8 alpha flag (1 = alpha-tested material)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from vk_gltf_renderer_amd import pathtracer as ptmod  # noqa: E402

name, out = sys.argv[1], sys.argv[2]
cut = int(sys.argv[3]) if len(sys.argv) > 3 else bench.ALPHA_CUT_DEFAULT
scene = ptmod.Scene(bench.scene_path(name, 0))
if cut > 0:
    scene.cut_alpha(cut)
d = scene.desc.contents
tris, flags = [], []
cam = scene.camera(0)
for n in range(d.numRenderNodes):
    rn = d.renderNodes[n]
    if d.renderNodeVisible and not d.renderNodeVisible[n]:
        continue
    rp = d.renderPrimitives[rn.renderPrimID]
    nv = rp.vertexCount
    if rp.triangleCount == 0:
        continue
    idx = np.ctypeslib.as_array(rp.indices, shape=(rp.triangleCount * 3,)).reshape(-1, 3)
    pos = np.ctypeslib.as_array(rp.positions, shape=(nv * 3,)).reshape(-1, 3)
    M = np.array(list(rn.objectToWorld), np.float32).reshape(4, 4)  # column-major like glm
    w = pos @ M[:3, :3] + M[3, :3]
    tris.append(w[idx].reshape(-1, 9).astype(np.float32))
    mat = d.materials[max(0, rn.materialID)]
    flags.append(np.full(len(idx), 1 if mat.alphaMode != 0 else 0, np.uint8))
T, F = np.concatenate(tris), np.concatenate(flags)
with open(out, "wb") as f:
    f.write(np.int32(len(T)).tobytes())
    f.write(T.tobytes())
    f.write(F.tobytes())
    f.write(np.array(list(cam.eye) + list(cam.center), np.float32).tobytes())
print(name, "triangles", len(T), "alpha", int(F.sum()), "eye", list(cam.eye))
