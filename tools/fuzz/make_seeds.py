"""Seed GLBs for tools/fuzz/fuzz_host.py: every image container the loader decodes, and small generated scenes."""
import sys, os, io, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import test_host_loader as T
from vk_gltf_renderer_amd import scenegen
from PIL import Image
d=sys.argv[1]; os.makedirs(d, exist_ok=True)
rng = np.random.default_rng(3)
rgba = rng.integers(0, 255, (16, 16, 4), dtype=np.uint8)
def scene(name, data, mime, ext=None):
    b = scenegen.GlbBuilder()
    img = b.image_bytes(data, mime)
    tex = {"source": img} if ext is None else {"extensions": {ext: {"source": img}}}
    b.doc.setdefault("textures", []).append(tex)
    if ext: b.ext_used.add(ext)
    m = b.material({"pbrMetallicRoughness": {"baseColorTexture": {"index": 0}}})
    pos, nrm, uv, idx = scenegen.grid(1, 1)
    b.node(mesh=b.mesh([b.primitive(pos, idx, nrm, uv, material=m)]))
    b.save(os.path.join(d, name))
# jpeg baseline + progressive
for prog in (False, True):
    buf = io.BytesIO(); Image.fromarray(rgba[..., :3]).save(buf, "JPEG", quality=80, progressive=prog)
    scene(f"jpeg{int(prog)}.glb", buf.getvalue(), "image/jpeg")
buf = io.BytesIO(); Image.fromarray(rgba[..., :3]).convert("CMYK").save(buf, "JPEG", quality=80)
scene("jpeg_cmyk.glb", buf.getvalue(), "image/jpeg")
# png variants
for mode in ("RGBA", "L", "P"):
    buf = io.BytesIO(); Image.fromarray(rgba).convert(mode).save(buf, "PNG")
    scene(f"png_{mode}.glb", buf.getvalue(), "image/png")
# dds bc1 / uncompressed, ktx2 raw / zlib, webp
scene("dds_bc1.glb", T._dds(16, 16, bytes(rng.integers(0, 255, 16 * 8, dtype=np.uint8)), fourcc=b"DXT1"), "image/vnd-ms.dds", "MSFT_texture_dds")
scene("dds_bc7.glb", T._dds(16, 16, bytes(rng.integers(0, 255, 16 * 16, dtype=np.uint8)), fourcc=b"DX10", dxgi=98), "image/vnd-ms.dds", "MSFT_texture_dds")
scene("ktx2_raw.glb", T._ktx2(16, 16, 37, rgba.tobytes()), "image/ktx2", "KHR_texture_basisu")
scene("ktx2_zlib.glb", T._ktx2(16, 16, 37, zlib.compress(rgba.tobytes()), scheme=3), "image/ktx2", "KHR_texture_basisu")
buf = io.BytesIO(); Image.fromarray(rgba).save(buf, "WEBP", lossless=True)
scene("webp.glb", buf.getvalue(), "image/webp", "EXT_texture_webp")
scenegen.scene_animated(d + '/anim.glb')
scenegen.scene_atrium_class(d + '/atrium.glb', seed=5, detail=0.05, tex_size=16)
scenegen.scene_material_zoo(d + '/zoo.glb', 'texture_transform', tess=6, tex_size=8)
scenegen.scene_helmet_class(d + '/helmet.glb', seed=1, tess=8, tex_size=16)
# geometry that exists only as EXT / KHR_meshopt_compression streams (attributes, triangles with both codec versions, the octahedral filter)
import test_meshopt as M
def meshopt_scene():
    b = scenegen.GlbBuilder()
    for k, (nx, ny) in enumerate(((5, 4), (20, 17))):
        pos, nrm, uv, idx = scenegen.grid(nx, ny, (2.0, 1.5), "y")
        b.node(mesh=b.mesh([b.primitive(pos, idx, np.broadcast_to(nrm, pos.shape), uv, material=b.material({}))]), translation=[3.0 * k, 0, 0])
    return b
M._pack_meshopt(meshopt_scene(), d + '/meshopt_v0.glb', "EXT_meshopt_compression", 0)
M._pack_meshopt(meshopt_scene(), d + '/meshopt_v1.glb', "KHR_meshopt_compression", 1, vertex_version=1)
M._pack_meshopt(meshopt_scene(), d + '/meshopt_oct.glb', "EXT_meshopt_compression", 1, oct_normals=True)
print(sorted(os.listdir(d)))
