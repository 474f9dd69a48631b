"""An INDEPENDENT check of the scene front end.  Every parity test feeds the HIP renderer and the CPU oracle from the same libmi_host.so
tables, so a loader bug is invisible to them (round-3 review).  Here the same files are decoded by a reader that shares nothing with
csrc/host/gltf_scene.cpp -- json + numpy, written from the glTF 2.0 specification (accessors with byteStride / normalisation,
node hierarchy with TRS or matrix, EXT_mesh_gpu_instancing, core material + the KHR factor extensions) -- and compared with
mi_scene_desc(): vertex streams, index buffers, render-node matrices and their inverses, material ids and material factors.
Reference behaviour restated: one RenderNode per (node, primitive) in depth-first scene order (src/gltf_scene.cpp:2139-2165,
:2338-2429), GltfShadeMaterial defaults and factors (src/gltf_material_cache.cpp:33-260)."""
import json
import os
import struct

import numpy as np
import pytest

from vk_gltf_renderer_amd import pathtracer as ptmod
from vk_gltf_renderer_amd import scenegen

_COMP = {5120: np.int8, 5121: np.uint8, 5122: np.int16, 5123: np.uint16, 5125: np.uint32, 5126: np.float32}
_NUM = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT4": 16}


class MiniGltf:
    """glTF 2.0 / GLB reader: just enough of the specification to flatten a scene."""

    def __init__(self, path):
        raw = open(path, "rb").read()
        self.dir = os.path.dirname(path)
        self.glb_bin = None
        if raw[:4] == b"glTF":
            _, _, total = struct.unpack_from("<4sII", raw, 0)
            off = 12
            while off < total:
                n, kind = struct.unpack_from("<I4s", raw, off)
                chunk = raw[off + 8:off + 8 + n]
                if kind == b"JSON":
                    self.doc = json.loads(chunk.decode())
                elif kind == b"BIN\0":
                    self.glb_bin = chunk
                off += 8 + n
        else:
            self.doc = json.loads(raw.decode())
        self._buffers = {}

    def buffer(self, i):
        if i not in self._buffers:
            b = self.doc["buffers"][i]
            self._buffers[i] = open(os.path.join(self.dir, b["uri"]), "rb").read() if "uri" in b else self.glb_bin
        return self._buffers[i]

    def accessor(self, i):
        a = self.doc["accessors"][i]
        comp, n = np.dtype(_COMP[a["componentType"]]), _NUM[a["type"]]
        bv = self.doc["bufferViews"][a["bufferView"]]
        base = bv.get("byteOffset", 0) + a.get("byteOffset", 0)
        stride = bv.get("byteStride", 0) or comp.itemsize * n
        data = self.buffer(bv["buffer"])
        out = np.empty((a["count"], n), comp)
        for k in range(a["count"]):  # (explicit: strides and offsets as the specification words them)
            out[k] = np.frombuffer(data, comp, n, base + k * stride)
        if a.get("normalized"):
            info = np.iinfo(comp)
            out = np.maximum(out.astype(np.float64) / info.max, -1.0) if info.min < 0 else out.astype(np.float64) / info.max
        return out

    @staticmethod
    def local_matrix(node):
        if "matrix" in node:
            return np.array(node["matrix"], np.float64).reshape(4, 4).T  # column-major in the file
        t, r, s = node.get("translation", [0, 0, 0]), node.get("rotation", [0, 0, 0, 1]), node.get("scale", [1, 1, 1])
        return MiniGltf.trs(t, r, s)

    @staticmethod
    def trs(t, r, s):
        x, y, z, w = (float(v) for v in r)
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                      [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                      [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        M = np.eye(4)
        M[:3, :3] = R * np.asarray(s, np.float64)[None, :]
        M[:3, 3] = t
        return M

    def flatten(self):
        """[(world matrix, mesh index, primitive index)] in depth-first order of the default scene; instancing expanded in place."""
        out = []

        def visit(ni, parent):
            node = self.doc["nodes"][ni]
            world = parent @ self.local_matrix(node)
            if "mesh" in node:
                inst = node.get("extensions", {}).get("EXT_mesh_gpu_instancing")
                mats = [world]
                if inst:
                    at = inst["attributes"]
                    cnt = self.doc["accessors"][next(iter(at.values()))]["count"]
                    T = self.accessor(at["TRANSLATION"]) if "TRANSLATION" in at else np.zeros((cnt, 3))
                    R = self.accessor(at["ROTATION"]) if "ROTATION" in at else np.tile([0, 0, 0, 1.0], (cnt, 1))
                    S = self.accessor(at["SCALE"]) if "SCALE" in at else np.ones((cnt, 3))
                    mats = [world @ self.trs(T[k], R[k], S[k]) for k in range(cnt)]
                for pi in range(len(self.doc["meshes"][node["mesh"]]["primitives"])):  # primitive-major: every instance of a primitive, then the
                    for m in mats:                                                       # next primitive (src/gltf_scene.cpp:2345-2370)
                        out.append((m, node["mesh"], pi))
            for c in node.get("children", []):
                visit(c, world)

        for root in self.doc["scenes"][self.doc.get("scene", 0)]["nodes"]:
            visit(root, np.eye(4))
        return out


def _prim_streams(desc, prim_id):
    p = desc.renderPrimitives[prim_id]
    nt, nv = p.triangleCount, p.vertexCount
    grab = lambda ptr, n, dt=np.float32: np.ctypeslib.as_array(ptr, shape=(nv * n,)).reshape(nv, n).astype(dt) if bool(ptr) else None
    return {"indices": np.ctypeslib.as_array(p.indices, shape=(nt * 3,)).copy(), "POSITION": grab(p.positions, 3), "NORMAL": grab(p.normals, 3),
            "TEXCOORD_0": grab(p.texCoords0, 2), "TEXCOORD_1": grab(p.texCoords1, 2), "TANGENT": grab(p.tangents, 4)}


def _expect_material(m):
    """Core factors + the KHR factor extensions as the glTF specification (and src/gltf_material_cache.cpp) default them."""
    pbr, ext = m.get("pbrMetallicRoughness", {}), m.get("extensions", {})
    e = {"pbrBaseColorFactor": pbr.get("baseColorFactor", [1, 1, 1, 1]), "pbrMetallicFactor": pbr.get("metallicFactor", 1.0),
         "pbrRoughnessFactor": pbr.get("roughnessFactor", 1.0), "emissiveFactor": m.get("emissiveFactor", [0, 0, 0]),
         "alphaMode": {"OPAQUE": 0, "MASK": 1, "BLEND": 2}[m.get("alphaMode", "OPAQUE")], "alphaCutoff": m.get("alphaCutoff", 0.5),
         "doubleSided": int(bool(m.get("doubleSided", False))),
         "transmissionFactor": ext.get("KHR_materials_transmission", {}).get("transmissionFactor", 0.0),
         "ior": ext.get("KHR_materials_ior", {}).get("ior", 1.5),
         "clearcoatFactor": ext.get("KHR_materials_clearcoat", {}).get("clearcoatFactor", 0.0),
         "clearcoatRoughness": ext.get("KHR_materials_clearcoat", {}).get("clearcoatRoughnessFactor", 0.0),
         "sheenColorFactor": ext.get("KHR_materials_sheen", {}).get("sheenColorFactor", [0, 0, 0]),
         "iridescenceFactor": ext.get("KHR_materials_iridescence", {}).get("iridescenceFactor", 0.0),
         "thicknessFactor": ext.get("KHR_materials_volume", {}).get("thicknessFactor", 0.0),
         "unlit": int("KHR_materials_unlit" in ext)}
    if "KHR_materials_emissive_strength" in ext:
        e["emissiveFactor"] = [v * ext["KHR_materials_emissive_strength"].get("emissiveStrength", 1.0) for v in e["emissiveFactor"]]
    return e


def _check_scene(path):
    g = MiniGltf(path)
    sc = ptmod.Scene(path)
    d = sc.desc.contents
    flat = g.flatten()
    assert d.numRenderNodes == len(flat), (d.numRenderNodes, len(flat))
    materials = g.doc.get("materials", [])
    tris = 0
    for i, (world, mesh, pi) in enumerate(flat):
        rn = d.renderNodes[i]
        prim = g.doc["meshes"][mesh]["primitives"][pi]
        # ---- matrices (column-major float[16] like glm) and the inverse the hit shader uses for normals
        o2w = np.array(rn.objectToWorld[:], np.float64).reshape(4, 4).T
        w2o = np.array(rn.worldToObject[:], np.float64).reshape(4, 4).T
        scale = max(1.0, np.abs(world).max())
        assert np.allclose(o2w, world, atol=2e-6 * scale), (i, o2w, world)
        assert np.allclose(w2o @ world, np.eye(4), atol=3e-5), i
        # ---- material id (a primitive without a material gets the default appended by the loader)
        assert rn.materialID == prim.get("material", len(materials) if "material" not in prim else 0), i
        # ---- vertex streams, bit for bit where the file holds float32, and the index buffer widened to u32
        s = _prim_streams(d, rn.renderPrimID)
        at = prim["attributes"]
        pos = g.accessor(at["POSITION"])
        assert s["POSITION"].shape == pos.shape and np.array_equal(s["POSITION"], pos.astype(np.float32)), i
        idx = g.accessor(prim["indices"]).reshape(-1).astype(np.uint32) if "indices" in prim else np.arange(len(pos), dtype=np.uint32)
        assert np.array_equal(s["indices"], idx), i
        tris += len(idx) // 3
        for name in ("NORMAL", "TEXCOORD_0", "TEXCOORD_1"):
            if name in at:
                ref = g.accessor(at[name]).astype(np.float32)
                assert s[name] is not None and np.allclose(s[name], ref, atol=1e-7), (i, name)
        if "TANGENT" in at:
            assert np.allclose(s["TANGENT"], g.accessor(at["TANGENT"]).astype(np.float32), atol=1e-7), i
    assert sc.num_triangles == tris
    # ---- material table
    assert d.numMaterials >= max(1, len(materials))  # (+ the default material the loader appends for primitives without one)
    for k, m in enumerate(materials):
        mat = d.materials[k]
        for key, val in _expect_material(m).items():
            got = getattr(mat, key)
            got = list(got) if hasattr(got, "__len__") else got
            assert got == pytest.approx(val, rel=1e-6, abs=1e-7), (k, key, got, val)
    return len(flat), tris


def test_in_tree_assets_against_an_independent_reader(built, assets):
    assert _check_scene(os.path.join(assets, "Box.glb")) == (1, 12)
    assert _check_scene(os.path.join(assets, "shader_ball.gltf")) == (1, 9450)


def test_generated_scenes_against_an_independent_reader(built, tmp_path):
    """Hierarchies, EXT_mesh_gpu_instancing with TRANSLATION + ROTATION + SCALE, shared accessors, 130 materials (street class);
    alpha-MASK foliage (atrium class); transmission / volume / ior (glass class); the material zoo's extension factors."""
    n, t = _check_scene(scenegen.scene_street_class(str(tmp_path / "street.glb"), seed=3, detail=0.1, tex_size=16))
    assert n > 900 and t > 50000
    _check_scene(scenegen.scene_atrium_class(str(tmp_path / "atrium.glb"), seed=5, detail=0.2, tex_size=32))
    _check_scene(scenegen.scene_glass_class(str(tmp_path / "glass.glb"), seed=9, tess=12))
    for group in ("clearcoat", "sheen", "iridescence"):
        _check_scene(scenegen.scene_material_zoo(str(tmp_path / f"zoo_{group}.glb"), group))


def test_strided_and_normalised_accessors_against_an_independent_reader(built, tmp_path):
    """What the generator never writes: an interleaved vertex buffer (byteStride), normalised UNSIGNED_SHORT texture coordinates,
    UNSIGNED_BYTE indices, a node given by `matrix` under a TRS parent."""
    pos = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]], np.float32)
    nrm = np.tile(np.array([0, 0, 1], np.float32), (4, 1))
    inter = np.concatenate([pos, nrm], axis=1).astype(np.float32).tobytes()  # 24-byte stride
    uv16 = np.array([[0, 0], [65535, 0], [65535, 65535], [0, 65535]], np.uint16).tobytes()
    idx8 = np.array([0, 1, 2, 0, 2, 3], np.uint8).tobytes() + b"\0\0"
    blob = inter + uv16 + idx8
    doc = {"asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": [0]}],
           "nodes": [{"children": [1], "translation": [1, 2, 3], "rotation": [0, 0.38268343, 0, 0.92387953], "scale": [2, 2, 2]},
                     {"mesh": 0, "matrix": [1, 0, 0, 0, 0, 0, 1, 0, 0, -1, 0, 0, 0.5, 0, 0, 1]}],
           "meshes": [{"primitives": [{"attributes": {"POSITION": 0, "NORMAL": 1, "TEXCOORD_0": 2}, "indices": 3, "material": 0}]}],
           "materials": [{"pbrMetallicRoughness": {"baseColorFactor": [0.2, 0.4, 0.6, 1.0], "metallicFactor": 0.25, "roughnessFactor": 0.75}, "emissiveFactor": [0.1, 0.2, 0.3]}],
           "buffers": [{"byteLength": len(blob)}],
           "bufferViews": [{"buffer": 0, "byteOffset": 0, "byteLength": len(inter), "byteStride": 24, "target": 34962},
                           {"buffer": 0, "byteOffset": len(inter), "byteLength": len(uv16), "target": 34962},
                           {"buffer": 0, "byteOffset": len(inter) + len(uv16), "byteLength": 6, "target": 34963}],
           "accessors": [{"bufferView": 0, "byteOffset": 0, "componentType": 5126, "count": 4, "type": "VEC3", "min": [0, 0, 0], "max": [1, 1, 0]},
                         {"bufferView": 0, "byteOffset": 12, "componentType": 5126, "count": 4, "type": "VEC3"},
                         {"bufferView": 1, "componentType": 5123, "normalized": True, "count": 4, "type": "VEC2"},
                         {"bufferView": 2, "componentType": 5121, "count": 6, "type": "SCALAR"}]}
    js = json.dumps(doc).encode()
    js += b" " * ((4 - len(js) % 4) % 4)
    blob += b"\0" * ((4 - len(blob) % 4) % 4)
    path = str(tmp_path / "strided.glb")
    with open(path, "wb") as f:
        f.write(struct.pack("<4sII", b"glTF", 2, 12 + 8 + len(js) + 8 + len(blob)))
        f.write(struct.pack("<I4s", len(js), b"JSON") + js + struct.pack("<I4s", len(blob), b"BIN\0") + blob)
    assert _check_scene(path) == (1, 2)


def _world_of_node(g, target):
    found = {}

    def visit(ni, parent):
        world = parent @ MiniGltf.local_matrix(g.doc["nodes"][ni])
        found[ni] = world
        for c in g.doc["nodes"][ni].get("children", []):
            visit(c, world)

    for root in g.doc["scenes"][g.doc.get("scene", 0)]["nodes"]:
        visit(root, np.eye(4))
    return found[target]


def test_lights_cameras_and_texture_transforms_against_the_specifications(built, tmp_path):
    """KHR_lights_punctual (a light shines down its node's -Z; spot cone angles; range -> 1 / range; the reference's conversion
    src/gltf_scene_vk.cpp:1354-1392), the camera of a node (eye = origin of the node, looking down -Z, yfov), and KHR_texture_transform
    (uv' = offset + R(rotation) * (scale * uv), checked where the specification and the reference's packing agree: no rotation, or a
    uniform scale) -- decoded independently from the file and compared with the loader's tables."""
    b = scenegen.GlbBuilder()
    img = np.full((4, 4, 4), 200, np.uint8)
    tex = b.texture(b.image(img))
    transforms = [dict(offset=[0.25, -0.5], scale=[2.0, 3.0]), dict(offset=[0.1, 0.2], rotation=0.7, scale=[1.5, 1.5]), dict(rotation=-1.1)]
    mats = [b.material({"pbrMetallicRoughness": {"baseColorTexture": {"index": tex, "texCoord": k % 2, "extensions": {"KHR_texture_transform": t}}}})
            for k, t in enumerate(transforms)]
    b.ext_used.add("KHR_texture_transform")
    pos, nrm, uv, idx = scenegen.grid(1, 1, (1.0, 1.0), "y")
    for m in mats:
        b.node(mesh=b.mesh([b.primitive(pos, idx, nrm, uv, uv1=uv * 0.5, material=m)]))
    lights = [{"type": "spot", "color": [1.0, 0.5, 0.25], "intensity": 7.0, "range": 4.0, "spot": {"innerConeAngle": 0.2, "outerConeAngle": 0.6}},
              {"type": "point", "intensity": 3.0, "extras": {"radius": 0.05}}, {"type": "directional", "color": [0.9, 0.9, 1.0], "intensity": 2.5}]
    light_nodes = []
    parent = b.node(translation=[1.0, 2.0, 3.0], rotation=[0.0, 0.38268343, 0.0, 0.92387953], children=[])
    for k, li in enumerate(lights):
        n = b.node(root=False, extensions={"KHR_lights_punctual": {"light": b.light(li)}}, translation=[0.5 * k, 1.0, -0.25 * k],
                   rotation=[float(np.sin(0.3 + 0.2 * k)), 0.0, 0.0, float(np.cos(0.3 + 0.2 * k))])
        b.doc["nodes"][parent]["children"].append(n)
        light_nodes.append(n)
    cam_node = b.camera_node((2.0, 1.5, 4.0), (0.0, 0.25, 0.0), yfov=0.6)
    path = b.save(str(tmp_path / "lights.glb"))
    g = MiniGltf(path)
    sc = ptmod.Scene(path)
    d = sc.desc.contents
    # ---- lights
    assert d.numLights == 3
    for k, (li, node) in enumerate(zip(lights, light_nodes)):
        L = d.lights[k]
        world = _world_of_node(g, node)
        assert np.allclose(list(L.position), world[:3, 3], atol=1e-5)
        assert np.allclose(list(L.direction), (world @ np.array([0, 0, -1.0, 0]))[:3], atol=1e-5)  # down the node's -Z
        assert L.type == {"directional": 1, "spot": 2, "point": 3}[li["type"]]
        assert np.allclose(list(L.color), li.get("color", [1, 1, 1])) and L.intensity == pytest.approx(li["intensity"])
        if li["type"] == "spot":
            assert L.innerAngle == pytest.approx(0.2) and L.outerAngle == pytest.approx(0.6)
        if li["type"] != "directional":
            assert L.angularSizeOrInvRange == pytest.approx(1.0 / li["range"] if "range" in li else 0.0)
        assert L.radius == pytest.approx(li.get("extras", {}).get("radius", 0.0))
    # ---- camera: eye at the node's origin, looking down its -Z, vertical field of view
    cam = sc.camera(0)
    world = _world_of_node(g, cam_node)
    eye, fwd = np.array(list(cam.eye)), np.array(list(cam.center)) - np.array(list(cam.eye))
    assert np.allclose(eye, world[:3, 3], atol=1e-5)
    assert np.allclose(fwd / np.linalg.norm(fwd), (world @ np.array([0, 0, -1.0, 0]))[:3], atol=1e-4)
    assert cam.fovDegrees == pytest.approx(np.degrees(0.6), abs=1e-3)
    # ---- texture transforms: the packed float3x2 applied as the shader applies it, against the specification's offset + R * (scale * uv)
    uvs = np.random.default_rng(1).uniform(-1, 2, (50, 2))
    for k, t in enumerate(transforms):
        mat = d.materials[k]
        ti = d.textureInfos[mat.pbrBaseColorTexture]
        assert ti.index == tex and ti.texCoord == k % 2
        m = np.array(list(ti.uvTransform), np.float64)
        got = np.stack([uvs[:, 0] * m[0] + uvs[:, 1] * m[2] + m[4], uvs[:, 0] * m[1] + uvs[:, 1] * m[3] + m[5]], 1)
        r, (sx, sy), (ox, oy) = t.get("rotation", 0.0), t.get("scale", [1.0, 1.0]), t.get("offset", [0.0, 0.0])
        su, sv = uvs[:, 0] * sx, uvs[:, 1] * sy
        want = np.stack([np.cos(r) * su + np.sin(r) * sv + ox, -np.sin(r) * su + np.cos(r) * sv + oy], 1)
        assert np.allclose(got, want, atol=1e-6), k
