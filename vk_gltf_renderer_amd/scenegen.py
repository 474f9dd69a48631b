"""Seeded procedural glTF (.glb) scenes.

The benchmark assets BASELINE.json names (DamagedHelmet, Sponza, BistroExterior, TransmissionTest, DragonDispersion)
are not available offline, so every config other than Box.glb uses a *synthetic stand-in of the same class* written by
this module as a real .glb file: the CPU oracle and the HIP tracer then load identical bytes through the same front end.
All generators take an explicit integer seed; numpy's PCG64 stream makes the bytes reproducible.

Also used by the tests to build small analytic scenes (planes, spheres, material sweeps).
"""
import json
import struct
import zlib

import numpy as np

_COMPONENT = {np.dtype(np.float32): 5126, np.dtype(np.uint32): 5125, np.dtype(np.uint16): 5123, np.dtype(np.uint8): 5121,
              np.dtype(np.int16): 5122, np.dtype(np.int8): 5120}
_TYPE = {1: "SCALAR", 2: "VEC2", 3: "VEC3", 4: "VEC4", 16: "MAT4"}


def png_bytes(rgba):
    """Encode an (H, W, 4) uint8 array as a PNG (filter 0, zlib level 6)."""
    rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
    h, w, c = rgba.shape
    assert c == 4
    raw = np.concatenate([np.zeros((h, 1), np.uint8), rgba.reshape(h, w * 4)], axis=1).tobytes()

    def chunk(tag, data):
        body = tag + data
        return struct.pack(">I", len(data)) + body + struct.pack(">I", zlib.crc32(body) & 0xFFFFFFFF)

    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 6))
            + chunk(b"IEND", b""))


class GlbBuilder:
    def __init__(self, generator="vk_gltf_renderer_amd.scenegen"):
        self.doc = {"asset": {"version": "2.0", "generator": generator}, "scene": 0, "scenes": [{"nodes": []}], "nodes": [], "meshes": [],
                    "materials": [], "accessors": [], "bufferViews": [], "buffers": [{"byteLength": 0}]}
        self.bin = bytearray()
        self.ext_used = set()

    # ---- raw data ------------------------------------------------------------------------------------------------------
    def _view(self, data, target=None):
        while len(self.bin) % 4:
            self.bin.append(0)
        off = len(self.bin)
        self.bin += data
        bv = {"buffer": 0, "byteOffset": off, "byteLength": len(data)}
        if target:
            bv["target"] = target
        self.doc["bufferViews"].append(bv)
        return len(self.doc["bufferViews"]) - 1

    def accessor(self, arr, target=None, normalized=False, minmax=False):
        arr = np.ascontiguousarray(arr)
        ncomp = 1 if arr.ndim == 1 else arr.shape[1]
        acc = {"bufferView": self._view(arr.tobytes(), target), "componentType": _COMPONENT[arr.dtype], "count": int(arr.shape[0]),
               "type": _TYPE[ncomp]}
        if normalized:
            acc["normalized"] = True
        if minmax:
            acc["min"] = [float(v) for v in np.atleast_1d(arr.min(axis=0))]
            acc["max"] = [float(v) for v in np.atleast_1d(arr.max(axis=0))]
        self.doc["accessors"].append(acc)
        return len(self.doc["accessors"]) - 1

    # ---- images / textures -----------------------------------------------------------------------------------------------
    def image(self, rgba):
        self.doc.setdefault("images", []).append({"bufferView": self._view(png_bytes(rgba)), "mimeType": "image/png"})
        return len(self.doc["images"]) - 1

    def image_bytes(self, data, mime):
        """An already encoded image (e.g. JPEG bytes) embedded as is."""
        self.doc.setdefault("images", []).append({"bufferView": self._view(bytes(data)), "mimeType": mime})
        return len(self.doc["images"]) - 1

    def sampler(self, mag=9729, min_=9987, wrap_s=10497, wrap_t=10497):
        self.doc.setdefault("samplers", []).append({"magFilter": mag, "minFilter": min_, "wrapS": wrap_s, "wrapT": wrap_t})
        return len(self.doc["samplers"]) - 1

    def texture(self, image, sampler=None):
        t = {"source": image}
        if sampler is not None:
            t["sampler"] = sampler
        self.doc.setdefault("textures", []).append(t)
        return len(self.doc["textures"]) - 1

    # ---- materials / meshes / nodes ----------------------------------------------------------------------------------------
    def material(self, mat):
        for k in mat.get("extensions", {}):
            self.ext_used.add(k)
        self.doc["materials"].append(mat)
        return len(self.doc["materials"]) - 1

    def primitive(self, positions, indices=None, normals=None, uv0=None, uv1=None, colors=None, tangents=None, material=None):
        attrs = {"POSITION": self.accessor(np.asarray(positions, np.float32), 34962, minmax=True)}
        if normals is not None:
            attrs["NORMAL"] = self.accessor(np.asarray(normals, np.float32), 34962)
        if uv0 is not None:
            attrs["TEXCOORD_0"] = self.accessor(np.asarray(uv0, np.float32), 34962)
        if uv1 is not None:
            attrs["TEXCOORD_1"] = self.accessor(np.asarray(uv1, np.float32), 34962)
        if tangents is not None:
            attrs["TANGENT"] = self.accessor(np.asarray(tangents, np.float32), 34962)
        if colors is not None:
            colors = np.asarray(colors)
            attrs["COLOR_0"] = self.accessor(colors, 34962, normalized=colors.dtype != np.float32)
        prim = {"attributes": attrs, "mode": 4}
        if indices is not None:
            idx = np.asarray(indices).reshape(-1)
            idx = idx.astype(np.uint16) if idx.max(initial=0) < 65535 else idx.astype(np.uint32)
            prim["indices"] = self.accessor(idx, 34963)
        if material is not None:
            prim["material"] = material
        return prim

    def mesh(self, primitives):
        self.doc["meshes"].append({"primitives": primitives})
        return len(self.doc["meshes"]) - 1

    def node(self, root=True, **kw):
        for k in kw.get("extensions", {}):
            self.ext_used.add(k)
        self.doc["nodes"].append(kw)
        idx = len(self.doc["nodes"]) - 1
        if root:
            self.doc["scenes"][0]["nodes"].append(idx)
        return idx

    def camera(self, yfov, znear, zfar, aspect=16 / 9):
        self.doc.setdefault("cameras", []).append({"type": "perspective", "perspective": {"yfov": yfov, "znear": znear, "zfar": zfar, "aspectRatio": aspect}})
        return len(self.doc["cameras"]) - 1

    def camera_ortho(self, xmag, ymag, znear, zfar):
        self.doc.setdefault("cameras", []).append({"type": "orthographic", "orthographic": {"xmag": xmag, "ymag": ymag, "znear": znear, "zfar": zfar}})
        return len(self.doc["cameras"]) - 1

    def camera_node(self, eye, center, up=(0, 1, 0), yfov=0.7854, znear=0.05, zfar=1000.0, ortho=None):
        cam = self.camera(yfov, znear, zfar) if ortho is None else self.camera_ortho(ortho[0], ortho[1], znear, zfar)
        eye, center, up = (np.asarray(v, np.float64) for v in (eye, center, up))
        f = center - eye
        f /= np.linalg.norm(f)
        s = np.cross(f, up)
        s /= np.linalg.norm(s)
        u = np.cross(s, f)
        m = np.eye(4)
        m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = s, u, -f, eye
        return self.node(camera=cam, matrix=[float(v) for v in m.T.reshape(-1)],
                         extras={"camera::eye": eye.tolist(), "camera::center": center.tolist(), "camera::up": up.tolist()})

    def light(self, light):
        self.ext_used.add("KHR_lights_punctual")
        ext = self.doc.setdefault("extensions", {}).setdefault("KHR_lights_punctual", {"lights": []})
        ext["lights"].append(light)
        return len(ext["lights"]) - 1

    def animation(self, channels, name=None):
        """channels: [(node, path, times, values, interpolation)]; CUBICSPLINE values are (keys, 3, n): in-tangent, value, out-tangent."""
        anim = {"samplers": [], "channels": []}
        if name:
            anim["name"] = name
        for node, path, times, values, interp in channels:
            values = np.asarray(values, np.float32)
            out = values.reshape(-1, values.shape[-1])
            anim["samplers"].append({"input": self.accessor(np.asarray(times, np.float32), minmax=True), "output": self.accessor(out),
                                     "interpolation": interp})
            anim["channels"].append({"sampler": len(anim["samplers"]) - 1, "target": {"node": node, "path": path}})
        self.doc.setdefault("animations", []).append(anim)
        return len(self.doc["animations"]) - 1

    def save(self, path):
        doc = dict(self.doc)
        if self.ext_used:
            doc["extensionsUsed"] = sorted(self.ext_used)
        while len(self.bin) % 4:
            self.bin.append(0)
        doc["buffers"] = [{"byteLength": len(self.bin)}]
        js = json.dumps(doc, separators=(",", ":")).encode()
        js += b" " * ((4 - len(js) % 4) % 4)
        total = 12 + 8 + len(js) + 8 + len(self.bin)
        with open(path, "wb") as f:
            f.write(struct.pack("<4sII", b"glTF", 2, total))
            f.write(struct.pack("<I4s", len(js), b"JSON") + js)
            f.write(struct.pack("<I4s", len(self.bin), b"BIN\0") + bytes(self.bin))
        return path


# ---- geometry helpers -------------------------------------------------------------------------------------------------------
def grid(nx, ny, size=(1.0, 1.0), axis="y"):
    """A (nx x ny)-quad plane centred at the origin, normal +axis, uv in [0,1]^2."""
    u, v = np.meshgrid(np.linspace(0, 1, nx + 1), np.linspace(0, 1, ny + 1), indexing="xy")
    u, v = u.reshape(-1), v.reshape(-1)
    a, b = (u - 0.5) * size[0], (v - 0.5) * size[1]
    z = np.zeros_like(a)
    if axis == "y":
        pos, nrm = np.stack([a, z, -b], 1), np.array([0, 1, 0], np.float32)
    elif axis == "z":
        pos, nrm = np.stack([a, b, z], 1), np.array([0, 0, 1], np.float32)
    else:
        pos, nrm = np.stack([z, b, -a], 1), np.array([1, 0, 0], np.float32)
    i = (np.arange(ny)[:, None] * (nx + 1) + np.arange(nx)[None, :]).reshape(-1)
    idx = np.stack([i, i + 1, i + nx + 2, i, i + nx + 2, i + nx + 1], 1).reshape(-1, 3)
    return pos.astype(np.float32), np.tile(nrm, (pos.shape[0], 1)).astype(np.float32), np.stack([u, 1 - v], 1).astype(np.float32), idx.astype(np.uint32)


def uv_sphere(nu, nv, radius=1.0, displace=None):
    """Lat-long sphere with nu x nv quads; optional radial displacement callback f(dir)->scale."""
    th = np.linspace(0, np.pi, nv + 1)
    ph = np.linspace(0, 2 * np.pi, nu + 1)
    T, Pm = np.meshgrid(th, ph, indexing="ij")
    d = np.stack([np.sin(T) * np.cos(Pm), np.cos(T), np.sin(T) * np.sin(Pm)], -1).reshape(-1, 3)
    r = radius * (displace(d) if displace is not None else 1.0)
    pos = d * np.reshape(r, (-1, 1)) if np.ndim(r) else d * r
    uv = np.stack([Pm / (2 * np.pi), T / np.pi], -1).reshape(-1, 2)
    i = (np.arange(nv)[:, None] * (nu + 1) + np.arange(nu)[None, :]).reshape(-1)
    idx = np.stack([i, i + nu + 1, i + nu + 2, i, i + nu + 2, i + 1], 1).reshape(-1, 3)
    # outward-facing counter-clockwise winding
    idx = idx[:, [0, 2, 1]]
    nrm = d.copy()
    if displace is not None:
        nrm = _vertex_normals(pos, idx)
    return pos.astype(np.float32), nrm.astype(np.float32), uv.astype(np.float32), idx.astype(np.uint32)


def _vertex_normals(pos, idx):
    n = np.zeros_like(pos, dtype=np.float64)
    fn = np.cross(pos[idx[:, 1]] - pos[idx[:, 0]], pos[idx[:, 2]] - pos[idx[:, 0]])
    for k in range(3):
        np.add.at(n, idx[:, k], fn)
    ln = np.linalg.norm(n, axis=1, keepdims=True)
    return n / np.maximum(ln, 1e-20)


def box(size=(1, 1, 1), inward=False):
    """Axis-aligned box made of 6 quads (24 vertices)."""
    sx, sy, sz = (0.5 * s for s in size)
    faces = [((1, 0, 0), (0, 1, 0), (0, 0, 1)), ((-1, 0, 0), (0, 1, 0), (0, 0, -1)), ((0, 1, 0), (0, 0, 1), (1, 0, 0)),
             ((0, -1, 0), (0, 0, -1), (1, 0, 0)), ((0, 0, 1), (1, 0, 0), (0, 1, 0)), ((0, 0, -1), (-1, 0, 0), (0, 1, 0))]
    pos, nrm, uv, idx = [], [], [], []
    for n, a, b in faces:
        n, a, b = (np.array(v, np.float64) for v in (n, a, b))
        ext = np.array([sx, sy, sz])
        c = n * ext
        a, b = a * ext, b * ext
        base = len(pos)
        for (u, v) in ((-1, -1), (1, -1), (1, 1), (-1, 1)):
            pos.append(c + u * a + v * b)
            nrm.append(-n if inward else n)
            uv.append(((u + 1) / 2, (1 - v) / 2))
        quad = [[0, 1, 2], [0, 2, 3]]
        for t in quad:
            t = [base + k for k in t]
            # a x b == n for every face above, so (0,1,2) is CCW seen from outside
            idx.append(t[::-1] if inward else t)
    return np.array(pos, np.float32), np.array(nrm, np.float32), np.array(uv, np.float32), np.array(idx, np.uint32)


def value_noise(rng, size, octaves=5, channels=1):
    """Tileable value noise in [0,1], (size, size, channels)."""
    out = np.zeros((size, size, channels), np.float64)
    amp, total = 1.0, 0.0
    for o in range(octaves):
        n = min(4 << o, size)
        g = rng.random((n, n, channels))
        rep = size // n
        # bilinear upsample with wrap
        xs = (np.arange(size) + 0.5) / rep - 0.5
        x0 = np.floor(xs).astype(int)
        t = xs - x0
        t = t * t * (3 - 2 * t)
        x0m, x1m = x0 % n, (x0 + 1) % n
        rows = g[x0m] * (1 - t)[:, None, None] + g[x1m] * t[:, None, None]
        up = rows[:, x0m] * (1 - t)[None, :, None] + rows[:, x1m] * t[None, :, None]
        out += amp * up
        total += amp
        amp *= 0.5
    return out / total


# ---- analytic test scenes -------------------------------------------------------------------------------------------------
def lambert_material(color=(1, 1, 1), **extra):
    """Pure Lambertian: KHR_materials_specular.specularFactor = 0 removes the dielectric specular lobe."""
    m = {"pbrMetallicRoughness": {"baseColorFactor": [*color, 1.0], "metallicFactor": 0.0, "roughnessFactor": 1.0},
         "extensions": {"KHR_materials_specular": {"specularFactor": 0.0}}}
    m.update(extra)
    return m


def scene_sphere(path, material, nu=64, nv=32, eye=(0, 0, 3.0), radius=1.0):
    b = GlbBuilder()
    m = b.material(material)
    pos, nrm, uv, idx = uv_sphere(nu, nv, radius)
    b.node(mesh=b.mesh([b.primitive(pos, idx, nrm, uv, material=m)]))
    b.camera_node(eye, (0, 0, 0), yfov=0.8)
    return b.save(path)


def scene_plane_with_light(path, albedo=0.5, light=None, size=20.0):
    """Large Lambert quad in the xz plane, camera above looking down, one punctual light."""
    b = GlbBuilder()
    m = b.material(lambert_material((albedo, albedo, albedo)))
    pos, nrm, uv, idx = grid(1, 1, (size, size), "y")
    b.node(mesh=b.mesh([b.primitive(pos, idx, nrm, uv, material=m)]))
    b.camera_node((0, 4.0, 0.0), (0, 0, 0), up=(0, 0, -1), yfov=0.5)
    if light is not None:
        li = b.light(light["def"])
        b.node(extensions={"KHR_lights_punctual": {"light": li}}, **light.get("node", {}))
    return b.save(path)


# ---- benchmark-class stand-ins ----------------------------------------------------------------------------------------------
def scene_helmet_class(path, seed=1234, tess=272, tex_size=1024):
    """DamagedHelmet-class: ONE mesh (~2*tess*tess/2 triangles), ONE material with five textures (baseColor sRGB,
    metallicRoughness, normal, occlusion, emissive), procedural value noise.  tess=272 -> 73 984 triangles."""
    rng = np.random.default_rng(seed)
    bumps = rng.normal(size=(24, 3))
    bumps /= np.linalg.norm(bumps, axis=1, keepdims=True)
    amp = rng.uniform(0.03, 0.12, 24)

    def displace(d):
        s = np.ones(d.shape[0])
        for k in range(24):
            s += amp[k] * np.exp(-((1 - d @ bumps[k]) * 14.0))
        return s

    pos, nrm, uv, idx = uv_sphere(tess, tess // 2, 1.0, displace)
    b = GlbBuilder()
    smp = b.sampler()
    n1 = value_noise(rng, tex_size, 6, 3)
    n2 = value_noise(rng, tex_size, 5, 1)[..., 0]
    base = np.clip(0.25 + 0.7 * n1 * np.array([0.9, 0.75, 0.6]), 0, 1)
    rough = np.clip(0.15 + 0.8 * n2, 0, 1)
    metal = (value_noise(rng, tex_size, 3, 1)[..., 0] > 0.5).astype(np.float64)
    occl = np.clip(0.6 + 0.4 * value_noise(rng, tex_size, 4, 1)[..., 0], 0, 1)
    hgt = value_noise(rng, tex_size, 6, 1)[..., 0]
    gx = np.roll(hgt, -1, 1) - np.roll(hgt, 1, 1)
    gy = np.roll(hgt, -1, 0) - np.roll(hgt, 1, 0)
    nmap = np.stack([-gx * 8.0, -gy * 8.0, np.ones_like(gx)], -1)
    nmap /= np.linalg.norm(nmap, axis=-1, keepdims=True)
    emis = np.clip((value_noise(rng, tex_size, 3, 1)[..., 0] - 0.72) * 6.0, 0, 1)[..., None] * np.array([0.2, 0.6, 1.0])

    def tex(rgb, alpha=None):
        a = np.ones(rgb.shape[:2]) if alpha is None else alpha
        img = np.concatenate([rgb, a[..., None]], -1)
        return b.texture(b.image((np.clip(img, 0, 1) * 255 + 0.5).astype(np.uint8)), smp)

    t_base, t_mr = tex(base), tex(np.stack([occl, rough, metal], -1))
    t_nrm, t_emis = tex(nmap * 0.5 + 0.5), tex(emis)
    mat = b.material({"pbrMetallicRoughness": {"baseColorTexture": {"index": t_base}, "metallicRoughnessTexture": {"index": t_mr}},
                      "normalTexture": {"index": t_nrm, "scale": 1.0}, "occlusionTexture": {"index": t_mr, "strength": 1.0},
                      "emissiveTexture": {"index": t_emis}, "emissiveFactor": [1.0, 1.0, 1.0]})
    # tangents along +u
    d = pos / np.linalg.norm(pos, axis=1, keepdims=True)
    tan = np.stack([-d[:, 2], np.zeros(len(d)), d[:, 0]], 1)
    ln = np.linalg.norm(tan, axis=1, keepdims=True)
    tan = np.where(ln > 1e-6, tan / np.maximum(ln, 1e-6), np.array([1.0, 0, 0]))
    tan = np.concatenate([tan, np.ones((len(tan), 1))], 1).astype(np.float32)
    b.node(mesh=b.mesh([b.primitive(pos, idx, nrm, uv, tangents=tan, material=mat)]))
    b.camera_node((0.0, 0.3, 3.2), (0, 0, 0), yfov=0.7)
    return b.save(path)


def _beam(p0, p1, width, up=(0.0, 1.0, 0.0)):
    """A thin box (12 triangles, each as long as the beam) from p0 to p1 with a square cross-section of edge `width`."""
    p0, p1 = np.asarray(p0, np.float64), np.asarray(p1, np.float64)
    ax = p1 - p0
    ln = np.linalg.norm(ax)
    ax /= ln
    u = np.cross(ax, np.asarray(up, np.float64))
    if np.linalg.norm(u) < 1e-6:
        u = np.cross(ax, np.array([1.0, 0.0, 0.0]))
    u /= np.linalg.norm(u)
    v = np.cross(ax, u)
    pos, nrm, uv, idx = box((ln, width, width))
    R = np.stack([ax, u, v], 1)  # box x -> beam axis
    return (pos.astype(np.float64) @ R.T + (p0 + p1) / 2).astype(np.float32), (nrm.astype(np.float64) @ R.T).astype(np.float32), uv * np.array([ln, 1.0], np.float32), idx


def scene_atrium_class(path, seed=4321, detail=1.0, tex_size=512, sliver=False):
    """Sponza-class interior: a long two-storey hall with columns, arches, drapes and alpha-MASK foliage, ~25 materials in
    ~100 primitives, one directional light + sky.  detail=1.0 gives ~262k triangles.
    sliver=True: the triangle-size distribution SURVEY 8(d) asks for ("triangle sizes log-uniform") and what makes the real asset hard for a BVH:
    every wall / floor / ceiling is a handful of long strips or two hall-sized triangles, long thin beams, rails and diagonal ropes run between
    finely tessellated columns, and half of the clutter is four times finer than the rest -- triangle edges from ~5 mm to 36 m (3.9 decades) at the
    same triangle count (+-2 %) and the same materials, light and camera."""
    rng = np.random.default_rng(seed)
    b = GlbBuilder()
    smp = b.sampler()

    def tex(rgb, alpha=None):
        a = np.ones(rgb.shape[:2]) if alpha is None else alpha
        img = np.concatenate([rgb, a[..., None]], -1)
        return b.texture(b.image((np.clip(img, 0, 1) * 255 + 0.5).astype(np.uint8)), smp)

    mats = []
    for k in range(22):
        hue = rng.uniform(0.55, 0.95, 3) * rng.uniform(0.75, 1.0)  # sRGB; ~0.25 linear albedo like scanned stone/fabric
        n = value_noise(rng, tex_size, 5, 3)
        t = tex(np.clip(hue * (0.55 + 0.45 * n), 0, 1))
        m = {"pbrMetallicRoughness": {"baseColorTexture": {"index": t}, "metallicFactor": float(rng.random() < 0.15),
                                      "roughnessFactor": float(rng.uniform(0.25, 0.95))}}
        if k % 5 == 0:
            m["doubleSided"] = True
        mats.append(b.material(m))
    # foliage: alpha-masked, double sided
    leaf_n = value_noise(rng, tex_size, 4, 1)[..., 0]
    yy, xx = np.mgrid[0:tex_size, 0:tex_size] / tex_size - 0.5
    leaf_a = ((np.sqrt(xx * xx * 3 + yy * yy) + 0.15 * (leaf_n - 0.5)) < 0.42).astype(np.float64)
    leaf_rgb = np.clip(np.stack([0.15 + 0.2 * leaf_n, 0.45 + 0.4 * leaf_n, 0.1 + 0.1 * leaf_n], -1), 0, 1)
    t_leaf = tex(leaf_rgb, leaf_a)
    m_leaf = b.material({"pbrMetallicRoughness": {"baseColorTexture": {"index": t_leaf}, "metallicFactor": 0.0, "roughnessFactor": 0.7},
                         "alphaMode": "MASK", "alphaCutoff": 0.5, "doubleSided": True})
    m_drape = b.material({"pbrMetallicRoughness": {"baseColorFactor": [0.6, 0.08, 0.08, 1], "metallicFactor": 0.0, "roughnessFactor": 0.9},
                          "doubleSided": True})

    L, Wd, Hh = 36.0, 14.0, 12.0
    g = max(2, int(24 * detail))

    def add(prim_args, material, **node):
        b.node(mesh=b.mesh([b.primitive(*prim_args, material=material)]), **node)

    def quad_grid(nx, ny, size, axis, flip=False):
        pos, nrm, uv, idx = grid(nx, ny, size, axis)
        if flip:
            idx, nrm = idx[:, [0, 2, 1]], -nrm
        return pos, idx, nrm, uv * np.array([size[0] / 4, size[1] / 4], np.float32)

    # shell: floor, ceiling (with a skylight slot), walls
    if sliver:
        # hall-long strips and two-triangle walls (the shell loses ~10 k triangles; the finer half of the clutter below takes them)
        add(quad_grid(1, g, (L, Wd), "y"), mats[0])
        for s in (-1, 1):
            add(quad_grid(1, 3, (L, Wd * 0.32), "y", flip=True), mats[1], translation=[0, Hh, s * Wd * 0.34])
            add(quad_grid(1, 1, (L, Hh), "z", flip=(s > 0)), mats[2], translation=[0, Hh / 2, s * Wd / 2])
            add(quad_grid(1, 1, (Wd, Hh), "x", flip=(s > 0)), mats[3], translation=[s * L / 2, Hh / 2, 0])
            add(quad_grid(1, 2, (L, 2.6), "y"), mats[4], translation=[0, Hh * 0.5, s * (Wd / 2 - 1.3)])
        # beams between the columns, hand rails along the galleries, ropes across the hall: long thin boxes, most of them not axis aligned
        beams = []
        for s in (-1, 1):
            z = s * (Wd / 2 - 2.6)
            for i in range(9):
                x0, x1 = -L / 2 + (i + 0.5) * L / 10, -L / 2 + (i + 1.5) * L / 10
                for y in (Hh * 0.47, Hh * 0.97):
                    beams.append(_beam((x0, y, z), (x1, y, z), 0.22))
                beams.append(_beam((x0, Hh * 0.5, z), (x1, Hh * 0.97, z), 0.05))  # diagonal brace
            for y in (Hh * 0.5 + 0.9, Hh * 0.5 + 0.5):
                beams.append(_beam((-L / 2, y, s * (Wd / 2 - 2.55)), (L / 2, y, s * (Wd / 2 - 2.55)), 0.04))  # 36-m rails
        for k in range(120):
            a = np.array([rng.uniform(-L / 2 + 1, L / 2 - 1), rng.uniform(Hh * 0.55, Hh * 0.98), -(Wd / 2 - 2.6)])
            c = np.array([a[0] + rng.uniform(-6, 6), rng.uniform(Hh * 0.55, Hh * 0.98), Wd / 2 - 2.6])
            beams.append(_beam(a, c, float(np.exp(rng.uniform(np.log(0.005), np.log(0.03))))))
        bp = np.concatenate([bm[0] for bm in beams]); bn = np.concatenate([bm[1] for bm in beams]); bu = np.concatenate([bm[2] for bm in beams])
        bi = np.concatenate([bm[3] + 24 * k for k, bm in enumerate(beams)]).astype(np.uint32)
        add((bp, bi, bn, bu), mats[9])
    else:
      add(quad_grid(g * 3, g, (L, Wd), "y"), mats[0])
      for s in (-1, 1):
        add(quad_grid(g * 3, g // 2, (L, Wd * 0.32), "y", flip=True), mats[1], translation=[0, Hh, s * Wd * 0.34])
        add(quad_grid(g * 3, g, (L, Hh), "z", flip=(s > 0)), mats[2], translation=[0, Hh / 2, s * Wd / 2])
        add(quad_grid(g, g, (Wd, Hh), "x", flip=(s > 0)), mats[3], translation=[s * L / 2, Hh / 2, 0])
        # gallery floor of the upper storey
        add(quad_grid(g * 3, 4, (L, 2.6), "y"), mats[4], translation=[0, Hh * 0.5, s * (Wd / 2 - 1.3)])
    # columns (tessellated cylinders = stretched spheres) and arches (tori segments)
    ncol = 10
    cu, cv = max(8, int(40 * detail)), max(8, int(56 * detail))
    for i in range(ncol):
        x = -L / 2 + (i + 0.5) * L / ncol
        for s in (-1, 1):
            for storey in range(2):
                pos, nrm, uv, idx = uv_sphere(cu, cv, 1.0)
                pos = pos * np.array([0.42, Hh * 0.25, 0.42], np.float32)
                nn = nrm / np.array([0.42, Hh * 0.25, 0.42], np.float32)
                nn /= np.linalg.norm(nn, axis=1, keepdims=True)
                add((pos, idx, nn.astype(np.float32), uv * np.array([2, 6], np.float32)), mats[5 + (i + storey) % 6],
                    translation=[x, Hh * (0.25 + 0.5 * storey), s * (Wd / 2 - 2.6)])
    # drapes: wavy vertical sheets
    for i in range(6):
        x = -L / 2 + (i + 0.75) * L / 6
        nx, ny = max(6, int(48 * detail)), max(6, int(64 * detail))
        pos, nrm, uv, idx = grid(nx, ny, (3.0, 5.0), "z")
        phase = rng.uniform(0, 6.28)
        pos[:, 2] += (0.18 * np.sin(pos[:, 0] * 7.0 + phase) * (0.3 + (2.5 - pos[:, 1]) / 5.0)).astype(np.float32)
        nrm = _vertex_normals(pos.astype(np.float64), idx).astype(np.float32)
        add((pos, idx, nrm, uv), m_drape, translation=[x, Hh * 0.72, (-1) ** i * (Wd / 2 - 2.2)])
    # foliage: clusters of alpha-masked quads (~10 % of the triangles)
    nleaf = int(13000 * detail)
    for cl in range(8):
        c = np.array([-L / 2 + 6.0 + cl * (L - 9.0) / 8.0, 0.0, (-1) ** cl * (Wd / 2 - 4.2) + rng.uniform(-0.4, 0.4)])
        n = nleaf // 8
        centers = c + rng.normal(size=(n, 3)) * np.array([0.9, 0.7, 0.9]) + np.array([0, 1.6, 0])
        ax = rng.normal(size=(n, 3))
        ax /= np.linalg.norm(ax, axis=1, keepdims=True)
        bx = np.cross(ax, rng.normal(size=(n, 3)))
        bx /= np.linalg.norm(bx, axis=1, keepdims=True)
        sz = np.exp(rng.uniform(np.log(0.08), np.log(0.35), (n, 1)))
        quad = np.stack([centers - ax * sz - bx * sz, centers + ax * sz - bx * sz, centers + ax * sz + bx * sz, centers - ax * sz + bx * sz], 1)
        pos = quad.reshape(-1, 3)
        nrm = np.repeat(np.cross(ax, bx), 4, 0)
        uv = np.tile(np.array([[0, 1], [1, 1], [1, 0], [0, 0]], np.float32), (n, 1))
        base = np.arange(n)[:, None] * 4
        idx = np.concatenate([base + np.array([0, 1, 2]), base + np.array([0, 2, 3])], 1).reshape(-1, 3)
        add((pos.astype(np.float32), idx.astype(np.uint32), nrm.astype(np.float32), uv), m_leaf)
    # clutter: displaced blobs standing in for statues/vases, varied materials
    for k in range(16):
        amp = rng.uniform(0.05, 0.25)
        fr = rng.normal(size=(6, 3)) * 3.0

        def displace(d, amp=amp, fr=fr):
            return 1.0 + amp * np.sin(d @ fr.T).sum(1) / 3.0

        radius = rng.uniform(0.5, 1.1)
        fine = sliver and k % 2 == 1  # every other blob: a third of the size at 1.3 x the tessellation (edges ~4 mm .. 3 cm instead of 3 .. 9 cm)
        pos, nrm, uv, idx = uv_sphere(max(8, int(96 * detail * (1.3 if fine else 0.72 if sliver else 1.0))),
                                      max(6, int(48 * detail * (1.3 if fine else 0.72 if sliver else 1.0))), radius * (0.3 if fine else 1.0), displace)
        add((pos, idx, nrm, uv), mats[11 + k % 11], translation=[float(rng.uniform(-L / 2 + 2, L / 2 - 2)), float(rng.uniform(0.6, 1.2)) * (0.3 if fine else 1.0),
                                                                float(rng.uniform(-Wd / 2 + 3.4, Wd / 2 - 3.4))])
    li = b.light({"type": "directional", "intensity": 12.0, "color": [1.0, 0.96, 0.9]})
    # light direction = -Z of the node: tilt so it shines through the skylight slot
    ang = 0.35
    b.node(extensions={"KHR_lights_punctual": {"light": li}}, rotation=[-float(np.sin((np.pi / 2 - ang) / 2)), 0.0, 0.0, float(np.cos((np.pi / 2 - ang) / 2))])
    b.camera_node((-L / 2 + 2.5, 2.2, 0.6), (L / 2, 3.8, -0.4), yfov=1.0, znear=0.05, zfar=200.0)
    return b.save(path)


def scene_street_class(path, seed=777, detail=1.0, tex_size=256, sliver=False):
    """BistroExterior-class street: two rows of buildings along a street, every building an instance of
    one of 24 facade meshes (EXT_mesh_gpu_instancing: ~1000 render nodes from ~40 glTF nodes), street furniture and trees with
    alpha-MASK foliage, ~130 materials over 24 textures, sun + sky.  detail=1.27 gives ~2.8 M triangles.
    sliver=True: the same street with the triangle shapes of a modelled one -- the road is 240-m strips, every facade is tessellated 16 : 1 along one
    direction (ledges and pilasters: triangles 10 m long and centimetres wide), overhead cables sag across the street; same triangle count (-3 %)."""
    rng = np.random.default_rng(seed)
    b = GlbBuilder()
    b.ext_used.add("EXT_mesh_gpu_instancing")
    smp = b.sampler()

    def tex(rgb, alpha=None):
        a = np.ones(rgb.shape[:2]) if alpha is None else alpha
        img = np.concatenate([rgb, a[..., None]], -1)
        return b.texture(b.image((np.clip(img, 0, 1) * 255 + 0.5).astype(np.uint8)), smp)

    textures = []
    for k in range(23):
        hue = rng.uniform(0.45, 0.95, 3) * rng.uniform(0.7, 1.0)
        textures.append(tex(np.clip(hue * (0.5 + 0.5 * value_noise(rng, tex_size, 5, 3)), 0, 1)))
    mats = []
    for k in range(128):
        m = {"pbrMetallicRoughness": {"baseColorTexture": {"index": textures[k % len(textures)]},
                                      "baseColorFactor": [float(v) for v in rng.uniform(0.6, 1.0, 3)] + [1.0],
                                      "metallicFactor": float(rng.random() < 0.12), "roughnessFactor": float(rng.uniform(0.2, 0.95))}}
        if k % 7 == 0:
            m["doubleSided"] = True
        if k % 16 == 3:
            m["emissiveFactor"] = [float(v) for v in rng.uniform(0.0, 0.6, 3)]
        mats.append(b.material(m))
    leaf_n = value_noise(rng, tex_size, 4, 1)[..., 0]
    yy, xx = np.mgrid[0:tex_size, 0:tex_size] / tex_size - 0.5
    leaf_a = ((np.sqrt(xx * xx * 3 + yy * yy) + 0.15 * (leaf_n - 0.5)) < 0.42).astype(np.float64)
    t_leaf = tex(np.clip(np.stack([0.15 + 0.2 * leaf_n, 0.4 + 0.4 * leaf_n, 0.1 + 0.1 * leaf_n], -1), 0, 1), leaf_a)
    m_leaf = b.material({"pbrMetallicRoughness": {"baseColorTexture": {"index": t_leaf}, "metallicFactor": 0.0, "roughnessFactor": 0.7},
                         "alphaMode": "MASK", "alphaCutoff": 0.5, "doubleSided": True})
    m_road = b.material({"pbrMetallicRoughness": {"baseColorTexture": {"index": textures[0]}, "baseColorFactor": [0.35, 0.35, 0.37, 1.0],
                                                  "metallicFactor": 0.0, "roughnessFactor": 0.85}})

    def instanced(mesh, translations, rotations_y=None, scales=None):
        """One glTF node carrying an EXT_mesh_gpu_instancing attribute set."""
        attrs = {"TRANSLATION": b.accessor(np.asarray(translations, np.float32))}
        if rotations_y is not None:
            h = np.asarray(rotations_y, np.float64) / 2
            attrs["ROTATION"] = b.accessor(np.stack([np.zeros_like(h), np.sin(h), np.zeros_like(h), np.cos(h)], 1).astype(np.float32))
        if scales is not None:
            attrs["SCALE"] = b.accessor(np.asarray(scales, np.float32))
        b.node(mesh=mesh, extensions={"EXT_mesh_gpu_instancing": {"attributes": attrs}})

    L, Wd = 240.0, 18.0
    g = max(2, int(40 * detail))
    pos, nrm, uv, idx = grid(1 if sliver else g * 4, g, (L, Wd * 3), "y")
    b.node(mesh=b.mesh([b.primitive(pos, idx, nrm, uv * np.array([L / 4, Wd], np.float32), material=m_road)]))
    if sliver:  # overhead cables: chains of 12-m beams a centimetre or two thick, sagging, most of them diagonal across the street
        beams = []
        for k in range(60):
            x0, x1 = sorted(rng.uniform(-L / 2, L / 2, 2))
            x1 = min(x1, x0 + 60.0)
            z0, z1 = rng.choice([-1, 1]) * Wd / 2, rng.choice([-1, 1]) * Wd / 2
            y, w, n = rng.uniform(6.0, 9.0), float(np.exp(rng.uniform(np.log(0.008), np.log(0.03)))), 5
            t = np.linspace(0, 1, n + 1)
            pts = np.stack([x0 + (x1 - x0) * t, y - 1.2 * np.sin(np.pi * t), z0 + (z1 - z0) * t], 1)
            beams += [_beam(pts[i], pts[i + 1], w) for i in range(n)]
        bp = np.concatenate([bm[0] for bm in beams]); bn = np.concatenate([bm[1] for bm in beams]); bu = np.concatenate([bm[2] for bm in beams])
        bi = np.concatenate([bm[3] + 24 * k for k, bm in enumerate(beams)]).astype(np.uint32)
        b.node(mesh=b.mesh([b.primitive(bp, bi, bn, bu, material=mats[7])]))
    # 24 building types: a displaced facade slab with window bays (4 primitives of different materials each)
    fx, fy = max(4, int(28 * detail)), max(4, int(40 * detail))
    building_meshes = []
    for t in range(24):
        prims = []
        w, h, d = rng.uniform(8, 14), rng.uniform(10, 26), rng.uniform(8, 12)
        for face, (sz, axis, off, flip) in enumerate((((w, h), "z", (0, h / 2, d / 2), False), ((w, h), "z", (0, h / 2, -d / 2), True),
                                                      ((d, h), "x", (w / 2, h / 2, 0), False), ((d, h), "x", (-w / 2, h / 2, 0), True))):
            if sliver:  # 16 : 1 strips, alternately horizontal (ledges) and vertical (pilasters)
                pos, nrm, uv, idx = grid(fx * 4, max(1, fy // 4), sz, axis) if (t + face) % 2 == 0 else grid(max(1, fx // 4), fy * 4, sz, axis)
            else:
                pos, nrm, uv, idx = grid(fx, fy, sz, axis)
            bays = np.sin(uv[:, 0] * np.pi * rng.integers(3, 7)) * np.sin(uv[:, 1] * np.pi * rng.integers(4, 10))
            relief = (0.12 * np.clip(bays, 0, 1) + 0.03 * np.sin(uv[:, 1] * 90.0)).astype(np.float32)
            pos = pos + nrm * relief[:, None]
            if flip:
                idx = idx[:, [0, 2, 1]]
                pos = pos * np.array([-1, 1, -1], np.float32) if axis == "x" else pos * np.array([1, 1, -1], np.float32)
            n2 = _vertex_normals(pos.astype(np.float64), idx).astype(np.float32)
            prims.append(b.primitive(pos + np.array(off, np.float32), idx, n2,
                                     uv * np.array([sz[0] / 3, sz[1] / 3], np.float32), material=mats[(t * 5 + face) % 120]))
        pos, nrm, uv, idx = grid(4, 4, (w, d), "y")
        prims.append(b.primitive(pos + np.array([0, h, 0], np.float32), idx, nrm, uv, material=mats[(t * 5 + 4) % 120]))
        building_meshes.append((b.mesh(prims), w))
    # two rows of buildings, ~20 instances per type
    per_type = [[] for _ in building_meshes]
    for side in (-1, 1):
        x = -L / 2
        while x < L / 2:
            t = int(rng.integers(len(building_meshes)))
            w = building_meshes[t][1]
            per_type[t].append(((x + w / 2, 0.0, side * (Wd / 2 + 6.0)), 0.0 if side < 0 else np.pi))
            x += w + rng.uniform(0.2, 1.5)
        # a second row behind, seen through the gaps and above the roofs
        x = -L / 2
        while x < L / 2:
            t = int(rng.integers(len(building_meshes)))
            w = building_meshes[t][1]
            per_type[t].append(((x + w / 2, 0.0, side * (Wd / 2 + 22.0)), 0.0 if side < 0 else np.pi))
            x += w + rng.uniform(1.0, 4.0)
    for (mesh, _), inst in zip(building_meshes, per_type):
        if inst:
            instanced(mesh, [i[0] for i in inst], [i[1] for i in inst])
    # street furniture: lamp posts, bollards, planters (instanced displaced blobs)
    for k in range(8):
        fr = rng.normal(size=(5, 3)) * 3.0
        amp = rng.uniform(0.05, 0.3)

        def displace(d, amp=amp, fr=fr):
            return 1.0 + amp * np.sin(d @ fr.T).sum(1) / 3.0

        pos, nrm, uv, idx = uv_sphere(max(8, int(36 * detail)), max(6, int(20 * detail)), 1.0, displace)
        pos = pos * np.array([0.25, rng.uniform(0.5, 2.5), 0.25], np.float32)
        nrm = _vertex_normals(pos.astype(np.float64), idx).astype(np.float32)
        mesh = b.mesh([b.primitive(pos, idx, nrm, uv, material=mats[120 + k])])
        n = 60
        tr = np.stack([rng.uniform(-L / 2, L / 2, n), np.full(n, 1.0), rng.choice([-1, 1], n) * rng.uniform(Wd / 2 - 2.5, Wd / 2 - 0.8, n)], 1)
        instanced(mesh, tr, rng.uniform(0, 6.28, n), np.tile(rng.uniform(0.6, 1.4, (n, 1)), (1, 3)))
    # trees: trunk + a cloud of alpha-masked leaf quads, 3 tree meshes x 16 instances
    for k in range(3):
        n = int(4500 * detail)
        centers = rng.normal(size=(n, 3)) * np.array([1.6, 1.2, 1.6]) + np.array([0, 5.0, 0])
        ax = rng.normal(size=(n, 3))
        ax /= np.linalg.norm(ax, axis=1, keepdims=True)
        bx = np.cross(ax, rng.normal(size=(n, 3)))
        bx /= np.linalg.norm(bx, axis=1, keepdims=True)
        sz = np.exp(rng.uniform(np.log(0.12), np.log(0.45), (n, 1)))
        quad = np.stack([centers - ax * sz - bx * sz, centers + ax * sz - bx * sz, centers + ax * sz + bx * sz, centers - ax * sz + bx * sz], 1)
        base = np.arange(n)[:, None] * 4
        idx = np.concatenate([base + np.array([0, 1, 2]), base + np.array([0, 2, 3])], 1).reshape(-1, 3)
        leaves = b.primitive(quad.reshape(-1, 3).astype(np.float32), idx.astype(np.uint32), np.repeat(np.cross(ax, bx), 4, 0).astype(np.float32),
                             np.tile(np.array([[0, 1], [1, 1], [1, 0], [0, 0]], np.float32), (n, 1)), material=m_leaf)
        pos, nrm, uv, idx = uv_sphere(16, 24, 1.0)
        pos = pos * np.array([0.22, 2.6, 0.22], np.float32) + np.array([0, 2.6, 0], np.float32)
        trunk = b.primitive(pos, idx, _vertex_normals(pos.astype(np.float64), idx).astype(np.float32), uv, material=mats[100 + k])
        n_inst = 16
        tr = np.stack([np.linspace(-L / 2 + 8, L / 2 - 8, n_inst) + rng.uniform(-2, 2, n_inst), np.zeros(n_inst),
                       np.where(np.arange(n_inst) % 2 == 0, -1, 1) * (Wd / 2 - 1.6)], 1)
        instanced(b.mesh([leaves, trunk]), tr, rng.uniform(0, 6.28, n_inst), np.tile(rng.uniform(0.8, 1.3, (n_inst, 1)), (1, 3)))
    li = b.light({"type": "directional", "intensity": 6.0, "color": [1.0, 0.95, 0.88]})
    ang = 0.9
    b.node(extensions={"KHR_lights_punctual": {"light": li}},
           rotation=[-float(np.sin((np.pi / 2 - ang) / 2)) * 0.94, 0.34 * float(np.sin((np.pi / 2 - ang) / 2)), 0.0, float(np.cos((np.pi / 2 - ang) / 2))])
    b.camera_node((-L / 2 + 6.0, 1.8, 1.2), (L / 2, 6.0, -0.6), yfov=0.95, znear=0.05, zfar=1000.0)
    return b.save(path)


def scene_glass_class(path, seed=99, tess=48, dragon=0, tex_size=512):
    """TransmissionTest-class: a grid of spheres sweeping transmission / roughness / IOR / attenuation / dispersion /
    volume scatter, plus opaque reference spheres, on a diffuse floor with one point light.
    dragon > 0 adds the DragonDispersion half of BASELINE configs[4]: ONE high-poly dispersive glass blob of
    2 * dragon * (dragon // 2) triangles (dragon = 932: 868 624) with KHR_materials_dispersion + a coloured volume behind the grid, and four
    TEXTURED glass slabs (base-colour, metallic-roughness, transmission and thickness maps) standing around it; the sphere grid alone stays the
    unit-test size."""
    rng = np.random.default_rng(seed)
    b = GlbBuilder()
    floor = b.material(lambert_material((0.55, 0.55, 0.55)))
    pos, nrm, uv, idx = grid(8, 8, (16, 16), "y")
    b.node(mesh=b.mesh([b.primitive(pos, idx, nrm, uv, material=floor)]))
    sp = uv_sphere(tess, tess // 2, 0.45)
    k = 0
    for iy in range(3):
        for ix in range(6):
            ext = {"KHR_materials_transmission": {"transmissionFactor": float([1.0, 0.9, 0.6][iy])}, "KHR_materials_ior": {"ior": float(1.1 + 0.15 * ix)}}
            if ix % 2 == 0:
                ext["KHR_materials_volume"] = {"thicknessFactor": 0.9, "attenuationDistance": float(0.4 + 0.4 * ix),
                                               "attenuationColor": [float(v) for v in rng.uniform(0.2, 1.0, 3)]}
            if ix == 3:
                ext["KHR_materials_dispersion"] = {"dispersion": 8.0}
                ext.setdefault("KHR_materials_volume", {"thicknessFactor": 0.9})
            if ix == 5 and iy == 2:
                ext["KHR_materials_volume"] = {"thicknessFactor": 0.9, "attenuationDistance": 0.6, "attenuationColor": [0.9, 0.5, 0.3]}
                ext["KHR_materials_volume_scatter"] = {"multiscatterColor": [0.8, 0.6, 0.4], "scatterAnisotropy": 0.3}
            m = b.material({"pbrMetallicRoughness": {"baseColorFactor": [*[float(v) for v in rng.uniform(0.7, 1.0, 3)], 1.0], "metallicFactor": 0.0,
                                                     "roughnessFactor": float([0.0, 0.15, 0.4][iy])}, "extensions": ext})
            b.node(mesh=b.mesh([b.primitive(sp[0], sp[3], sp[1], sp[2], material=m)]), translation=[-3.0 + 1.2 * ix, 0.46, -1.2 + 1.2 * iy])
            k += 1
    if dragon > 0:
        bumps = rng.normal(size=(40, 3))
        bumps /= np.linalg.norm(bumps, axis=1, keepdims=True)
        amp, sharp = rng.uniform(0.05, 0.22, 40), rng.uniform(6.0, 40.0, 40)
        fr = rng.normal(size=(8, 3)) * 9.0

        def displace(d):
            sc = np.ones(d.shape[0])
            for k in range(40):
                sc += amp[k] * np.exp(-((1 - d @ bumps[k]) * sharp[k]))
            return sc + 0.012 * np.sin(d @ fr.T).sum(1)  # scales

        pos, nrm, uv, idx = uv_sphere(dragon, dragon // 2, 0.95, displace)
        m = b.material({"pbrMetallicRoughness": {"baseColorFactor": [0.95, 0.98, 0.96, 1.0], "metallicFactor": 0.0, "roughnessFactor": 0.02},
                        "extensions": {"KHR_materials_transmission": {"transmissionFactor": 1.0}, "KHR_materials_ior": {"ior": 1.55},
                                       "KHR_materials_dispersion": {"dispersion": 12.0},
                                       "KHR_materials_volume": {"thicknessFactor": 1.0, "attenuationDistance": 1.4, "attenuationColor": [0.55, 0.9, 0.7]}}})
        b.node(mesh=b.mesh([b.primitive(pos, idx, nrm, uv, material=m)]), translation=[0.0, 1.25, -3.1])
        smp = b.sampler()

        def tex(img):
            return b.texture(b.image((np.clip(img, 0, 1) * 255 + 0.5).astype(np.uint8)), smp)

        for k in range(4):
            n3, n1 = value_noise(rng, tex_size, 5, 3), value_noise(rng, tex_size, 4, 1)[..., 0]
            hue = rng.uniform(0.6, 1.0, 3)
            one = np.ones_like(n1)
            t_base = tex(np.concatenate([np.clip(hue * (0.6 + 0.4 * n3), 0, 1), one[..., None]], -1))
            t_mr = tex(np.stack([one, np.clip(0.02 + 0.5 * n1 * n1, 0, 1), np.zeros_like(n1), one], -1))  # g: roughness, b: metallic
            t_tr = tex(np.stack([np.clip(0.35 + 0.65 * value_noise(rng, tex_size, 3, 1)[..., 0], 0, 1), 0.5 + 0.5 * n1, one, one], -1))  # r: transmission, g: thickness
            ext = {"KHR_materials_transmission": {"transmissionFactor": 1.0, "transmissionTexture": {"index": t_tr}}, "KHR_materials_ior": {"ior": float(1.3 + 0.1 * k)},
                   "KHR_materials_volume": {"thicknessFactor": 0.25, "thicknessTexture": {"index": t_tr}, "attenuationDistance": float(0.5 + 0.5 * k),
                                            "attenuationColor": [float(v) for v in rng.uniform(0.4, 1.0, 3)]}}
            m = b.material({"pbrMetallicRoughness": {"baseColorTexture": {"index": t_base}, "metallicRoughnessTexture": {"index": t_mr}, "metallicFactor": 1.0,
                                                     "roughnessFactor": 1.0}, "extensions": ext})
            sp_ = box((1.6, 2.2, 0.12))
            ang = [0.5, -0.5, 0.25, -0.25][k]
            b.node(mesh=b.mesh([b.primitive(sp_[0], sp_[3], sp_[1], sp_[2], material=m)]), translation=[[-3.6, 3.6, -2.0, 2.0][k], 1.11, [-1.6, -1.6, -3.6, -3.6][k]],
                   rotation=[0.0, float(np.sin(ang / 2)), 0.0, float(np.cos(ang / 2))])
    li = b.light({"type": "point", "intensity": 300.0, "color": [1, 1, 1], "extras": {"radius": 0.25}})
    b.node(extensions={"KHR_lights_punctual": {"light": li}}, translation=[0.0, 5.0, 2.0])
    b.camera_node((0.0, 3.2, 5.2), (0, 0.3, 0), yfov=0.75)
    return b.save(path)


def scene_mixed_alpha_glass(path, seed=17, tess=20, tex_size=64):
    """Every kind of non-opaque instance side by side, for the walks' candidate handling: clear glass (transmissive, alphaMode OPAQUE: the alpha
    draw always commits), BLEND glass (transmissive AND a real alpha test), a diffuse-transmission sphere (non-opaque, alphaMode OPAQUE, not
    transmissive for shadow rays), alpha-MASK and alpha-BLEND textured cards, opaque spheres; floor, point light."""
    rng = np.random.default_rng(seed)
    b = GlbBuilder()
    floor = b.material(lambert_material((0.6, 0.6, 0.6)))
    pos, nrm, uv, idx = grid(6, 6, (10, 10), "y")
    b.node(mesh=b.mesh([b.primitive(pos, idx, nrm, uv, material=floor)]))
    sp = uv_sphere(tess, tess // 2, 0.42)
    a = value_noise(rng, tex_size, 4, 1)[..., 0]
    rgba = np.concatenate([np.clip(value_noise(rng, tex_size, 3, 3), 0, 1), (a > 0.5).astype(np.float64)[..., None]], -1)
    mask_tex = b.texture(b.image((rgba * 255 + 0.5).astype(np.uint8)), b.sampler())
    rgba[..., 3] = np.clip(a * 1.4, 0, 1)
    blend_tex = b.texture(b.image((rgba * 255 + 0.5).astype(np.uint8)), b.sampler())

    def pbr(color, rough=0.2, **extra):
        m = {"pbrMetallicRoughness": {"baseColorFactor": [*[float(c) for c in color], 1.0], "metallicFactor": 0.0, "roughnessFactor": float(rough)}}
        m.update(extra)
        return m
    mats = [
        pbr((0.95, 0.97, 1.0), 0.0, extensions={"KHR_materials_transmission": {"transmissionFactor": 1.0}, "KHR_materials_ior": {"ior": 1.5}}),
        pbr((0.9, 0.6, 0.5), 0.1, extensions={"KHR_materials_transmission": {"transmissionFactor": 0.7}, "KHR_materials_volume": {
            "thicknessFactor": 0.8, "attenuationDistance": 0.7, "attenuationColor": [0.8, 0.4, 0.3]}}),
        {**pbr((0.7, 0.9, 0.8), 0.05, extensions={"KHR_materials_transmission": {"transmissionFactor": 0.9}}), "alphaMode": "BLEND",
         "pbrMetallicRoughness": {"baseColorFactor": [0.7, 0.9, 0.8, 0.6], "metallicFactor": 0.0, "roughnessFactor": 0.05}},
        pbr((0.8, 0.7, 0.3), 0.5, extensions={"KHR_materials_diffuse_transmission": {"diffuseTransmissionFactor": 0.6, "diffuseTransmissionColorFactor": [0.9, 0.8, 0.5]}}),
        pbr((0.7, 0.2, 0.2), 0.6), pbr((0.2, 0.3, 0.8), 0.3),
    ]
    for k, m in enumerate(mats):
        b.node(mesh=b.mesh([b.primitive(sp[0], sp[3], sp[1], sp[2], material=b.material(m))]), translation=[-2.5 + 1.0 * k, 0.43, 0.4 * ((k % 2) * 2 - 1)])
    # cards between the spheres and the light: MASK and BLEND alpha from a texture, double sided
    cp, cn, cuv, cidx = grid(3, 3, (1.6, 1.1), "z")
    mask = b.material({"pbrMetallicRoughness": {"baseColorTexture": {"index": mask_tex}, "metallicFactor": 0.0, "roughnessFactor": 0.8}, "alphaMode": "MASK", "alphaCutoff": 0.5,
                       "doubleSided": True})
    blend = b.material({"pbrMetallicRoughness": {"baseColorTexture": {"index": blend_tex}, "metallicFactor": 0.0, "roughnessFactor": 0.8}, "alphaMode": "BLEND", "doubleSided": True})
    b.node(mesh=b.mesh([b.primitive(cp, cidx, cn, cuv, material=mask)]), translation=[-1.3, 1.5, 0.9], rotation=_quat((1, 0, 0), -0.9))
    b.node(mesh=b.mesh([b.primitive(cp, cidx, cn, cuv, material=blend)]), translation=[1.3, 1.5, 0.9], rotation=_quat((1, 0, 0), -0.9))
    li = b.light({"type": "point", "intensity": 220.0, "color": [1, 1, 1], "extras": {"radius": 0.2}})
    b.node(extensions={"KHR_lights_punctual": {"light": li}}, translation=[0.0, 4.0, 2.5])
    b.camera_node((0.0, 2.6, 5.0), (0, 0.4, 0), yfov=0.7)
    return b.save(path)


# ---- material zoo: one small scene per glTF material extension / renderer feature (parity tests of every BSDF lobe) -------------
def _sphere_tangents(pos):
    d = pos / np.maximum(np.linalg.norm(pos, axis=1, keepdims=True), 1e-12)
    tan = np.stack([-d[:, 2], np.zeros(len(d)), d[:, 0]], 1)
    ln = np.linalg.norm(tan, axis=1, keepdims=True)
    tan = np.where(ln > 1e-6, tan / np.maximum(ln, 1e-6), np.array([1.0, 0, 0]))
    return np.concatenate([tan, np.ones((len(tan), 1))], 1).astype(np.float32)


ZOO_GROUPS = ("clearcoat", "sheen", "iridescence", "anisotropy", "specgloss", "specular", "diffuse_transmission", "retroreflection",
              "unlit_emissive", "blend", "texture_transform", "vertex_streams")


def zoo_materials(b, group, rng, tex_size=64):
    """The materials of one zoo group (list of glTF material dicts); textures are added to builder `b`."""
    smp = b.sampler()
    smp_clamp = b.sampler(wrap_s=33071, wrap_t=33648)  # CLAMP_TO_EDGE / MIRRORED_REPEAT

    def tex(rgb, alpha=None, sampler=smp):
        rgb = np.asarray(rgb, np.float64)
        if rgb.ndim == 2:
            rgb = np.repeat(rgb[..., None], 3, -1)
        a = np.ones(rgb.shape[:2]) if alpha is None else alpha
        img = np.concatenate([rgb, a[..., None]], -1)
        return b.texture(b.image((np.clip(img, 0, 1) * 255 + 0.5).astype(np.uint8)), sampler)

    def noise(ch=3, oct_=4):
        n = value_noise(rng, tex_size, oct_, ch)
        return n if ch > 1 else n[..., 0]

    def nmap(strength=6.0):
        h = noise(1, 5)
        gx, gy = np.roll(h, -1, 1) - np.roll(h, 1, 1), np.roll(h, -1, 0) - np.roll(h, 1, 0)
        n = np.stack([-gx * strength, -gy * strength, np.ones_like(gx)], -1)
        n /= np.linalg.norm(n, axis=-1, keepdims=True)
        return n * 0.5 + 0.5

    def mr(color, metallic, roughness, **extra):
        m = {"pbrMetallicRoughness": {"baseColorFactor": [*[float(c) for c in color], 1.0], "metallicFactor": float(metallic), "roughnessFactor": float(roughness)}}
        m.update(extra)
        return m

    if group == "clearcoat":
        return [
            mr((0.8, 0.1, 0.1), 0.0, 0.6, extensions={"KHR_materials_clearcoat": {"clearcoatFactor": 1.0, "clearcoatRoughnessFactor": 0.05}}),
            mr((0.1, 0.3, 0.8), 1.0, 0.4, extensions={"KHR_materials_clearcoat": {"clearcoatFactor": 0.7, "clearcoatRoughnessFactor": 0.3}}),
            mr((0.2, 0.7, 0.2), 0.0, 0.5, normalTexture={"index": tex(nmap()), "scale": 0.8},
               extensions={"KHR_materials_clearcoat": {"clearcoatFactor": 1.0, "clearcoatRoughnessFactor": 0.6, "clearcoatTexture": {"index": tex(noise(3))},
                                                       "clearcoatRoughnessTexture": {"index": tex(noise(3))}, "clearcoatNormalTexture": {"index": tex(nmap(3.0))}}}),
        ]
    if group == "sheen":
        return [
            mr((0.3, 0.05, 0.3), 0.0, 0.8, extensions={"KHR_materials_sheen": {"sheenColorFactor": [0.9, 0.8, 1.0], "sheenRoughnessFactor": 0.3}}),
            mr((0.05, 0.2, 0.4), 0.0, 0.9, extensions={"KHR_materials_sheen": {"sheenColorFactor": [1.0, 1.0, 1.0], "sheenRoughnessFactor": 0.8,
                                                                                "sheenColorTexture": {"index": tex(0.3 + 0.7 * noise(3))},
                                                                                "sheenRoughnessTexture": {"index": tex(noise(3), 0.2 + 0.8 * noise(1))}}}),
            mr((0.5, 0.5, 0.5), 0.6, 0.5, extensions={"KHR_materials_sheen": {"sheenColorFactor": [0.2, 0.9, 0.4], "sheenRoughnessFactor": 0.05}}),
        ]
    if group == "iridescence":
        return [
            mr((0.05, 0.05, 0.05), 0.0, 0.2, extensions={"KHR_materials_iridescence": {"iridescenceFactor": 1.0, "iridescenceIor": 1.3, "iridescenceThicknessMaximum": 400.0}}),
            mr((0.9, 0.8, 0.7), 1.0, 0.15, extensions={"KHR_materials_iridescence": {"iridescenceFactor": 0.8, "iridescenceIor": 1.8, "iridescenceThicknessMinimum": 150.0,
                                                                                     "iridescenceThicknessMaximum": 900.0, "iridescenceTexture": {"index": tex(0.4 + 0.6 * noise(3))},
                                                                                     "iridescenceThicknessTexture": {"index": tex(noise(3))}}}),
            mr((0.6, 0.6, 0.9), 0.3, 0.4, extensions={"KHR_materials_iridescence": {"iridescenceFactor": 0.5, "iridescenceThicknessMaximum": 250.0},
                                                       "KHR_materials_specular": {"specularColorFactor": [1.0, 0.6, 0.3]}}),
        ]
    if group == "anisotropy":
        an = np.concatenate([0.5 + 0.5 * np.cos(noise(1)[..., None] * 6.28), 0.5 + 0.5 * np.sin(noise(1)[..., None] * 6.28), noise(1)[..., None]], -1)
        return [
            mr((0.9, 0.9, 0.9), 1.0, 0.35, extensions={"KHR_materials_anisotropy": {"anisotropyStrength": 0.9, "anisotropyRotation": 0.0}}),
            mr((0.9, 0.6, 0.2), 1.0, 0.25, extensions={"KHR_materials_anisotropy": {"anisotropyStrength": 0.7, "anisotropyRotation": 1.1}}),
            mr((0.2, 0.2, 0.8), 0.0, 0.3, normalTexture={"index": tex(nmap(3.0))},
               extensions={"KHR_materials_anisotropy": {"anisotropyStrength": 1.0, "anisotropyRotation": 0.4, "anisotropyTexture": {"index": tex(an)}}}),
        ]
    if group == "specgloss":
        return [
            {"extensions": {"KHR_materials_pbrSpecularGlossiness": {"diffuseFactor": [0.7, 0.3, 0.1, 1.0], "specularFactor": [0.04, 0.04, 0.04], "glossinessFactor": 0.6}}},
            {"extensions": {"KHR_materials_pbrSpecularGlossiness": {"diffuseFactor": [0.05, 0.05, 0.05, 1.0], "specularFactor": [0.9, 0.7, 0.3], "glossinessFactor": 0.8}}},
            {"extensions": {"KHR_materials_pbrSpecularGlossiness": {"diffuseFactor": [1.0, 1.0, 1.0, 1.0], "specularFactor": [1.0, 1.0, 1.0], "glossinessFactor": 1.0,
                                                                   "diffuseTexture": {"index": tex(0.2 + 0.8 * noise(3))},
                                                                   "specularGlossinessTexture": {"index": tex(0.3 * noise(3), 0.3 + 0.7 * noise(1))}}}},
        ]
    if group == "specular":
        return [
            mr((0.7, 0.2, 0.2), 0.0, 0.3, extensions={"KHR_materials_specular": {"specularFactor": 0.3, "specularColorFactor": [0.2, 0.6, 1.0]}}),
            mr((0.2, 0.2, 0.2), 0.0, 0.2, extensions={"KHR_materials_specular": {"specularFactor": 1.0, "specularColorFactor": [1.0, 1.0, 1.0],
                                                                                  "specularTexture": {"index": tex(noise(3), noise(1))},
                                                                                  "specularColorTexture": {"index": tex(0.2 + 0.8 * noise(3))}},
                                                       "KHR_materials_ior": {"ior": 1.9}}),
            mr((0.8, 0.8, 0.8), 0.0, 0.5, occlusionTexture={"index": tex(0.3 + 0.7 * noise(3)), "strength": 0.7},
               emissiveFactor=[1.0, 1.0, 1.0], emissiveTexture={"index": tex(np.clip((noise(3) - 0.6) * 4.0, 0, 1))},
               extensions={"KHR_materials_emissive_strength": {"emissiveStrength": 2.5}}),
        ]
    if group == "diffuse_transmission":
        return [
            mr((0.8, 0.7, 0.3), 0.0, 0.7, doubleSided=True, extensions={"KHR_materials_diffuse_transmission": {"diffuseTransmissionFactor": 0.7,
                                                                                                              "diffuseTransmissionColorFactor": [0.9, 0.5, 0.2]}}),
            mr((0.3, 0.7, 0.3), 0.0, 0.5, doubleSided=True,
               extensions={"KHR_materials_diffuse_transmission": {"diffuseTransmissionFactor": 1.0, "diffuseTransmissionColorFactor": [1.0, 1.0, 1.0],
                                                                  "diffuseTransmissionTexture": {"index": tex(noise(3), 0.2 + 0.8 * noise(1))},
                                                                  "diffuseTransmissionColorTexture": {"index": tex(0.3 + 0.7 * noise(3))}}}),
            mr((0.6, 0.6, 0.9), 0.0, 0.4, extensions={"KHR_materials_diffuse_transmission": {"diffuseTransmissionFactor": 0.4},
                                                       "KHR_materials_volume": {"thicknessFactor": 0.5, "attenuationDistance": 0.8, "attenuationColor": [0.9, 0.4, 0.4]}}),
        ]
    if group == "retroreflection":
        return [
            mr((0.8, 0.8, 0.1), 0.0, 0.3, extensions={"KHR_materials_retroreflection": {"retroreflectionFactor": 1.0}}),
            mr((0.9, 0.9, 0.9), 1.0, 0.25, extensions={"KHR_materials_retroreflection": {"retroreflectionFactor": 0.6, "retroreflectionTexture": {"index": tex(noise(3))}}}),
            mr((0.2, 0.3, 0.8), 0.0, 0.5, extensions={"KHR_materials_retroreflection": {"retroreflectionFactor": 0.5},
                                                       "KHR_materials_clearcoat": {"clearcoatFactor": 0.8, "clearcoatRoughnessFactor": 0.1},
                                                       "KHR_materials_sheen": {"sheenColorFactor": [0.5, 0.5, 0.5], "sheenRoughnessFactor": 0.5}}),
        ]
    if group == "unlit_emissive":
        return [
            mr((0.2, 0.8, 0.9), 0.0, 0.5, extensions={"KHR_materials_unlit": {}}),
            {"pbrMetallicRoughness": {"baseColorTexture": {"index": tex(0.2 + 0.8 * noise(3))}, "metallicFactor": 0.0}, "extensions": {"KHR_materials_unlit": {}}},
            mr((0.1, 0.1, 0.1), 0.0, 0.6, emissiveFactor=[1.0, 0.5, 0.2], extensions={"KHR_materials_emissive_strength": {"emissiveStrength": 4.0}}),
        ]
    if group == "blend":
        a = (noise(1, 3) > 0.45).astype(np.float64) * 0.8 + 0.1
        return [
            {"pbrMetallicRoughness": {"baseColorFactor": [0.9, 0.2, 0.2, 0.45], "metallicFactor": 0.0, "roughnessFactor": 0.6}, "alphaMode": "BLEND", "doubleSided": True},
            {"pbrMetallicRoughness": {"baseColorFactor": [1.0, 1.0, 1.0, 0.9], "baseColorTexture": {"index": tex(0.3 + 0.7 * noise(3), a)}, "metallicFactor": 0.0,
                                      "roughnessFactor": 0.4}, "alphaMode": "BLEND"},
            {"pbrMetallicRoughness": {"baseColorFactor": [0.3, 0.8, 0.3, 1.0], "baseColorTexture": {"index": tex(np.ones((tex_size, tex_size, 3)), noise(1, 3))},
                                      "metallicFactor": 0.0, "roughnessFactor": 0.7}, "alphaMode": "MASK", "alphaCutoff": 0.55, "doubleSided": True},
        ]
    if group == "texture_transform":
        chk = ((np.indices((tex_size, tex_size)).sum(0) // max(1, tex_size // 8)) % 2).astype(np.float64)
        base = np.stack([0.2 + 0.7 * chk, 0.3 + 0.5 * noise(1), 0.9 - 0.6 * chk], -1)
        t0, t1 = tex(base), tex(base, sampler=smp_clamp)
        return [
            {"pbrMetallicRoughness": {"baseColorTexture": {"index": t0, "extensions": {"KHR_texture_transform": {"offset": [0.25, 0.1], "scale": [3.0, 2.0], "rotation": 0.5}}},
                                      "metallicFactor": 0.0, "roughnessFactor": 0.6}},
            {"pbrMetallicRoughness": {"baseColorTexture": {"index": t1, "extensions": {"KHR_texture_transform": {"offset": [-0.4, 0.3], "scale": [2.5, 2.5]}}},
                                      "metallicFactor": 0.0, "roughnessFactor": 0.6}},
            {"pbrMetallicRoughness": {"baseColorTexture": {"index": t0, "texCoord": 0, "extensions": {"KHR_texture_transform": {"texCoord": 1, "rotation": -0.8, "scale": [1.5, 4.0]}}},
                                      "metallicRoughnessTexture": {"index": tex(noise(3)), "extensions": {"KHR_texture_transform": {"scale": [4.0, 4.0]}}}, "metallicFactor": 1.0}},
        ]
    if group == "vertex_streams":
        return [
            mr((1.0, 1.0, 1.0), 0.0, 0.6),  # COLOR_0 (u8 normalised) modulates base colour
            {"pbrMetallicRoughness": {"baseColorTexture": {"index": tex(0.2 + 0.8 * noise(3)), "texCoord": 1}, "metallicFactor": 0.0, "roughnessFactor": 0.5},
             "occlusionTexture": {"index": tex(0.3 + 0.7 * noise(3)), "texCoord": 1}},
            {"pbrMetallicRoughness": {"baseColorFactor": [1.0, 1.0, 1.0, 1.0], "metallicFactor": 0.0, "roughnessFactor": 0.5}, "alphaMode": "BLEND"},  # COLOR_0 alpha (float)
        ]
    raise KeyError(group)


def scene_material_zoo(path, group, seed=21, tess=32, tex_size=64, lights="point", camera="perspective"):
    """Three spheres carrying the materials of `group` over a Lambert floor + a back wall.  lights: "point" (sphere light + spot), "none"
    (environment only).  camera: "perspective" | "ortho"."""
    rng = np.random.default_rng(seed)
    b = GlbBuilder()
    mats = [b.material(m) for m in zoo_materials(b, group, rng, tex_size)]
    floor = b.material(lambert_material((0.5, 0.5, 0.5)))
    pos, nrm, uv, idx = grid(4, 4, (12, 12), "y")
    b.node(mesh=b.mesh([b.primitive(pos, idx, nrm, uv, material=floor)]))
    wall = b.material(lambert_material((0.7, 0.6, 0.5)))
    pos, nrm, uv, idx = grid(4, 4, (12, 6), "z")
    b.node(mesh=b.mesh([b.primitive(pos, idx, nrm, uv, material=wall)]), translation=[0, 3.0, -2.2])
    sp_pos, sp_nrm, sp_uv, sp_idx = uv_sphere(tess, tess // 2, 0.8)
    tan = _sphere_tangents(sp_pos)
    for k, m in enumerate(mats):
        kw = {}
        if group == "vertex_streams":
            d = sp_pos / 0.8
            if k == 0:
                kw["colors"] = (np.clip(np.concatenate([0.5 + 0.5 * d, np.ones((len(d), 1))], 1), 0, 1) * 255 + 0.5).astype(np.uint8)
            if k == 1:
                kw["uv1"] = np.stack([sp_uv[:, 1] * 2.0, sp_uv[:, 0] * 3.0], 1).astype(np.float32)
            if k == 2:
                kw["colors"] = np.concatenate([np.full((len(d), 3), 0.8), 0.15 + 0.85 * np.clip(0.5 + 0.5 * d[:, 1:2], 0, 1)], 1).astype(np.float32)
        elif group == "texture_transform":
            kw["uv1"] = np.stack([sp_uv[:, 1], sp_uv[:, 0]], 1).astype(np.float32)
        b.node(mesh=b.mesh([b.primitive(sp_pos, sp_idx, sp_nrm, sp_uv, tangents=tan, material=m, **kw)]), translation=[-1.9 + 1.9 * k, 0.81, 0.0],
               rotation=[0.0, float(np.sin(0.3 * k)), 0.0, float(np.cos(0.3 * k))])
    if lights == "point":
        li = b.light({"type": "point", "intensity": 260.0, "color": [1.0, 0.95, 0.9], "extras": {"radius": 0.3}})
        b.node(extensions={"KHR_lights_punctual": {"light": li}}, translation=[1.5, 4.5, 3.0])
        ls = b.light({"type": "spot", "intensity": 500.0, "color": [0.8, 0.9, 1.0], "spot": {"innerConeAngle": 0.25, "outerConeAngle": 0.5}})
        # light direction = -Z of the node: aim at the middle sphere from the upper left
        f = np.array([0.0, 0.8, 0.0]) - np.array([-3.0, 4.0, 2.5])
        f /= np.linalg.norm(f)
        zaxis = -f
        xaxis = np.cross([0, 1, 0], zaxis)
        xaxis /= np.linalg.norm(xaxis)
        yaxis = np.cross(zaxis, xaxis)
        m = np.eye(4)
        m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = xaxis, yaxis, zaxis, [-3.0, 4.0, 2.5]
        b.node(extensions={"KHR_lights_punctual": {"light": ls}}, matrix=[float(v) for v in m.T.reshape(-1)])
    if camera == "ortho":
        # zfar kept small: the reference's getRay puts an orthographic ray's origin on the clip-space z = -1 plane, i.e. (zfar - 2 znear)
        # BEHIND the eye under Vulkan's [0, 1] depth range, and every hit inherits |origin| * 2^-24 of rounding from that distance
        b.camera_node((0.0, 2.6, 6.0), (0, 0.7, 0), ortho=(3.4, 2.55), znear=0.05, zfar=24.0)
    else:
        b.camera_node((0.0, 2.6, 6.0), (0, 0.7, 0), yfov=0.62)
    return b.save(path)


def _quat(axis, angle):
    axis = np.asarray(axis, np.float64)
    axis = axis / np.linalg.norm(axis)
    return [float(v) for v in (*(axis * np.sin(angle / 2)), np.cos(angle / 2))]


def scene_animated(path, seed=5, tess=12):
    """A small kinetic sculpture for the animation path: arm (LINEAR rotation + translation) -> elbow (CUBICSPLINE translation and
    rotation) -> hand (STEP scale) with an instanced mesh, a point light riding on the elbow, a static floor, and a second clip."""
    rng = np.random.default_rng(seed)
    b = GlbBuilder()
    red = b.material(lambert_material((0.8, 0.25, 0.2)))
    blue = b.material({"pbrMetallicRoughness": {"baseColorFactor": [0.2, 0.3, 0.8, 1], "metallicFactor": 0.6, "roughnessFactor": 0.35}})
    grey = b.material(lambert_material((0.6, 0.6, 0.6)))
    sp, sn, _, si = uv_sphere(tess * 2, tess, 0.35)
    bp, bn, _, bi = box((0.5, 0.3, 0.4))
    fp, fn, _, fi = grid(4, 4, (8, 8))
    ball = b.mesh([b.primitive(sp, si, normals=sn, material=red)])
    brick = b.mesh([b.primitive(bp, bi, normals=bn, material=blue)])
    floor = b.mesh([b.primitive(fp, fi, normals=fn, material=grey)])
    b.node(mesh=floor, translation=[0, -1.2, 0])
    inst_t = b.accessor(np.asarray([[0, 0, 0], [0.9, 0.1, 0], [-0.9, 0.1, 0.2]], np.float32))
    inst_s = b.accessor(np.asarray([[1, 1, 1], [0.5, 0.5, 0.5], [0.7, 0.4, 0.7]], np.float32))
    hand = b.node(root=False, mesh=brick, translation=[0, -0.8, 0], extensions={"EXT_mesh_gpu_instancing": {"attributes": {"TRANSLATION": inst_t, "SCALE": inst_s}}})
    li = b.light({"type": "point", "color": [1, 0.9, 0.7], "intensity": 60.0})
    lamp = b.node(root=False, translation=[0.3, 0.5, 0.6], extensions={"KHR_lights_punctual": {"light": li}})
    elbow = b.node(root=False, mesh=ball, translation=[1.2, 0, 0], rotation=_quat((0, 0, 1), 0.3), children=[hand, lamp])
    arm = b.node(mesh=ball, translation=[-0.5, 0.4, 0], scale=[1.0, 1.0, 1.0], children=[elbow])
    still = b.node(mesh=brick, translation=[-2.0, -0.6, -1.0], rotation=_quat((0, 1, 0), 0.7))
    b.camera_node((0.5, 1.2, 6.0), (0.2, -0.2, 0))
    b.light({"type": "directional", "intensity": 2.0})
    b.node(rotation=_quat((1, 0, 0), -1.0), extensions={"KHR_lights_punctual": {"light": 1}})
    t4 = [0.0, 0.5, 1.25, 2.0]
    rot = [_quat((0, 1, 0), a) for a in (0.0, 1.4, 3.6, 6.0)]  # 3.6 - 1.4 > pi: the slerp must take the short way round
    cub_t = rng.uniform(-0.5, 0.5, (4, 3, 3)).astype(np.float32)
    cub_t[:, 1, :] += [1.2, 0, 0]
    cub_r = np.zeros((4, 3, 4), np.float32)
    for k, a in enumerate((0.3, -0.8, 1.1, 0.3)):
        cub_r[k, 1] = _quat((0, 0.3, 1), a)
        cub_r[k, 0] = rng.uniform(-0.3, 0.3, 4)
        cub_r[k, 2] = rng.uniform(-0.3, 0.3, 4)
    b.animation([(arm, "rotation", t4, rot, "LINEAR"),
                 (arm, "translation", [0.25, 1.0, 1.75], [[-0.5, 0.4, 0], [0.3, 0.9, -0.4], [-0.5, 0.0, 0.5]], "LINEAR"),
                 (elbow, "translation", t4, cub_t, "CUBICSPLINE"),
                 (elbow, "rotation", t4, cub_r, "CUBICSPLINE"),
                 (hand, "scale", [0.0, 0.7, 1.4, 2.0], [[1, 1, 1], [1.5, 0.6, 1.0], [0.5, 1.4, 0.8], [1, 1, 1]], "STEP")], name="sculpture")
    b.animation([(still, "translation", [1.0, 3.0], [[-2.0, -0.6, -1.0], [-2.0, 0.8, -1.0]], "LINEAR")], name="lift")
    b.ext_used.add("EXT_mesh_gpu_instancing")
    return b.save(path)
