"""mi_scene_cut_alpha (csrc/host/alpha_cut.cpp): the load-time bake for alpha-MASK geometry -- this renderer's counterpart of the
reference's opacity micro-map bake (src/gltf_scene_omm.cpp).  Checked on the CPU: (1) no point of a triangle where the alpha test
passes -- evaluated independently in numpy with the sampler's filter and address modes -- loses its geometry, on quads with tiled,
clamped, nearest- and bilinear-filtered alpha textures; (2) the CPU oracle renders the cut and the uncut atrium-class scene to the same
image; the GPU test checks the same against the HIP renderer and reports what the cut saves."""
import ctypes as C

import numpy as np
import pytest

from vk_gltf_renderer_amd import scenegen


def _alpha_texture(size, rng):
    """RGBA8: a disc with a soft rim plus speckles, so that thresholds, rims and isolated texels all occur."""
    y, x = np.mgrid[0:size, 0:size]
    r = np.hypot((x + 0.5) / size - 0.5, (y + 0.5) / size - 0.45)
    a = np.clip((0.33 - r) / 0.06, 0, 1)
    a = np.where(rng.random((size, size)) < 0.002, 1.0, a)
    img = np.zeros((size, size, 4), np.uint8)
    img[..., :3] = 180
    img[..., 3] = np.round(a * 255)
    return img


def _wrap(i, n, mode):
    if mode == 33071:  # CLAMP_TO_EDGE
        return np.clip(i, 0, n - 1)
    return np.mod(i, n)  # REPEAT


def _alpha_at(img, uv, linear, wrap_s, wrap_t):
    """alpha of SampleLevel(uv, 0) as pt_shading.h getOpacityFast / the oracle evaluate it (float64 here)."""
    h, w = img.shape[:2]
    a = img[..., 3].astype(np.float64) / 255.0
    fx, fy = uv[:, 0] * w, uv[:, 1] * h
    if not linear:
        return a[_wrap(np.floor(fy).astype(int), h, wrap_t), _wrap(np.floor(fx).astype(int), w, wrap_s)]
    fx, fy = fx - 0.5, fy - 0.5
    x0, y0 = np.floor(fx).astype(int), np.floor(fy).astype(int)
    tx, ty = fx - x0, fy - y0
    X0, X1, Y0, Y1 = _wrap(x0, w, wrap_s), _wrap(x0 + 1, w, wrap_s), _wrap(y0, h, wrap_t), _wrap(y0 + 1, h, wrap_t)
    return (a[Y0, X0] * (1 - tx) + a[Y0, X1] * tx) * (1 - ty) + (a[Y1, X0] * (1 - tx) + a[Y1, X1] * tx) * ty


def _quad_scene(path, img, uv_scale, uv_offset, mag, wrap, cutoff=0.5, factor=1.0):
    b = scenegen.GlbBuilder()
    tex = b.texture(b.image(img), b.sampler(mag=mag, min_=9729 if mag == 9729 else 9728, wrap_s=wrap, wrap_t=wrap))
    m = b.material({"pbrMetallicRoughness": {"baseColorTexture": {"index": tex}, "baseColorFactor": [1, 1, 1, factor]}, "alphaMode": "MASK", "alphaCutoff": cutoff,
                    "doubleSided": True})
    pos, nrm, uv, idx = scenegen.grid(2, 1, (2.0, 1.0), "z")
    b.node(mesh=b.mesh([b.primitive(pos, idx, nrm, uv * uv_scale + uv_offset, material=m)]))
    b.camera_node((0, 0, 3.0), (0, 0, 0))
    return b.save(path)


def _prim_arrays(scene, prim=0):
    d = scene.desc.contents
    p = d.renderPrimitives[prim]
    nt, nv = p.triangleCount, p.vertexCount
    idx = np.ctypeslib.as_array(p.indices, shape=(nt * 3,)).reshape(nt, 3).copy()
    uv = np.ctypeslib.as_array(p.texCoords0, shape=(nv * 2,)).reshape(nv, 2).copy()
    pos = np.ctypeslib.as_array(p.positions, shape=(nv * 3,)).reshape(nv, 3).copy()
    return idx, uv, pos


@pytest.mark.parametrize("case", ["bilinear_repeat_tiled", "nearest_repeat", "bilinear_clamp_offset", "factor_and_cutoff"])
def test_no_passing_point_loses_its_geometry(built, tmp_path, case):
    from vk_gltf_renderer_amd.pathtracer import Scene
    rng = np.random.default_rng(5)
    img = _alpha_texture(64, rng)
    scale, offset, mag, wrap, cutoff, factor = {"bilinear_repeat_tiled": (2.5, -0.7, 9729, 10497, 0.5, 1.0), "nearest_repeat": (1.0, 0.0, 9728, 10497, 0.5, 1.0),
                                                "bilinear_clamp_offset": (1.6, -0.3, 9729, 33071, 0.5, 1.0), "factor_and_cutoff": (1.0, 0.0, 9729, 10497, 0.3, 0.7)}[case]
    path = _quad_scene(str(tmp_path / f"{case}.glb"), img, scale, offset, mag, wrap, cutoff, factor)
    ref = Scene(path)
    idx0, uv0, pos0 = _prim_arrays(ref)
    cut = Scene(path)
    dropped = cut.cut_alpha(8)
    assert dropped > 0
    tris = cut.num_triangles
    assert cut.cut_alpha(8) == 0 and cut.num_triangles == tris  # once per loaded scene
    idx1, uv1, pos1 = _prim_arrays(cut)
    assert idx1.max() < len(uv1) and len(idx1) != len(idx0)
    # the new vertices lie on the original triangles' planes (z = 0 quad) and inside the quad
    assert np.abs(pos1[:, 2]).max() < 1e-6 and np.abs(pos1[:, 0]).max() <= 1.0 + 1e-6 and np.abs(pos1[:, 1]).max() <= 0.5 + 1e-6
    # random points of the original triangles, in uv space (the quad's uv map is affine and one to one)
    n = 60000
    t = rng.integers(0, len(idx0), n)
    b = rng.dirichlet((1, 1, 1), n)
    uv = (uv0[idx0[t]] * b[..., None]).sum(axis=1)
    alpha = factor * _alpha_at(img, uv, mag == 9729, wrap, wrap)
    passing = uv[alpha >= cutoff]
    assert len(passing) > 500
    # is every passing point inside a kept sub-triangle?  (uv-space point-in-triangle with a tolerance of a thousandth of a texel)
    A, B, Cc = uv1[idx1[:, 0]], uv1[idx1[:, 1]], uv1[idx1[:, 2]]
    d = (B[:, 0] - A[:, 0]) * (Cc[:, 1] - A[:, 1]) - (B[:, 1] - A[:, 1]) * (Cc[:, 0] - A[:, 0])
    covered = np.zeros(len(passing), bool)
    for k in range(len(idx1)):
        if abs(d[k]) < 1e-12:
            continue
        px, py = passing[:, 0] - A[k, 0], passing[:, 1] - A[k, 1]
        u = (px * (Cc[k, 1] - A[k, 1]) - py * (Cc[k, 0] - A[k, 0])) / d[k]
        v = (py * (B[k, 0] - A[k, 0]) - px * (B[k, 1] - A[k, 1])) / d[k]
        covered |= (u >= -1e-5) & (v >= -1e-5) & (u + v <= 1 + 1e-5)
    assert covered.all(), (case, int((~covered).sum()))
    # the other half of the classification: triangles [0, opaqueTriangleCount) are those the walks will NOT alpha-test -- every
    # point of them must pass the test (random points inside each, alpha evaluated as above)
    n_opaque = cut.desc.contents.renderPrimitives[0].opaqueTriangleCount
    assert 0 <= n_opaque <= len(idx1)
    if case != "bilinear_repeat_tiled":
        assert n_opaque > 0, case
    if n_opaque:
        to = rng.integers(0, n_opaque, 40000)
        bo = rng.dirichlet((1, 1, 1), 40000)
        uvo = (uv1[idx1[to]] * bo[..., None]).sum(axis=1)
        ao = factor * _alpha_at(img, uvo, mag == 9729, wrap, wrap)
        assert (ao >= cutoff).all(), (case, int((ao < cutoff).sum()), float(ao.min()))
        print(case, "opaque triangles", n_opaque, "of", len(idx1))
    # and the cut is worth something: a good part of the failing area is gone
    area = lambda i, p: 0.5 * np.abs(np.cross(p[i[:, 1]] - p[i[:, 0]], p[i[:, 2]] - p[i[:, 0]])[:, 2]).sum()
    print(case, "triangles", len(idx0), "->", len(idx1), "area kept", area(idx1, pos1) / area(idx0, pos0), "passing fraction", (alpha >= cutoff).mean())
    assert area(idx1, pos1) < (0.97 if case == "bilinear_repeat_tiled" else 0.9) * area(idx0, pos0)  # (2.5 tiles per quad: each sub-triangle sees most of the texture)


def test_primitives_the_bake_cannot_be_sure_about_are_left_alone(built, tmp_path):
    from vk_gltf_renderer_amd.pathtracer import Scene
    rng = np.random.default_rng(7)
    img = _alpha_texture(32, rng)
    # MIRRORED_REPEAT sampler, BLEND mode, and a MASK material without a texture: nothing may change
    for name, kw, mode in (("mirror", dict(wrap=33648), "MASK"), ("blend", dict(wrap=10497), "BLEND")):
        b = scenegen.GlbBuilder()
        tex = b.texture(b.image(img), b.sampler(wrap_s=kw["wrap"], wrap_t=kw["wrap"]))
        m = b.material({"pbrMetallicRoughness": {"baseColorTexture": {"index": tex}}, "alphaMode": mode, "alphaCutoff": 0.5})
        pos, nrm, uv, idx = scenegen.grid(1, 1, (1.0, 1.0), "z")
        b.node(mesh=b.mesh([b.primitive(pos, idx, nrm, uv, material=m)]))
        s = Scene(b.save(str(tmp_path / f"{name}.glb")))
        before = s.num_triangles
        assert s.cut_alpha(8) == 0 and s.num_triangles == before


def _selection_agrees(scene, uncut, cut):
    """The selection image (TraceLow: every triangle opaque, raytracer_interface.h.slang:124-137) may differ in one way only: where the
    uncut scene picks an alpha-MASK instance through a part of it that the bake removed, the baked scene picks what is seen there."""
    d = scene.desc.contents
    masked = np.zeros(d.numRenderNodes + 1, bool)
    for n in range(d.numRenderNodes):
        masked[n + 1] = d.materials[max(0, d.renderNodes[n].materialID)].alphaMode == 1  # MI_ALPHA_MASK
    return ((uncut == cut) | masked[uncut]).mean()


def test_oracle_renders_the_cut_scene_like_the_uncut_one(built, tmp_path):
    """Atrium-class scene (alpha-MASK foliage cards): same image from the CPU oracle with and without the bake."""
    import parity_util as pu
    path = scenegen.scene_atrium_class(str(tmp_path / "atrium.glb"), seed=5, detail=0.2, tex_size=64)
    a = pu.Setup(path, 160, 96, max_depth=4)
    b = pu.Setup(path, 160, 96, max_depth=4)
    before = b.scene.num_triangles
    dropped = b.scene.cut_alpha(8)
    print("atrium-class: triangles", before, "->", b.scene.num_triangles, "dropped (sub-)triangles", dropped)
    assert dropped > 0
    oa, ob = pu.render_oracle(a, 4), pu.render_oracle(b, 4)
    m = pu.compare_images(oa["accum"], ob["accum"])
    print(m)
    assert _selection_agrees(a.scene, oa["selection"], ob["selection"]) > 0.999
    assert m["frac_within_1e-4"] > 0.97 and m["rel_l2"] < 5e-3 and m["alpha_max_abs"] <= 0.25 + 1e-6


@pytest.mark.gpu
def test_gpu_renders_the_cut_scene_like_the_oracle_renders_the_uncut_one(built, tmp_path):
    """End to end: HIP renderer on the baked scene against the CPU oracle on the scene as loaded -- the bake must not be visible.
    Atrium-class (alpha-MASK foliage under sun + sky) and street-class (instanced alpha-MASK trees)."""
    import parity_util as pu
    for name, path, depth in (("atrium", scenegen.scene_atrium_class(str(tmp_path / "atrium.glb"), seed=5, detail=0.2, tex_size=64), 4),
                              ("street", scenegen.scene_street_class(str(tmp_path / "street.glb"), seed=11, detail=0.14, tex_size=32), 4)):
        plain = pu.Setup(path, 160, 96, max_depth=depth)
        baked = pu.Setup(path, 160, 96, max_depth=depth)
        before = baked.scene.num_triangles
        assert baked.scene.cut_alpha(4) > 0
        o, g = pu.render_oracle(plain, 4), pu.render_gpu(baked, 4, collect_counters=True)
        h = pu.render_gpu(plain, 4, collect_counters=True)
        m = pu.compare_images(o["accum"], g["accum"])
        print(name, "triangles", before, "->", baked.scene.num_triangles, m, "triangle tests per frame", h["stats"]["trisClosest"] + h["stats"]["trisShadow"], "->",
              g["stats"]["trisClosest"] + g["stats"]["trisShadow"])
        assert _selection_agrees(plain.scene, o["selection"], g["selection"]) > 0.999
        assert m["rel_l2"] < 6e-3 and m["frac_within_1e-2"] > 0.99 and m["frac_within_1e-4"] > 0.95
        # the point of the bake: fewer surface-less candidates.  Paths are the same ones (same segments up to rounding-induced turns)
        assert abs(g["stats"]["segments"] - h["stats"]["segments"]) <= 0.002 * h["stats"]["segments"]


@pytest.mark.gpu
def test_gpu_opaque_class_changes_no_bit(built, tmp_path):
    """A switch that only moves work around: the OPAQUE class of the bake -- triangles that cannot fail their alpha test skip it in the walks --
    against the same baked scene with every triangle alpha-tested (MI_PT_DIAG_NO_OPAQUE_TRIS).  Same paths, same arithmetic: bit-identical
    images, selection ids and path counters.  (The window sort of the SIMPLE shade kernel that this test also flipped was removed in round 5.)"""
    import os
    import parity_util as pu
    path = scenegen.scene_atrium_class(str(tmp_path / "atrium.glb"), seed=5, detail=0.2, tex_size=64)
    baked = pu.Setup(path, 160, 96, max_depth=6, alpha_cut=8)
    d = baked.scene.desc.contents
    assert sum(d.renderPrimitives[i].opaqueTriangleCount for i in range(d.numRenderPrimitives)) > 0

    def render(**env):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            return pu.render_gpu(baked, 6, collect_counters=True, in_flight=3)
        finally:
            for k, v in old.items():
                if v is None:
                    del os.environ[k]
                else:
                    os.environ[k] = v

    ref = render()
    tested = render(MI_PT_DIAG_NO_OPAQUE_TRIS="1")
    assert (ref["accum"] == tested["accum"]).all() and (ref["selection"] == tested["selection"]).all()
    for k in ("segments", "surfaceHits", "shadowRays", "textureTaps"):
        assert ref["stats"][k] == tested["stats"][k], k


def _translucent_card_scene(path, img, transmission):
    """A floor, an alpha-MASK card hovering over it (doubleSided, optionally KHR_materials_transmission) and a directional light from
    above: the card's shadow on the floor is black for a plain MASK material and tinted for a transmissive one."""
    b = scenegen.GlbBuilder()
    tex = b.texture(b.image(img), b.sampler(mag=9729, min_=9729, wrap_s=33071, wrap_t=33071))
    card = {"pbrMetallicRoughness": {"baseColorTexture": {"index": tex}, "baseColorFactor": [0.3, 0.9, 0.4, 1.0], "roughnessFactor": 0.4, "metallicFactor": 0.0},
            "alphaMode": "MASK", "alphaCutoff": 0.5, "doubleSided": True}
    if transmission > 0:
        card["extensions"] = {"KHR_materials_transmission": {"transmissionFactor": transmission}}
    floor = b.material({"pbrMetallicRoughness": {"baseColorFactor": [0.8, 0.8, 0.8, 1.0], "roughnessFactor": 0.9, "metallicFactor": 0.0}})
    pos, nrm, uv, idx = scenegen.grid(1, 1, (6.0, 6.0), "y")
    b.node(mesh=b.mesh([b.primitive(pos, idx, nrm, uv, material=floor)]))
    pos, nrm, uv, idx = scenegen.grid(2, 2, (2.0, 2.0), "y")
    b.node(mesh=b.mesh([b.primitive(pos, idx, nrm, uv, material=b.material(card))]), translation=[0.0, 1.0, 0.0])
    li = b.light({"type": "directional", "intensity": 4.0, "color": [1.0, 1.0, 1.0]})
    b.node(extensions={"KHR_lights_punctual": {"light": li}}, rotation=[-0.7071068, 0.0, 0.0, 0.7071068])  # -z turned to -y: shining down
    b.camera_node((0.0, 4.0, 5.0), (0, 0.3, 0))
    return b.save(path)


def test_transmissive_mask_material_gets_no_opaque_class(built, tmp_path):
    """Round-3 advisor finding: an OPAQUE-classified triangle is FORCE_OPAQUE to the shadow walk (occluded outright), so a MASK material
    that also transmits must keep the alpha test -- and with it the ordered transmissive pass -- on every piece that survives the bake."""
    from vk_gltf_renderer_amd.pathtracer import Scene
    img = _alpha_texture(64, np.random.default_rng(3))
    counts = {}
    for name, tr in (("plain", 0.0), ("translucent", 0.6)):
        s = Scene(_translucent_card_scene(str(tmp_path / f"{name}.glb"), img, tr))
        assert s.cut_alpha(8) > 0
        d = s.desc.contents
        counts[name] = sum(d.renderPrimitives[i].opaqueTriangleCount for i in range(d.numRenderPrimitives))
    assert counts["plain"] > 0 and counts["translucent"] == 0, counts


@pytest.mark.gpu
def test_gpu_translucent_mask_card_casts_the_same_tinted_shadow_with_the_cut(built, tmp_path):
    """The shadow of a transmissive alpha-MASK card: HIP renderer on the baked scene == HIP renderer on the scene as loaded (same paths;
    the bake only removes pieces that cannot pass) and both agree with the CPU oracle; the shadow is tinted, not black."""
    import parity_util as pu
    img = _alpha_texture(64, np.random.default_rng(3))
    path = _translucent_card_scene(str(tmp_path / "translucent.glb"), img, 0.6)
    plain, baked = pu.Setup(path, 160, 120, max_depth=4), pu.Setup(path, 160, 120, max_depth=4, alpha_cut=8)
    assert baked.alpha_cut_dropped > 0
    o, g0, g1 = pu.render_oracle(plain, 8), pu.render_gpu(plain, 8), pu.render_gpu(baked, 8)
    m0, m1 = pu.compare_images(o["accum"], g0["accum"]), pu.compare_images(o["accum"], g1["accum"])
    print(m0, m1)
    for m in (m0, m1):
        assert m["rel_l2"] < 4e-3 and m["frac_within_1e-2"] > 0.99
    assert g0["stats"]["shadowRays"] == g1["stats"]["shadowRays"] and g0["stats"]["segments"] == g1["stats"]["segments"]
    # the card's shadow on the floor is lit through the card: compare with the opaque-card version of the scene
    opaque = pu.Setup(_translucent_card_scene(str(tmp_path / "opaque.glb"), img, 0.0), 160, 120, max_depth=4, alpha_cut=8)
    g2 = pu.render_gpu(opaque, 8)
    lum_opaque, lum_translucent = g2["accum"][..., :3].mean(-1), g1["accum"][..., :3].mean(-1)
    dark = lum_opaque <= np.quantile(lum_opaque, 0.1)  # the opaque card's shadow (the oracle gives 0.0078 there against 0.0109 through the translucent card)
    assert lum_translucent[dark].mean() > 1.15 * lum_opaque[dark].mean(), (lum_translucent[dark].mean(), lum_opaque[dark].mean())


def test_cracks_of_the_adaptive_cut_stay_below_a_stated_measure(built, tmp_path):
    """The adaptive merge leaves T-junctions between coarse and refined cells (alpha_cut.cpp header): a refined cell's edge vertex is a
    float32 interpolation that need not lie exactly on the coarse neighbour's edge.  Since the OPAQUE class commits hits without an alpha
    test, a ray through such a crack would pass through the solid interior of a card -- this bounds how often: points of the card whose
    alpha passes with a wide margin (0.75 against a cutoff of 0.5: not near the rim), tested EXACTLY (float64 on the float32 vertices,
    no tolerance) for membership in some triangle of the baked geometry."""
    from vk_gltf_renderer_amd.pathtracer import Scene
    rng = np.random.default_rng(11)
    img = _alpha_texture(64, rng)
    path = _quad_scene(str(tmp_path / "card.glb"), img, 1.0, 0.0, 9729, 33071, 0.5, 1.0)
    ref, cut = Scene(path), Scene(path)
    idx0, uv0, pos0 = _prim_arrays(ref)
    assert cut.cut_alpha(8) > 0
    idx1, uv1, pos1 = _prim_arrays(cut)
    n = 400000
    t = rng.integers(0, len(idx0), n)
    b = rng.dirichlet((1, 1, 1), n)
    uv = (uv0[idx0[t]] * b[..., None]).sum(axis=1)
    p = (pos0[idx0[t]].astype(np.float64) * b[..., None]).sum(axis=1)[:, :2]
    solid = p[_alpha_at(img, uv, True, 33071, 33071) >= 0.75]
    assert len(solid) > 50000
    P = pos1.astype(np.float64)[:, :2]
    A, B, Cc = P[idx1[:, 0]], P[idx1[:, 1]], P[idx1[:, 2]]
    covered = np.zeros(len(solid), bool)
    for k in range(len(idx1)):
        e0 = (B[k, 0] - A[k, 0]) * (solid[:, 1] - A[k, 1]) - (B[k, 1] - A[k, 1]) * (solid[:, 0] - A[k, 0])
        e1 = (Cc[k, 0] - B[k, 0]) * (solid[:, 1] - B[k, 1]) - (Cc[k, 1] - B[k, 1]) * (solid[:, 0] - B[k, 0])
        e2 = (A[k, 0] - Cc[k, 0]) * (solid[:, 1] - Cc[k, 1]) - (A[k, 1] - Cc[k, 1]) * (solid[:, 0] - Cc[k, 0])
        covered |= ((e0 >= 0) & (e1 >= 0) & (e2 >= 0)) | ((e0 <= 0) & (e1 <= 0) & (e2 <= 0))
    leak = float((~covered).mean())
    print("solid points", len(solid), "not inside any baked triangle:", int((~covered).sum()), "fraction", leak)
    assert leak <= 2e-5, leak
