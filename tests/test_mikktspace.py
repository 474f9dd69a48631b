"""Mikkelsen tangent spaces (vk_gltf_renderer_amd/csrc/host/mikktspace_tangents.cpp, an independent implementation of the published
method) against the reference's OWN third_party/MikkTSpace/mikktspace.c:
  * where /root/reference exists, oracle/_ref/libmikk_ref.so (built from the reference's source where it lies, oracle/Makefile) is run
    on the same arrays -- and the committed fixture tests/golden/mikk_cases.npz is checked to be what it produces;
  * everywhere (the GPU box has no /root/reference), the implementation is compared with that fixture.
Then the glTF-level wrapper (recomputeTangents with vertex splitting, src/gltf_create_tangent.cpp:512-612)."""
import ctypes as C
import os

import numpy as np
import pytest

from vk_gltf_renderer_amd import _capi as capi
from vk_gltf_renderer_amd import pathtracer as ptmod
from vk_gltf_renderer_amd import scenegen

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "mikk_cases.npz")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libmikk_ref.so")
F32P, U32P = C.POINTER(C.c_float), C.POINTER(C.c_uint32)


def _unit(v):
    return v / np.maximum(np.linalg.norm(v, axis=-1, keepdims=True), 1e-30)


def mesh_zoo():
    """name -> (positions, normals, uvs, indices): smooth and faceted meshes, UV seams, mirrored UVs, degenerate triangles (equal
    positions, zero uv area), duplicated vertices that weld, a non-manifold fan (three triangles on one edge), random soups."""
    rng = np.random.default_rng(12)
    out = {}
    # uv sphere with a seam (duplicated column) and pole fans
    p, n, uv, idx = scenegen.uv_sphere(16, 8, 1.0)
    out["sphere"] = (p, n, uv, idx.reshape(-1))
    # height-field grid with a smooth normal field, half of it uv-mirrored (orientation flips across the middle)
    g = 9
    x, y = np.meshgrid(np.linspace(-1, 1, g), np.linspace(-1, 1, g))
    z = 0.3 * np.sin(2.5 * x) * np.cos(1.7 * y)
    pos = np.stack([x, y, z], -1).reshape(-1, 3)
    nrm = _unit(np.stack([-0.75 * np.cos(2.5 * x) * np.cos(1.7 * y), 0.51 * np.sin(2.5 * x) * np.sin(1.7 * y), np.ones_like(x)], -1).reshape(-1, 3))
    u = np.abs(x)  # mirrored about x = 0
    uvm = np.stack([u, (y + 1) / 2], -1).reshape(-1, 2)
    tri = []
    for j in range(g - 1):
        for i in range(g - 1):
            a = j * g + i
            tri += [[a, a + 1, a + g], [a + 1, a + g + 1, a + g]]
    out["mirrored_grid"] = (pos, nrm, uvm, np.array(tri).reshape(-1))
    # the same grid unindexed (every corner its own vertex: welding has to find the sharing) with face normals on a quarter of it
    tri = np.array(tri)
    pos2, nrm2, uv2 = pos[tri.reshape(-1)].copy(), nrm[tri.reshape(-1)].copy(), uvm[tri.reshape(-1)].copy()
    fn = _unit(np.cross(pos[tri[:, 1]] - pos[tri[:, 0]], pos[tri[:, 2]] - pos[tri[:, 0]]))
    nrm2[:len(nrm2) // 4] = np.repeat(fn, 3, 0)[:len(nrm2) // 4]
    out["unindexed_mixed_normals"] = (pos2, nrm2, uv2, np.arange(len(pos2)))
    # degenerates: repeated positions, zero uv area, a triangle with two equal indices; they share vertices with good triangles
    pos3 = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0.2], [0, 0, 0], [2, 0, 0.1], [2, 1, 0]], np.float64)
    nrm3 = _unit(np.array([[0, 0, 1], [0.1, 0, 1], [0, 0.1, 1], [0.1, 0.1, 1], [0, 0, 1], [0.2, 0, 1], [0.1, 0.2, 1]], np.float64))
    uv3 = np.array([[0, 0], [1, 0], [0, 1], [1, 1], [0.5, 0.5], [2, 0], [2, 0]], np.float64)  # 5 and 6 share a uv: zero-area with 1 or 3
    out["degenerates"] = (pos3, nrm3, uv3, np.array([0, 1, 2, 1, 3, 2, 0, 4, 1, 1, 5, 3, 5, 6, 3, 2, 2, 3, 1, 5, 6]))
    # three triangles sharing one edge, two of them with opposite uv orientation
    pos4 = np.array([[0, 0, 0], [0, 0, 1], [1, 0, 0.5], [-1, 0.3, 0.5], [0, 1, 0.5]], np.float64)
    nrm4 = _unit(np.array([[0, 1, 0.2], [0, 1, -0.2], [0.3, 1, 0], [-0.3, 1, 0], [0.2, 0.5, 0]], np.float64))
    uv4 = np.array([[0, 0], [0, 1], [1, 0.5], [1, 0.4], [0.7, 0.5]], np.float64)
    out["butterfly"] = (pos4, nrm4, uv4, np.array([0, 1, 2, 1, 0, 3, 0, 1, 4]))
    # random soups: shared vertices by chance (small index range), arbitrary normals and uvs
    for k, (nv, nt) in enumerate(((12, 40), (60, 150), (300, 900))):
        out[f"soup{k}"] = (rng.normal(size=(nv, 3)), _unit(rng.normal(size=(nv, 3))), rng.uniform(-2, 2, (nv, 2)), rng.integers(0, nv, nt * 3))
    return {k: (np.ascontiguousarray(p, np.float32), np.ascontiguousarray(n, np.float32), np.ascontiguousarray(t, np.float32), np.ascontiguousarray(i, np.uint32))
            for k, (p, n, t, i) in out.items()}


def ours(p, n, t, i):
    out = np.zeros((len(i), 4), np.float32)
    rc = capi.host_lib().mi_mikktspace(p.ctypes.data_as(F32P), n.ctypes.data_as(F32P), t.ctypes.data_as(F32P), len(p), i.ctypes.data_as(U32P), len(i) // 3,
                                       out.ctypes.data_as(F32P))
    assert rc == 0
    return out


def reference(p, n, t, i):
    lib = C.CDLL(REF_SO)
    lib.mikk_ref.argtypes = [F32P, F32P, F32P, U32P, C.c_int, F32P]
    out = np.zeros((len(i), 4), np.float32)
    assert lib.mikk_ref(p.ctypes.data_as(F32P), n.ctypes.data_as(F32P), t.ctypes.data_as(F32P), i.ctypes.data_as(U32P), len(i) // 3, out.ctypes.data_as(F32P))
    return out


def _same(a, b, name):
    assert (a[:, 3] == b[:, 3]).all(), (name, np.nonzero(a[:, 3] != b[:, 3])[0][:8])
    d = np.abs(a[:, :3] - b[:, :3]).max(-1)
    assert d.max() <= 2e-6, (name, int(np.argmax(d)), float(d.max()), a[np.argmax(d)], b[np.argmax(d)])


@pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref is built only where /root/reference exists")
def test_matches_the_reference_library_and_the_fixture_is_current(built):
    zoo = mesh_zoo()
    gold = np.load(GOLD) if os.path.exists(GOLD) else None
    for name, m in zoo.items():
        want = reference(*m)
        _same(ours(*m), want, name)
        assert gold is not None and np.array_equal(gold[name], want), f"{name}: regenerate tests/golden/mikk_cases.npz (python tests/test_mikktspace.py)"
    assert any((reference(*m)[:, 3] < 0).any() and (reference(*m)[:, 3] > 0).any() for m in zoo.values())  # both orientations occur


def test_matches_the_committed_reference_outputs(built):
    gold = np.load(GOLD)
    zoo = mesh_zoo()
    assert set(gold.files) == set(zoo)
    for name, m in zoo.items():
        _same(ours(*m), gold[name], name)


def test_recompute_tangents_splits_vertices_at_discontinuities(built, tmp_path):
    """The glTF-level entry (recomputeTangents(model, force, mikktspace), src/gltf_create_tangent.cpp:619-677): a uv-mirrored grid has
    vertices on the mirror line whose two sides disagree in handedness -> those vertices are duplicated, every corner then points at
    a vertex with a compatible tangent (within ~11 degrees, same handedness), all other streams are copied, geometry is unchanged."""
    zoo = mesh_zoo()
    pos, nrm, uv, idx = zoo["mirrored_grid"]
    b = scenegen.GlbBuilder()
    m = b.material(scenegen.lambert_material((0.8, 0.8, 0.8)))
    b.node(mesh=b.mesh([b.primitive(pos, idx.reshape(-1, 3), nrm, uv, material=m)]))
    sc = ptmod.Scene(b.save(str(tmp_path / "grid.glb")))
    d0 = sc.desc.contents.renderPrimitives[0]
    nv0, nt = d0.vertexCount, d0.triangleCount
    assert not d0.tangents  # no normal map: the loader creates none
    assert sc.recompute_tangents(False, True) == 0  # not forced: primitives without tangents are left alone
    added = sc.recompute_tangents(True, True)
    assert added == 9, added  # the nine vertices on the mirror line x = 0
    d = sc.desc.contents.renderPrimitives[0]
    assert d.vertexCount == nv0 + added and d.triangleCount == nt
    P = np.ctypeslib.as_array(d.positions, shape=(d.vertexCount, 3))
    T = np.ctypeslib.as_array(d.tangents, shape=(d.vertexCount, 4))
    I = np.ctypeslib.as_array(d.indices, shape=(nt * 3,))
    assert np.array_equal(P[I], pos[idx])  # same triangles
    raw = ours(pos, nrm, uv, idx)
    want = np.concatenate([raw[:, :3], -raw[:, 3:]], 1)  # handedness flipped for this renderer's bitangent convention
    got = T[I]
    assert (np.sign(got[:, 3]) == np.sign(want[:, 3])).all()
    assert ((got[:, :3] * want[:, :3]).sum(-1) >= 0.98 - 1e-6).all()
    assert np.allclose(np.linalg.norm(T[:, :3], axis=1), 1.0, atol=1e-5)
    # the simple method on the same scene keeps the vertex count
    assert sc.recompute_tangents(True, False) == 0
    assert sc.desc.contents.renderPrimitives[0].vertexCount == nv0 + added


if __name__ == "__main__":  # regenerates the fixture from the reference library
    np.savez_compressed(GOLD, **{k: reference(*m) for k, m in mesh_zoo().items()})
    print("wrote", GOLD)
