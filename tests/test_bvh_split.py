"""Triangle pre-splitting of the device BVH builder (csrc/device/bvh_split.h: the work of ONE thread, the function the kernels of bvh_build.hip call),
compiled for the host through tests/host_shim -- no GPU needed.  Checked: the reference boxes of a triangle COVER it (a ray that hits the triangle is
inside one of them -- what keeps the image independent of the split), stay inside the triangle's own box, respect the area threshold until the depth
bound, and degenerate or unsplittable input gets exactly one reference."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CAP = 2048


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("host_shim_split") / "libsplit_on_host.so")
    shim = os.path.join(ROOT, "tests", "host_shim")
    subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-I" + shim, "-I" + os.path.join(ROOT, "vk_gltf_renderer_amd", "csrc", "device"), "-o", out,
                    os.path.join(shim, "split_on_host.cpp")], check=True)
    L = C.CDLL(out)
    L.dev_split_triangle.restype = C.c_int
    L.dev_split_triangle.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_int]
    return L


def split(lib, tri, threshold, depth=8, splittable=True):
    tri = np.ascontiguousarray(tri, np.float32).reshape(3, 3)
    box = np.concatenate([tri.min(0), tri.max(0)]).astype(np.float32)
    out = np.zeros((CAP, 6), np.float32)
    n = lib.dev_split_triangle(tri.ctypes.data, box.ctypes.data, int(splittable), C.c_float(threshold), depth, out.ctypes.data, CAP)
    assert 1 <= n <= CAP
    return out[:n], box


def half_area(b):
    e = b[..., 3:] - b[..., :3]
    return e[..., 0] * e[..., 1] + e[..., 1] * e[..., 2] + e[..., 2] * e[..., 0]


def covered(boxes, pts):
    inside = (pts[:, None, :] >= boxes[None, :, :3]).all(-1) & (pts[:, None, :] <= boxes[None, :, 3:]).all(-1)
    return inside.any(1)


def sample_points(tri, rng, n=4000):
    u = rng.random((n, 2))
    flip = u.sum(1) > 1
    u[flip] = 1 - u[flip]
    # points as the intersector would see them: p0 + u e1 + v e2 in float32, plus the vertices and edge midpoints
    tri = tri.astype(np.float32)
    p = (tri[0] + u[:, :1].astype(np.float32) * (tri[1] - tri[0]) + u[:, 1:].astype(np.float32) * (tri[2] - tri[0])).astype(np.float32)
    extra = np.array([tri[0], tri[1], tri[2], (tri[0] + tri[1]) / 2, (tri[1] + tri[2]) / 2, (tri[0] + tri[2]) / 2], np.float32)
    return np.concatenate([p, extra])


def test_references_cover_the_triangle_and_stay_inside_its_box(lib):
    rng = np.random.default_rng(7)
    total = 0
    for k in range(300):
        scale = 10.0 ** rng.uniform(-2, 2)
        kind = k % 4
        tri = rng.normal(size=(3, 3)) * scale + rng.normal(size=3) * 10.0 ** rng.uniform(-1, 3)
        if kind == 1:  # long thin sliver, not axis aligned
            d = rng.normal(size=3)
            tri = np.stack([tri[0], tri[0] + d * scale * 40, tri[0] + d * scale * 40 + rng.normal(size=3) * scale * 0.01])
        elif kind == 2:  # axis-aligned wall (zero extent on one axis)
            tri[:, k % 3] = tri[0, k % 3]
        tri = tri.astype(np.float32)
        boxes, box = split(lib, tri, threshold=half_area(box_of(tri)) / 10.0 ** rng.uniform(0.5, 2.5))
        total += len(boxes)
        assert (boxes[:, :3] >= box[:3]).all() and (boxes[:, 3:] <= box[3:]).all() and (boxes[:, :3] <= boxes[:, 3:]).all()
        pts = sample_points(tri, rng)
        # a sampled point may lie an ulp outside the triangle's own float box (p0 + u e1 + v e2 rounds): clamp like the builder's box does
        pts = np.clip(pts, box[:3], box[3:])
        assert covered(boxes, pts).all(), (k, kind, len(boxes))
    assert total > 300 * 4  # (the thresholds above really split)


def box_of(tri):
    return np.concatenate([tri.min(0), tri.max(0)]).astype(np.float32)


def test_parts_obey_the_threshold_until_the_depth_bound(lib):
    tri = np.array([[0, 0, 0], [36, 0, 0], [0, 12, 0.5]], np.float32)
    whole = float(half_area(box_of(tri)))
    for depth in (1, 3, 8, 10):
        boxes, _ = split(lib, tri, threshold=whole / 64, depth=depth)
        assert len(boxes) <= 2 ** depth
        if depth >= 8:  # deep enough for every part to get below the threshold (plus the padding)
            assert (half_area(boxes) <= whole / 64 * 1.001).all()
    # the references of a split triangle have far less box area in total than ... each alone would: the point of the exercise
    boxes, _ = split(lib, tri, threshold=whole / 64)
    assert half_area(boxes).max() < whole / 32 and 8 <= len(boxes) <= 256


def test_one_reference_where_nothing_is_to_be_split(lib):
    tri = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    for kw in (dict(threshold=10.0), dict(threshold=1e-6, splittable=False), dict(threshold=1e-6, depth=0)):
        boxes, box = split(lib, tri, **kw)
        assert len(boxes) == 1 and (boxes[0] == box).all()
    # degenerate input: a point, a segment, coincident vertices far from the origin -- one reference or a few, never none, always covering
    for tri in (np.zeros((3, 3)), np.array([[0, 0, 0], [5, 5, 5], [10, 10, 10]]), np.full((3, 3), 1e6), np.array([[1e6, 0, 0], [1e6 + 0.25, 0, 0], [1e6, 0.25, 0]])):
        boxes, box = split(lib, tri.astype(np.float32), threshold=1e-9)
        assert len(boxes) >= 1 and covered(boxes, tri.astype(np.float32)).all()


def test_a_hall_sized_wall_next_to_centimetre_detail(lib):
    """The case of the sliver stand-in: a 36 x 12 m wall triangle with the scene's mean box area at a few square centimetres gets the full 2^8 budget, every
    part covering its piece of the wall."""
    rng = np.random.default_rng(3)
    tri = np.array([[-18, 0, 7], [18, 0, 7], [18, 12, 7]], np.float32)
    boxes, box = split(lib, tri, threshold=16 * 0.004, depth=8)
    assert 128 <= len(boxes) <= 256
    assert covered(boxes, np.clip(sample_points(tri, rng, 20000), box[:3], box[3:])).all()
    assert half_area(boxes).sum() < 3.0 * half_area(box)  # (flat wall: box area = 2-D area, the parts tile it with overlap only from the padding and the diagonal)
