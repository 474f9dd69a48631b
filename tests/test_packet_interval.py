"""The packet walk's interval node test (csrc/device/pt_packet.h) must enter every child that the per-ray test
(pt_bvh8.h: bvh8TestChildrenPlanes) enters for some ray of the packet -- otherwise a camera ray could lose its hit -- and should enter
few others.  Checked on the CPU through tests/host_shim on random nodes and packets: the kind the kernel sees (64 samples of one
pixel, a pinhole or a thin-lens camera), wider ones (an 8x8 pixel block), and rays that graze, start inside or point along an axis."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dev(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("host_shim_packet") / "libdevice_on_host.so")
    shim = os.path.join(ROOT, "tests", "host_shim")
    subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-shared", "-I" + shim, "-I" + os.path.join(ROOT, "include"),
                    "-I" + os.path.join(ROOT, "vk_gltf_renderer_amd", "csrc", "device"), "-o", out, os.path.join(shim, "device_on_host.cpp")], check=True)
    lib = C.CDLL(out)
    fp, up = C.POINTER(C.c_float), C.POINTER(C.c_uint32)
    lib.dev_packet_masks.argtypes = [fp, fp, fp, C.c_int, fp, fp, fp, up, up]
    lib.dev_packet_masks.restype = C.c_int
    return lib


def _node(rng):
    """A node as the builder quantises it: origin, power-of-two scale per axis, eight child boxes as bytes (some slots empty = inverted)."""
    P = rng.uniform(-20, 20, 3).astype(np.float32)
    s = np.exp2(rng.integers(-12, 2, 3)).astype(np.float32)
    lo = rng.integers(0, 230, (3, 8))
    hi = np.minimum(lo + rng.integers(1, 120, (3, 8)), 255)
    empty = rng.random(8) < 0.25
    lo[:, empty], hi[:, empty] = 255, 0
    planes = np.zeros(48, np.float32)
    for a in range(3):
        planes[(2 * a) * 8:(2 * a) * 8 + 8] = lo[a]
        planes[(2 * a + 1) * 8:(2 * a + 1) * 8 + 8] = hi[a]
    return P, s, planes


def _packet(rng, P, s, kind):
    centre = P + s * 128.0
    extent = float(np.max(s) * 255.0)
    if kind == "inside":
        eye = (P + s * rng.uniform(0, 255, 3)).astype(np.float32)
    else:
        eye = (centre + rng.normal(size=3) * extent * rng.uniform(0.5, 6.0)).astype(np.float32)
    target = centre + rng.uniform(-1.2, 1.2, 3) * s * 128.0
    d0 = target - eye
    d0 /= max(np.linalg.norm(d0), 1e-20)
    if kind == "axis":  # no travel along one axis (inverse direction at its 1e30 clamp), from inside the node's slab on that axis
        a = int(rng.integers(0, 3))
        eye[a] = np.float32(P[a] + s[a] * rng.uniform(0, 255))
        d0[a] = rng.choice([-1.0, 1.0]) * 10.0 ** rng.uniform(-40, -12)
        d0 /= np.linalg.norm(d0)
    spread = {"pixel": 5e-4, "block": 8e-3, "lens": 5e-4, "inside": 5e-4, "axis": 5e-4, "graze": 5e-4}[kind]
    n = 64
    dirs = d0[None, :] + rng.normal(size=(n, 3)) * spread
    if kind == "axis":
        dirs[:, a] = d0[a] * rng.uniform(0.5, 2.0, n)
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    orgs = np.repeat(eye[None, :], n, 0)
    if kind == "lens":
        orgs = orgs + rng.normal(size=(n, 3)) * extent * 0.01
    tmax = np.full(n, np.inf, np.float32)
    if rng.random() < 0.5:
        tmax = (np.linalg.norm(centre - eye) * rng.uniform(0.3, 1.5, n)).astype(np.float32)
    return orgs.astype(np.float32), dirs.astype(np.float32), tmax


@pytest.mark.parametrize("kind", ["pixel", "block", "lens", "inside", "axis", "graze"])
def test_interval_mask_contains_the_per_ray_masks(dev, kind):
    rng = np.random.default_rng({"pixel": 1, "block": 2, "lens": 3, "inside": 4, "axis": 5, "graze": 6}[kind])
    fp = C.POINTER(C.c_float)
    used = extra = exact_bits = 0
    for _ in range(4000):
        P, s, planes = _node(rng)
        orgs, dirs, tmax = _packet(rng, P, s, kind)
        if kind == "graze":  # aim along a face of a child box
            c = int(rng.integers(0, 8))
            a = int(rng.integers(0, 3))
            face = P[a] + s[a] * planes[(2 * a + int(rng.integers(0, 2))) * 8 + c]
            orgs[:, a] = face + rng.normal() * s[a] * 1e-3
            dirs[:, a] = rng.normal(size=64) * 1e-6
            dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
            dirs = dirs.astype(np.float32)
            if len({bool(x) for x in (dirs[:, a] < 0)}) > 1:
                dirs[:, a] = np.abs(dirs[:, a])
        ex, iv = C.c_uint32(0), C.c_uint32(0)
        ok = dev.dev_packet_masks(P.ctypes.data_as(fp), s.ctypes.data_as(fp), planes.ctypes.data_as(fp), 64, np.ascontiguousarray(orgs).ctypes.data_as(fp),
                                  np.ascontiguousarray(dirs).ctypes.data_as(fp), tmax.ctypes.data_as(fp), C.byref(ex), C.byref(iv))
        if not ok:
            continue
        used += 1
        assert ex.value & ~iv.value == 0, (kind, P, s, planes, hex(ex.value), hex(iv.value))
        exact_bits += bin(ex.value).count("1")
        extra += bin(iv.value & ~ex.value).count("1")
    assert used > 1000, used
    # how loose the interval test is: children entered without need, per child entered by need
    ratio = extra / max(exact_bits, 1)
    print(kind, "packets", used, "children needed", exact_bits, "extra", extra, f"ratio {ratio:.4f}")
    # the kernel uses the interval test for packets of one pixel's samples from a pinhole camera ("pixel", "inside", "axis", "graze"); an
    # 8x8 pixel block or a thin lens would make it loose ("block", "lens": containment only), those packets keep the per-ray test
    limit = {"pixel": 0.3, "inside": 0.1, "axis": 0.3, "graze": 0.4}.get(kind)
    assert limit is None or ratio < limit, ratio
