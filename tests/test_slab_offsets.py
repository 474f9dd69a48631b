"""The per-lane node test of the 8-wide walk after round 4 (csrc/device/pt_bvh8.h), restated in numpy float32 and checked on the CPU:
 (1) slabOffsets -- the pad now goes onto the plane TIME, B = fl(P * idir) -+ 2^-21 (255 |A| + |P * idir|), the same expression whatever the
     direction's sign -- keeps the test CONSERVATIVE: a child box that a ray hits in exact arithmetic (planes p + q * 2^e, float64) is
     never reported missed, for rays outside, inside, grazing a face, and with no travel along an axis;
 (2) the leaf word: doubling the 8-bit child hit mask bit by bit and ANDing the node's valid16 gives exactly the triangles of the hit
     leaf children, and leafPop's index arithmetic (base + popcount of the valid bits below) enumerates them in storage order."""
import numpy as np
import pytest

F = np.float32
K = F(4.76837158e-7)  # 2^-21


def _fma32(a, b, c):
    """fl32(a * b + c) for float32 inputs (the product is exact in float64; so is the sum unless the exponents lie > 29 bits apart)."""
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(F)


def _device_mask(P, s, qlo, qhi, org, dirv, tmax):
    """bvh8TestChildren: child i is hit unless sign(tf - tn) is set; near / far planes chosen by the sign of idir."""
    eps = F(1e-30)
    d = np.where(np.abs(dirv) < eps, np.copysign(eps, dirv), dirv).astype(F)
    idir = (F(1.0) / d).astype(F)
    tn, tf = np.zeros(8, F), np.full(8, tmax, F)
    for a in range(3):
        Pa = F(P[a] - org[a])
        A = F(s[a] * idir[a])
        t0 = F(Pa * idir[a])
        e = F(K * _fma32(np.array(255.0, F), np.abs(A)[None], np.abs(t0)[None])[0])
        Bn, Bf = F(t0 - e), F(t0 + e)
        neg = idir[a] < 0
        qn, qf = (qhi[a], qlo[a]) if neg else (qlo[a], qhi[a])
        tn = np.maximum(tn, _fma32(qn.astype(F), np.full(8, A, F), np.full(8, Bn, F)))
        tf = np.minimum(tf, _fma32(qf.astype(F), np.full(8, A, F), np.full(8, Bf, F)))
    return ~np.signbit((tf - tn).astype(F))


def _exact_mask(P, s, qlo, qhi, org, dirv, tmax):
    """The same boxes in float64: lo = p + qlo * s, hi = p + qhi * s; closed slabs, a ray parallel to a slab must start inside it."""
    hit = np.ones(8, bool)
    t0, t1 = np.zeros(8), np.full(8, float(tmax))
    for a in range(3):
        lo = float(P[a]) + qlo[a].astype(np.float64) * float(s[a])
        hi = float(P[a]) + qhi[a].astype(np.float64) * float(s[a])
        o, d = float(org[a]), float(dirv[a])
        if d == 0.0:
            hit &= (o >= lo) & (o <= hi)
            continue
        ta, tb = (lo - o) / d, (hi - o) / d
        t0 = np.maximum(t0, np.minimum(ta, tb))
        t1 = np.minimum(t1, np.maximum(ta, tb))
    return hit & (t0 <= t1) & (qlo[0] <= qhi[0])


@pytest.mark.parametrize("kind", ["outside", "inside", "graze", "axis", "far"])
def test_slab_offsets_keep_the_node_test_conservative(kind):
    rng = np.random.default_rng({"outside": 1, "inside": 2, "graze": 3, "axis": 4, "far": 5}[kind])
    missed = checked = entered = 0
    for _ in range(3000):
        P = rng.uniform(-50, 50, 3).astype(F)
        s = np.exp2(rng.integers(-14, 3, 3)).astype(F)
        qlo = rng.integers(0, 230, (3, 8))
        qhi = np.minimum(qlo + rng.integers(0, 120, (3, 8)), 255)
        empty = rng.random(8) < 0.2
        qlo[:, empty], qhi[:, empty] = 255, 0
        centre = P.astype(np.float64) + s * 128.0
        ext = float(np.max(s) * 255.0)
        if kind == "inside":
            org = (P + s * rng.uniform(0, 255, 3)).astype(F)
        elif kind == "far":
            org = (centre + rng.normal(size=3) * ext * rng.uniform(1e3, 1e5)).astype(F)
        else:
            org = (centre + rng.normal(size=3) * ext * rng.uniform(0.6, 8.0)).astype(F)
        c = int(rng.integers(0, 8))
        target = P.astype(np.float64) + s * rng.uniform(qlo[:, c], np.maximum(qhi[:, c], qlo[:, c] + 1))
        if kind == "graze":  # aim at a corner / edge of a child box
            target = P.astype(np.float64) + s * np.where(rng.random(3) < 0.5, qlo[:, c], qhi[:, c])
        d = target - org
        d /= max(np.linalg.norm(d), 1e-30)
        if kind == "axis":
            a = int(rng.integers(0, 3))
            org[a] = F(P[a] + s[a] * rng.uniform(0, 255))
            d[a] = rng.choice([-1.0, 1.0]) * 10.0 ** rng.uniform(-40, -10)
        d = d.astype(F)
        tmax = F(np.inf) if rng.random() < 0.5 else F(np.linalg.norm(centre - org) * rng.uniform(0.3, 2.0))
        dev = _device_mask(P, s, qlo, qhi, org, d, tmax)
        ex = _exact_mask(P, s, qlo, qhi, org, d, tmax)
        missed += int((ex & ~dev).sum())
        checked += int(ex.sum())
        entered += int(dev.sum())
    assert checked > 500, (kind, checked)
    assert missed == 0, (kind, missed, checked)
    # ... and it is not conservative by entering everything (from 1e3-1e5 node extents away the pad, 2^-21 of the distance, is a visible
    # part of a child box: twice the exact count there)
    assert entered <= (3.0 if kind == "far" else 1.6) * checked + 200, (kind, entered, checked)


def _spread(hm):
    x = hm
    x = (x | (x << 4)) & 0x0F0F
    x = (x | (x << 2)) & 0x3333
    x = (x | (x << 1)) & 0x5555
    return x | (x << 1)


def test_leaf_word_is_the_triangles_of_the_hit_leaf_children():
    rng = np.random.default_rng(7)
    for _ in range(5000):
        counts = rng.integers(0, 3, 8)  # triangles of the child in each slot (0: inner child or empty slot)
        valid = sum(((3 if c == 2 else 1) << (2 * i)) for i, c in enumerate(counts) if c)
        hm = int(rng.integers(0, 256))
        word = ((_spread(hm) & valid) << 16) | valid
        # storage order: the children's triangles one after the other in slot order
        first = np.concatenate([[0], np.cumsum(counts)[:-1]])
        expect = [int(first[i]) + k for i in range(8) if (hm >> i) & 1 for k in range(counts[i])]
        got, w = [], word
        while w > 0xFFFF:  # leafPending / leafPop
            bit = (w & 0xFFFF0000 & -(w & 0xFFFF0000)).bit_length() - 1
            w &= ~(1 << bit)
            got.append(bin(w & ((1 << (bit - 16)) - 1)).count("1"))
        assert got == expect, (counts, hm, got, expect)
        assert bin(word >> 16).count("1") == len(expect)  # leafCount
