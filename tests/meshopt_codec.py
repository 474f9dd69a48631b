"""TEST INFRASTRUCTURE ONLY: encoders for the EXT_meshopt_compression bitstreams and filters, written from the extension's specification, so that
tests/test_meshopt.py can round-trip csrc/host/meshopt_decoder.cpp.  No third-party encoder exists in this image: these encoders and the decoder come
from the same reading of the specification, which is why the decoder's own end-of-stream checks matter (a stream it misreads fails, it does not decode
to something else)."""
import numpy as np


# ---- mode ATTRIBUTES -----------------------------------------------------------------------------------------------------------------------
def encode_vertices(v):
    """v: uint8 array (count, stride), stride a multiple of 4 and <= 256 -> bytes (codec version 0)"""
    v = np.ascontiguousarray(v, np.uint8)
    count, stride = v.shape
    assert stride % 4 == 0 and 0 < stride <= 256
    out = bytearray([0xA0])
    block = min(256, (8192 // stride) & ~15)
    first = v[0].copy() if count else np.zeros(stride, np.uint8)
    last = first.copy()
    for off in range(0, count, block):
        blk = v[off:off + block]
        n = len(blk)
        aligned = (n + 15) & ~15
        for k in range(stride):
            col = blk[:, k].astype(np.int32)
            prev = np.concatenate([[int(last[k])], col[:-1]])
            d = ((col - prev + 128) & 0xFF) - 128          # signed byte difference
            z = ((d << 1) ^ (d >> 31)) & 0xFF              # zigzag
            z = np.concatenate([z, np.zeros(aligned - n, np.int32)])
            header = bytearray((aligned // 16 + 3) // 4)
            body = bytearray()
            for g in range(aligned // 16):
                grp = z[g * 16:(g + 1) * 16]
                if not grp.any():
                    mode = 0
                else:
                    cost = {1: 4 + int((grp >= 3).sum()), 2: 8 + int((grp >= 15).sum()), 3: 16}
                    mode = min(cost, key=lambda m: (cost[m], m))
                header[g // 4] |= mode << ((g % 4) * 2)
                if mode == 3:
                    body += bytes(int(x) for x in grp)
                elif mode in (1, 2):
                    bits = 2 if mode == 1 else 4
                    sentinel = (1 << bits) - 1
                    per = 8 // bits
                    packed, extra = bytearray(), bytearray()
                    for b in range(16 // per):
                        byte = 0
                        for x in grp[b * per:(b + 1) * per]:
                            byte = (byte << bits) | (sentinel if x >= sentinel else int(x))
                            if x >= sentinel:
                                extra.append(int(x))
                        packed.append(byte)
                    body += packed + extra
            out += header + body
        last = blk[-1].copy()
    tail = max(32, stride)
    out += bytes(tail - stride) + bytes(first)
    return bytes(out)


def _encode_plane(z, widths):
    """z: zigzag values padded to a multiple of 16 -> (header bytes + body) with the group widths table `widths` (bits per value; 0 = zeros, 8 = plain)"""
    header = bytearray((len(z) // 16 + 3) // 4)
    body = bytearray()
    for g in range(len(z) // 16):
        grp = z[g * 16:(g + 1) * 16]
        best = None
        for i, bits in enumerate(widths):
            if bits == 0:
                if grp.any():
                    continue
                cost = 0
            elif bits == 8:
                cost = 16
            else:
                cost = 2 * bits + int((grp >= (1 << bits) - 1).sum())
            if best is None or cost < best[0]:
                best = (cost, i, bits)
        _, i, bits = best
        header[g // 4] |= i << ((g % 4) * 2)
        if bits == 8:
            body += bytes(int(x) for x in grp)
        elif bits:
            sentinel, per = (1 << bits) - 1, 8 // bits
            packed, extra = bytearray(), bytearray()
            for b in range(16 // per):
                byte = 0
                for x in grp[b * per:(b + 1) * per]:
                    byte = (byte << bits) | (sentinel if x >= sentinel else int(x))
                    if x >= sentinel:
                        extra.append(int(x))
                packed.append(byte)
            body += packed + extra
    return bytes(header) + bytes(body)


def encode_vertices_v1(v, channels=None):
    """Codec version 1 (KHR_meshopt_compression).  channels: one byte per 4-byte component -- low two bits 0 = byte differences, 1 = 16-bit differences,
    2 = 32-bit XOR rotated left by the high nibble."""
    v = np.ascontiguousarray(v, np.uint8)
    count, stride = v.shape
    assert stride % 4 == 0 and 0 < stride <= 256
    channels = list(channels) if channels is not None else [0] * (stride // 4)
    out = bytearray([0xA1])
    block = min(256, (8192 // stride) & ~15)
    first = v[0].copy() if count else np.zeros(stride, np.uint8)
    last = first.copy()
    for off in range(0, count, block):
        blk = v[off:off + block].astype(np.int64)
        n = len(blk)
        aligned = (n + 15) & ~15
        control, data = bytearray(stride // 4), bytearray()
        for k in range(0, stride, 4):
            ch = channels[k // 4]
            mode, rot = ch & 3, ch >> 4
            planes = np.zeros((4, n), np.int64)
            if mode == 0:
                for j in range(4):
                    col = blk[:, k + j]
                    prev = np.concatenate([[int(last[k + j])], col[:-1]])
                    d = ((col - prev + 128) & 0xFF) - 128
                    planes[j] = ((d << 1) ^ (d >> 63)) & 0xFF
            elif mode == 1:
                for h in (0, 2):
                    col = blk[:, k + h] | (blk[:, k + h + 1] << 8)
                    prev = np.concatenate([[int(last[k + h]) | (int(last[k + h + 1]) << 8)], col[:-1]])
                    d = ((col - prev + 32768) & 0xFFFF) - 32768
                    z = ((d << 1) ^ (d >> 63)) & 0xFFFF
                    planes[h], planes[h + 1] = z & 0xFF, z >> 8
            else:
                col = blk[:, k] | (blk[:, k + 1] << 8) | (blk[:, k + 2] << 16) | (blk[:, k + 3] << 24)
                l32 = int(last[k]) | (int(last[k + 1]) << 8) | (int(last[k + 2]) << 16) | (int(last[k + 3]) << 24)
                prev = np.concatenate([[l32], col[:-1]])
                x = (col ^ prev) & 0xFFFFFFFF
                x = ((x << rot) | (x >> ((32 - rot) & 31))) & 0xFFFFFFFF if rot else x
                for j in range(4):
                    planes[j] = (x >> (8 * j)) & 0xFF
            for j in range(4):
                z = np.concatenate([planes[j], np.zeros(aligned - n, np.int64)])
                if not z.any():
                    ctrl, enc = 2, b""
                else:
                    options = [(len(e), c, e) for c, e in ((0, _encode_plane(z, (0, 1, 2, 4))), (1, _encode_plane(z, (1, 2, 4, 8))), (3, bytes(int(x) for x in planes[j])))]
                    _, ctrl, enc = min(options, key=lambda o: (o[0], o[1]))
                control[k // 4] |= ctrl << (j * 2)
                data += enc
        out += control + data
        last = v[off + n - 1].copy()
    used = stride + stride // 4
    out += bytes(max(24, used) - used) + bytes(first) + bytes(channels)
    return bytes(out)


# ---- mode TRIANGLES ------------------------------------------------------------------------------------------------------------------------
def _varint(v):
    out = bytearray()
    while True:
        if v < 128:
            out.append(v)
            return out
        out.append((v & 127) | 128)
        v >>= 7


def _free_index(value, last):
    d = (value - last) & 0xFFFFFFFF
    d = d - (1 << 32) if d >= (1 << 31) else d       # signed 32-bit difference
    z = ((d << 1) ^ (d >> 31)) & 0xFFFFFFFF
    return _varint(z)


TABLE = bytes([0x00, 0x76, 0x87, 0x56, 0x67, 0x78, 0xA9, 0x86, 0x65, 0x89, 0x68, 0x98, 0x01, 0x69, 0x00, 0x00])


def encode_triangles(indices, version=1, table=TABLE):
    """indices: flat sequence, a multiple of 3 -> bytes.  The codec may rotate a triangle's corners (never its winding)."""
    idx = [int(x) for x in indices]
    assert len(idx) % 3 == 0
    codes, data = bytearray(), bytearray()
    edges = [(-1, -1)] * 16
    recent = [-1] * 16
    state = {"edge": 0, "vert": 0}
    nxt, last = 0, 0
    first_free = 13 if version >= 1 else 15

    def push_edge(a, b):
        edges[state["edge"]] = (a, b)
        state["edge"] = (state["edge"] + 1) & 15

    def push_vertex(v, keep=True):
        recent[state["vert"]] = v
        state["vert"] = (state["vert"] + (1 if keep else 0)) & 15

    def find_vertex(v):
        for i in range(16):
            if recent[(state["vert"] - 1 - i) & 15] == v:
                return i
        return -1

    def find_edge(a, b, c):
        for i in range(16):
            e = edges[(state["edge"] - 1 - i) & 15]
            if e == (a, b):
                return i, 0
            if e == (b, c):
                return i, 1
            if e == (c, a):
                return i, 2
        return -1, 0

    for t in range(0, len(idx), 3):
        tri = idx[t:t + 3]
        fe, rot = find_edge(*tri)
        if 0 <= fe < 15:
            a, b, c = tri[rot], tri[(rot + 1) % 3], tri[(rot + 2) % 3]
            fc = find_vertex(c)
            if 1 <= fc < first_free:
                code = fc
            elif c == nxt:
                code = 0
                nxt += 1
            else:
                code = 15
            if code == 15 and version >= 1:
                if c + 1 == last:
                    code, last = 13, c
                elif c == last + 1:
                    code, last = 14, c
            codes.append((fe << 4) | code)
            if code == 15:
                data += _free_index(c, last)
                last = c
            if code == 0 or code >= first_free:
                push_vertex(c)
            push_edge(c, b)
            push_edge(a, c)
        else:
            rot = 1 if tri[1] == nxt else (2 if tri[2] == nxt else 0)
            a, b, c = tri[rot], tri[(rot + 1) % 3], tri[(rot + 2) % 3]
            reset = False
            if version >= 1 and (a, b, c) == (0, 1, 2) and nxt > 0:
                reset, nxt = True, 0
                for i in range(16):
                    recent[i] = -1
            fb, fc = find_vertex(b), find_vertex(c)
            if a == nxt:
                fea = 0
                nxt += 1
            else:
                fea = 15
            if 0 <= fb < 14:
                feb = fb + 1
            elif b == nxt:
                feb = 0
                nxt += 1
            else:
                feb = 15
            if 0 <= fc < 14:
                fec = fc + 1
            elif c == nxt:
                fec = 0
                nxt += 1
            else:
                fec = 15
            pair = (feb << 4) | fec
            slot = table.find(bytes([pair]), 0, 14) if (feb != 15 and fec != 15) else -1
            if fea == 0 and slot >= 0 and not reset:
                codes.append(0xF0 | slot)
            else:
                codes.append(0xFE if fea == 0 else 0xFF)
                data.append(pair)
                assert not (pair == 0 and not reset and fea == 0), "three fresh corners are in the table"
            for f, val in ((fea, a), (feb, b), (fec, c)):
                if f == 15:
                    data += _free_index(val, last)
                    last = val
            push_vertex(a)
            push_vertex(b, feb in (0, 15))
            push_vertex(c, fec in (0, 15))
            push_edge(b, a)
            push_edge(c, b)
            push_edge(a, c)
    return bytes([0xE0 | version]) + bytes(codes) + bytes(data) + bytes(table)


# ---- mode INDICES --------------------------------------------------------------------------------------------------------------------------
def encode_sequence(indices, version=1):
    out = bytearray([0xD0 | version])
    last = [0, 0]
    current = 0
    for v in (int(x) for x in indices):
        # the baseline that is nearer (a large jump switches to the other one)
        cd = abs(v - last[current])
        if cd >= 30 and abs(v - last[1 - current]) < cd:
            current = 1 - current
        d = v - last[current]
        z = ((d << 1) ^ (d >> 63)) & 0xFFFFFFFF
        out += _varint((z << 1) | current)
        last[current] = v
    return bytes(out) + bytes(4)


# ---- filters (what an encoder stores; the decoder turns it back into normalised integers / floats) ---------------------------------------------
def filter_oct_encode(n, bits, stride):
    """unit vectors (count, 3) (+ an optional 4th column kept as is) -> int8 / int16 (count, 4) of `bits` precision"""
    n = np.asarray(n, np.float64)
    v, w = n[:, :3], (n[:, 3] if n.shape[1] > 3 else np.zeros(len(n)))
    s = 1.0 / np.abs(v).sum(1)
    x, y, z = v[:, 0] * s, v[:, 1] * s, v[:, 2] * s
    u = np.where(z >= 0, x, (1 - np.abs(y)) * np.where(x >= 0, 1.0, -1.0))
    t = np.where(z >= 0, y, (1 - np.abs(x)) * np.where(y >= 0, 1.0, -1.0))
    scale = float((1 << (bits - 1)) - 1)
    q = np.stack([np.rint(u * scale), np.rint(t * scale), np.full(len(n), scale), np.rint(w * scale)], 1)
    return q.astype(np.int8 if stride == 4 else np.int16)


def filter_quat_encode(q, bits):
    """unit quaternions (count, 4) -> int16 (count, 4)"""
    q = np.asarray(q, np.float64)
    out = np.zeros((len(q), 4), np.int16)
    scale = float((1 << (bits - 1)) - 1)
    for i, r in enumerate(q):
        qc = int(np.argmax(np.abs(r)))
        sign = -1.0 if r[qc] < 0 else 1.0
        for k in range(3):
            out[i, k] = int(np.rint(r[(qc + 1 + k) & 3] * np.sqrt(2.0) * sign * scale))
        out[i, 3] = (int(scale) & ~3) | qc
    return out


def filter_exp_encode(f, bits):
    """floats (count, k) -> uint32 (count, k): 24-bit signed mantissa of `bits` significant bits, 8-bit signed exponent"""
    f = np.asarray(f, np.float32)
    out = np.zeros(f.shape, np.uint32)
    for idx in np.ndindex(f.shape):
        v = float(f[idx])
        _, e = np.frexp(v)
        exp = max(int(e) - (bits - 1), -100)
        m = int(np.rint(v * 2.0 ** (-exp)))
        out[idx] = (m & 0xFFFFFF) | ((exp & 0xFF) << 24)
    return out
