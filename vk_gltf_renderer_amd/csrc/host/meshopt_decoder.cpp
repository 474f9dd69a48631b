// EXT_meshopt_compression / KHR_meshopt_compression bitstream decoders (see meshopt_decoder.hpp).  The three stream layouts, as the extension
// specifies them:
//
// ATTRIBUTES  [0xA0 | version] [blocks ...] [tail: max(32, stride) bytes, the FIRST vertex in its last `stride` bytes]      (version 0; version 1: see decodeVertexBuffer)
//   Vertices come in blocks of min(256, (8192 / stride) & ~15); inside a block the data is stored byte plane by byte plane (all vertices' byte 0,
//   then byte 1, ...).  A plane is the zigzag-coded difference of every byte to the same byte of the previous vertex (the tail's vertex before the
//   first one), cut into groups of 16: two header bits per group (packed four groups to a byte in front of the plane) say whether the group is all
//   zero, 2-bit values (4 bytes), 4-bit values (8 bytes) -- in both, the all-ones value is an escape whose real byte follows the packed part, in
//   order -- or 16 plain bytes.
// TRIANGLES   [0xE0 | version] [one code byte per triangle] [data bytes ...] [tail: 16-byte table of common (b, c) codes]
//   A triangle reuses one of the last 16 edges (code high nibble) or starts fresh (high nibble 15); its remaining corners are the next unseen
//   index, one of the last 16 vertices, or a free index stored in the data bytes as a zigzag varint difference to the previous free index
//   (version 1 adds the two shortcuts "previous - 1" / "previous + 1" and a restart code).
// INDICES     [0xD0 | version] [varints ...] [4 zero bytes]
//   Every index is a zigzag varint difference to one of two running baselines, chosen by the varint's lowest bit.
#include "meshopt_decoder.hpp"

#include <cmath>
#include <cstring>

namespace meshopt {

namespace {

bool fail(std::string& err, const char* what)
{
  err = std::string("meshopt: ") + what;
  return false;
}

// ---- ATTRIBUTES ---------------------------------------------------------------------------------------------------------------------
constexpr size_t GROUP = 16, BLOCK_BYTES = 8192, BLOCK_MAX = 256, TAIL_MIN = 32;

// one group of 16 values with `bits` = 1, 2 or 4 per packed value, most significant value first; returns the position after the group or null
const uint8_t* unpackGroup(const uint8_t* p, const uint8_t* end, uint8_t* out, int bits)
{
  const size_t packed = GROUP * size_t(bits) / 8;
  if(size_t(end - p) < packed)
    return nullptr;
  const uint8_t* extra    = p + packed;
  const unsigned sentinel = (1u << bits) - 1u;
  const int      perByte  = 8 / bits;
  for(size_t b = 0; b < packed; ++b)
  {
    unsigned byte = p[b];
    for(int k = 0; k < perByte; ++k)
    {
      const unsigned v = (byte >> (8 - bits)) & sentinel;
      byte             = (byte << bits) & 0xffu;
      if(v == sentinel)
      {
        if(extra >= end)
          return nullptr;
        *out++ = *extra++;
      }
      else
        *out++ = uint8_t(v);
    }
  }
  return extra;
}

// a byte plane of `alignedCount` values in groups of 16; the two header bits of a group index `widths` (bits per value: 0 = all zero, 8 = plain bytes)
const uint8_t* decodePlane(const uint8_t* p, const uint8_t* end, uint8_t* out, size_t alignedCount, const int widths[4])
{
  const size_t groups      = alignedCount / GROUP;
  const size_t headerBytes = (groups + 3) / 4;
  if(size_t(end - p) < headerBytes)
    return nullptr;
  const uint8_t* header = p;
  p += headerBytes;
  for(size_t g = 0; g < groups; ++g)
  {
    const int bits = widths[(header[g / 4] >> ((g % 4) * 2)) & 3];
    uint8_t*  o    = out + g * GROUP;
    if(bits == 0)
      memset(o, 0, GROUP);
    else if(bits == 8)
    {
      if(size_t(end - p) < GROUP)
        return nullptr;
      memcpy(o, p, GROUP);
      p += GROUP;
    }
    else
    {
      p = unpackGroup(p, end, o, bits);
      if(!p)
        return nullptr;
    }
  }
  return p;
}

// ---- TRIANGLES / INDICES ---------------------------------------------------------------------------------------------------------------
// up to five bytes, seven bits each, least significant group first; the caller guarantees five readable bytes
uint32_t readVarint(const uint8_t*& p)
{
  uint32_t v = 0;
  for(int i = 0; i < 5; ++i)
  {
    const uint8_t b = *p++;
    v |= uint32_t(b & 127u) << (7 * i);
    if(b < 128u)
      break;
  }
  return v;
}
uint32_t readFreeIndex(const uint8_t*& p, uint32_t previous)
{
  const uint32_t v = readVarint(p);
  return previous + ((v >> 1) ^ (0u - (v & 1u)));
}
void store(uint8_t* dst, size_t i, size_t stride, uint32_t v)
{
  if(stride == 2)
  {
    const uint16_t s = uint16_t(v);
    memcpy(dst + i * 2, &s, 2);
  }
  else
    memcpy(dst + i * 4, &v, 4);
}

int roundSigned(float v) { return int(v + (v >= 0.0f ? 0.5f : -0.5f)); }

template <typename T>
void octahedral(T* d, size_t count)
{
  const float maxv = float((1 << (int(sizeof(T)) * 8 - 1)) - 1);
  for(size_t i = 0; i < count; ++i)
  {
    // x, y as stored; the third component holds the value that stands for 1.0, from which z follows
    float x = float(d[i * 4 + 0]), y = float(d[i * 4 + 1]);
    float z = float(d[i * 4 + 2]) - std::fabs(x) - std::fabs(y);
    // lower hemisphere: fold back
    const float t = z >= 0.0f ? 0.0f : z;
    x += x >= 0.0f ? t : -t;
    y += y >= 0.0f ? t : -t;
    const float len = std::sqrt(x * x + y * y + z * z);
    const float s   = maxv / len;
    d[i * 4 + 0]    = T(roundSigned(x * s));
    d[i * 4 + 1]    = T(roundSigned(y * s));
    d[i * 4 + 2]    = T(roundSigned(z * s));
  }
}

}  // namespace

bool decodeVertexBuffer(uint8_t* dst, size_t count, size_t stride, const uint8_t* src, size_t srcSize, std::string& err)
{
  if(stride == 0 || stride > 256 || stride % 4 != 0)
    return fail(err, "ATTRIBUTES: byteStride must be a multiple of 4 in [4, 256]");
  if(srcSize < 1)
    return fail(err, "ATTRIBUTES: empty stream");
  if((src[0] & 0xf0u) != 0xa0u)
    return fail(err, "ATTRIBUTES: not a vertex stream");
  const int version = src[0] & 0x0f;
  if(version > 1)
    return fail(err, "ATTRIBUTES: unknown vertex codec version");
  // Version 1 (KHR_meshopt_compression) keeps the layout and adds, per 4-byte component of the vertex, a CHANNEL byte in the tail -- how its
  // differences are formed: per byte, per 16-bit half, or as a rotated XOR of the 32-bit word -- and, at the head of every block, a CONTROL byte whose
  // two bits per byte plane pick the plane's coding: group widths {0, 1, 2, 4} or {1, 2, 4, 8}, all zero, or plain bytes.
  const size_t channelBytes = version == 0 ? 0 : stride / 4;
  const size_t tailUsed = stride + channelBytes, tailMin = version == 0 ? 32 : 24;
  const size_t tail     = tailUsed < tailMin ? tailMin : tailUsed;
  if(srcSize < 1 + tail)
    return fail(err, "ATTRIBUTES: stream shorter than its header and tail");
  uint8_t        previous[256];
  const uint8_t* channels = src + srcSize - channelBytes;
  memcpy(previous, src + srcSize - tailUsed, stride);
  for(size_t c = 0; c < channelBytes; ++c)
    if((channels[c] & 3u) == 3u)
      return fail(err, "ATTRIBUTES: unknown channel mode");
  size_t blockMax = (BLOCK_BYTES / stride) & ~(GROUP - 1);
  if(blockMax > BLOCK_MAX)
    blockMax = BLOCK_MAX;
  static const int widthsV0[4] = {0, 2, 4, 8}, widthsV1[5] = {0, 1, 2, 4, 8};
  const uint8_t *p = src + 1, *end = src + srcSize;
  uint8_t        planes[4][BLOCK_MAX];
  for(size_t first = 0; first < count; first += blockMax)
  {
    const size_t n       = count - first < blockMax ? count - first : blockMax;
    const size_t aligned = (n + GROUP - 1) & ~(GROUP - 1);
    uint8_t*     out     = dst + first * stride;
    if(size_t(end - p) < channelBytes)
      return fail(err, "ATTRIBUTES: stream ends inside a block");
    const uint8_t* control = p;
    p += channelBytes;
    for(size_t k = 0; k < stride; k += 4)
    {
      for(int j = 0; j < 4; ++j)
      {
        const int ctrl = version == 0 ? 0 : (control[k / 4] >> (j * 2)) & 3;
        if(version != 0 && ctrl == 3)
        {
          if(size_t(end - p) < n)
            return fail(err, "ATTRIBUTES: stream ends inside a block");
          memcpy(planes[j], p, n);
          p += n;
        }
        else if(version != 0 && ctrl == 2)
          memset(planes[j], 0, n);
        else
        {
          p = decodePlane(p, end, planes[j], aligned, version == 0 ? widthsV0 : widthsV1 + ctrl);
          if(!p)
            return fail(err, "ATTRIBUTES: stream ends inside a block");
        }
      }
      const unsigned channel = version == 0 ? 0u : channels[k / 4];
      const unsigned mode    = channel & 3u;
      if(mode == 0)
      {
        for(int j = 0; j < 4; ++j)
        {
          uint8_t prev = previous[k + size_t(j)];
          for(size_t i = 0; i < n; ++i)
          {
            const uint8_t z = planes[j][i];
            prev            = uint8_t(uint8_t((z >> 1) ^ (0u - (z & 1u))) + prev);
            out[i * stride + k + size_t(j)] = prev;
          }
        }
      }
      else if(mode == 1)
      {
        for(int h = 0; h < 4; h += 2)
        {
          uint16_t prev = uint16_t(previous[k + size_t(h)] | (previous[k + size_t(h) + 1] << 8));
          for(size_t i = 0; i < n; ++i)
          {
            const uint16_t z = uint16_t(planes[h][i] | (planes[h + 1][i] << 8));
            prev             = uint16_t(uint16_t((z >> 1) ^ (0u - (z & 1u))) + prev);
            out[i * stride + k + size_t(h)]     = uint8_t(prev);
            out[i * stride + k + size_t(h) + 1] = uint8_t(prev >> 8);
          }
        }
      }
      else
      {
        const unsigned rot = (32u - (channel >> 4)) & 31u;  // the encoder rotated the XOR left by channel >> 4
        uint32_t       prev = uint32_t(previous[k]) | (uint32_t(previous[k + 1]) << 8) | (uint32_t(previous[k + 2]) << 16) | (uint32_t(previous[k + 3]) << 24);
        for(size_t i = 0; i < n; ++i)
        {
          const uint32_t d = uint32_t(planes[0][i]) | (uint32_t(planes[1][i]) << 8) | (uint32_t(planes[2][i]) << 16) | (uint32_t(planes[3][i]) << 24);
          prev             = ((d << rot) | (d >> ((32u - rot) & 31u))) ^ prev;
          for(int j = 0; j < 4; ++j)
            out[i * stride + k + size_t(j)] = uint8_t(prev >> (8 * j));
        }
      }
    }
    memcpy(previous, out + (n - 1) * stride, stride);
  }
  if(size_t(end - p) != tail)
    return fail(err, "ATTRIBUTES: the blocks do not end where the tail begins");
  return true;
}

bool decodeIndexBuffer(uint8_t* dst, size_t count, size_t stride, const uint8_t* src, size_t srcSize, std::string& err)
{
  if(count % 3 != 0 || (stride != 2 && stride != 4))
    return fail(err, "TRIANGLES: count must be a multiple of 3 and byteStride 2 or 4");
  if(srcSize < 1 + count / 3 + 16)
    return fail(err, "TRIANGLES: stream shorter than its codes and table");
  if((src[0] & 0xf0u) != 0xe0u)
    return fail(err, "TRIANGLES: not an index stream");
  const int version = src[0] & 0x0f;
  if(version > 1)
    return fail(err, "TRIANGLES: unknown index codec version");
  const int      firstFree = version >= 1 ? 13 : 15;  // corner codes from here on are not FIFO references
  uint32_t       edges[16][2], recent[16];
  memset(edges, 0xff, sizeof(edges));
  memset(recent, 0xff, sizeof(recent));
  size_t         edgeAt = 0, recentAt = 0;
  uint32_t       next = 0, lastFree = 0;
  const uint8_t* code    = src + 1;
  const uint8_t* data    = code + count / 3;
  const uint8_t* dataEnd = src + srcSize - 16;  // the table; a triangle reads at most 16 data bytes, so checking against it keeps every read inside
  const uint8_t* table   = dataEnd;
  auto pushEdge   = [&](uint32_t a, uint32_t b) { edges[edgeAt][0] = a; edges[edgeAt][1] = b; edgeAt = (edgeAt + 1) & 15; };
  auto pushVertex = [&](uint32_t v, bool keep = true) { recent[recentAt] = v; recentAt = (recentAt + (keep ? 1 : 0)) & 15; };
  for(size_t i = 0; i < count; i += 3)
  {
    if(data > dataEnd)
      return fail(err, "TRIANGLES: data bytes run into the table");
    const unsigned tri = *code++;
    uint32_t       a, b, c;
    if(tri < 0xf0u)
    {
      // an edge of the last sixteen, plus one corner
      const size_t e = (edgeAt - 1 - (tri >> 4)) & 15;
      a              = edges[e][0];
      b              = edges[e][1];
      const int cc   = int(tri & 15u);
      if(cc < firstFree)
      {
        const bool fresh = cc == 0;
        c                = fresh ? next : recent[(recentAt - 1 - size_t(cc)) & 15];
        next += fresh ? 1u : 0u;
        pushVertex(c, fresh);
      }
      else
      {
        c = lastFree = (cc != 15) ? lastFree + uint32_t(cc - (cc ^ 3)) : readFreeIndex(data, lastFree);  // 13 -> previous - 1, 14 -> previous + 1
        pushVertex(c);
      }
      pushEdge(c, b);
      pushEdge(a, c);
    }
    else
    {
      if(tri < 0xfeu)
      {
        // a fresh first corner, the other two from the stream's table of common pairs (no free indices there)
        const unsigned pair = table[tri & 15u];
        const int      bc = int(pair >> 4), cc = int(pair & 15u);
        a                 = next++;
        const bool bFresh = bc == 0;
        b                 = bFresh ? next : recent[(recentAt - size_t(bc)) & 15];
        next += bFresh ? 1u : 0u;
        const bool cFresh = cc == 0;
        c                 = cFresh ? next : recent[(recentAt - size_t(cc)) & 15];
        next += cFresh ? 1u : 0u;
        pushVertex(a);
        pushVertex(b, bFresh);
        pushVertex(c, cFresh);
      }
      else
      {
        // the pair spelled out in a data byte; 0xff: the first corner is a free index too; a zero pair restarts the numbering
        const unsigned pair = *data++;
        const int      ac = tri == 0xfeu ? 0 : 15, bc = int(pair >> 4), cc = int(pair & 15u);
        if(pair == 0)
          next = 0;
        a = ac == 0 ? next++ : 0u;
        b = bc == 0 ? next++ : recent[(recentAt - size_t(bc)) & 15];
        c = cc == 0 ? next++ : recent[(recentAt - size_t(cc)) & 15];
        if(ac == 15)
          a = lastFree = readFreeIndex(data, lastFree);
        if(bc == 15)
          b = lastFree = readFreeIndex(data, lastFree);
        if(cc == 15)
          c = lastFree = readFreeIndex(data, lastFree);
        pushVertex(a);
        pushVertex(b, bc == 0 || bc == 15);
        pushVertex(c, cc == 0 || cc == 15);
      }
      pushEdge(b, a);
      pushEdge(c, b);
      pushEdge(a, c);
    }
    store(dst, i, stride, a);
    store(dst, i + 1, stride, b);
    store(dst, i + 2, stride, c);
  }
  if(data != dataEnd)
    return fail(err, "TRIANGLES: the data bytes do not end where the table begins");
  return true;
}

bool decodeIndexSequence(uint8_t* dst, size_t count, size_t stride, const uint8_t* src, size_t srcSize, std::string& err)
{
  if(stride != 2 && stride != 4)
    return fail(err, "INDICES: byteStride must be 2 or 4");
  if(srcSize < 1 + count + 4)
    return fail(err, "INDICES: stream shorter than one byte per index and its tail");
  if((src[0] & 0xf0u) != 0xd0u)
    return fail(err, "INDICES: not an index sequence");
  if((src[0] & 0x0fu) > 1)
    return fail(err, "INDICES: unknown codec version");
  const uint8_t* data    = src + 1;
  const uint8_t* dataEnd = src + srcSize - 4;  // a varint reads at most 5 bytes: starting before the 4-byte tail keeps it inside
  uint32_t       baseline[2] = {0, 0};
  for(size_t i = 0; i < count; ++i)
  {
    if(data >= dataEnd)
      return fail(err, "INDICES: varints run into the tail");
    uint32_t       v     = readVarint(data);
    const unsigned which = v & 1u;
    v >>= 1;
    baseline[which] += (v >> 1) ^ (0u - (v & 1u));
    store(dst, i, stride, baseline[which]);
  }
  if(data != dataEnd)
    return fail(err, "INDICES: the varints do not end where the tail begins");
  return true;
}

bool filterOctahedral(uint8_t* data, size_t count, size_t stride, std::string& err)
{
  if(stride == 4)
    octahedral(reinterpret_cast<int8_t*>(data), count);
  else if(stride == 8)
    octahedral(reinterpret_cast<int16_t*>(data), count);
  else
    return fail(err, "OCTAHEDRAL filter: byteStride must be 4 or 8");
  return true;
}

bool filterQuaternion(uint8_t* data, size_t count, size_t stride, std::string& err)
{
  if(stride != 8)
    return fail(err, "QUATERNION filter: byteStride must be 8");
  int16_t*    d     = reinterpret_cast<int16_t*>(data);
  const float scale = 1.0f / std::sqrt(2.0f);
  for(size_t i = 0; i < count; ++i)
  {
    // the last component carries the range of the other three (its low two bits: which component was dropped)
    const int   range = d[i * 4 + 3] | 3;
    const float ss    = scale / float(range);
    const float x = float(d[i * 4 + 0]) * ss, y = float(d[i * 4 + 1]) * ss, z = float(d[i * 4 + 2]) * ss;
    const float ww = 1.0f - x * x - y * y - z * z;
    const float w  = std::sqrt(ww >= 0.0f ? ww : 0.0f);
    const int   xf = roundSigned(x * 32767.0f), yf = roundSigned(y * 32767.0f), zf = roundSigned(z * 32767.0f), wf = int(w * 32767.0f + 0.5f);
    const int   dropped = d[i * 4 + 3] & 3;
    d[i * 4 + ((dropped + 1) & 3)] = int16_t(xf);
    d[i * 4 + ((dropped + 2) & 3)] = int16_t(yf);
    d[i * 4 + ((dropped + 3) & 3)] = int16_t(zf);
    d[i * 4 + ((dropped + 0) & 3)] = int16_t(wf);
  }
  return true;
}

bool filterExponential(uint8_t* data, size_t count, size_t stride, std::string& err)
{
  if(stride == 0 || stride % 4 != 0)
    return fail(err, "EXPONENTIAL filter: byteStride must be a multiple of 4");
  const size_t words = count * (stride / 4);
  for(size_t i = 0; i < words; ++i)
  {
    uint32_t v;
    memcpy(&v, data + i * 4, 4);
    const int32_t  m = int32_t(v << 8) >> 8;  // 24-bit signed mantissa
    const int32_t  e = int32_t(v) >> 24;      // 8-bit signed exponent
    const uint32_t pw = uint32_t(e + 127) << 23;
    float          p2;
    memcpy(&p2, &pw, 4);
    const float f = p2 * float(m);
    memcpy(data + i * 4, &f, 4);
  }
  return true;
}

}  // namespace meshopt
