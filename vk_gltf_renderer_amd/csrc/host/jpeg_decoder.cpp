// JPEG (ITU-T T.81) decode for the glTF front end: baseline / extended-sequential and progressive Huffman, 8 bit, 1 or 3
// components, any sampling factors, restart intervals -> RGBA8.
//
// The reference gets its JPEG pixels from stb_image (src/gltf_image_loader.cpp:163-236 `loadStb`; the library is a third-party
// dependency that is not part of the reference tree, version unpinned).  The entropy decoding below follows T.81 (any correct
// decoder produces the same coefficients); the three places where decoders legitimately differ -- the inverse DCT, the chroma
// upsampling and the YCbCr -> RGB conversion -- restate stb_image's published integer arithmetic so that texels match what the
// reference uploads: 12-bit fixed-point "islow"-style IDCT with the column pass kept at 2 extra bits, (3 near + 1 far)
// triangle upsampling, 20-bit fixed-point colour conversion.
#include "image_loader.hpp"

#include <algorithm>
#include <cstring>

namespace mihost {

namespace {

const uint8_t kZigzag[64 + 15] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13,
                                  6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31,
                                  39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
                                  // run past the end on corrupt data lands here
                                  63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};

struct Huffman
{
  bool    present = false;
  uint8_t vals[256];
  int     mincode[17], maxcode[18], valptr[17];  // per code length, T.81 F.2.2.3
};

struct Component
{
  int                  id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
  int                  dcPred = 0;
  int                  blocksW = 0, blocksH = 0;    // blocks covering the component itself (non-interleaved scans)
  int                  paddedW = 0, paddedH = 0;    // blocks covering whole MCUs (storage)
  std::vector<int16_t> coef;                        // paddedW * paddedH * 64, natural order
  std::vector<uint8_t> plane;                       // paddedW*8 x paddedH*8 samples
};

struct Decoder
{
  const uint8_t* p;
  const uint8_t* end;
  size_t         fileSize = 0;
  std::string*   error;
  uint16_t       quant[4][64] = {};  // natural order
  Huffman        dc[4], ac[4];
  int            width = 0, height = 0, ncomp = 0;
  bool           progressive = false;
  Component      comp[4];
  int            hmax = 1, vmax = 1, mcusX = 0, mcusY = 0;
  int            restartInterval = 0;
  // entropy-coded segment reader
  uint32_t bitbuf = 0;
  int      bitcnt = 0;
  bool     marker = false;  // ran into a marker: feed zeros
  int      eobrun = 0;

  bool fail(const char* msg)
  {
    if(error)
      *error = std::string("JPEG: ") + msg;
    return false;
  }

  // ---- bits --------------------------------------------------------------------------------------------------------------
  void fill()
  {
    while(bitcnt <= 24)
    {
      uint32_t b = 0;
      if(!marker && p < end)
      {
        b = *p;
        if(b == 0xff)
        {
          if(p + 1 < end && p[1] == 0x00)
            p += 2;  // stuffed zero
          else
          {
            marker = true;  // RSTn / EOI / next segment: stays at p
            b      = 0;
          }
        }
        else
          ++p;
      }
      else
        marker = true;
      bitbuf |= b << (24 - bitcnt);
      bitcnt += 8;
    }
  }
  int getBits(int n)
  {
    if(n == 0)
      return 0;
    if(bitcnt < n)
      fill();
    int v = int(bitbuf >> (32 - n));
    bitbuf <<= n;
    bitcnt -= n;
    return v;
  }
  int getBit() { return getBits(1); }
  void resetBits()
  {
    bitbuf = 0;
    bitcnt = 0;
    marker = false;
    eobrun = 0;
  }
  int decodeHuff(const Huffman& h)
  {
    int code = 0;
    for(int len = 1; len <= 16; ++len)
    {
      code = (code << 1) | getBit();
      if(code <= h.maxcode[len] && h.maxcode[len] >= 0)
        return h.vals[h.valptr[len] + code - h.mincode[len]];
    }
    return -1;
  }
  static int extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }  // T.81 F.2.2.1
  int receiveExtend(int s) { return s ? extend(getBits(s), s) : 0; }

  // ---- tables ------------------------------------------------------------------------------------------------------------
  bool readDqt(const uint8_t* s, int len)
  {
    while(len > 0)
    {
      int pq = s[0] >> 4, tq = s[0] & 15;
      if(tq > 3 || pq > 1 || len < 1 + 64 * (pq + 1))
        return fail("bad DQT");
      for(int i = 0; i < 64; ++i)
        quant[tq][kZigzag[i]] = pq ? uint16_t((s[1 + 2 * i] << 8) | s[2 + 2 * i]) : s[1 + i];
      s += 1 + 64 * (pq + 1);
      len -= 1 + 64 * (pq + 1);
    }
    return true;
  }
  bool readDht(const uint8_t* s, int len)
  {
    while(len > 0)
    {
      if(len < 17)
        return fail("bad DHT");
      int tc = s[0] >> 4, th = s[0] & 15;
      if(tc > 1 || th > 3)
        return fail("bad DHT id");
      Huffman& h     = tc ? ac[th] : dc[th];
      int      total = 0;
      for(int i = 1; i <= 16; ++i)
        total += s[i];
      if(total > 256 || len < 17 + total)
        return fail("bad DHT size");
      std::memcpy(h.vals, s + 17, size_t(total));
      int code = 0, k = 0;
      for(int l = 1; l <= 16; ++l)
      {
        h.valptr[l]  = k;
        h.mincode[l] = code;
        k += s[l];
        code += s[l];
        h.maxcode[l] = s[l] ? code - 1 : -1;
        code <<= 1;
      }
      h.maxcode[17] = 0x7fffffff;
      h.present     = true;
      s += 17 + total;
      len -= 17 + total;
    }
    return true;
  }
  bool readSof(const uint8_t* s, int len, bool prog)
  {
    if(len < 6 || s[0] != 8)
      return fail("only 8-bit samples are supported");
    height = (s[1] << 8) | s[2];
    width  = (s[3] << 8) | s[4];
    ncomp  = s[5];
    // (the coefficient and plane arrays below are sized by these: an entropy-coded 8x8 block takes at least a few bits, so a file
    // of `fileSize` bytes cannot describe more than some hundred pixels per byte)
    if(width <= 0 || height <= 0 || !saneImageSize(uint64_t(width), uint64_t(height)) || uint64_t(width) * uint64_t(height) > uint64_t(fileSize) * 512ull + 65536ull)
      return fail("bad dimensions");
    if((ncomp != 1 && ncomp != 3 && ncomp != 4) || len < 6 + 3 * ncomp)
      return fail("only 1-, 3- or 4-component images are supported");
    progressive = prog;
    for(int i = 0; i < ncomp; ++i)
    {
      Component& c = comp[i];
      c.id         = s[6 + 3 * i];
      c.h          = s[7 + 3 * i] >> 4;
      c.v          = s[7 + 3 * i] & 15;
      c.tq         = s[8 + 3 * i];
      if(c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3)
        return fail("bad component");
      hmax = std::max(hmax, c.h);
      vmax = std::max(vmax, c.v);
    }
    mcusX = (width + 8 * hmax - 1) / (8 * hmax);
    mcusY = (height + 8 * vmax - 1) / (8 * vmax);
    for(int i = 0; i < ncomp; ++i)
    {
      Component& c = comp[i];
      c.blocksW    = ((width * c.h + hmax - 1) / hmax + 7) / 8;
      c.blocksH    = ((height * c.v + vmax - 1) / vmax + 7) / 8;
      c.paddedW    = mcusX * c.h;
      c.paddedH    = mcusY * c.v;
      c.coef.assign(size_t(c.paddedW) * c.paddedH * 64, 0);
    }
    return true;
  }

  // ---- blocks ------------------------------------------------------------------------------------------------------------
  bool decodeBlockSequential(Component& c, int16_t* b)
  {
    const Huffman &hd = dc[c.td], &ha = ac[c.ta];
    int            t  = decodeHuff(hd);
    if(t < 0 || t > 15)
      return fail("bad DC code");
    c.dcPred += receiveExtend(t);
    b[0] = int16_t(c.dcPred);
    for(int k = 1; k < 64;)
    {
      int rs = decodeHuff(ha);
      if(rs < 0)
        return fail("bad AC code");
      int r = rs >> 4, s = rs & 15;
      if(s == 0)
      {
        if(r != 15)
          break;
        k += 16;
      }
      else
      {
        k += r;
        b[kZigzag[k]] = int16_t(receiveExtend(s));
        ++k;
      }
    }
    return true;
  }
  bool decodeBlockDcProgressive(Component& c, int16_t* b, int ah, int al)
  {
    if(ah == 0)
    {
      int t = decodeHuff(dc[c.td]);
      if(t < 0 || t > 15)
        return fail("bad DC code");
      c.dcPred += receiveExtend(t);
      b[0] = int16_t(c.dcPred * (1 << al));
    }
    else if(getBit())
      b[0] = int16_t(b[0] + (1 << al));  // stb and libjpeg: |= for the positive case, add for two's complement in general
    return true;
  }
  bool decodeBlockAcProgressive(Component& c, int16_t* b, int ss, int se, int ah, int al)
  {
    const Huffman& ha = ac[c.ta];
    if(ah == 0)
    {
      if(eobrun)
      {
        --eobrun;
        return true;
      }
      for(int k = ss; k <= se;)
      {
        int rs = decodeHuff(ha);
        if(rs < 0)
          return fail("bad AC code");
        int r = rs >> 4, s = rs & 15;
        if(s == 0)
        {
          if(r < 15)
          {
            eobrun = (1 << r) - 1 + (r ? getBits(r) : 0);
            break;
          }
          k += 16;
        }
        else
        {
          k += r;
          b[kZigzag[k]] = int16_t(receiveExtend(s) * (1 << al));
          ++k;
        }
      }
      return true;
    }
    // refinement scan (T.81 G.1.2.3)
    const int p1 = 1 << al, m1 = -(1 << al);
    int       k  = ss;
    if(eobrun == 0)
    {
      for(; k <= se; ++k)
      {
        int rs = decodeHuff(ha);
        if(rs < 0)
          return fail("bad AC code");
        int r = rs >> 4, s = rs & 15, val = 0;
        if(s == 0)
        {
          if(r < 15)
          {
            eobrun = (1 << r) + (r ? getBits(r) : 0);
            break;
          }
        }
        else
          val = getBit() ? p1 : m1;  // s is 1 in a conforming stream
        while(k <= se)
        {
          int16_t& cf = b[kZigzag[k]];
          if(cf != 0)
          {
            if(getBit() && (cf & p1) == 0)
              cf = int16_t(cf + (cf >= 0 ? p1 : m1));
          }
          else
          {
            if(r == 0)
              break;
            --r;
          }
          ++k;
        }
        if(s && k <= se)
          b[kZigzag[k]] = int16_t(val);
      }
    }
    if(eobrun > 0)
    {
      for(; k <= se; ++k)
      {
        int16_t& cf = b[kZigzag[k]];
        if(cf != 0 && getBit() && (cf & p1) == 0)
          cf = int16_t(cf + (cf >= 0 ? p1 : m1));
      }
      --eobrun;
    }
    return true;
  }

  // consumes an RSTn marker if one is pending at p
  void restart()
  {
    while(p + 1 < end && !(p[0] == 0xff && p[1] >= 0xd0 && p[1] <= 0xd7))
    {
      if(p[0] == 0xff && p[1] != 0 && p[1] != 0xff)
        break;  // some other marker: leave it to the segment loop
      ++p;
    }
    if(p + 1 < end && p[0] == 0xff && p[1] >= 0xd0 && p[1] <= 0xd7)
      p += 2;
    resetBits();
    for(int i = 0; i < ncomp; ++i)
      comp[i].dcPred = 0;
  }

  bool readScan(const uint8_t* s, int len)
  {
    int ns = len >= 1 ? s[0] : 0;
    if(ns < 1 || ns > ncomp || len < 1 + 2 * ns + 3)
      return fail("bad SOS");
    Component* sc[4];
    for(int i = 0; i < ns; ++i)
    {
      sc[i] = nullptr;
      for(int j = 0; j < ncomp; ++j)
        if(comp[j].id == s[1 + 2 * i])
          sc[i] = &comp[j];
      if(!sc[i])
        return fail("SOS names an unknown component");
      sc[i]->td = s[2 + 2 * i] >> 4;
      sc[i]->ta = s[2 + 2 * i] & 15;
      if(sc[i]->td > 3 || sc[i]->ta > 3)
        return fail("bad table selector");
    }
    const int ss = s[1 + 2 * ns], se = s[2 + 2 * ns], ah = s[3 + 2 * ns] >> 4, al = s[3 + 2 * ns] & 15;
    if(progressive)
    {
      if(ss > 63 || se > 63 || ss > se || ah > 13 || al > 13 || (ss == 0 && se != 0) || (ss > 0 && ns != 1))
        return fail("bad progressive scan parameters");
    }
    for(int i = 0; i < ns; ++i)
    {
      const bool needDc = !progressive || ss == 0, needAc = !progressive || ss > 0;
      if((needDc && (!progressive || ah == 0) && !dc[sc[i]->td].present) || (needAc && !ac[sc[i]->ta].present))
        return fail("scan uses an undefined Huffman table");
      sc[i]->dcPred = 0;
    }
    resetBits();
    auto block = [&](Component& c, int bx, int by) -> bool {
      int16_t* b = c.coef.data() + (size_t(by) * c.paddedW + bx) * 64;
      if(!progressive)
        return decodeBlockSequential(c, b);
      return ss == 0 ? decodeBlockDcProgressive(c, b, ah, al) : decodeBlockAcProgressive(c, b, ss, se, ah, al);
    };
    int todo = restartInterval ? restartInterval : 0x7fffffff;
    if(ns == 1)
    {
      Component& c = *sc[0];
      for(int by = 0; by < c.blocksH; ++by)
        for(int bx = 0; bx < c.blocksW; ++bx)
        {
          if(!block(c, bx, by))
            return false;
          if(--todo <= 0)
          {
            restart();
            todo = restartInterval;
          }
        }
    }
    else
    {
      for(int my = 0; my < mcusY; ++my)
        for(int mx = 0; mx < mcusX; ++mx)
        {
          for(int i = 0; i < ns; ++i)
            for(int y = 0; y < sc[i]->v; ++y)
              for(int x = 0; x < sc[i]->h; ++x)
                if(!block(*sc[i], mx * sc[i]->h + x, my * sc[i]->v + y))
                  return false;
          if(--todo <= 0)
          {
            restart();
            todo = restartInterval;
          }
        }
    }
    return true;
  }

  // ---- inverse DCT: stb_image's integer transform (constants scaled by 4096) -----------------------------------------------
  static int f2f(double x) { return int(x * 4096 + 0.5); }
  static int fsh(int x) { return x * 4096; }
  static uint8_t clamp8(int x) { return uint8_t(x < 0 ? 0 : (x > 255 ? 255 : x)); }
  struct Idct1d
  {
    int x0, x1, x2, x3, t0, t1, t2, t3;
  };
  static Idct1d idct1d(int s0, int s1, int s2, int s3, int s4, int s5, int s6, int s7)
  {
    Idct1d o;
    int    p2 = s2, p3 = s6;
    int    p1 = (p2 + p3) * f2f(0.5411961);
    int    t2 = p1 + p3 * f2f(-1.847759065);
    int    t3 = p1 + p2 * f2f(0.765366865);
    p2        = s0;
    p3        = s4;
    int t0    = fsh(p2 + p3);
    int t1    = fsh(p2 - p3);
    o.x0      = t0 + t3;
    o.x3      = t0 - t3;
    o.x1      = t1 + t2;
    o.x2      = t1 - t2;
    t0        = s7;
    t1        = s5;
    t2        = s3;
    t3        = s1;
    p3        = t0 + t2;
    int p4    = t1 + t3;
    p1        = t0 + t3;
    p2        = t1 + t2;
    int p5    = (p3 + p4) * f2f(1.175875602);
    t0        = t0 * f2f(0.298631336);
    t1        = t1 * f2f(2.053119869);
    t2        = t2 * f2f(3.072711026);
    t3        = t3 * f2f(1.501321110);
    p1        = p5 + p1 * f2f(-0.899976223);
    p2        = p5 + p2 * f2f(-2.562915447);
    p3        = p3 * f2f(-1.961570560);
    p4        = p4 * f2f(-0.390180644);
    o.t3      = t3 + p1 + p4;
    o.t2      = t2 + p2 + p3;
    o.t1      = t1 + p2 + p4;
    o.t0      = t0 + p1 + p3;
    return o;
  }
  static void idctBlock(uint8_t* out, int stride, const int16_t* coef, const uint16_t* q)
  {
    int  val[64];
    int  d[64];
    for(int i = 0; i < 64; ++i)
      d[i] = int(int16_t(int(coef[i]) * int(q[i])));  // stb_image keeps dequantised coefficients in 16 bits
    for(int i = 0; i < 8; ++i)  // columns
    {
      const int* c = d + i;
      int*       v = val + i;
      if(c[8] == 0 && c[16] == 0 && c[24] == 0 && c[32] == 0 && c[40] == 0 && c[48] == 0 && c[56] == 0)
      {
        int dcterm = c[0] * 4;
        v[0] = v[8] = v[16] = v[24] = v[32] = v[40] = v[48] = v[56] = dcterm;
        continue;
      }
      Idct1d o = idct1d(c[0], c[8], c[16], c[24], c[32], c[40], c[48], c[56]);
      o.x0 += 512; o.x1 += 512; o.x2 += 512; o.x3 += 512;
      v[0]  = (o.x0 + o.t3) >> 10;
      v[56] = (o.x0 - o.t3) >> 10;
      v[8]  = (o.x1 + o.t2) >> 10;
      v[48] = (o.x1 - o.t2) >> 10;
      v[16] = (o.x2 + o.t1) >> 10;
      v[40] = (o.x2 - o.t1) >> 10;
      v[24] = (o.x3 + o.t0) >> 10;
      v[32] = (o.x3 - o.t0) >> 10;
    }
    for(int i = 0; i < 8; ++i)  // rows
    {
      const int* v = val + i * 8;
      uint8_t*   o8 = out + i * stride;
      Idct1d     o  = idct1d(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
      const int  bias = 65536 + (128 << 17);
      o.x0 += bias; o.x1 += bias; o.x2 += bias; o.x3 += bias;
      o8[0] = clamp8((o.x0 + o.t3) >> 17);
      o8[7] = clamp8((o.x0 - o.t3) >> 17);
      o8[1] = clamp8((o.x1 + o.t2) >> 17);
      o8[6] = clamp8((o.x1 - o.t2) >> 17);
      o8[2] = clamp8((o.x2 + o.t1) >> 17);
      o8[5] = clamp8((o.x2 - o.t1) >> 17);
      o8[3] = clamp8((o.x3 + o.t0) >> 17);
      o8[4] = clamp8((o.x3 - o.t0) >> 17);
    }
  }

  void reconstruct()
  {
    for(int i = 0; i < ncomp; ++i)
    {
      Component& c      = comp[i];
      const int  stride = c.paddedW * 8;
      c.plane.assign(size_t(stride) * c.paddedH * 8, 0);
      for(int by = 0; by < c.paddedH; ++by)
        for(int bx = 0; bx < c.paddedW; ++bx)
          idctBlock(c.plane.data() + size_t(by) * 8 * stride + bx * 8, stride, c.coef.data() + (size_t(by) * c.paddedW + bx) * 64, quant[c.tq]);
    }
  }

  // ---- chroma upsampling (stb_image's resamplers) ------------------------------------------------------------------------
  // one output row of `w` full-resolution samples of component c
  void upsampleRow(const Component& c, int y, uint8_t* out, std::vector<uint8_t>& tmp) const
  {
    const int      hs = hmax / c.h, vs = vmax / c.v;
    const int      stride = c.paddedW * 8;
    const int      wLo = (width + hs - 1) / hs, hLo = (height + vs - 1) / vs;
    const bool     exactH = hmax % c.h == 0, exactV = vmax % c.v == 0;
    const uint8_t* plane = c.plane.data();
    if(hs == 1 && vs == 1 && exactH && exactV)
    {
      std::memcpy(out, plane + size_t(y) * stride, size_t(width));
      return;
    }
    if(exactH && exactV && ((hs == 2 && (vs == 1 || vs == 2)) || (hs == 1 && vs == 2)))
    {
      // near / far low-resolution rows of this output row
      int k = y / vs;
      int nearRow = k, farRow = k;
      if(vs == 2)
        farRow = (y & 1) ? std::min(k + 1, hLo - 1) : std::max(k - 1, 0);
      const uint8_t* in_near = plane + size_t(nearRow) * stride;
      const uint8_t* in_far  = plane + size_t(farRow) * stride;
      if(hs == 1)  // vertical only
      {
        for(int i = 0; i < width; ++i)
          out[i] = uint8_t((3 * in_near[i] + in_far[i] + 2) >> 2);
        return;
      }
      tmp.resize(size_t(wLo) * 2 + 2);
      uint8_t* o = tmp.data();
      if(vs == 1)  // horizontal only
      {
        const uint8_t* in = in_near;
        if(wLo == 1)
          o[0] = o[1] = in[0];
        else
        {
          o[0] = in[0];
          o[1] = uint8_t((in[0] * 3 + in[1] + 2) >> 2);
          int i;
          for(i = 1; i < wLo - 1; ++i)
          {
            int n        = 3 * in[i] + 2;
            o[i * 2 + 0] = uint8_t((n + in[i - 1]) >> 2);
            o[i * 2 + 1] = uint8_t((n + in[i + 1]) >> 2);
          }
          o[i * 2 + 0] = uint8_t((in[wLo - 2] * 3 + in[wLo - 1] + 2) >> 2);  // (sic: stb_image weights the far sample here)
          o[i * 2 + 1] = in[wLo - 1];
        }
      }
      else  // both
      {
        if(wLo == 1)
          o[0] = o[1] = uint8_t((3 * in_near[0] + in_far[0] + 2) >> 2);
        else
        {
          int t1 = 3 * in_near[0] + in_far[0];
          o[0]   = uint8_t((t1 + 2) >> 2);
          for(int i = 1; i < wLo; ++i)
          {
            int t0       = t1;
            t1           = 3 * in_near[i] + in_far[i];
            o[i * 2 - 1] = uint8_t((3 * t0 + t1 + 8) >> 4);
            o[i * 2]     = uint8_t((3 * t1 + t0 + 8) >> 4);
          }
          o[wLo * 2 - 1] = uint8_t((t1 + 2) >> 2);
        }
      }
      std::memcpy(out, o, size_t(width));
      return;
    }
    // any other ratio: nearest
    const uint8_t* in = plane + size_t(std::min(y * c.v / vmax, c.paddedH * 8 - 1)) * stride;
    for(int i = 0; i < width; ++i)
      out[i] = in[std::min(i * c.h / hmax, stride - 1)];
  }

  // Colour model as stb_image decides it (the reference decodes JPEG through tinygltf's stb_image): three components are RGB as stored when their ids
  // are 'R', 'G', 'B' or when an Adobe segment says "no transform" and there is no JFIF header, else YCbCr; four components are CMYK (Adobe
  // transform 0: the values are stored inverted, so a channel is c * k / 255), YCCK (transform 2: YCbCr to RGB, then (255 - rgb) * k / 255), or
  // YCbCr with a fourth channel that is ignored.
  int  adobeTransform = -1;
  bool jfif           = false;
  static uint8_t mul8(int x, int y)
  {
    const unsigned t = unsigned(x) * unsigned(y) + 128u;
    return uint8_t((t + (t >> 8)) >> 8);
  }
  bool output(Image& img) const
  {
    img.width  = width;
    img.height = height;
    img.rgba.assign(size_t(width) * height * 4, 255);
    std::vector<uint8_t> rows[4], tmp;
    for(int i = 0; i < ncomp; ++i)
      rows[i].resize(size_t(width));
    auto fixed = [](double x) { return int(x * 4096.0 + 0.5) << 8; };
    const bool plainRgb = ncomp == 3 && ((comp[0].id == 'R' && comp[1].id == 'G' && comp[2].id == 'B') || (adobeTransform == 0 && !jfif));
    const bool cmyk     = ncomp == 4 && adobeTransform == 0;
    const bool ycck     = ncomp == 4 && adobeTransform == 2;
    for(int y = 0; y < height; ++y)
    {
      for(int i = 0; i < ncomp; ++i)
        upsampleRow(comp[i], y, rows[i].data(), tmp);
      uint8_t* o = img.rgba.data() + size_t(y) * width * 4;
      if(ncomp == 1)
      {
        for(int x = 0; x < width; ++x)
          o[4 * x] = o[4 * x + 1] = o[4 * x + 2] = rows[0][size_t(x)];
        continue;
      }
      if(plainRgb || cmyk)
      {
        for(int x = 0; x < width; ++x)
          for(int c = 0; c < 3; ++c)
            o[4 * x + c] = cmyk ? mul8(rows[c][size_t(x)], rows[3][size_t(x)]) : rows[c][size_t(x)];
        continue;
      }
      for(int x = 0; x < width; ++x)
      {
        int yFixed = (int(rows[0][size_t(x)]) << 20) + (1 << 19);
        int cr = int(rows[2][size_t(x)]) - 128, cb = int(rows[1][size_t(x)]) - 128;
        int r = yFixed + cr * fixed(1.40200);
        int g = yFixed + cr * -fixed(0.71414) + int((unsigned(cb * -fixed(0.34414))) & 0xffff0000u);
        int b = yFixed + cb * fixed(1.77200);
        r >>= 20; g >>= 20; b >>= 20;
        o[4 * x] = clamp8(r); o[4 * x + 1] = clamp8(g); o[4 * x + 2] = clamp8(b);
        if(ycck)
          for(int c = 0; c < 3; ++c)
            o[4 * x + c] = mul8(255 - o[4 * x + c], rows[3][size_t(x)]);
      }
    }
    return true;
  }

  bool run(Image& img)
  {
    if(end - p < 4 || p[0] != 0xff || p[1] != 0xd8)
      return fail("not a JPEG stream");
    p += 2;
    bool haveFrame = false, haveScan = false;
    while(p + 4 <= end)
    {
      if(p[0] != 0xff)
      {
        ++p;  // stray bytes between segments (and whatever an entropy-coded segment left behind)
        continue;
      }
      int m = p[1];
      if(m == 0xff || m == 0x00 || (m >= 0xd0 && m <= 0xd7) || m == 0x01)
      {
        p += (m == 0xff) ? 1 : 2;
        continue;
      }
      if(m == 0xd9)
        break;
      int len = (p[2] << 8) | p[3];
      if(len < 2 || p + 2 + len > end)
        return fail("truncated segment");
      const uint8_t* s = p + 4;
      p += 2 + len;
      len -= 2;
      switch(m)
      {
        case 0xdb:
          if(!readDqt(s, len))
            return false;
          break;
        case 0xc4:
          if(!readDht(s, len))
            return false;
          break;
        case 0xc0:
        case 0xc1:
        case 0xc2:
          if(haveFrame)
            return fail("more than one frame");
          if(!readSof(s, len, m == 0xc2))
            return false;
          haveFrame = true;
          break;
        case 0xc3: case 0xc5: case 0xc6: case 0xc7: case 0xc9: case 0xca: case 0xcb: case 0xcd: case 0xce: case 0xcf:
          return fail("lossless / hierarchical / arithmetic-coded JPEG is not supported");
        case 0xdd:
          if(len < 2)
            return fail("bad DRI");
          restartInterval = (s[0] << 8) | s[1];
          break;
        case 0xe0:
          if(len >= 5 && std::memcmp(s, "JFIF", 5) == 0)
            jfif = true;
          break;
        case 0xee:
          if(len >= 12 && std::memcmp(s, "Adobe", 5) == 0)
            adobeTransform = s[11];
          break;
        case 0xda:
          if(!haveFrame)
            return fail("scan before frame header");
          if(!readScan(s, len))
            return false;
          haveScan = true;
          break;
        default:
          break;  // APPn, COM, ...
      }
    }
    if(!haveFrame || !haveScan)
      return fail("no image data");
    reconstruct();
    return output(img);
  }
};

}  // namespace

bool isJpeg(const uint8_t* data, size_t size)
{
  return size >= 3 && data[0] == 0xff && data[1] == 0xd8 && data[2] == 0xff;
}

bool decodeJpeg(const uint8_t* data, size_t size, Image& out, std::string* error)
{
  Decoder d;
  d.p     = data;
  d.end      = data + size;
  d.fileSize = size;
  d.error    = error;
  return d.run(out);
}

}  // namespace mihost
