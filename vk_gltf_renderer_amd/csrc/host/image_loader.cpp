// Image decode for the glTF front end: PNG (all colour types / bit depths, non-interlaced and Adam7) -> RGBA8 here, JPEG in
// jpeg_decoder.cpp, plus mip-chain generation.  Mirrors what the reference obtains from stb_image in src/gltf_image_loader.cpp:163-236 and
// from the blit chain in src/gltf_scene_vk.cpp:1247-1347 (nvvk::cmdGenerateMipmaps: each level is a LINEAR-filtered
// 2:1 blit of the previous one; for sRGB formats the hardware filters in linear light).  DDS: dds_decoder.cpp.
// KTX2/WebP are a "next" row; an undecodable image becomes the reference's 1x1 magenta
// (src/gltf_scene_vk.cpp:1057-1060).
#include "image_loader.hpp"

#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstring>

namespace mihost {

namespace {

uint32_t be32(const uint8_t* p)
{
  return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | uint32_t(p[3]);
}

int paeth(int a, int b, int c)
{
  int p  = a + b - c;
  int pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
  if(pa <= pb && pa <= pc)
    return a;
  return pb <= pc ? b : c;
}

// Undo PNG scanline filtering in place. `data` holds h rows of (1 + rowBytes) bytes.
bool unfilter(uint8_t* data, size_t rowBytes, size_t h, size_t bpp)
{
  std::vector<uint8_t> zero(rowBytes, 0);
  const uint8_t*       prev = zero.data();
  for(size_t y = 0; y < h; ++y)
  {
    uint8_t* row  = data + y * (rowBytes + 1);
    uint8_t  type = row[0];
    uint8_t* cur  = row + 1;
    switch(type)
    {
      case 0:
        break;
      case 1:
        for(size_t i = bpp; i < rowBytes; ++i)
          cur[i] = uint8_t(cur[i] + cur[i - bpp]);
        break;
      case 2:
        for(size_t i = 0; i < rowBytes; ++i)
          cur[i] = uint8_t(cur[i] + prev[i]);
        break;
      case 3:
        for(size_t i = 0; i < rowBytes; ++i)
        {
          int a  = i >= bpp ? cur[i - bpp] : 0;
          cur[i] = uint8_t(cur[i] + ((a + prev[i]) >> 1));
        }
        break;
      case 4:
        for(size_t i = 0; i < rowBytes; ++i)
        {
          int a  = i >= bpp ? cur[i - bpp] : 0;
          int c  = i >= bpp ? prev[i - bpp] : 0;
          cur[i] = uint8_t(cur[i] + paeth(a, prev[i], c));
        }
        break;
      default:
        return false;
    }
    prev = cur;
  }
  return true;
}

struct PngInfo
{
  uint32_t w = 0, h = 0;
  int      depth = 0, colorType = 0, interlace = 0;
  uint8_t  palette[256 * 3] = {};
  uint8_t  paletteAlpha[256];
  int      paletteCount = 0;
  bool     hasTrns      = false;
  uint16_t trnsGray = 0, trnsR = 0, trnsG = 0, trnsB = 0;
};

int channelsOf(int colorType)
{
  switch(colorType)
  {
    case 0: return 1;
    case 2: return 3;
    case 3: return 1;
    case 4: return 2;
    case 6: return 4;
  }
  return 0;
}

uint32_t sampleAt(const uint8_t* row, size_t index, int depth)
{
  switch(depth)
  {
    case 8: return row[index];
    case 16: return (uint32_t(row[2 * index]) << 8) | row[2 * index + 1];
    case 4: return (row[index >> 1] >> ((1 - (index & 1)) * 4)) & 0xF;
    case 2: return (row[index >> 2] >> ((3 - (index & 3)) * 2)) & 0x3;
    case 1: return (row[index >> 3] >> (7 - (index & 7))) & 0x1;
  }
  return 0;
}

uint8_t to8(uint32_t v, int depth)
{
  switch(depth)
  {
    case 8: return uint8_t(v);
    case 16: return uint8_t(v >> 8);
    case 4: return uint8_t(v * 17);
    case 2: return uint8_t(v * 85);
    case 1: return uint8_t(v * 255);
  }
  return 0;
}

// Convert one defiltered pass (pw x ph) into the RGBA8 image at the Adam7 positions given by (x0,y0,dx,dy).
void emitPass(const PngInfo& info, const uint8_t* data, size_t rowBytes, uint32_t pw, uint32_t ph, uint32_t x0,
              uint32_t y0, uint32_t dx, uint32_t dy, uint8_t* rgba)
{
  const int ch = channelsOf(info.colorType);
  for(uint32_t y = 0; y < ph; ++y)
  {
    const uint8_t* row = data + size_t(y) * (rowBytes + 1) + 1;
    for(uint32_t x = 0; x < pw; ++x)
    {
      uint8_t* out = rgba + (size_t(y0 + y * dy) * info.w + (x0 + x * dx)) * 4;
      switch(info.colorType)
      {
        case 0: {
          uint32_t g = sampleAt(row, x, info.depth);
          out[0] = out[1] = out[2] = to8(g, info.depth);
          out[3]                   = (info.hasTrns && g == info.trnsGray) ? 0 : 255;
          break;
        }
        case 2: {
          uint32_t r = sampleAt(row, size_t(x) * ch + 0, info.depth), g = sampleAt(row, size_t(x) * ch + 1, info.depth),
                   b = sampleAt(row, size_t(x) * ch + 2, info.depth);
          out[0] = to8(r, info.depth);
          out[1] = to8(g, info.depth);
          out[2] = to8(b, info.depth);
          out[3] = (info.hasTrns && r == info.trnsR && g == info.trnsG && b == info.trnsB) ? 0 : 255;
          break;
        }
        case 3: {
          uint32_t i = sampleAt(row, x, info.depth);
          if(int(i) >= info.paletteCount)
            i = 0;
          out[0] = info.palette[3 * i + 0];
          out[1] = info.palette[3 * i + 1];
          out[2] = info.palette[3 * i + 2];
          out[3] = info.paletteAlpha[i];
          break;
        }
        case 4: {
          out[0] = out[1] = out[2] = to8(sampleAt(row, size_t(x) * 2, info.depth), info.depth);
          out[3]                   = to8(sampleAt(row, size_t(x) * 2 + 1, info.depth), info.depth);
          break;
        }
        case 6: {
          for(int c = 0; c < 4; ++c)
            out[c] = to8(sampleAt(row, size_t(x) * 4 + c, info.depth), info.depth);
          break;
        }
      }
    }
  }
}

}  // namespace

bool isPng(const uint8_t* data, size_t size)
{
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  return size >= 8 && memcmp(data, sig, 8) == 0;
}

bool decodePng(const uint8_t* data, size_t size, Image& out, std::string* error)
{
  auto fail = [&](const char* msg) {
    if(error)
      *error = msg;
    return false;
  };
  if(!isPng(data, size))
    return fail("not a PNG");
  PngInfo info;
  memset(info.paletteAlpha, 255, sizeof(info.paletteAlpha));
  std::vector<uint8_t> idat;
  size_t               pos     = 8;
  bool                 gotIhdr = false, gotEnd = false;
  while(pos + 12 <= size && !gotEnd)
  {
    uint32_t       len  = be32(data + pos);
    const uint8_t* type = data + pos + 4;
    const uint8_t* body = data + pos + 8;
    if(pos + 12 + size_t(len) > size)
      return fail("truncated PNG chunk");
    if(!memcmp(type, "IHDR", 4))
    {
      if(len < 13)
        return fail("bad IHDR");
      info.w         = be32(body);
      info.h         = be32(body + 4);
      info.depth     = body[8];
      info.colorType = body[9];
      info.interlace = body[12];
      gotIhdr        = true;
    }
    else if(!memcmp(type, "PLTE", 4))
    {
      info.paletteCount = int(std::min<uint32_t>(len / 3, 256));
      memcpy(info.palette, body, size_t(info.paletteCount) * 3);
    }
    else if(!memcmp(type, "tRNS", 4))
    {
      info.hasTrns = true;
      if(info.colorType == 3)
        memcpy(info.paletteAlpha, body, std::min<uint32_t>(len, 256));
      else if(info.colorType == 0 && len >= 2)
        info.trnsGray = uint16_t((body[0] << 8) | body[1]);
      else if(info.colorType == 2 && len >= 6)
      {
        info.trnsR = uint16_t((body[0] << 8) | body[1]);
        info.trnsG = uint16_t((body[2] << 8) | body[3]);
        info.trnsB = uint16_t((body[4] << 8) | body[5]);
      }
    }
    else if(!memcmp(type, "IDAT", 4))
      idat.insert(idat.end(), body, body + len);
    else if(!memcmp(type, "IEND", 4))
      gotEnd = true;
    pos += 12 + size_t(len);
  }
  if(!gotIhdr || info.w == 0 || info.h == 0 || channelsOf(info.colorType) == 0)
    return fail("bad PNG header");
  if(!saneImageSize(info.w, info.h))
    return fail("PNG: image dimensions out of range");
  if(info.colorType == 3 && info.hasTrns == false)
    memset(info.paletteAlpha, 255, sizeof(info.paletteAlpha));

  const int    ch           = channelsOf(info.colorType);
  const size_t bitsPerPixel = size_t(ch) * size_t(info.depth);
  const size_t bpp          = std::max<size_t>(1, bitsPerPixel / 8);
  auto         rowBytesOf   = [&](uint32_t w) { return (size_t(w) * bitsPerPixel + 7) / 8; };

  struct Pass
  {
    uint32_t x0, y0, dx, dy;
  };
  std::vector<Pass> passes;
  if(info.interlace)
    passes = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
  else
    passes = {{0, 0, 1, 1}};
  size_t total = 0;
  for(const Pass& p : passes)
  {
    uint32_t pw = (info.w > p.x0) ? (info.w - p.x0 + p.dx - 1) / p.dx : 0;
    uint32_t ph = (info.h > p.y0) ? (info.h - p.y0 + p.dy - 1) / p.dy : 0;
    if(pw && ph)
      total += (rowBytesOf(pw) + 1) * ph;
  }
  // deflate expands by at most 1032 : 1: a stream that short cannot hold the image the header claims (checked before allocating)
  if(total > idat.size() * 1032 + 1024)
    return fail("PNG: the compressed data cannot hold an image of the declared size");
  std::vector<uint8_t> raw(total);
  uLongf               destLen = uLongf(total);
  int                  zr      = uncompress(raw.data(), &destLen, idat.data(), uLong(idat.size()));
  if(zr != Z_OK || destLen != total)
    return fail("PNG inflate failed");

  out.width  = int(info.w);
  out.height = int(info.h);
  out.rgba.assign(size_t(info.w) * info.h * 4, 0);
  size_t off = 0;
  for(const Pass& p : passes)
  {
    uint32_t pw = (info.w > p.x0) ? (info.w - p.x0 + p.dx - 1) / p.dx : 0;
    uint32_t ph = (info.h > p.y0) ? (info.h - p.y0 + p.dy - 1) / p.dy : 0;
    if(!pw || !ph)
      continue;
    size_t rb = rowBytesOf(pw);
    if(!unfilter(raw.data() + off, rb, ph, bpp))
      return fail("bad PNG filter type");
    emitPass(info, raw.data() + off, rb, pw, ph, p.x0, p.y0, p.dx, p.dy, out.rgba.data());
    off += (rb + 1) * ph;
  }
  return true;
}

bool decodeImage(const uint8_t* data, size_t size, Image& out, std::string* error)
{
  if(isPng(data, size))
    return decodePng(data, size, out, error);
  if(isJpeg(data, size))
    return decodeJpeg(data, size, out, error);
  if(isDds(data, size))
    return decodeDds(data, size, out, error);
  if(isKtx(data, size))
    return decodeKtx(data, size, out, error);
  if(isWebp(data, size))
    return decodeWebp(data, size, out, error);
  if(error)
    *error = "unsupported image container (PNG, JPEG, DDS, KTX, KTX2 and WebP are decoded)";
  return false;
}

Image magentaImage()
{
  Image img;
  img.width  = 1;
  img.height = 1;
  img.rgba   = {255, 0, 255, 255};
  return img;
}

// ---- sRGB transfer (IEC 61966-2-1), as the texture unit applies it for *_SRGB formats -------------------------------
float srgbToLinear(uint8_t v)
{
  static float table[256];
  static bool  init = false;
  if(!init)
  {
    for(int i = 0; i < 256; ++i)
    {
      float c  = float(i) / 255.0f;
      table[i] = c <= 0.04045f ? c / 12.92f : std::pow((c + 0.055f) / 1.055f, 2.4f);
    }
    init = true;
  }
  return table[v];
}

uint8_t linearToSrgb8(float c)
{
  c       = std::min(std::max(c, 0.0f), 1.0f);
  float s = c <= 0.0031308f ? c * 12.92f : 1.055f * std::pow(c, 1.0f / 2.4f) - 0.055f;
  return uint8_t(std::lround(s * 255.0f));
}

// One 2:1 LINEAR blit level (src -> dst = max(1, src/2)): every destination texel centre is sampled with a bilinear
// filter in the source, which for even sizes is the 2x2 box average.
void downsample(const std::vector<uint8_t>& src, int sw, int sh, std::vector<uint8_t>& dst, int dw, int dh, bool srgb)
{
  dst.resize(size_t(dw) * dh * 4);
  for(int y = 0; y < dh; ++y)
    for(int x = 0; x < dw; ++x)
    {
      // destination texel centre in source texel space
      float fx = (x + 0.5f) * float(sw) / float(dw) - 0.5f;
      float fy = (y + 0.5f) * float(sh) / float(dh) - 0.5f;
      int   x0 = int(std::floor(fx)), y0 = int(std::floor(fy));
      float tx = fx - float(x0), ty = fy - float(y0);
      int   x1 = std::min(x0 + 1, sw - 1), y1 = std::min(y0 + 1, sh - 1);
      x0 = std::max(x0, 0);
      y0 = std::max(y0, 0);
      const uint8_t* p00 = &src[(size_t(y0) * sw + x0) * 4];
      const uint8_t* p10 = &src[(size_t(y0) * sw + x1) * 4];
      const uint8_t* p01 = &src[(size_t(y1) * sw + x0) * 4];
      const uint8_t* p11 = &src[(size_t(y1) * sw + x1) * 4];
      uint8_t*       o   = &dst[(size_t(y) * dw + x) * 4];
      for(int c = 0; c < 4; ++c)
      {
        bool  decode = srgb && c < 3;
        float a      = decode ? srgbToLinear(p00[c]) : p00[c] / 255.0f;
        float b      = decode ? srgbToLinear(p10[c]) : p10[c] / 255.0f;
        float cc     = decode ? srgbToLinear(p01[c]) : p01[c] / 255.0f;
        float d      = decode ? srgbToLinear(p11[c]) : p11[c] / 255.0f;
        float v      = (a * (1 - tx) + b * tx) * (1 - ty) + (cc * (1 - tx) + d * tx) * ty;
        o[c]         = decode ? linearToSrgb8(v) : uint8_t(std::lround(std::min(std::max(v, 0.0f), 1.0f) * 255.0f));
      }
    }
}

void buildMipChain(const Image& base, bool srgb, std::vector<std::vector<uint8_t>>& levels)
{
  levels.clear();
  levels.push_back(base.rgba);
  int w = base.width, h = base.height;
  while(w > 1 || h > 1)
  {
    int nw = std::max(1, w / 2), nh = std::max(1, h / 2);
    levels.emplace_back();
    downsample(levels[levels.size() - 2], w, h, levels.back(), nw, nh, srgb);
    w = nw;
    h = nh;
  }
}

}  // namespace mihost
