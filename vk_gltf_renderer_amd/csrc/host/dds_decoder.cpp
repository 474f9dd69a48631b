// DDS (MSFT_texture_dds) decode for the glTF front end -> RGBA8 of the top mip level.
//
// The reference reads DDS through nv_dds (src/gltf_image_loader.cpp:69-160; external library) and uploads block-compressed
// data as is for the GPU's texture unit to decode.  There is no texture unit on this path, so the blocks are
// decoded on the host: BC1, BC2, BC3 (colour endpoints expanded to 8 bits, palette entries (2a+b+1)/3 and (a+b+1)/2),
// BC4 / BC5 (eight-entry ramps, rounded to nearest) -- the arithmetic of the D3D functional specification, which GPUs
// implement to within a unit of the last place -- plus the uncompressed 8-bit layouts (RGBA, BGRA, BGRX, RGB, BGR, L, LA,
// R, RG), and BC7 (bc7_decoder.cpp).  BC6H (HDR), cube maps and volume textures are not decoded (the loader falls back to the
// texture's core `source` image or to the reference's 1x1 magenta).  The block decode is shared with the KTX reader (ktx_decoder.cpp).  Mip levels stored in the file are ignored: the chain is rebuilt from
// level 0 like for every other image (image_loader.cpp buildMipChain).
#include "image_loader.hpp"

#include <cstring>

namespace mihost {

namespace {

uint32_t le32(const uint8_t* p) { return uint32_t(p[0]) | (uint32_t(p[1]) << 8) | (uint32_t(p[2]) << 16) | (uint32_t(p[3]) << 24); }
constexpr uint32_t fourcc(char a, char b, char c, char d)
{
  return uint32_t(uint8_t(a)) | (uint32_t(uint8_t(b)) << 8) | (uint32_t(uint8_t(c)) << 16) | (uint32_t(uint8_t(d)) << 24);
}

enum class Layout
{
  Unknown,
  BC1,
  BC2,
  BC3,
  BC4,
  BC5,
  BC7,
  Masks,  // uncompressed, described by bit masks
};

void expand565(uint16_t c, int* rgb)
{
  int r = (c >> 11) & 31, g = (c >> 5) & 63, b = c & 31;
  rgb[0] = (r << 3) | (r >> 2);
  rgb[1] = (g << 2) | (g >> 4);
  rgb[2] = (b << 3) | (b >> 2);
}

// 8 bytes of BC1-style colour data -> 16 RGBA texels; `punchThrough`: the c0 <= c1 three-colour + transparent mode is honoured
void decodeColorBlock(const uint8_t* s, uint8_t out[16][4], bool punchThrough)
{
  const uint16_t c0 = uint16_t(s[0] | (s[1] << 8)), c1 = uint16_t(s[2] | (s[3] << 8));
  int            pal[4][4];
  expand565(c0, pal[0]);
  expand565(c1, pal[1]);
  pal[0][3] = pal[1][3] = 255;
  if(c0 > c1 || !punchThrough)
  {
    for(int k = 0; k < 3; ++k)
    {
      pal[2][k] = (2 * pal[0][k] + pal[1][k] + 1) / 3;
      pal[3][k] = (pal[0][k] + 2 * pal[1][k] + 1) / 3;
    }
    pal[2][3] = pal[3][3] = 255;
  }
  else
  {
    for(int k = 0; k < 3; ++k)
    {
      pal[2][k] = (pal[0][k] + pal[1][k] + 1) / 2;
      pal[3][k] = 0;
    }
    pal[2][3] = 255;
    pal[3][3] = 0;
  }
  const uint32_t idx = le32(s + 4);
  for(int i = 0; i < 16; ++i)
  {
    const int* p = pal[(idx >> (2 * i)) & 3];
    out[i][0] = uint8_t(p[0]); out[i][1] = uint8_t(p[1]); out[i][2] = uint8_t(p[2]); out[i][3] = uint8_t(p[3]);
  }
}

// 8 bytes of BC4-style data -> 16 values
void decodeRampBlock(const uint8_t* s, uint8_t out[16])
{
  const int a0 = s[0], a1 = s[1];
  int       pal[8];
  pal[0] = a0;
  pal[1] = a1;
  if(a0 > a1)
    for(int i = 1; i < 7; ++i)
      pal[1 + i] = ((7 - i) * a0 + i * a1 + 3) / 7;
  else
  {
    for(int i = 1; i < 5; ++i)
      pal[1 + i] = ((5 - i) * a0 + i * a1 + 2) / 5;
    pal[6] = 0;
    pal[7] = 255;
  }
  uint64_t bits = 0;
  for(int i = 0; i < 6; ++i)
    bits |= uint64_t(s[2 + i]) << (8 * i);
  for(int i = 0; i < 16; ++i)
    out[i] = uint8_t(pal[(bits >> (3 * i)) & 7]);
}

int maskShift(uint32_t m)
{
  int s = 0;
  while(m && !(m & 1u))
  {
    m >>= 1;
    ++s;
  }
  return s;
}
int maskBits(uint32_t m)
{
  int n = 0;
  for(m >>= maskShift(m); m & 1u; m >>= 1)
    ++n;
  return n;
}
uint8_t extract(uint32_t v, uint32_t mask)
{
  if(!mask)
    return 0;
  const int      bits = maskBits(mask);
  const uint32_t x    = (v & mask) >> maskShift(mask);
  if(bits >= 8)
    return uint8_t(x >> (bits - 8));
  // replicate the bits (exact for 8-bit channels, the usual expansion for 4/5/6-bit ones)
  uint32_t r = x << (8 - bits);
  for(int b = bits; b < 8; b += bits)
    r |= r >> b;
  return uint8_t(r);
}

}  // namespace

// width x height texels of 4x4 blocks in row-major block order -> RGBA8 (blocks beyond the image edge are cropped)
bool decodeBlocks(BlockFormat format, const uint8_t* s, size_t size, int width, int height, Image& out)
{
  const uint32_t bw = uint32_t(width + 3) / 4, bh = uint32_t(height + 3) / 4;
  const size_t   blockBytes = (format == BlockFormat::BC1 || format == BlockFormat::BC4) ? 8 : 16;
  if(width <= 0 || height <= 0 || size < size_t(bw) * bh * blockBytes)
    return false;
  out.width  = width;
  out.height = height;
  out.rgba.assign(size_t(width) * size_t(height) * 4, 255);
  for(uint32_t by = 0; by < bh; ++by)
    for(uint32_t bx = 0; bx < bw; ++bx, s += blockBytes)
    {
      uint8_t px[16][4];
      uint8_t ramp[16], ramp2[16];
      switch(format)
      {
        case BlockFormat::BC1:
          decodeColorBlock(s, px, true);
          break;
        case BlockFormat::BC2:
          decodeColorBlock(s + 8, px, false);
          for(int i = 0; i < 16; ++i)
          {
            const int a = (s[i >> 1] >> (4 * (i & 1))) & 15;
            px[i][3]    = uint8_t(a * 17);
          }
          break;
        case BlockFormat::BC3:
          decodeColorBlock(s + 8, px, false);
          decodeRampBlock(s, ramp);
          for(int i = 0; i < 16; ++i)
            px[i][3] = ramp[i];
          break;
        case BlockFormat::BC4:
          decodeRampBlock(s, ramp);
          for(int i = 0; i < 16; ++i)
          {
            px[i][0] = ramp[i]; px[i][1] = 0; px[i][2] = 0; px[i][3] = 255;  // (r, 0, 0, 1) as the texture unit returns it
          }
          break;
        case BlockFormat::BC5:
          decodeRampBlock(s, ramp);
          decodeRampBlock(s + 8, ramp2);
          for(int i = 0; i < 16; ++i)
          {
            px[i][0] = ramp[i]; px[i][1] = ramp2[i]; px[i][2] = 0; px[i][3] = 255;
          }
          break;
        default:
          decodeBc7Block(s, px);
          break;
      }
      for(uint32_t y = 0; y < 4 && by * 4 + y < uint32_t(height); ++y)
        for(uint32_t x = 0; x < 4 && bx * 4 + x < uint32_t(width); ++x)
          std::memcpy(out.rgba.data() + (size_t(by * 4 + y) * size_t(width) + bx * 4 + x) * 4, px[y * 4 + x], 4);
    }
  return true;
}

bool isDds(const uint8_t* data, size_t size)
{
  return size >= 128 && data[0] == 'D' && data[1] == 'D' && data[2] == 'S' && data[3] == ' ';
}

bool decodeDds(const uint8_t* data, size_t size, Image& out, std::string* error)
{
  auto fail = [&](const char* msg) {
    if(error)
      *error = std::string("DDS: ") + msg;
    return false;
  };
  if(!isDds(data, size) || le32(data + 4) != 124)
    return fail("bad header");
  const uint32_t height = le32(data + 12), width = le32(data + 16), depth = le32(data + 24);
  const uint32_t pfFlags = le32(data + 80), pfFourCC = le32(data + 84), pfBits = le32(data + 88);
  uint32_t       rMask = le32(data + 92), gMask = le32(data + 96), bMask = le32(data + 100), aMask = le32(data + 104);
  const uint32_t caps2 = le32(data + 112);
  if(!saneImageSize(width, height))
    return fail("bad dimensions");
  if((caps2 & 0x200u) || ((caps2 & 0x200000u) && depth > 1))
    return fail("cube maps and volume textures are not supported");
  size_t   offset = 128;
  Layout   layout = Layout::Unknown;
  uint32_t bytesPerPixel = 0;
  if(pfFlags & 0x4u)  // DDPF_FOURCC
  {
    if(pfFourCC == fourcc('D', 'X', 'T', '1'))
      layout = Layout::BC1;
    else if(pfFourCC == fourcc('D', 'X', 'T', '2') || pfFourCC == fourcc('D', 'X', 'T', '3'))
      layout = Layout::BC2;
    else if(pfFourCC == fourcc('D', 'X', 'T', '4') || pfFourCC == fourcc('D', 'X', 'T', '5'))
      layout = Layout::BC3;
    else if(pfFourCC == fourcc('A', 'T', 'I', '1') || pfFourCC == fourcc('B', 'C', '4', 'U'))
      layout = Layout::BC4;
    else if(pfFourCC == fourcc('A', 'T', 'I', '2') || pfFourCC == fourcc('B', 'C', '5', 'U'))
      layout = Layout::BC5;
    else if(pfFourCC == fourcc('D', 'X', '1', '0'))
    {
      if(size < 148)
        return fail("truncated DX10 header");
      const uint32_t dxgi = le32(data + 128), dim = le32(data + 132), arraySize = le32(data + 140);
      offset              = 148;
      if(dim != 3 /* TEXTURE2D */ || arraySize > 1 || (le32(data + 136) & 0x4u) /* cube */)
        return fail("only plain 2D textures are supported");
      switch(dxgi)
      {
        case 70: case 71: case 72: layout = Layout::BC1; break;
        case 73: case 74: case 75: layout = Layout::BC2; break;
        case 76: case 77: case 78: layout = Layout::BC3; break;
        case 79: case 80: layout = Layout::BC4; break;
        case 82: case 83: layout = Layout::BC5; break;
        case 97: case 98: case 99: layout = Layout::BC7; break;
        case 27: case 28: case 29: layout = Layout::Masks; bytesPerPixel = 4; rMask = 0xffu; gMask = 0xff00u; bMask = 0xff0000u; aMask = 0xff000000u; break;
        case 87: case 90: case 91: layout = Layout::Masks; bytesPerPixel = 4; bMask = 0xffu; gMask = 0xff00u; rMask = 0xff0000u; aMask = 0xff000000u; break;
        case 88: case 92: case 93: layout = Layout::Masks; bytesPerPixel = 4; bMask = 0xffu; gMask = 0xff00u; rMask = 0xff0000u; aMask = 0; break;
        case 61: layout = Layout::Masks; bytesPerPixel = 1; rMask = 0xffu; gMask = bMask = aMask = 0; break;
        case 49: layout = Layout::Masks; bytesPerPixel = 2; rMask = 0xffu; gMask = 0xff00u; bMask = aMask = 0; break;
        default: return fail("unsupported DXGI format (BC6H, float and >8-bit formats are not decoded)");
      }
    }
    else
      return fail("unsupported FourCC");
  }
  else if(pfFlags & (0x40u | 0x20000u | 0x2u))  // DDPF_RGB | DDPF_LUMINANCE | DDPF_ALPHA
  {
    if(pfBits != 8 && pfBits != 16 && pfBits != 24 && pfBits != 32)
      return fail("unsupported bit count");
    layout        = Layout::Masks;
    bytesPerPixel = pfBits / 8;
    if(!(pfFlags & 0x1u) && !(pfFlags & 0x2u))  // no DDPF_ALPHAPIXELS / DDPF_ALPHA
      aMask = 0;
    if(pfFlags & 0x20000u)  // luminance: replicate into rgb
      gMask = bMask = rMask;
    if((pfFlags & 0x2u) && !(pfFlags & 0x40u) && !(pfFlags & 0x20000u))  // alpha only
      rMask = gMask = bMask = 0;
  }
  else
    return fail("unsupported pixel format");

  out.width  = int(width);
  out.height = int(height);
  out.rgba.assign(size_t(width) * height * 4, 255);
  if(layout == Layout::Masks)
  {
    const size_t need = size_t(width) * height * bytesPerPixel;
    if(size < offset + need)
      return fail("truncated pixel data");
    const uint8_t* s = data + offset;
    for(size_t i = 0; i < size_t(width) * height; ++i, s += bytesPerPixel)
    {
      uint32_t v = 0;
      for(uint32_t b = 0; b < bytesPerPixel; ++b)
        v |= uint32_t(s[b]) << (8 * b);
      uint8_t* o = out.rgba.data() + i * 4;
      o[0] = extract(v, rMask); o[1] = extract(v, gMask); o[2] = extract(v, bMask);
      o[3] = aMask ? extract(v, aMask) : 255;
    }
    return true;
  }
  const BlockFormat bf = layout == Layout::BC1 ? BlockFormat::BC1 : layout == Layout::BC2 ? BlockFormat::BC2 : layout == Layout::BC3 ? BlockFormat::BC3
                         : layout == Layout::BC4 ? BlockFormat::BC4 : layout == Layout::BC5 ? BlockFormat::BC5 : BlockFormat::BC7;
  if(!decodeBlocks(bf, data + offset, size - offset, int(width), int(height), out))
    return fail("truncated block data");
  return true;
}

}  // namespace mihost
