// KTX 1 / KTX 2 containers (KHR_texture_basisu images and plain .ktx2 textures) -> RGBA8 of the top mip level; WebP at the end.
//
// The reference reads both through nv_ktx (src/gltf_image_loader.cpp:123-160; external library) and uploads the payload in its
// VkFormat for the texture unit to filter.  There is no texture unit on this path, so level 0 is decoded on the host:
//   * containers: KTX 2.0 (80-byte header + level index, https://registry.khronos.org/KTX/specs/2.0/ktxspec.v2.html) and KTX 1.1
//     (64-byte header, little-endian files);
//   * supercompression (KTX 2): none, Zstandard (scheme 2; libzstd.so.1 is loaded at run time, there is no header in this image)
//     and ZLIB (scheme 3).  BasisLZ (scheme 1) and UASTC payloads (vkFormat = UNDEFINED) need the Basis Universal transcoder and
//     are reported as unsupported: the loader then falls back to the texture's core `source` like the reference does for a
//     container it cannot read;
//   * formats: R8, R8G8, R8G8B8, B8G8R8, R8G8B8A8, B8G8R8A8 (UNORM / SRGB), BC1-BC5 and BC7 (block decode shared with the DDS
//     reader).  The transfer function comes from the glTF slot (src/gltf_image_loader.cpp `tryForceVkFormatTransferFunction`), so
//     the UNORM / SRGB variants decode to the same bytes.
// Only plain 2D images (one face, one layer, depth 0/1); the file's own mip levels are ignored and the chain rebuilt from level 0
// like for every other container (image_loader.cpp buildMipChain).
#include "image_loader.hpp"

#include <dlfcn.h>
#include <zlib.h>

#include <cstring>

namespace mihost {

namespace {

uint32_t le32(const uint8_t* p) { return uint32_t(p[0]) | (uint32_t(p[1]) << 8) | (uint32_t(p[2]) << 16) | (uint32_t(p[3]) << 24); }
uint64_t le64(const uint8_t* p) { return uint64_t(le32(p)) | (uint64_t(le32(p + 4)) << 32); }

const uint8_t kKtx1Id[12] = {0xAB, 0x4B, 0x54, 0x58, 0x20, 0x31, 0x31, 0xBB, 0x0D, 0x0A, 0x1A, 0x0A};
const uint8_t kKtx2Id[12] = {0xAB, 0x4B, 0x54, 0x58, 0x20, 0x32, 0x30, 0xBB, 0x0D, 0x0A, 0x1A, 0x0A};

// what a level's bytes are
struct PixelLayout
{
  bool        block = false;
  BlockFormat blockFormat = BlockFormat::BC1;
  int         channels = 0;   // uncompressed: bytes per texel
  bool        bgr = false;    // blue first
};

bool layoutOfVkFormat(uint32_t f, PixelLayout& L)
{
  switch(f)
  {
    case 9: case 15: L.channels = 1; return true;                       // R8_UNORM / R8_SRGB
    case 16: case 22: L.channels = 2; return true;                      // R8G8
    case 23: case 29: L.channels = 3; return true;                      // R8G8B8
    case 30: case 36: L.channels = 3; L.bgr = true; return true;        // B8G8R8
    case 37: case 43: L.channels = 4; return true;                      // R8G8B8A8
    case 44: case 50: L.channels = 4; L.bgr = true; return true;        // B8G8R8A8
    case 131: case 132: case 133: case 134: L.block = true; L.blockFormat = BlockFormat::BC1; return true;
    case 135: case 136: L.block = true; L.blockFormat = BlockFormat::BC2; return true;
    case 137: case 138: L.block = true; L.blockFormat = BlockFormat::BC3; return true;
    case 139: L.block = true; L.blockFormat = BlockFormat::BC4; return true;
    case 141: L.block = true; L.blockFormat = BlockFormat::BC5; return true;
    case 145: case 146: L.block = true; L.blockFormat = BlockFormat::BC7; return true;
    default: return false;
  }
}
bool layoutOfGlFormat(uint32_t glInternal, PixelLayout& L)
{
  switch(glInternal)
  {
    case 0x8229: L.channels = 1; return true;                                    // GL_R8
    case 0x822B: L.channels = 2; return true;                                    // GL_RG8
    case 0x8051: case 0x8C41: L.channels = 3; return true;                       // GL_RGB8 / GL_SRGB8
    case 0x8058: case 0x8C43: L.channels = 4; return true;                       // GL_RGBA8 / GL_SRGB8_ALPHA8
    case 0x83F0: case 0x83F1: case 0x8C4C: case 0x8C4D: L.block = true; L.blockFormat = BlockFormat::BC1; return true;  // S3TC DXT1
    case 0x83F2: case 0x8C4E: L.block = true; L.blockFormat = BlockFormat::BC2; return true;
    case 0x83F3: case 0x8C4F: L.block = true; L.blockFormat = BlockFormat::BC3; return true;
    case 0x8DBB: L.block = true; L.blockFormat = BlockFormat::BC4; return true;  // GL_COMPRESSED_RED_RGTC1
    case 0x8DBD: L.block = true; L.blockFormat = BlockFormat::BC5; return true;  // GL_COMPRESSED_RG_RGTC2
    case 0x8E8C: case 0x8E8D: L.block = true; L.blockFormat = BlockFormat::BC7; return true;  // BPTC UNORM
    default: return false;
  }
}

bool decodeLevel(const PixelLayout& L, const uint8_t* s, size_t size, int width, int height, size_t rowStride, Image& out)
{
  if(L.block)
    return decodeBlocks(L.blockFormat, s, size, width, height, out);
  const size_t row = rowStride ? rowStride : size_t(width) * size_t(L.channels);
  if(size < row * size_t(height - 1) + size_t(width) * size_t(L.channels))
    return false;
  out.width  = width;
  out.height = height;
  out.rgba.assign(size_t(width) * size_t(height) * 4, 255);
  for(int y = 0; y < height; ++y)
    for(int x = 0; x < width; ++x)
    {
      const uint8_t* p = s + size_t(y) * row + size_t(x) * size_t(L.channels);
      uint8_t*       o = out.rgba.data() + (size_t(y) * size_t(width) + size_t(x)) * 4;
      // the texture unit returns (r, 0, 0, 1) / (r, g, 0, 1) for one- and two-channel formats
      o[0] = p[L.bgr ? 2 : 0];
      o[1] = L.channels >= 2 ? p[1] : 0;
      o[2] = L.channels >= 3 ? p[L.bgr ? 0 : 2] : 0;
      o[3] = L.channels >= 4 ? p[3] : 255;
    }
  return true;
}

// Zstandard through the run-time library (stable one-shot API of libzstd >= 1.0)
bool zstdDecompress(const uint8_t* src, size_t srcSize, std::vector<uint8_t>& dst, size_t dstSize, std::string& why)
{
  typedef size_t (*DecompressFn)(void*, size_t, const void*, size_t);
  typedef unsigned (*IsErrorFn)(size_t);
  static void*        lib        = dlopen("libzstd.so.1", RTLD_NOW | RTLD_LOCAL);
  static DecompressFn decompress = lib ? reinterpret_cast<DecompressFn>(dlsym(lib, "ZSTD_decompress")) : nullptr;
  static IsErrorFn    isError    = lib ? reinterpret_cast<IsErrorFn>(dlsym(lib, "ZSTD_isError")) : nullptr;
  if(!decompress || !isError)
  {
    why = "Zstandard supercompression needs libzstd.so.1, which could not be loaded";
    return false;
  }
  dst.resize(dstSize);
  const size_t r = decompress(dst.data(), dstSize, src, srcSize);
  if(isError(r) || r != dstSize)
  {
    why = "corrupt Zstandard level data";
    return false;
  }
  return true;
}

}  // namespace

bool isKtx(const uint8_t* data, size_t size)
{
  return size >= 12 && (memcmp(data, kKtx1Id, 12) == 0 || memcmp(data, kKtx2Id, 12) == 0);
}

bool decodeKtx(const uint8_t* data, size_t size, Image& out, std::string* error)
{
  auto fail = [&](const std::string& msg) {
    if(error)
      *error = "KTX: " + msg;
    return false;
  };
  if(!isKtx(data, size))
    return fail("bad identifier");
  PixelLayout L;
  if(memcmp(data, kKtx2Id, 12) == 0)
  {
    if(size < 80 + 24)
      return fail("truncated header");
    const uint32_t vkFormat = le32(data + 12), width = le32(data + 20), height = le32(data + 24), depth = le32(data + 28);
    const uint32_t layers = le32(data + 32), faces = le32(data + 36), levels = le32(data + 40), scheme = le32(data + 44);
    if(!saneImageSize(width, height))
      return fail("bad dimensions");
    if(depth > 1 || layers > 1 || faces != 1)
      return fail("only plain 2D images are supported (no volume, array or cube textures)");
    if(scheme == 1 || vkFormat == 0)
      return fail("BasisLZ / UASTC payloads need the Basis Universal transcoder, which is not part of this front end");
    if(!layoutOfVkFormat(vkFormat, L))
      return fail("unsupported VkFormat " + std::to_string(vkFormat) + " (8-bit R / RG / RGB / RGBA, BC1-BC5 and BC7 are decoded)");
    if(levels > 32 || size < 80 + size_t(std::max(levels, 1u)) * 24)
      return fail("truncated level index");
    // level index entry 0 = the base level: byteOffset, byteLength, uncompressedByteLength
    const uint64_t off = le64(data + 80), len = le64(data + 88), ulen = le64(data + 96);
    if(off > size || len > size - off)
      return fail("level 0 lies outside the file");
    const uint8_t*       payload = data + off;
    size_t               payloadSize = size_t(len);
    std::vector<uint8_t> inflated;
    if(scheme == 2 || scheme == 3)
    {
      // the inflated size of a level is known from the header: anything else is a corrupt or hostile file (and would size a buffer)
      const uint64_t expect = L.block ? uint64_t((width + 3) / 4) * ((height + 3) / 4) * ((L.blockFormat == BlockFormat::BC1 || L.blockFormat == BlockFormat::BC4) ? 8 : 16)
                                      : uint64_t(width) * height * uint64_t(L.channels);
      if(ulen != expect)
        return fail("uncompressed length does not match the image size");
      if(scheme == 2)
      {
        std::string why;
        if(!zstdDecompress(payload, payloadSize, inflated, size_t(ulen), why))
          return fail(why);
      }
      else
      {
        inflated.resize(size_t(ulen));
        uLongf n = uLongf(ulen);
        if(uncompress(inflated.data(), &n, payload, uLong(payloadSize)) != Z_OK || n != uLongf(ulen))
          return fail("corrupt ZLIB level data");
      }
      payload     = inflated.data();
      payloadSize = inflated.size();
    }
    else if(scheme != 0)
      return fail("unknown supercompression scheme " + std::to_string(scheme));
    if(!decodeLevel(L, payload, payloadSize, int(width), int(height), 0, out))
      return fail("truncated level data");
    return true;
  }
  // ---- KTX 1.1
  if(size < 64 + 4)
    return fail("truncated header");
  if(le32(data + 12) != 0x04030201u)
    return fail("big-endian KTX 1 files are not supported");
  const uint32_t glType = le32(data + 16), glInternal = le32(data + 28), width = le32(data + 36), height = le32(data + 40), depth = le32(data + 44);
  const uint32_t elements = le32(data + 48), faces = le32(data + 52), kvBytes = le32(data + 60);
  if(!saneImageSize(width, height))
    return fail("bad dimensions");
  if(depth > 1 || elements > 1 || faces != 1)
    return fail("only plain 2D images are supported (no volume, array or cube textures)");
  if(!layoutOfGlFormat(glInternal, L) || (!L.block && glType != 0x1401u /* GL_UNSIGNED_BYTE */))
    return fail("unsupported glInternalFormat");
  if(!L.block)
  {
    // the CLIENT format (glFormat, offset 24) gives the channel order of the stored bytes: GL_BGR / GL_BGRA data under an RGB8 /
    // RGBA8 internal format is blue first; anything but the plain 8-bit orders with the channel count of the internal format is refused
    const uint32_t glFormat = le32(data + 24);
    const bool     rgbOrder = glFormat == 0x1903u /* GL_RED */ || glFormat == 0x8227u /* GL_RG */ || glFormat == 0x1907u /* GL_RGB */ || glFormat == 0x1908u /* GL_RGBA */
                          || glFormat == 0x1909u /* GL_LUMINANCE */ || glFormat == 0x190Au /* GL_LUMINANCE_ALPHA */;
    const bool bgrOrder = (glFormat == 0x80E0u /* GL_BGR */ && L.channels == 3) || (glFormat == 0x80E1u /* GL_BGRA */ && L.channels == 4);
    if(bgrOrder)
      L.bgr = true;
    else if(!rgbOrder)
      return fail("unsupported glFormat");
  }
  if(kvBytes > size - 64 - 4)
    return fail("truncated key/value data");
  const size_t   at = 64 + size_t(kvBytes);
  const uint32_t imageSize = le32(data + at);
  if(imageSize > size - at - 4)
    return fail("level 0 lies outside the file");
  // rows of uncompressed KTX 1 images are padded to 4 bytes (GL_UNPACK_ALIGNMENT 4)
  const size_t rowStride = L.block ? 0 : ((size_t(width) * size_t(L.channels) + 3) & ~size_t(3));
  if(!decodeLevel(L, data + at + 4, imageSize, int(width), int(height), rowStride, out))
    return fail("truncated level data");
  return true;
}

// ---- WebP (EXT_texture_webp) -----------------------------------------------------------------------------------------------
// The reference decodes WebP with libwebp (webPLoadCallback, src/renderer.cpp:106-131: WebPGetInfo + WebPDecodeRGBAInto); so does
// this front end, through the run-time library (libwebp.so.7; no header in this image, the two entry points are its stable ABI).
bool isWebp(const uint8_t* data, size_t size)
{
  return size >= 12 && memcmp(data, "RIFF", 4) == 0 && memcmp(data + 8, "WEBP", 4) == 0;
}

bool decodeWebp(const uint8_t* data, size_t size, Image& out, std::string* error)
{
  typedef int (*GetInfoFn)(const uint8_t*, size_t, int*, int*);
  typedef uint8_t* (*DecodeIntoFn)(const uint8_t*, size_t, uint8_t*, size_t, int);
  static void*        lib        = [] { void* h = dlopen("libwebp.so.7", RTLD_NOW | RTLD_LOCAL); return h ? h : dlopen("libwebp.so", RTLD_NOW | RTLD_LOCAL); }();
  static GetInfoFn    getInfo    = lib ? reinterpret_cast<GetInfoFn>(dlsym(lib, "WebPGetInfo")) : nullptr;
  static DecodeIntoFn decodeInto = lib ? reinterpret_cast<DecodeIntoFn>(dlsym(lib, "WebPDecodeRGBAInto")) : nullptr;
  auto fail = [&](const char* msg) {
    if(error)
      *error = std::string("WebP: ") + msg;
    return false;
  };
  if(!getInfo || !decodeInto)
    return fail("libwebp could not be loaded");
  int w = 0, h = 0;
  if(!getInfo(data, size, &w, &h) || w <= 0 || h <= 0 || !saneImageSize(uint64_t(w), uint64_t(h)))
    return fail("bad header");
  out.width  = w;
  out.height = h;
  out.rgba.assign(size_t(w) * size_t(h) * 4, 0);
  if(!decodeInto(data, size, out.rgba.data(), out.rgba.size(), w * 4))
    return fail("corrupt image data");
  return true;
}

}  // namespace mihost
