#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace mihost {

struct Image
{
  int                  width = 0, height = 0;
  std::vector<uint8_t> rgba;  // width*height*4
};

// Image files are untrusted input: a header may claim any size.  Every decoder checks the claimed dimensions here before it allocates:
// sides up to 32768 (the device limit is 65535) and at most 2^28 texels (1 GiB of RGBA8), so that a few bytes cannot ask for terabytes.
inline bool saneImageSize(uint64_t w, uint64_t h)
{
  return w >= 1 && h >= 1 && w <= 32768 && h <= 32768 && w * h <= (1ull << 28);
}

bool    isPng(const uint8_t* data, size_t size);
bool    decodePng(const uint8_t* data, size_t size, Image& out, std::string* error);
bool    isJpeg(const uint8_t* data, size_t size);
bool    decodeJpeg(const uint8_t* data, size_t size, Image& out, std::string* error);  // jpeg_decoder.cpp
bool    isDds(const uint8_t* data, size_t size);
bool    decodeDds(const uint8_t* data, size_t size, Image& out, std::string* error);  // dds_decoder.cpp
// 4x4 block formats shared by the DDS and KTX containers (decoded on the host: there is no texture unit on this path)
enum class BlockFormat { BC1, BC2, BC3, BC4, BC5, BC7 };
bool    decodeBlocks(BlockFormat format, const uint8_t* blocks, size_t size, int width, int height, Image& out);  // dds_decoder.cpp
void    decodeBc7Block(const uint8_t* block16, uint8_t out[16][4]);                                             // bc7_decoder.cpp
bool    isKtx(const uint8_t* data, size_t size);
bool    decodeKtx(const uint8_t* data, size_t size, Image& out, std::string* error);  // ktx_decoder.cpp (KTX 1 and KTX 2)
bool    isWebp(const uint8_t* data, size_t size);
bool    decodeWebp(const uint8_t* data, size_t size, Image& out, std::string* error);  // ktx_decoder.cpp (through libwebp, like the reference)
bool    decodeImage(const uint8_t* data, size_t size, Image& out, std::string* error);
Image   magentaImage();
float   srgbToLinear(uint8_t v);
uint8_t linearToSrgb8(float c);
void    buildMipChain(const Image& base, bool srgb, std::vector<std::vector<uint8_t>>& levels);

}  // namespace mihost
