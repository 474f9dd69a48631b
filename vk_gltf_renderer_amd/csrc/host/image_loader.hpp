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

bool    isPng(const uint8_t* data, size_t size);
bool    decodePng(const uint8_t* data, size_t size, Image& out, std::string* error);
bool    isJpeg(const uint8_t* data, size_t size);
bool    decodeJpeg(const uint8_t* data, size_t size, Image& out, std::string* error);  // jpeg_decoder.cpp
bool    isDds(const uint8_t* data, size_t size);
bool    decodeDds(const uint8_t* data, size_t size, Image& out, std::string* error);  // dds_decoder.cpp
bool    decodeImage(const uint8_t* data, size_t size, Image& out, std::string* error);
Image   magentaImage();
float   srgbToLinear(uint8_t v);
uint8_t linearToSrgb8(float c);
void    buildMipChain(const Image& base, bool srgb, std::vector<std::vector<uint8_t>>& levels);

}  // namespace mihost
