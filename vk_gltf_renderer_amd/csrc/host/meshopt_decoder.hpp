// Decoders of the EXT_meshopt_compression / KHR_meshopt_compression bitstreams (the reference calls meshoptimizer for them:
// src/gltf_scene.cpp:372-470, decompressMeshoptExtension).  Written from the extension's bitstream specification; every function is
// bounds-checked on untrusted input and verifies that the stream ends where its layout says it must, so a misread stream is an error,
// not silent geometry.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>

namespace meshopt {

// mode ATTRIBUTES: `count` vertices of `stride` bytes (a multiple of 4, at most 256).  Codec versions 0 (EXT_meshopt_compression) and 1 (KHR_).
bool decodeVertexBuffer(uint8_t* dst, size_t count, size_t stride, const uint8_t* src, size_t srcSize, std::string& err);
// mode TRIANGLES: `count` indices (a multiple of 3) of `stride` = 2 or 4 bytes.  Codec versions 0 and 1.
bool decodeIndexBuffer(uint8_t* dst, size_t count, size_t stride, const uint8_t* src, size_t srcSize, std::string& err);
// mode INDICES: `count` indices of `stride` = 2 or 4 bytes.
bool decodeIndexSequence(uint8_t* dst, size_t count, size_t stride, const uint8_t* src, size_t srcSize, std::string& err);

// filters, in place on the decoded data
bool filterOctahedral(uint8_t* data, size_t count, size_t stride, std::string& err);  // stride 4 (int8 x 4) or 8 (int16 x 4)
bool filterQuaternion(uint8_t* data, size_t count, size_t stride, std::string& err);  // stride 8 (int16 x 4)
bool filterExponential(uint8_t* data, size_t count, size_t stride, std::string& err); // stride a multiple of 4

}  // namespace meshopt
