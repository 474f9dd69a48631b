// TEST INFRASTRUCTURE ONLY: C entry points over csrc/host/meshopt_decoder.cpp for tests/test_meshopt.py (the product reaches the decoders through the glTF
// loader only).  Every function returns 1 on success, 0 on a refused stream, and copies the decoder's message into `err` (256 bytes).
#include "meshopt_decoder.hpp"

#include <cstring>

namespace {
int done(bool ok, const std::string& e, char* err)
{
  if(err)
  {
    strncpy(err, e.c_str(), 255);
    err[255] = 0;
  }
  return ok ? 1 : 0;
}
}  // namespace

extern "C" {
__attribute__((visibility("default"))) int mo_vertices(uint8_t* dst, size_t count, size_t stride, const uint8_t* src, size_t n, char* err)
{
  std::string e;
  return done(meshopt::decodeVertexBuffer(dst, count, stride, src, n, e), e, err);
}
__attribute__((visibility("default"))) int mo_triangles(uint8_t* dst, size_t count, size_t stride, const uint8_t* src, size_t n, char* err)
{
  std::string e;
  return done(meshopt::decodeIndexBuffer(dst, count, stride, src, n, e), e, err);
}
__attribute__((visibility("default"))) int mo_sequence(uint8_t* dst, size_t count, size_t stride, const uint8_t* src, size_t n, char* err)
{
  std::string e;
  return done(meshopt::decodeIndexSequence(dst, count, stride, src, n, e), e, err);
}
__attribute__((visibility("default"))) int mo_filter(int which, uint8_t* data, size_t count, size_t stride, char* err)
{
  std::string e;
  const bool  ok = which == 0 ? meshopt::filterOctahedral(data, count, stride, e) : (which == 1 ? meshopt::filterQuaternion(data, count, stride, e) : meshopt::filterExponential(data, count, stride, e));
  return done(ok, e, err);
}
}
