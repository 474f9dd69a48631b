// Small column-major float math used by the host layer (the reference uses glm, which is not vendored in
// /root/reference). mat4 layout is glm's: m[4*c + r].
#pragma once
#include <array>
#include <cmath>
#include <cstring>

namespace mx {

struct vec3
{
  float x = 0, y = 0, z = 0;
};
inline vec3  operator+(vec3 a, vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline vec3  operator-(vec3 a, vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline vec3  operator*(vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline vec3  cross(vec3 a, vec3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float length(vec3 a) { return std::sqrt(dot(a, a)); }
inline vec3  normalize(vec3 a)
{
  float l = length(a);
  return l > 0 ? a * (1.0f / l) : a;
}

struct mat4
{
  float m[16];
  float&       at(int c, int r) { return m[4 * c + r]; }
  const float& at(int c, int r) const { return m[4 * c + r]; }
};

inline mat4 identity()
{
  mat4 r{};
  for(int i = 0; i < 16; ++i)
    r.m[i] = (i % 5 == 0) ? 1.0f : 0.0f;
  return r;
}

inline mat4 mul(const mat4& a, const mat4& b)
{
  mat4 r{};
  for(int c = 0; c < 4; ++c)
    for(int rr = 0; rr < 4; ++rr)
    {
      float s = 0;
      for(int k = 0; k < 4; ++k)
        s += a.at(k, rr) * b.at(c, k);
      r.at(c, rr) = s;
    }
  return r;
}

inline vec3 transformPoint(const mat4& m, vec3 p)
{
  return {m.at(0, 0) * p.x + m.at(1, 0) * p.y + m.at(2, 0) * p.z + m.at(3, 0),
          m.at(0, 1) * p.x + m.at(1, 1) * p.y + m.at(2, 1) * p.z + m.at(3, 1),
          m.at(0, 2) * p.x + m.at(1, 2) * p.y + m.at(2, 2) * p.z + m.at(3, 2)};
}

inline mat4 translate(vec3 t)
{
  mat4 r     = identity();
  r.at(3, 0) = t.x;
  r.at(3, 1) = t.y;
  r.at(3, 2) = t.z;
  return r;
}
inline mat4 scale(vec3 s)
{
  mat4 r     = identity();
  r.at(0, 0) = s.x;
  r.at(1, 1) = s.y;
  r.at(2, 2) = s.z;
  return r;
}
// quaternion (x, y, z, w) -> rotation matrix (glm::mat4_cast)
inline mat4 fromQuat(float x, float y, float z, float w)
{
  mat4  r  = identity();
  float xx = x * x, yy = y * y, zz = z * z, xz = x * z, xy = x * y, yz = y * z, wx = w * x, wy = w * y, wz = w * z;
  r.at(0, 0) = 1 - 2 * (yy + zz);
  r.at(0, 1) = 2 * (xy + wz);
  r.at(0, 2) = 2 * (xz - wy);
  r.at(1, 0) = 2 * (xy - wz);
  r.at(1, 1) = 1 - 2 * (xx + zz);
  r.at(1, 2) = 2 * (yz + wx);
  r.at(2, 0) = 2 * (xz + wy);
  r.at(2, 1) = 2 * (yz - wx);
  r.at(2, 2) = 1 - 2 * (xx + yy);
  return r;
}

// General 4x4 inverse by cofactors (what glm::inverse does).
inline mat4 inverse(const mat4& a)
{
  const float* m = a.m;
  float        inv[16];
  inv[0]  = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
  inv[4]  = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
  inv[8]  = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
  inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
  inv[1]  = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
  inv[5]  = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
  inv[9]  = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
  inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
  inv[2]  = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
  inv[6]  = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
  inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
  inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
  inv[3]  = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
  inv[7]  = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
  inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
  inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
  float det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
  mat4  r{};
  float id = (det != 0.0f) ? 1.0f / det : 0.0f;
  for(int i = 0; i < 16; ++i)
    r.m[i] = inv[i] * id;
  return r;
}

// glm::lookAtRH
inline mat4 lookAt(vec3 eye, vec3 center, vec3 up)
{
  vec3 f     = normalize(center - eye);
  vec3 s     = normalize(cross(f, up));
  vec3 u     = cross(s, f);
  mat4 r     = identity();
  r.at(0, 0) = s.x;
  r.at(1, 0) = s.y;
  r.at(2, 0) = s.z;
  r.at(0, 1) = u.x;
  r.at(1, 1) = u.y;
  r.at(2, 1) = u.z;
  r.at(0, 2) = -f.x;
  r.at(1, 2) = -f.y;
  r.at(2, 2) = -f.z;
  r.at(3, 0) = -dot(s, eye);
  r.at(3, 1) = -dot(u, eye);
  r.at(3, 2) = dot(f, eye);
  return r;
}

// glm::perspectiveRH_ZO with the Vulkan y flip that nvutils::CameraManipulator::getPerspectiveMatrix applies.
inline mat4 perspectiveVk(float fovyRad, float aspect, float zNear, float zFar)
{
  float t = std::tan(fovyRad * 0.5f);
  mat4  r{};
  std::memset(r.m, 0, sizeof(r.m));
  r.at(0, 0) = 1.0f / (aspect * t);
  r.at(1, 1) = -1.0f / t;
  r.at(2, 2) = zFar / (zNear - zFar);
  r.at(2, 3) = -1.0f;
  r.at(3, 2) = -(zFar * zNear) / (zFar - zNear);
  return r;
}

// glm::orthoRH_ZO with the same y flip.
inline mat4 orthoVk(float left, float right, float bottom, float top, float zNear, float zFar)
{
  mat4 r     = identity();
  r.at(0, 0) = 2.0f / (right - left);
  r.at(1, 1) = -2.0f / (top - bottom);
  r.at(2, 2) = -1.0f / (zFar - zNear);
  r.at(3, 0) = -(right + left) / (right - left);
  r.at(3, 1) = (top + bottom) / (top - bottom);
  r.at(3, 2) = -zNear / (zFar - zNear);
  return r;
}

}  // namespace mx
