// Mikkelsen tangent spaces for a triangle list (mikktspace_tangents.cpp).
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace mihost {

// positions / normals: 3 floats per vertex, texCoords: 2 floats per vertex, indices: 3 per triangle.  cornerTangents receives 4 floats
// per triangle corner: the unit tangent (direction of increasing s in the vertex's tangent plane) and +1 when the uv mapping
// preserves orientation at that corner, -1 otherwise -- the values Mikkelsen's `setTSpaceBasic` callback reports.
void mikkTangentSpaces(const float* positions, const float* normals, const float* texCoords, const uint32_t* indices, size_t numTriangles,
                       std::vector<float>& cornerTangents);

}  // namespace mihost
