// TriangleUtils::TriangleData — the 148-byte per-triangle record of the reference (include/SdfLib/utils/TriangleUtils.h:18-71):
// local frame, in-plane edge directions and the edge / vertex pseudonormals.  Produced on the device (sdfhip_mesh_create);
// this header only carries the layout so ExactOctreeSdf::getTrianglesData() and the .bin payload have a C++ type.
#ifndef SDFLIB_TRIANGLE_UTILS_H
#define SDFLIB_TRIANGLE_UTILS_H
#include <array>
#include "glm_compat.h"

namespace sdflib {
namespace TriangleUtils {
struct TriangleData {
    glm::vec3 origin;
    float transform[9];                    // glm::mat3, column-major
    float b[2], c[2];
    float v2;
    float v3[2];
    std::array<glm::vec3, 3> edgesNormal;
    std::array<glm::vec3, 3> verticesNormal;
    glm::vec3 getTriangleNormal() const { return glm::vec3(transform[2], transform[5], transform[8]); }
};
static_assert(sizeof(TriangleData) == 148, "TriangleData must stay 148 bytes (TriangleUtils.h:56-71)");
}  // namespace TriangleUtils
}  // namespace sdflib
#endif
