// sdflib::BoundingBox / sdflib::Mesh — API-compatible subset of the reference's include/SdfLib/utils/Mesh.h:16-106
// (the assimp loader is out of scope; meshes come from memory).
#ifndef SDFLIB_MESH_H
#define SDFLIB_MESH_H
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>
#include "glm_compat.h"

namespace sdflib {
struct BoundingBox {
    BoundingBox() : min(INFINITY), max(-INFINITY) {}
    BoundingBox(glm::vec3 mn, glm::vec3 mx) : min(mn), max(mx) {}
    glm::vec3 min, max;
    glm::vec3 getSize() const { return max - min; }
    glm::vec3 getCenter() const { return min + 0.5f * getSize(); }
    void addMargin(float margin) { min = min - glm::vec3(margin); max = max + glm::vec3(margin); }
};

class Mesh {
public:
    Mesh() {}
    // same semantics as the reference's raw-pointer constructor (src/utils/Mesh.cpp:34-42): copies, does not compute the bbox
    Mesh(glm::vec3* vertices, uint32_t numVertices, uint32_t* indices, uint32_t numIndices) {
        mVertices.resize(numVertices); std::memcpy(mVertices.data(), vertices, sizeof(glm::vec3) * numVertices);
        mIndices.resize(numIndices); std::memcpy(mIndices.data(), indices, sizeof(uint32_t) * numIndices);
    }
    std::vector<glm::vec3>& getVertices() { return mVertices; }
    const std::vector<glm::vec3>& getVertices() const { return mVertices; }
    std::vector<uint32_t>& getIndices() { return mIndices; }
    const std::vector<uint32_t>& getIndices() const { return mIndices; }
    const BoundingBox& getBoundingBox() const { return mBBox; }
    void computeBoundingBox() {
        glm::vec3 mn(INFINITY), mx(-INFINITY);
        for (const glm::vec3& v : mVertices) {
            mn.x = std::fmin(mn.x, v.x); mx.x = std::fmax(mx.x, v.x);
            mn.y = std::fmin(mn.y, v.y); mx.y = std::fmax(mx.y, v.y);
            mn.z = std::fmin(mn.z, v.z); mx.z = std::fmax(mx.z, v.z);
        }
        mBBox = BoundingBox(mn, mx);
    }
private:
    std::vector<glm::vec3> mVertices;
    std::vector<uint32_t> mIndices;
    BoundingBox mBBox;
};
}  // namespace sdflib
#endif
