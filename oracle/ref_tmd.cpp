// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// oracle/_ref/libtmd_ref.so: the REAL reference's nearest-triangle search, compiled from the reference's own
// header where it lies (no copy, no stand-in: the header needs only the C++ standard library),
//   /root/reference/libs/InteractiveComputerGraphics/InteractiveComputerGraphics/TriangleMeshDistance.h
// behind a small C ABI, used exactly the way SdfLib's OctreeSdf build uses it:
//   ICG::ICG                 include/SdfLib/TrianglesInfluence.h:884-889  (float vertices widened to double,
//                            indices reinterpreted as std::array<int,3>, the std::vector constructor)
//   ICG::getNearestTriangle  include/SdfLib/TrianglesInfluence.h:898-905  (signed_distance({x, y, z}).triangle_id)
// It pins oracle/orc_bvh.h: tests/test_oracle_ref_pin.py compares triangle ids, distances
// and the whole BVH node array with the restatement's.  The node array is a private member of the class; it is
// read here through the explicit-instantiation access rule (no edit of, and no macro games around, the header).
#include <cstdint>
#include <cstddef>
#include <cstring>
#include <vector>
#include <array>
#include <new>
#include "TriangleMeshDistance.h"

namespace {
// [temp.explicit]/12: access checking is not applied to the arguments of an explicit instantiation, which lets a
// friend function defined in the instantiated template hand out a pointer to a private data member.
template <class Tag, auto Member>
struct Expose {
    friend auto exposed(Tag) { return Member; }
};
struct NodesTag { friend auto exposed(NodesTag); };
struct RootTag { friend auto exposed(RootTag); };
template struct Expose<NodesTag, &tmd::TriangleMeshDistance::nodes>;
template struct Expose<RootTag, &tmd::TriangleMeshDistance::root_bv>;

struct Handle {
    tmd::TriangleMeshDistance* mesh;
};
}  // namespace

extern "C" {

// as ICG's constructor: toDoubleVector(vertices) + the index vector seen as std::array<int, 3> triples
void* tmdref_create(const float* xyz, uint32_t numVertices, const uint32_t* indices, uint32_t numTriangles) {
    std::vector<std::array<double, 3>> verts(numVertices);
    for (uint32_t i = 0; i < numVertices; i++)
        verts[i] = {static_cast<double>(xyz[3 * i]), static_cast<double>(xyz[3 * i + 1]), static_cast<double>(xyz[3 * i + 2])};
    std::vector<std::array<int, 3>> tris(numTriangles);
    std::memcpy(tris.data(), indices, sizeof(int) * 3 * (size_t)numTriangles);
    Handle* h = new Handle;
    h->mesh = new tmd::TriangleMeshDistance(verts, tris);
    return h;
}

void tmdref_destroy(void* handle) {
    Handle* h = static_cast<Handle*>(handle);
    if (!h) return;
    delete h->mesh;
    delete h;
}

// as ICG::getNearestTriangle / ICG::getDistance: the float point goes into the std::array<double, 3> overload
void tmdref_nearest(void* handle, const float* pts, uint64_t n, uint32_t* outIds, double* outSignedDist /* nullable */) {
    const tmd::TriangleMeshDistance& m = *static_cast<Handle*>(handle)->mesh;
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t i = 0; i < (int64_t)n; i++) {
        const float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
        tmd::Result r = m.signed_distance({x, y, z});
        outIds[i] = static_cast<uint32_t>(r.triangle_id);
        if (outSignedDist) outSignedDist[i] = r.distance;
    }
}

uint64_t tmdref_num_nodes(void* handle) {
    const tmd::TriangleMeshDistance& m = *static_cast<Handle*>(handle)->mesh;
    return (m.*exposed(NodesTag{})).size();
}

// spheres: 8 doubles per node = left (cx, cy, cz, r), right (cx, cy, cz, r); leftRight: 2 ints per node
// (left == -1: leaf, right = triangle id) — the layout of oracle's orc_bvh_export.  rootSphere: 4 doubles.
void tmdref_export_nodes(void* handle, double* spheres, int32_t* leftRight, double* rootSphere /* nullable */) {
    const tmd::TriangleMeshDistance& m = *static_cast<Handle*>(handle)->mesh;
    const auto& nodes = m.*exposed(NodesTag{});
    for (size_t i = 0; i < nodes.size(); i++) {
        const auto& nd = nodes[i];
        double* s = spheres + 8 * i;
        s[0] = nd.bv_left.center[0]; s[1] = nd.bv_left.center[1]; s[2] = nd.bv_left.center[2]; s[3] = nd.bv_left.radius;
        s[4] = nd.bv_right.center[0]; s[5] = nd.bv_right.center[1]; s[6] = nd.bv_right.center[2]; s[7] = nd.bv_right.radius;
        leftRight[2 * i] = nd.left;
        leftRight[2 * i + 1] = nd.right;
    }
    if (rootSphere) {
        const auto& r = m.*exposed(RootTag{});
        rootSphere[0] = r.center[0]; rootSphere[1] = r.center[1]; rootSphere[2] = r.center[2]; rootSphere[3] = r.radius;
    }
}

}  // extern "C"
