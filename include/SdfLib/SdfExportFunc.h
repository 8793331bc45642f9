// Handle-style C interface of the reference's Unity plugin (src/tools/SdfLibUnity/SdfExportFunc.h:16-58), implemented on
// top of the C++ classes above.  createOctreeSdf uses the CONTINUITY builder like the reference (SdfExportFunc.cpp:84-113).
// One difference, on purpose: deleteSdf frees every format (the reference leaks OCTREE objects, SdfExportFunc.cpp:170-182).
// Include in exactly one translation unit of the plugin (tools/SdfLibUnity/SdfExportFunc.cpp -> libSdfLibUnity.so).
// getDistances (batched) is an addition: one device launch instead of one per point.
#ifndef SDFLIB_EXPORT_FUNC_H
#define SDFLIB_EXPORT_FUNC_H
#include "OctreeSdf.h"
#include "ExactOctreeSdf.h"
#ifndef EXPORT
#define EXPORT extern "C" __attribute__((visibility("default")))
#endif
#include <cmath>
#include <cstdio>

// The reference's entry points never throw (its classes log and carry on); the classes here report failures (no HIP device,
// depth beyond the lattice limit, unreadable file ...) by exception.  None may unwind through the C ABI into Unity / ctypes:
// every body runs under this guard, logs to stderr and returns the neutral value (nullptr / NaN / 0).
namespace sdflib { namespace detail {
template <class R, class F> R guarded(const char* what, R onError, F&& f) noexcept {
    try { return f(); }
    catch (const std::exception& e) { std::fprintf(stderr, "[SdfLib] %s: %s\n", what, e.what()); }
    catch (...) { std::fprintf(stderr, "[SdfLib] %s: unknown error\n", what); }
    return onError;
}
template <class F> void guardedVoid(const char* what, F&& f) noexcept { guarded<int>(what, 0, [&] { f(); return 0; }); }
}}

EXPORT void saveSdf(sdflib::SdfFunction* sdf, char* path) {
    sdflib::detail::guardedVoid("saveSdf", [&] { if (sdf && path) sdf->saveToFile(std::string(path)); });
}
EXPORT sdflib::SdfFunction* loadSdf(char* path) {
    return sdflib::detail::guarded<sdflib::SdfFunction*>("loadSdf", nullptr, [&]() -> sdflib::SdfFunction* {
        return path ? sdflib::SdfFunction::loadFromFile(std::string(path)).release() : nullptr; });
}
EXPORT sdflib::SdfFunction* createExactOctreeSdf(glm::vec3* vertices, uint32_t numVertices, uint32_t* indices, uint32_t numIndices,
                                                 float bbMinX, float bbMinY, float bbMinZ, float bbMaxX, float bbMaxY, float bbMaxZ,
                                                 uint32_t startOctreeDepth, uint32_t maxOctreeDepth, uint32_t minTrianglesPerNode, uint32_t numThreads) {
    return sdflib::detail::guarded<sdflib::SdfFunction*>("createExactOctreeSdf", nullptr, [&]() -> sdflib::SdfFunction* {
        sdflib::Mesh mesh(vertices, numVertices, indices, numIndices);
        return new sdflib::ExactOctreeSdf(mesh, sdflib::BoundingBox(glm::vec3(bbMinX, bbMinY, bbMinZ), glm::vec3(bbMaxX, bbMaxY, bbMaxZ)), maxOctreeDepth,
                                          startOctreeDepth, minTrianglesPerNode, numThreads);
    });
}
EXPORT sdflib::SdfFunction* createOctreeSdf(glm::vec3* vertices, uint32_t numVertices, uint32_t* indices, uint32_t numIndices,
                                            float bbMinX, float bbMinY, float bbMinZ, float bbMaxX, float bbMaxY, float bbMaxZ,
                                            uint32_t startOctreeDepth, uint32_t maxOctreeDepth, float maxError, uint32_t numThreads) {
    return sdflib::detail::guarded<sdflib::SdfFunction*>("createOctreeSdf", nullptr, [&]() -> sdflib::SdfFunction* {
        sdflib::Mesh mesh(vertices, numVertices, indices, numIndices);
        return new sdflib::OctreeSdf(mesh, sdflib::BoundingBox(glm::vec3(bbMinX, bbMinY, bbMinZ), glm::vec3(bbMaxX, bbMaxY, bbMaxZ)), maxOctreeDepth,
                                     startOctreeDepth, maxError, sdflib::OctreeSdf::InitAlgorithm::CONTINUITY, numThreads);
    });
}
EXPORT float getDistance(sdflib::SdfFunction* sdf, float x, float y, float z) {
    return sdflib::detail::guarded<float>("getDistance", NAN, [&] { return sdf->getDistance(glm::vec3(x, y, z)); });
}
EXPORT float getDistanceAndGradient(sdflib::SdfFunction* sdf, float x, float y, float z, glm::vec3* outGradient) {
    return sdflib::detail::guarded<float>("getDistanceAndGradient", NAN, [&] { return sdf->getDistance(glm::vec3(x, y, z), *outGradient); });
}
EXPORT void getDistances(sdflib::SdfFunction* sdf, const glm::vec3* points, uint64_t n, float* outDistances, glm::vec3* outGradients) {
    sdflib::detail::guardedVoid("getDistances", [&] { sdf->getDistances(points, n, outDistances, outGradients); });
}
EXPORT glm::vec3 getBBMinPoint(sdflib::SdfFunction* sdf) {
    return sdflib::detail::guarded<glm::vec3>("getBBMinPoint", glm::vec3(0.f), [&] { return sdf->getSampleArea().min; });
}
EXPORT glm::vec3 getBBSize(sdflib::SdfFunction* sdf) {
    return sdflib::detail::guarded<glm::vec3>("getBBSize", glm::vec3(0.f), [&] { return sdf->getSampleArea().getSize(); });
}
// OctreeSdf only, x component, 0 for every other format — as the reference (SdfExportFunc.cpp:140-145)
EXPORT uint32_t getStartGridSize(sdflib::SdfFunction* sdf) {
    return sdflib::detail::guarded<uint32_t>("getStartGridSize", 0u, [&] {
        return sdf->getFormat() == sdflib::SdfFunction::OCTREE ? (uint32_t)static_cast<sdflib::OctreeSdf*>(sdf)->getStartGridSize().x : 0u; });
}
EXPORT uint32_t getOctreeDataSize(sdflib::SdfFunction* sdf) {
    return sdflib::detail::guarded<uint32_t>("getOctreeDataSize", 0u, [&] {
        return sdf->getFormat() == sdflib::SdfFunction::OCTREE ? (uint32_t)static_cast<sdflib::OctreeSdf*>(sdf)->getOctreeData().size() : 0u; });
}
EXPORT void getOctreeData(sdflib::SdfFunction* sdf, uint32_t* dst) {
    sdflib::detail::guardedVoid("getOctreeData", [&] {
        if (sdf->getFormat() != sdflib::SdfFunction::OCTREE) return;
        const auto& d = static_cast<sdflib::OctreeSdf*>(sdf)->getOctreeData();
        std::memcpy(dst, d.data(), d.size() * sizeof(uint32_t));
    });
}
EXPORT void deleteSdf(sdflib::SdfFunction* sdf) { sdflib::detail::guardedVoid("deleteSdf", [&] { delete sdf; }); }
#endif
