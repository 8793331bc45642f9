// sdflib::SdfFunction — the reference's abstract query interface (include/SdfLib/SdfFunction.h:12-58), plus the batched
// form the GPU engine is built for.  saveToFile / loadFromFile (cereal format) are a "next" row and not provided.
#ifndef SDFLIB_SDF_FUNCTION_H
#define SDFLIB_SDF_FUNCTION_H
#include <cstddef>
#include <memory>
#include <stdexcept>
#include <string>
#include "utils/Mesh.h"
#include "../sdfhip.h"

namespace sdflib {
class SdfFunction {
public:
    enum SdfFormat { GRID, OCTREE, EXACT_OCTREE, NONE };
    virtual ~SdfFunction() = default;
    virtual float getDistance(glm::vec3 sample) const = 0;
    virtual float getDistance(glm::vec3 sample, glm::vec3& outGradient) const = 0;
    virtual BoundingBox getSampleArea() const = 0;
    virtual SdfFormat getFormat() const { return SdfFormat::NONE; }
    // batched getDistance on the GPU: n points (host pointers); outGradients may be null
    virtual void getDistances(const glm::vec3* samples, size_t n, float* outDistances, glm::vec3* outGradients = nullptr) const = 0;
};

namespace detail {
// one context per process and device, created on first use; throws if no HIP device exists (there is no CPU fallback)
inline sdfhip_ctx* defaultContext() {
    static sdfhip_ctx* ctx = nullptr;
    if (!ctx && sdfhip_ctx_create(0, nullptr, SDFHIP_STREAM_PRIVATE, &ctx) != SDFHIP_OK)
        throw std::runtime_error(std::string("sdfhip: ") + sdfhip_last_error());
    return ctx;
}
inline void check(int rc) { if (rc != SDFHIP_OK) throw std::runtime_error(std::string("sdfhip: ") + sdfhip_last_error()); }
}  // namespace detail
}  // namespace sdflib
#endif
