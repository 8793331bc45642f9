// sdflib::SdfFunction — the reference's abstract query interface (include/SdfLib/SdfFunction.h:12-58), plus the batched
// form the GPU engine is built for.  saveToFile / loadFromFile write and read the reference's on-disk layout (cereal 1.3.2
// PortableBinary, src/sdf/SdfFunction.cpp:9-79) restated without cereal: a flag byte 1 (little-endian archive), then the
// fields in declaration order as raw little-endian scalars, std::vector = u64 count + elements.  Parity unpinned: no file
// written by an upstream build exists in this environment to compare with.
#ifndef SDFLIB_SDF_FUNCTION_H
#define SDFLIB_SDF_FUNCTION_H
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <fstream>
#include <memory>
#include <vector>
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

    // src/sdf/SdfFunction.cpp:9-41 — false (and a message on stderr) if the file cannot be opened or the format is unknown
    bool saveToFile(const std::string& outputPath) {
        std::ofstream os(outputPath, std::ios::out | std::ios::binary);
        if (!os.is_open()) { std::fprintf(stderr, "[error] Cannot open file %s\n", outputPath.c_str()); return false; }
        const SdfFormat format = getFormat();
        if (format != SdfFormat::OCTREE && format != SdfFormat::EXACT_OCTREE) { std::fprintf(stderr, "[error] Unknown format to save\n"); return false; }
        const uint8_t littleEndian = 1; const int32_t f = (int32_t)format;
        os.write(reinterpret_cast<const char*>(&littleEndian), 1);
        os.write(reinterpret_cast<const char*>(&f), 4);
        writePayload(os);
        return os.good();
    }
    // src/sdf/SdfFunction.cpp:43-79 — empty pointer on failure; defined in SdfLoad.h (needs the concrete classes)
    static std::unique_ptr<SdfFunction> loadFromFile(const std::string& inputPath);

protected:
    virtual void writePayload(std::ostream&) const {}
};

namespace detail {
// one context per process and device, created on first use; throws if no HIP device exists (there is no CPU fallback)
inline sdfhip_ctx* defaultContext() {
    static sdfhip_ctx* ctx = nullptr;
    if (!ctx && sdfhip_ctx_create(0, nullptr, SDFHIP_STREAM_PRIVATE, &ctx) != SDFHIP_OK)
        throw std::runtime_error(std::string("sdfhip: ") + sdfhip_last_error());
    return ctx;
}
template <typename T> inline void put(std::ostream& os, const T& v) { os.write(reinterpret_cast<const char*>(&v), sizeof(T)); }
template <typename T> inline void putVec(std::ostream& os, const T* p, uint64_t n) { put(os, n); os.write(reinterpret_cast<const char*>(p), (std::streamsize)(n * sizeof(T))); }
template <typename T> inline bool get(std::istream& is, T& v) { is.read(reinterpret_cast<char*>(&v), sizeof(T)); return is.good(); }
template <typename T> inline bool getVec(std::istream& is, std::vector<T>& v) {
    uint64_t n = 0; if (!get(is, n) || n > (1ull << 36) / sizeof(T)) return false;
    v.resize(n); is.read(reinterpret_cast<char*>(v.data()), (std::streamsize)(n * sizeof(T))); return n == 0 || is.good();
}
inline void check(int rc) { if (rc != SDFHIP_OK) throw std::runtime_error(std::string("sdfhip: ") + sdfhip_last_error()); }
}  // namespace detail
}  // namespace sdflib
#endif
