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
#include <thread>
#include <cstdlib>
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
// Several GPUs for the constructors: SDFLIB_DEVICES="0,1,2,3" (or "all") in the environment makes OctreeSdf / ExactOctreeSdf build through
// sdfhip_multi_* — shards over the listed devices, RCCL all-gather, one replica of the tree per device; batched getDistances calls are
// then split over the replicas.  Unset (or one device): the single-device path.  The multi-process flavour is sdflib_amd/distributed.py.
inline sdfhip_multi* defaultMulti() {
    static sdfhip_multi* multi = nullptr; static bool tried = false;
    if (tried) return multi;
    tried = true;
    const char* env = std::getenv("SDFLIB_DEVICES");
    if (!env || !*env) return nullptr;
    std::vector<int> devs;
    if (std::string(env) == "all") { for (int d = 0; d < 64; d++) { sdfhip_ctx* c = nullptr; if (sdfhip_ctx_create(d, nullptr, SDFHIP_STREAM_PRIVATE, &c) != SDFHIP_OK) break; sdfhip_ctx_destroy(c); devs.push_back(d); } }
    else { std::string tok; for (const char* q = env;; q++) { if (*q == ',' || *q == 0) { if (!tok.empty()) devs.push_back(std::atoi(tok.c_str())); tok.clear(); if (!*q) break; } else tok.push_back(*q); } }
    if (devs.size() < 2) return nullptr;
    if (sdfhip_multi_create(devs.data(), (int)devs.size(), &multi) != SDFHIP_OK) throw std::runtime_error(std::string("sdfhip: ") + sdfhip_last_error());
    return multi;
}
// fn(part, begin, end) on one host thread per part of [0, n)
template <typename F> inline void splitOver(size_t parts, size_t n, F fn) {
    std::vector<std::thread> th; std::vector<std::string> err(parts);
    for (size_t k = 0; k < parts; k++) {
        const size_t b = n * k / parts, e = n * (k + 1) / parts;
        auto job = [&, k, b, e]() { try { fn(k, b, e); } catch (const std::exception& x) { err[k] = x.what(); } };
        if (k + 1 < parts) th.emplace_back(job); else job();
    }
    for (std::thread& t : th) t.join();
    for (const std::string& m : err) if (!m.empty()) throw std::runtime_error(m);
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
