// SdfFunction::loadFromFile (src/sdf/SdfFunction.cpp:43-79): reads the format tag and dispatches to the concrete class.
// Included at the end of OctreeSdf.h and ExactOctreeSdf.h; do not include directly.
#ifndef SDFLIB_SDF_LOAD_H
#define SDFLIB_SDF_LOAD_H
#include "OctreeSdf.h"
#include "ExactOctreeSdf.h"

namespace sdflib {
inline std::unique_ptr<SdfFunction> SdfFunction::loadFromFile(const std::string& inputPath) {
    std::ifstream is(inputPath, std::ios::binary);
    if (!is.is_open()) { std::fprintf(stderr, "[error] Cannot open file %s\n", inputPath.c_str()); return std::unique_ptr<SdfFunction>(); }
    uint8_t littleEndian = 0; int32_t format = (int32_t)SdfFormat::NONE;
    if (!detail::get(is, littleEndian) || littleEndian != 1 || !detail::get(is, format)) {
        std::fprintf(stderr, "[error] Unknown file format\n"); return std::unique_ptr<SdfFunction>();
    }
    if (format == (int32_t)SdfFormat::OCTREE) {
        std::unique_ptr<OctreeSdf> obj(new OctreeSdf());
        if (obj->readPayload(is)) return obj;
    } else if (format == (int32_t)SdfFormat::EXACT_OCTREE) {
        std::unique_ptr<ExactOctreeSdf> obj(new ExactOctreeSdf());
        if (obj->readPayload(is)) return obj;
    }
    std::fprintf(stderr, "[error] Unknown file format\n");      // GRID (UniformGridSdf) is outside the accelerated path
    return std::unique_ptr<SdfFunction>();
}
}  // namespace sdflib
#endif
