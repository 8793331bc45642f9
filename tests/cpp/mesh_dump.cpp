#include <cstdio>
#include "SdfLib/utils/Mesh.h"
int main(int argc, char** argv) {
    sdflib::Mesh m(argv[1]);
    FILE* f = std::fopen(argv[2], "wb");
    uint32_t nv = (uint32_t)m.getVertices().size(), ni = (uint32_t)m.getIndices().size();
    std::fwrite(&nv, 4, 1, f); std::fwrite(&ni, 4, 1, f);
    std::fwrite(m.getVertices().data(), 12, nv, f); std::fwrite(m.getIndices().data(), 4, ni, f);
    const sdflib::BoundingBox& b = m.getBoundingBox();
    std::fwrite(&b.min, 12, 1, f); std::fwrite(&b.max, 12, 1, f);
    std::fclose(f);
    return 0;
}
