// C++ drop-in check: the reference's class API (OctreeSdf / ExactOctreeSdf / SdfFunction) on top of libsdfhip.
// Built and run by tests/test_cpp_api.py (GPU) — prints values the Python side compares with the oracle.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "SdfLib/OctreeSdf.h"
#include "SdfLib/ExactOctreeSdf.h"

int main(int argc, char** argv) {
    if (argc < 4) { std::fprintf(stderr, "usage: %s vertices.bin indices.bin points.bin\n", argv[0]); return 2; }
    auto readAll = [](const char* path) { std::vector<char> b; FILE* f = std::fopen(path, "rb"); if (!f) std::exit(3); std::fseek(f, 0, SEEK_END); long n = std::ftell(f); std::fseek(f, 0, SEEK_SET); b.resize(n); if (std::fread(b.data(), 1, n, f) != (size_t)n) std::exit(3); std::fclose(f); return b; };
    std::vector<char> vb = readAll(argv[1]), ib = readAll(argv[2]), pb = readAll(argv[3]);
    sdflib::Mesh mesh(reinterpret_cast<glm::vec3*>(vb.data()), (uint32_t)(vb.size() / 12), reinterpret_cast<uint32_t*>(ib.data()), (uint32_t)(ib.size() / 4));
    mesh.computeBoundingBox();
    sdflib::BoundingBox box = mesh.getBoundingBox();
    const glm::vec3 s = box.getSize();
    box.addMargin(0.2f * std::fmax(std::fmax(s.x, s.y), s.z));
    const glm::vec3* pts = reinterpret_cast<const glm::vec3*>(pb.data());
    const size_t n = pb.size() / 12;

    sdflib::OctreeSdf oct(mesh, box, 5, 2, 1e-3f, sdflib::OctreeSdf::InitAlgorithm::NO_CONTINUITY, 2);
    std::printf("octree words %zu grid %d range %.9g border %.9g\n", oct.getOctreeData().size(), oct.getStartGridSize().x, oct.getOctreeValueRange(), oct.getOctreeMinBorderValue());
    std::vector<float> d(n); std::vector<glm::vec3> g(n);
    const sdflib::SdfFunction& f = oct;
    f.getDistances(pts, n, d.data(), g.data());
    size_t mism = 0;
    for (size_t i = 0; i < n; i++) {
        glm::vec3 gs; const float ds = f.getDistance(pts[i], gs);
        if (ds != d[i] || gs.x != g[i].x || gs.y != g[i].y || gs.z != g[i].z) mism++;
    }
    std::printf("octree scalar-vs-batched mismatches %zu\n", mism);
    FILE* fo = std::fopen(argv[4], "wb"); std::fwrite(d.data(), 4, n, fo); std::fclose(fo);

    sdflib::ExactOctreeSdf ex(mesh, box, 5, 1, 16);
    std::vector<float> de(n);
    ex.getDistances(pts, n, de.data());
    std::printf("exact nodes %zu maxleaf %u d0 %.9g scalar %.9g\n", ex.getOctreeData().size(), ex.getMaxTrianglesInLeafs(), de[0], ex.getDistance(pts[0]));
    fo = std::fopen(argv[5], "wb"); std::fwrite(de.data(), 4, n, fo); std::fclose(fo);
    return mism == 0 ? 0 : 1;
}
