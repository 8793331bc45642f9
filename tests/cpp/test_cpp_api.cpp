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
    std::printf("octree words %zu grid %d range %.9g border %.9g replicas %zu\n", oct.getOctreeData().size(), oct.getStartGridSize().x, oct.getOctreeValueRange(), oct.getOctreeMinBorderValue(),
                oct.getNumDeviceReplicas());
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

    // saveToFile / loadFromFile round trip (argv[6], argv[7]) + getDepthDensity + getTrianglesData
    size_t ioMism = 0;
    if (argc >= 8) {
        if (!oct.saveToFile(argv[6]) || !ex.saveToFile(argv[7])) { std::printf("save failed\n"); return 1; }
        std::unique_ptr<sdflib::SdfFunction> lo = sdflib::SdfFunction::loadFromFile(argv[6]), le = sdflib::SdfFunction::loadFromFile(argv[7]);
        if (!lo || !le || lo->getFormat() != sdflib::SdfFunction::OCTREE || le->getFormat() != sdflib::SdfFunction::EXACT_OCTREE) { std::printf("load failed\n"); return 1; }
        std::vector<float> d2(n), e2(n); std::vector<glm::vec3> g2(n);
        lo->getDistances(pts, n, d2.data(), g2.data());
        le->getDistances(pts, n, e2.data());
        for (size_t i = 0; i < n; i++) if (d2[i] != d[i] || g2[i].x != g[i].x || e2[i] != de[i] || lo->getDistance(pts[i]) != d[i]) ioMism++;
        std::printf("reloaded-vs-built mismatches %zu\n", ioMism);
        if (sdflib::SdfFunction::loadFromFile("/nonexistent/file.bin")) { std::printf("load of a missing file must fail\n"); return 1; }
        std::vector<float> dens; oct.getDepthDensity(dens);
        float total = 0.f; for (float v : dens) total += v;
        std::printf("depth density levels %zu total %.9g\n", dens.size(), total * 1.0f);
        const auto td = ex.getTrianglesData();
        std::printf("triangles %zu normal0 %.9g %.9g %.9g\n", td.size(), td[0].getTriangleNormal().x, td[0].getTriangleNormal().y, td[0].getTriangleNormal().z);
        // deep copies (the reference's classes are implicitly copyable: OctreeSdf.h:146-172, ExactOctreeSdf.h:91-93): a copy answers like the
        // original, has its own device tree (it survives the original) and its own host array
        size_t copyMism = 0;
        {
            std::unique_ptr<sdflib::OctreeSdf> oc2;
            std::unique_ptr<sdflib::ExactOctreeSdf> ex2;
            {
                sdflib::OctreeSdf octCopy(oct);              // copy construction
                sdflib::OctreeSdf octAssigned; octAssigned = octCopy;          // copy assignment
                oc2.reset(new sdflib::OctreeSdf(octAssigned));
                sdflib::ExactOctreeSdf exCopy(ex);
                sdflib::ExactOctreeSdf exAssigned; exAssigned = exCopy;
                ex2.reset(new sdflib::ExactOctreeSdf(exAssigned));
            }      // the intermediate copies are gone here
            std::vector<float> d3(n), e3(n); std::vector<glm::vec3> g3(n);
            oc2->getDistances(pts, n, d3.data(), g3.data());
            ex2->getDistances(pts, n, e3.data());
            for (size_t i = 0; i < n; i++) if (d3[i] != d[i] || g3[i].x != g[i].x || g3[i].y != g[i].y || g3[i].z != g[i].z || e3[i] != de[i] || oc2->getDistance(pts[i]) != d[i]) copyMism++;
            if (oc2->getOctreeData().size() != oct.getOctreeData().size() || oc2->getOctreeData().data() == oct.getOctreeData().data()) copyMism++;
            oc2->getOctreeData()[0].childrenIndex ^= 1u;              // the copy's array is its own
            if (oc2->getOctreeData()[0].childrenIndex == oct.getOctreeData()[0].childrenIndex) copyMism++;
            if (ex2->getMaxTrianglesInLeafs() != ex.getMaxTrianglesInLeafs() || ex2->getOctreeData().size() != ex.getOctreeData().size()) copyMism++;
        }
        std::printf("copy-vs-original mismatches %zu\n", copyMism);
        sdflib::ExactOctreeSdf moved = std::move(ex);
        std::printf("moved exact scalar %.9g\n", moved.getDistance(pts[0]));
    }
    // a reloaded tree may legitimately differ from the built one far from the origin (cell size from the stored box, OctreeSdf.h:233):
    // the Python side judges that count
    return mism == 0 ? 0 : 1;
}
