// sdflib::BoundingBox / sdflib::Mesh — API-compatible subset of the reference's include/SdfLib/utils/Mesh.h:16-106
// Meshes come from memory or from OBJ / PLY files (minimal readers standing in for the reference's assimp loader).
#ifndef SDFLIB_MESH_H
#define SDFLIB_MESH_H
#include <cctype>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
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

namespace detail {
inline void fanTriangulate(const std::vector<long long>& poly, size_t numVertices, std::vector<uint32_t>& indices) {
    for (size_t k = 1; k + 1 < poly.size(); k++) {
        const long long tri[3] = {poly[0], poly[k], poly[k + 1]};
        for (long long i : tri) indices.push_back((uint32_t)(i >= 0 ? i : (long long)numVertices + i));
    }
}
inline bool readObj(std::istream& is, std::vector<glm::vec3>& vertices, std::vector<uint32_t>& indices) {
    std::string line;
    while (std::getline(is, line)) {
        std::istringstream ls(line);
        std::string tag; ls >> tag;
        if (tag == "v") { glm::vec3 p; ls >> p.x >> p.y >> p.z; vertices.push_back(p); }
        else if (tag == "f") {
            std::vector<long long> poly; std::string tok;
            while (ls >> tok) { const long long i = std::atoll(tok.substr(0, tok.find('/')).c_str()); poly.push_back(i > 0 ? i - 1 : i); }   // 1-based; negative = relative
            fanTriangulate(poly, vertices.size(), indices);
        }
    }
    return !vertices.empty() && !indices.empty();
}
inline size_t plyTypeSize(const std::string& t) {
    if (t == "char" || t == "uchar" || t == "int8" || t == "uint8") return 1;
    if (t == "short" || t == "ushort" || t == "int16" || t == "uint16") return 2;
    if (t == "int" || t == "uint" || t == "float" || t == "int32" || t == "uint32" || t == "float32") return 4;
    if (t == "double" || t == "float64") return 8;
    return 0;
}
inline double plyScalar(const char* p, const std::string& t) {
    if (t == "char" || t == "int8") { int8_t v; std::memcpy(&v, p, 1); return v; }
    if (t == "uchar" || t == "uint8") { uint8_t v; std::memcpy(&v, p, 1); return v; }
    if (t == "short" || t == "int16") { int16_t v; std::memcpy(&v, p, 2); return v; }
    if (t == "ushort" || t == "uint16") { uint16_t v; std::memcpy(&v, p, 2); return v; }
    if (t == "int" || t == "int32") { int32_t v; std::memcpy(&v, p, 4); return v; }
    if (t == "uint" || t == "uint32") { uint32_t v; std::memcpy(&v, p, 4); return v; }
    if (t == "float" || t == "float32") { float v; std::memcpy(&v, p, 4); return v; }
    double v; std::memcpy(&v, p, 8); return v;
}
inline bool readPly(std::istream& is, std::vector<glm::vec3>& vertices, std::vector<uint32_t>& indices) {
    struct Prop { std::string type, countType, itemType, name; bool list; };
    struct Element { std::string name; size_t count; std::vector<Prop> props; };
    std::vector<Element> elements; std::string format, line;
    while (std::getline(is, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        std::istringstream ls(line); std::string tag; ls >> tag;
        if (tag == "format") ls >> format;
        else if (tag == "element") { Element e; ls >> e.name >> e.count; elements.push_back(e); }
        else if (tag == "property" && !elements.empty()) {
            Prop p; ls >> p.type; p.list = p.type == "list";
            if (p.list) ls >> p.countType >> p.itemType;
            ls >> p.name; elements.back().props.push_back(p);
        } else if (tag == "end_header") break;
    }
    if (format != "ascii" && format != "binary_little_endian" && format != "binary_big_endian") return false;
    const bool ascii = format == "ascii", swapBytes = format == "binary_big_endian";      // hosts are little-endian (x86-64)
    std::vector<char> buf;
    auto readScalar = [&](const std::string& t) -> double {
        if (ascii) { double v = 0; is >> v; return v; }
        const size_t n = plyTypeSize(t); buf.resize(8); is.read(buf.data(), (std::streamsize)n);
        if (swapBytes) for (size_t a = 0, b = n ? n - 1 : 0; a < b; a++, b--) std::swap(buf[a], buf[b]);
        return plyScalar(buf.data(), t);
    };
    for (const Element& e : elements) {
        for (size_t i = 0; i < e.count; i++) {
            glm::vec3 p; std::vector<long long> poly;
            for (const Prop& pr : e.props) {
                if (pr.list) {
                    const size_t n = (size_t)readScalar(pr.countType);
                    std::vector<long long> items(n);
                    for (size_t k = 0; k < n; k++) items[k] = (long long)readScalar(pr.itemType);
                    if (e.name == "face" && poly.empty()) poly = items;
                } else {
                    const double v = readScalar(pr.type);
                    if (e.name == "vertex") { if (pr.name == "x") p.x = (float)v; else if (pr.name == "y") p.y = (float)v; else if (pr.name == "z") p.z = (float)v; }
                }
            }
            if (!is) return false;
            if (e.name == "vertex") vertices.push_back(p);
            else if (e.name == "face") fanTriangulate(poly, vertices.size(), indices);
        }
    }
    return !vertices.empty() && !indices.empty();
}
inline bool readMeshFile(const std::string& path, std::vector<glm::vec3>& vertices, std::vector<uint32_t>& indices) {
    std::ifstream is(path, std::ios::binary);
    if (!is.is_open()) return false;
    std::string ext = path.size() >= 4 ? path.substr(path.size() - 4) : std::string();
    for (char& ch : ext) ch = (char)std::tolower((unsigned char)ch);
    bool ok = false;
    if (ext == ".obj") ok = readObj(is, vertices, indices);
    else if (ext == ".ply") ok = readPly(is, vertices, indices);
    if (!ok) return false;
    for (uint32_t i : indices) if (i >= vertices.size()) return false;
    return true;
}
}  // namespace detail

class Mesh {
public:
    Mesh() {}
    // same semantics as the reference's raw-pointer constructor (src/utils/Mesh.cpp:34-42): copies, does not compute the bbox
    Mesh(glm::vec3* vertices, uint32_t numVertices, uint32_t* indices, uint32_t numIndices) {
        mVertices.resize(numVertices); std::memcpy(mVertices.data(), vertices, sizeof(glm::vec3) * numVertices);
        mIndices.resize(numIndices); std::memcpy(mIndices.data(), indices, sizeof(uint32_t) * numIndices);
    }
    // The reference's file constructor (src/utils/Mesh.cpp:9-62) goes through assimp (first aiMesh, faces triangulated); here:
    // an OBJ file's objects / groups are read as ONE mesh (assimp would split them and the reference keep the first only);
    // OBJ and ASCII / binary-little-endian PLY, polygons fan-triangulated, then the bounding box — which, as in the reference,
    // is what later enables the seam welding of the triangle data.  On failure the mesh stays empty (the reference logs and returns).
    explicit Mesh(const std::string& filePath) {
        if (!detail::readMeshFile(filePath, mVertices, mIndices)) { std::fprintf(stderr, "[error] Error with import model %s\n", filePath.c_str()); mVertices.clear(); mIndices.clear(); return; }
        computeBoundingBox();
    }
    // src/utils/Mesh.cpp:131-139
    void applyTransform(glm::mat4 trans) {
        for (glm::vec3& vert : mVertices) { const glm::vec4 r = trans * glm::vec4(vert, 1.0f); vert = glm::vec3(r.x, r.y, r.z); }
        computeBoundingBox();
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
