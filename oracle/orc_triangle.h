// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_math.h header).
//
// Restates (reference file:line):
//   TriangleData ctor                         include/SdfLib/utils/TriangleUtils.h:23-42
//   getSqDistPointAndTriangle(point, data)     include/SdfLib/utils/TriangleUtils.h:76-135
//   getSignedDistPointAndTriangle(point, data) include/SdfLib/utils/TriangleUtils.h:137-196
//   ... (point, data, v1, v2, v3, outNormal)   include/SdfLib/utils/TriangleUtils.h:198-290
//   ... (point, data, outNormal)               include/SdfLib/utils/TriangleUtils.h:292-376
//   getSqDistPointAndTriangle(p, a, b, c)      include/SdfLib/utils/TriangleUtils.h:383-404
//   calculateMeshTriangleData (live branches)  src/utils/TriangleUtils.cpp:7-86, 422-427
// The four point/triangle variants of the reference repeat the same 2-D Voronoi-region tests; here the
// region is classified once (same comparisons, same operand order) and each variant switches on it.
// Not restated: the degenerate-triangle branches (disabled in the reference by `if(false && ...)`, TriangleUtils.cpp:45).
// The non-manifold seam welding (TriangleUtils.cpp:292-420) is restated and runs when a mesh bounding box is supplied:
// the reference reads mesh.getBoundingBox(), which only the file loader computes — a Mesh built from raw pointers
// (src/utils/Mesh.cpp:34-42) carries (+inf,-inf), which makes the welding threshold 0 and the pass a no-op.
#pragma once
#include "orc_math.h"
#include <vector>
#include <map>
#include <utility>
#include <algorithm>

namespace orc {

// 37 floats = 148 bytes, field order as in the reference struct (TriangleUtils.h:56-71).
struct TriangleData {
    V3 origin;
    M3 transform;
    V2 b, c;
    float v2;
    V2 v3;
    V3 edgesNormal[3];
    V3 verticesNormal[3];
    V3 normal() const { return V3{transform.c[0].z, transform.c[1].z, transform.c[2].z}; }
};
static_assert(sizeof(TriangleData) == 148, "TriangleData must be 148 bytes");

static inline TriangleData makeTriangleData(V3 p1, V3 p2, V3 p3) {
    TriangleData d;
    d.origin = p1;
    V3 sx = normalize(p2 - p1);
    V3 sz = normalize(cross(p2 - p1, p3 - p1));
    V3 sy = cross(sz, sx);
    M3 frame; frame.c[0] = sx; frame.c[1] = sy; frame.c[2] = sz;
    d.transform = inverse(frame);
    V3 e = mul(d.transform, p3 - p2);
    d.b = normalize(V2{e.x, e.y});
    e = mul(d.transform, p1 - p3);
    d.c = normalize(V2{e.x, e.y});
    d.v2 = mul(d.transform, p2 - d.origin).x;
    e = mul(d.transform, p3 - d.origin);
    d.v3 = V2{e.x, e.y};
    for (int k = 0; k < 3; k++) { d.edgesNormal[k] = v3(0.f, 0.f, 1.f); d.verticesNormal[k] = v3(0.f, 0.f, 1.f); }
    return d;
}

enum Region { R_V1, R_V2, R_V3, R_E1, R_E2, R_E3, R_F };

struct Proj { V3 p; float de1, de2, de3; Region r; };

static inline Proj classify(V3 point, const TriangleData& d) {
    Proj o;
    o.p = mul(d.transform, point - d.origin);
    const V3 p = o.p;
    o.de1 = -p.y;
    o.de2 = (p.x - d.v2) * d.b.y - p.y * d.b.x;
    o.de3 = p.x * d.c.y - p.y * d.c.x;
    if (o.de1 >= 0) {
        if (p.x <= 0) o.r = R_V1;
        else if (p.x >= d.v2) o.r = R_V2;
        else o.r = R_E1;
    } else if (o.de2 >= 0) {
        if ((p.x - d.v2) * d.b.x + p.y * d.b.y <= 0) o.r = R_V2;
        else if ((p.x - d.v3.x) * d.b.x + (p.y - d.v3.y) * d.b.y >= 0) o.r = R_V3;
        else o.r = R_E2;
    } else if (o.de3 >= 0) {
        if (p.x * d.c.x + p.y * d.c.y >= 0) o.r = R_V1;
        else if ((p.x - d.v3.x) * d.c.x + (p.y - d.v3.y) * d.c.y <= 0) o.r = R_V3;
        else o.r = R_E3;
    } else o.r = R_F;
    return o;
}

static inline float sqDistPointTriangle(V3 point, const TriangleData& d) {
    const Proj o = classify(point, d);
    const V3 p = o.p;
    switch (o.r) {
        case R_V1: return dot(p, p);
        case R_V2: { V3 q = p - v3(d.v2, 0.f, 0.f); return dot(q, q); }
        case R_V3: { V3 q = p - v3(d.v3.x, d.v3.y, 0.f); return dot(q, q); }
        case R_E1: return o.de1 * o.de1 + p.z * p.z;
        case R_E2: return o.de2 * o.de2 + p.z * p.z;
        case R_E3: return o.de3 * o.de3 + p.z * p.z;
        default:   return p.z * p.z;
    }
}

static inline float signedDistPointTriangle(V3 point, const TriangleData& d) {
    const Proj o = classify(point, d);
    const V3 p = o.p;
    switch (o.r) {
        case R_V1: return gsign(dot(d.verticesNormal[0], p)) * std::sqrt(dot(p, p));
        case R_V2: { V3 q = p - v3(d.v2, 0.f, 0.f); return gsign(dot(d.verticesNormal[1], q)) * std::sqrt(dot(q, q)); }
        case R_V3: { V3 q = p - v3(d.v3.x, d.v3.y, 0.f); return gsign(dot(d.verticesNormal[2], q)) * std::sqrt(dot(q, q)); }
        case R_E1: return gsign(dot(d.edgesNormal[0], p)) * std::sqrt(o.de1 * o.de1 + p.z * p.z);
        case R_E2: return gsign(dot(d.edgesNormal[1], p - v3(d.v2, 0.f, 0.f))) * std::sqrt(o.de2 * o.de2 + p.z * p.z);
        case R_E3: return gsign(dot(d.edgesNormal[2], p)) * std::sqrt(o.de3 * o.de3 + p.z * p.z);
        default:   return p.z;
    }
}

// Variant used by TriCubicInterpolation::calculatePointValues (needs the world-space vertices; a NaN
// direction falls back to the triangle normal, TriangleUtils.h:208-212).
static inline float signedDistPointTriangleGrad(V3 point, const TriangleData& d, V3 w1, V3 w2, V3 w3, V3& outN) {
    const Proj o = classify(point, d);
    const V3 p = o.p;
    auto nrm = [&d](V3 v) { V3 n = normalize(v); return std::isnan(n.x + n.y + n.z) ? d.normal() : n; };
    switch (o.r) {
        case R_V1: { float s = gsign(dot(d.verticesNormal[0], p)); outN = s * nrm(point - w1); return s * std::sqrt(dot(p, p)); }
        case R_V2: { V3 q = p - v3(d.v2, 0.f, 0.f); float s = gsign(dot(d.verticesNormal[1], q)); outN = s * nrm(point - w2); return s * std::sqrt(dot(q, q)); }
        case R_V3: { V3 q = p - v3(d.v3.x, d.v3.y, 0.f); float s = gsign(dot(d.verticesNormal[2], q)); outN = s * nrm(point - w3); return s * std::sqrt(dot(q, q)); }
        case R_E1: {
            float s = gsign(dot(d.edgesNormal[0], p));
            outN = s * nrm(mulT(d.transform, v3(0.f, p.y, p.z)));
            return s * std::sqrt(o.de1 * o.de1 + p.z * p.z);
        }
        case R_E2: {
            float s = gsign(dot(d.edgesNormal[1], p - v3(d.v2, 0.f, 0.f)));
            float t = (p.x - d.v2) * d.b.x + p.y * d.b.y;
            outN = s * nrm(mulT(d.transform, v3((p.x - d.v2) - t * d.b.x, p.y - t * d.b.y, p.z)));
            return s * std::sqrt(o.de2 * o.de2 + p.z * p.z);
        }
        case R_E3: {
            float s = gsign(dot(d.edgesNormal[2], p));
            float t = p.x * d.c.x + p.y * d.c.y;
            outN = s * nrm(mulT(d.transform, v3(p.x - t * d.c.x, p.y - t * d.c.y, p.z)));
            return s * std::sqrt(o.de3 * o.de3 + p.z * p.z);
        }
        default: outN = d.normal(); return p.z;
    }
}

// Variant used by ExactOctreeSdf::getDistance(sample, outGradient) (no vertex arguments, no NaN guard).
static inline float signedDistPointTriangleGradLocal(V3 point, const TriangleData& d, V3& outN) {
    const Proj o = classify(point, d);
    const V3 p = o.p;
    switch (o.r) {
        case R_V1: { float s = gsign(dot(d.verticesNormal[0], p)); outN = s * normalize(point - d.origin); return s * std::sqrt(dot(p, p)); }
        case R_V2: {
            V3 q = p - v3(d.v2, 0.f, 0.f); float s = gsign(dot(d.verticesNormal[1], q));
            outN = s * normalize(point - d.origin - mulT(d.transform, v3(d.v2, 0.f, 0.f)));
            return s * std::sqrt(dot(q, q));
        }
        case R_V3: {
            V3 q = p - v3(d.v3.x, d.v3.y, 0.f); float s = gsign(dot(d.verticesNormal[2], q));
            outN = s * normalize(point - d.origin - mulT(d.transform, v3(d.v3.x, d.v3.y, 0.f)));
            return s * std::sqrt(dot(q, q));
        }
        case R_E1: {
            float s = gsign(dot(d.edgesNormal[0], p));
            outN = s * normalize(mulT(d.transform, v3(0.f, p.y, p.z)));
            return s * std::sqrt(o.de1 * o.de1 + p.z * p.z);
        }
        case R_E2: {
            float s = gsign(dot(d.edgesNormal[1], p - v3(d.v2, 0.f, 0.f)));
            float t = (p.x - d.v2) * d.b.x + p.y * d.b.y;
            outN = s * normalize(mulT(d.transform, v3((p.x - d.v2) - t * d.b.x, p.y - t * d.b.y, p.z)));
            return s * std::sqrt(o.de2 * o.de2 + p.z * p.z);
        }
        case R_E3: {
            float s = gsign(dot(d.edgesNormal[2], p));
            float t = p.x * d.c.x + p.y * d.c.y;
            outN = s * normalize(mulT(d.transform, v3(p.x - t * d.c.x, p.y - t * d.c.y, p.z)));
            return s * std::sqrt(o.de3 * o.de3 + p.z * p.z);
        }
        default: outN = d.normal(); return p.z;
    }
}

// Raw-vertex squared distance (TriangleUtils.h:383-404); used by the reference's TriangleDistanceTest KAT.
static inline float sqDistPointTriangleRaw(V3 p, V3 a, V3 b, V3 c) {
    V3 ba = b - a, pa = p - a, cb = c - b, pb = p - b, ac = a - c, pc = p - c;
    V3 n = cross(ba, ac);
    auto dot2 = [](V3 v) { return dot(v, v); };
    float s = gsign(dot(cross(ba, n), pa)) + gsign(dot(cross(cb, n), pb)) + gsign(dot(cross(ac, n), pc));
    if (s < 2.0f) {
        float d1 = dot2(ba * gclamp(dot(ba, pa) / dot2(ba), 0.0f, 1.0f) - pa);
        float d2 = dot2(cb * gclamp(dot(cb, pb) / dot2(cb), 0.0f, 1.0f) - pb);
        float d3 = dot2(ac * gclamp(dot(ac, pc) / dot2(ac), 0.0f, 1.0f) - pc);
        return gmin(gmin(d1, d2), d3);
    }
    return dot(n, pa) * dot(n, pa) / dot(n, n);
}

// calculateMeshTriangleData, live branches only (see file header).
struct MeshBox { V3 min, max; };

static inline std::vector<TriangleData> meshTriangleData(const V3* vertices, uint32_t numVertices,
                                                         const uint32_t* indices, uint32_t numTriangles, const MeshBox* meshBox = nullptr) {
    std::vector<TriangleData> tris(numTriangles);
    const uint32_t numIndices = 3 * numTriangles;
    for (uint32_t i = 0, t = 0; i < numIndices; i += 3, t++)
        tris[t] = makeTriangleData(vertices[indices[i]], vertices[indices[i + 1]], vertices[indices[i + 2]]);

    std::map<std::pair<uint32_t, uint32_t>, uint32_t> openEdges;   // (vmin,vmax) -> 3*t+k of the first owner
    std::vector<V3> vertexNormal(numVertices, v3(0.f));
    for (uint32_t i = 0, t = 0; i < numIndices; i += 3, t++) {
        for (uint32_t k = 0; k < 3; k++) {
            const uint32_t a = indices[i + k], b = indices[i + ((k + 1) % 3)], c = indices[i + ((k + 2) % 3)];
            auto ins = openEdges.insert(std::make_pair(std::make_pair(gmin(a, b), gmax(a, b)), i + k));
            if (!ins.second) {
                const uint32_t t2 = ins.first->second / 3;
                V3 en = tris[t].normal() + tris[t2].normal();
                tris[t].edgesNormal[k] = mul(tris[t].transform, en);
                tris[t2].edgesNormal[ins.first->second % 3] = mul(tris[t2].transform, en);
                openEdges.erase(ins.first);
            }
            const float cosang = gclamp(dot(normalize(vertices[b] - vertices[a]), normalize(vertices[c] - vertices[a])), -1.0f, 1.0f);
            const float angle = std::acos(cosang);
            vertexNormal[a] += angle * tris[t].normal();
        }
    }
    if (!openEdges.empty() && meshBox) {
        // seam welding: vertices of single-owner edges that coincide (within 1e-5 / size) are merged, their edges re-paired
        std::map<uint32_t, uint32_t> vmap;
        auto parentOf = [&](uint32_t v) { auto it = vmap.find(v); while (it != vmap.end() && it->second != v) { v = it->second; it = vmap.find(v); } return v; };
        std::vector<uint32_t> nm(2 * openEdges.size());
        uint32_t fill = 0;
        for (auto& e : openEdges) { nm[fill++] = e.first.first; nm[fill++] = e.first.second; }
        std::sort(nm.begin(), nm.end()); nm.erase(std::unique(nm.begin(), nm.end()), nm.end());
        const V3 bb = meshBox->max - meshBox->min; const V3 start = meshBox->min;
        const uint32_t axisRes = 2048;
        const float big = gmax(bb.x, gmax(bb.y, bb.z));
        const float gridScale = static_cast<float>(axisRes) / big;
        const float threshold = 1e-5 / big;          // double quotient, rounded to float by the declaration (as in the reference)
        const float sqThr = threshold * threshold;
        std::map<uint64_t, std::vector<uint32_t>> set1, set2;
        // int + int * uint32: evaluated in uint32 and widened to the 64-bit key, exactly as the reference's expression is
        auto cellId = [axisRes](IV3 id) { return id.x + id.y * axisRes + id.z * axisRes * axisRes; };
        auto bucket = [](std::map<uint64_t, std::vector<uint32_t>>& m, uint64_t key) -> std::vector<uint32_t>& { return m.insert(std::make_pair(key, std::vector<uint32_t>())).first->second; };
        for (uint32_t i = 0; i < nm.size(); i++) {
            const V3 p = vertices[nm[i]];
            bucket(set1, cellId(iv3((p - start) * gridScale))).push_back(nm[i]);
            bucket(set2, cellId(iv3((p - start) * gridScale + 0.5f))).push_back(nm[i]);
        }
        std::map<uint64_t, std::vector<uint32_t>>* sets[2] = {&set1, &set2};
        for (uint32_t i = 0; i < nm.size(); i++) {
            float offset = 0.0f;
            for (std::map<uint64_t, std::vector<uint32_t>>* ps : sets) {
                const V3 p = vertices[nm[i]];
                auto it = ps->find(cellId(iv3((p - start) * gridScale + offset)));
                if (it != ps->end()) {
                    for (uint32_t other : it->second) {
                        const V3 d = p - vertices[other];
                        if (dot(d, d) < sqThr) {
                            const uint32_t p1 = parentOf(nm[i]), p2 = parentOf(other);
                            if (nm[i] == p1) vmap[p1] = p1;
                            vmap[p2] = p1;
                            break;
                        }
                    }
                }
                offset += 0.5f;
            }
        }
        std::map<std::pair<uint32_t, uint32_t>, uint32_t> repaired;
        for (auto it = openEdges.begin(); it != openEdges.end(); it++) {
            const uint32_t a = parentOf(it->first.first), b = parentOf(it->first.second);
            auto ins = repaired.insert(std::make_pair(std::make_pair(gmin(a, b), gmax(a, b)), it->second));
            if (!ins.second) {
                const uint32_t t = it->second / 3, t2 = ins.first->second / 3;
                V3 en = tris[t].normal() + tris[t2].normal();
                tris[t].edgesNormal[it->second % 3] = mul(tris[t].transform, en);
                tris[t2].edgesNormal[ins.first->second % 3] = mul(tris[t2].transform, en);
                repaired.erase(ins.first);
            }
        }
        for (uint32_t v : nm) { const uint32_t p = parentOf(v); if (p != v) vertexNormal[p] += vertexNormal[v]; }
        for (uint32_t v : nm) { const uint32_t p = parentOf(v); vertexNormal[v] = vertexNormal[p]; }
    }
    for (uint32_t i = 0; i < numIndices; i++)
        tris[i / 3].verticesNormal[i % 3] = mul(tris[i / 3].transform, vertexNormal[indices[i]]);
    return tris;
}

}  // namespace orc
