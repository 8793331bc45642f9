// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_math.h header).
//
// Restates the OctreeSdf NO_CONTINUITY construction and the OctreeSdf query (reference file:line):
//   OctreeSdf::buildOctree (cube-ify box, start grid)        src/sdf/OctreeSdf.cpp:37-86
//   OctreeSdf::initOctree<VHQueries<TriCubic>> + processNode src/sdf/OctreeSdfDepthFirst.h:32-558
//   VHQueries::calculateVerticesInfo (+ 32^3 vertex cache)   include/SdfLib/TrianglesInfluence.h:926-1012
//   TriCubicInterpolation::calculatePointValues              include/SdfLib/InterpolationMethods.h:273-290
//   OctreeSdf::getDistance (value / value+gradient)          src/sdf/OctreeSdf.cpp:93-152
//   BoundingBox::getDistance (both overloads, as written)    include/SdfLib/utils/Mesh.h:42-63
//   OctreeSdf::computeMinBorderValue                         src/sdf/OctreeSdf.cpp:155-230
//
// Two modes (SURVEY.md section 7 "hard part 0"):
//   vertexCache = true  + LAYOUT_GLOBAL_DFS : emulates the reference run with numThreads = 1 (sequential DFS,
//                         lattice-point cache hit/miss sequence reproduced).
//   vertexCache = false : "canonical" mode = reference semantics with the cache disabled, which is
//                         traversal-order independent; this is what the HIP path is specified against.
// Layouts: LAYOUT_GLOBAL_DFS = array produced by the numThreads<2 branch (OctreeSdfDepthFirst.h:394-416);
//          LAYOUT_SUBTREES   = array produced by the numThreads>=2 branch ([grid][cell 0 body][cell 1 body]...,
//                              OctreeSdfDepthFirst.h:417-503).
#pragma once
#include "orc_math.h"
#include "orc_triangle.h"
#include "orc_bvh.h"
#include "orc_tricubic.h"
#include <vector>
#include <array>
#include <cstring>
#include <limits>
#include <functional>

namespace orc {

enum { LAYOUT_GLOBAL_DFS = 0, LAYOUT_SUBTREES = 1 };

static const uint32_t LEAF_BIT = 1u << 31, MARK_BIT = 1u << 30, INDEX_MASK = ~(LEAF_BIT | MARK_BIT);
static inline uint32_t nodeWord(bool leaf, uint32_t index) { return (index & INDEX_MASK) | (leaf ? LEAF_BIT : 0u); }

struct Box { V3 min, max; V3 size() const { return max - min; } V3 center() const { return min + 0.5f * size(); } };

static inline float boxDistance(const Box& b, V3 p) {
    V3 q = gabs(p - b.center()) - 0.5f * b.size();
    return length(gmax(q, v3(0.f))) + gmin(gmax(q.x, gmax(q.y, q.z)), 0.0f);
}
// Reproduced as written in the reference, including its use of the full size and the uncentred point.
static inline float boxDistanceGrad(const Box& b, V3 p, V3& g) {
    V3 a = gabs(p) - b.size();
    int k = a[0] > a[1] ? 0 : 1;
    int l = a[2] > a[k] ? 2 : k;
    if (a[l] < 0) g[l] = p[l] / std::fabs(p[l]);
    else {
        V3 bb = gmax(a, v3(0.f));
        float c = length(bb);
        g[0] = a[0] > 0 ? bb[0] / c * p[0] / std::fabs(p[0]) : 0;
        g[1] = a[1] > 0 ? bb[1] / c * p[1] / std::fabs(p[1]) : 0;
        g[2] = a[2] > 0 ? bb[2] / c * p[2] / std::fabs(p[2]) : 0;
    }
    return boxDistance(b, p);
}

struct MeshView { const V3* vertices; uint32_t numVertices; const uint32_t* indices; uint32_t numTriangles; };

static inline void pointValues(V3 p, uint32_t t, const MeshView& mesh, const std::vector<TriangleData>& td, float out[8]) {
    V3 g;
    out[0] = signedDistPointTriangleGrad(p, td[t], mesh.vertices[mesh.indices[3 * t]], mesh.vertices[mesh.indices[3 * t + 1]],
                                         mesh.vertices[mesh.indices[3 * t + 2]], g);
    out[1] = g.x; out[2] = g.y; out[3] = g.z; out[4] = 0.f; out[5] = 0.f; out[6] = 0.f; out[7] = 0.f;
}

// VHQueries: nearest triangle through the fp64 BVH, with the optional direct-mapped lattice-point cache.
struct VHQueries {
    const SphereBvh* bvh = nullptr;
    bool useCache = false;
    struct Entry { uint32_t x, y, z, info; };
    std::vector<Entry> cache;
    V3 coordToId, minPoint;
    uint64_t numQueries = 0;
    void init(const SphereBvh* b, const Box& box, uint32_t maxDepth, bool cacheOn) {
        bvh = b; useCache = cacheOn;
        if (cacheOn) {
            const uint32_t inval = (1u << maxDepth) + 1u;
            cache.assign(32 * 32 * 32, Entry{inval, inval, inval, 0});
        }
        const float s = (float)(1 << maxDepth);
        const V3 sz = box.size();
        coordToId = V3{s / sz.x, s / sz.y, s / sz.z};
        minPoint = box.min;
    }
    uint32_t nearest(V3 p) {
        if (!useCache) { numQueries++; return bvh->nearestTriangle(p); }
        const V3 q = (p - minPoint) * coordToId;
        const uint32_t ix = (uint32_t)std::round(q.x), iy = (uint32_t)std::round(q.y), iz = (uint32_t)std::round(q.z);
        const uint32_t slot = ((iz & 31u) << 10) | ((iy & 31u) << 5) | (ix & 31u);
        Entry& e = cache[slot];
        if (e.x == ix && e.y == iy && e.z == iz) return e.info;
        numQueries++;
        const uint32_t t = bvh->nearestTriangle(p);
        e = Entry{ix, iy, iz, t};
        return t;
    }
    template <int N>
    void verticesInfo(V3 center, float half, const V3* rel, float outValues[][8], uint32_t* outInfo,
                      const MeshView& mesh, const std::vector<TriangleData>& td) {
        for (int i = 0; i < N; i++) {
            const V3 p = center + rel[i] * half;
            outInfo[i] = nearest(p);
            pointValues(p, outInfo[i], mesh, td, outValues[i]);
        }
    }
};

static const V3 CORNER_REL[8] = {
    {-1.f, -1.f, -1.f}, {1.f, -1.f, -1.f}, {-1.f, 1.f, -1.f}, {1.f, 1.f, -1.f},
    {-1.f, -1.f, 1.f},  {1.f, -1.f, 1.f},  {-1.f, 1.f, 1.f},  {1.f, 1.f, 1.f}};

struct OctreeSdfData {
    Box box;
    int startGridSize = 0, startGridXY = 0;
    float startGridCellSize = 0.f;
    uint32_t maxDepth = 0;
    float valueRange = 0.f, minBorderValue = 0.f;
    std::vector<uint32_t> data;     // OctreeNode words (u32 / float union)
    uint64_t numBvhQueries = 0;
};

struct BuildNode {
    uint32_t nodeIndex; uint32_t depth; V3 center; float size;
    float vv[8][8]; uint32_t vi[8];
};

static inline void computeMinBorder(OctreeSdfData& out);

struct OctreeBuilder {
    const MeshView mesh;
    const std::vector<TriangleData>& td;
    const SphereBvh& bvh;
    OctreeSdfData& out;
    uint32_t startDepth, maxDepth, startOctreeDepth;
    int rule; float sqThreshold, param1;

    OctreeBuilder(const MeshView& m, const std::vector<TriangleData>& t, const SphereBvh& b, OctreeSdfData& o)
        : mesh(m), td(t), bvh(b), out(o) {}

    // One DFS step (processNode, OctreeSdfDepthFirst.h:137-391). Children are pushed 0..7 (popped 7..0).
    void processNode(const BuildNode& node, VHQueries& q, std::vector<BuildNode>& stack, std::vector<uint32_t>& octree, float& valueRange) {
        const Stencil& st = stencil();
        float coeff[64];
        auto writeLeaf = [&]() {
            const uint32_t at = (uint32_t)octree.size();
            octree[node.nodeIndex] = nodeWord(true, at);
            tricubicFit(node.vv, 2.0f * node.size, coeff);
            octree.resize(octree.size() + 64);
            std::memcpy(&octree[at], coeff, 64 * sizeof(float));
            for (int i = 0; i < 8; i++) valueRange = gmax(valueRange, std::fabs(node.vv[i][0]));
        };
        if (node.depth < maxDepth) {
            float mid[19][8]; uint32_t midInfo[19];
            q.verticesInfo<19>(node.center, node.size, st.midRel, mid, midInfo, mesh, td);
            bool terminal = false;
            if (node.depth >= startDepth) {
                tricubicFit(node.vv, 2.0f * node.size, coeff);
                terminal = ruleValue(rule, coeff, mid, param1) < sqThreshold;
            }
            if (!terminal) {
                const float ns = 0.5f * node.size;
                const bool alloc = node.depth >= startDepth;
                const uint32_t childIndex = alloc ? (uint32_t)octree.size() : std::numeric_limits<uint32_t>::max();
                if (node.nodeIndex != std::numeric_limits<uint32_t>::max()) octree[node.nodeIndex] = nodeWord(false, childIndex);
                if (alloc) octree.resize(octree.size() + 8);
                for (int c = 0; c < 8; c++) {
                    BuildNode ch;
                    ch.nodeIndex = alloc ? childIndex + c : childIndex;
                    ch.depth = node.depth + 1;
                    ch.center = node.center + V3{(c & 1) ? ns : -ns, (c & 2) ? ns : -ns, (c & 4) ? ns : -ns};
                    ch.size = ns;
                    for (int j = 0; j < 8; j++) {
                        const int src = st.childSrc[c][j];
                        if (src >= 0) { std::memcpy(ch.vv[j], mid[src], 8 * sizeof(float)); ch.vi[j] = midInfo[src]; }
                        else { std::memcpy(ch.vv[j], node.vv[-src - 1], 8 * sizeof(float)); ch.vi[j] = node.vi[-src - 1]; }
                    }
                    stack.push_back(ch);
                }
            } else writeLeaf();
        } else writeLeaf();
    }

    void run(const Box& inBox, uint32_t depth, uint32_t startDepth_, int rule_, float p0, float p1, bool vertexCache, int layout) {
        startDepth = startDepth_; maxDepth = depth; rule = rule_; sqThreshold = p0 * p0; param1 = p1;
        out.maxDepth = depth;
        const V3 bs = inBox.size();
        const float maxSize = gmax(gmax(bs.x, bs.y), bs.z);
        out.box.min = inBox.center() - 0.5f * maxSize;
        out.box.max = inBox.center() + 0.5f * maxSize;
        out.startGridSize = 1 << startDepth;
        out.startGridXY = out.startGridSize * out.startGridSize;
        out.startGridCellSize = maxSize / (float)out.startGridSize;
        startOctreeDepth = startDepth < 1u ? startDepth : 1u;

        VHQueries mainQ; mainQ.init(&bvh, out.box, maxDepth, vertexCache);
        std::vector<BuildNode> stack;
        {
            const float newSize = (float)(0.5f * out.box.size().x * std::pow(0.5f, startOctreeDepth));
            const V3 startCenter = out.box.min + newSize;
            const uint32_t vpa = 1u << startOctreeDepth;
            for (uint32_t k = 0; k < vpa; k++) for (uint32_t j = 0; j < vpa; j++) for (uint32_t i = 0; i < vpa; i++) {
                BuildNode n;
                n.nodeIndex = std::numeric_limits<uint32_t>::max(); n.depth = startOctreeDepth;
                n.center = startCenter + V3{(float)i, (float)j, (float)k} * 2.0f * newSize;
                n.size = newSize;
                mainQ.verticesInfo<8>(n.center, n.size, CORNER_REL, n.vv, n.vi, mesh, td);
                stack.push_back(n);
            }
        }
        const uint32_t G = (uint32_t)out.startGridSize, G3 = G * G * G;
        auto gridIndex = [&](const BuildNode& n) {
            V3 f = (n.center - out.box.min) / out.startGridCellSize;
            int x = (int)std::floor(f.x), y = (int)std::floor(f.y), z = (int)std::floor(f.z);
            return (uint32_t)(z * out.startGridXY + y * out.startGridSize + x);
        };
        float valueRange = 0.f;
        if (layout == LAYOUT_GLOBAL_DFS) {
            out.data.assign(G3, 0u);
            while (!stack.empty()) {
                BuildNode n = stack.back(); stack.pop_back();
                if (n.depth == startDepth) n.nodeIndex = gridIndex(n);
                processNode(n, mainQ, stack, out.data, valueRange);
            }
            out.numBvhQueries = mainQ.numQueries;
        } else {
            // upper levels on the "main thread", then one independent sub-octree per start-grid cell
            std::vector<BuildNode> cellRoots(G3);
            std::vector<uint32_t> dummy;
            while (!stack.empty()) {
                BuildNode n = stack.back(); stack.pop_back();
                if (n.depth == startDepth) cellRoots[gridIndex(n)] = n;
                else processNode(n, mainQ, stack, dummy, valueRange);
            }
            std::vector<std::vector<uint32_t>> sub(G3);
            std::vector<float> ranges(G3, 0.f);
            std::vector<uint64_t> nq(G3, 0);
            #pragma omp parallel for schedule(dynamic, 1)
            for (int64_t ci = 0; ci < (int64_t)G3; ci++) {
                VHQueries q; q.init(&bvh, out.box, maxDepth, false);   // per-cell state; cache never used in this layout
                std::vector<BuildNode> st2;
                BuildNode root = cellRoots[ci]; root.nodeIndex = 0;
                sub[ci].assign(1, 0u);
                st2.push_back(root);
                while (!st2.empty()) {
                    BuildNode n = st2.back(); st2.pop_back();
                    processNode(n, q, st2, sub[ci], ranges[ci]);
                }
                nq[ci] = q.numQueries;
            }
            out.data.assign(G3, 0u);
            out.numBvhQueries = mainQ.numQueries;
            for (uint32_t ci = 0; ci < G3; ci++) {
                std::vector<uint32_t>& s = sub[ci];
                const uint32_t startIndex = (uint32_t)out.data.size();
                // rebase every node word of the sub-octree by startIndex - 1 (the root moves to grid slot ci)
                std::function<void(uint32_t)> visit = [&](uint32_t at) {
                    const uint32_t w = s[at];
                    const bool leaf = (w & LEAF_BIT) != 0;
                    if (!leaf) for (uint32_t c = 0; c < 8; c++) visit((w & INDEX_MASK) + c);
                    s[at] = nodeWord(leaf, (w & INDEX_MASK) + startIndex - 1);
                };
                visit(0);
                out.data[ci] = s[0];
                out.data.insert(out.data.end(), s.begin() + 1, s.end());
                valueRange = gmax(valueRange, ranges[ci]);
                out.numBvhQueries += nq[ci];
            }
        }
        out.valueRange = valueRange;
        computeMinBorderValue();
    }

    void computeMinBorderValue() { computeMinBorder(out); }
};

static inline void computeMinBorder(OctreeSdfData& out) {
    {
        const std::vector<uint32_t>& d = out.data;
        std::function<float(uint32_t, V3, float)> rec = [&](uint32_t at, V3 pos, float half) -> float {
            float mn = INFINITY;
            if (!(d[at] & LEAF_BIT)) {
                for (uint32_t i = 0; i < 8; i++) {
                    const V3 cp = pos + 0.5f * half * CORNER_REL[i];
                    if (cp.x < half || cp.y < half || cp.z < half || cp.x > (1.0f - half) || cp.y > (1.0f - half) || cp.z > (1.0f - half))
                        mn = gmin(mn, rec((d[at] & INDEX_MASK) + i, cp, 0.5f * half));
                }
            } else {
                for (uint32_t i = 0; i < 8; i++) {
                    const V3 sp = pos + half * CORNER_REL[i];
                    if (sp.x < 1e-4 || sp.y < 1e-4 || sp.z < 1e-4 || sp.x > (1.0f - 1e-4) || sp.y > (1.0f - 1e-4) || sp.z > (1.0f - 1e-4)) {
                        const float* c = reinterpret_cast<const float*>(&d[d[at] & INDEX_MASK]);
                        mn = gmin(mn, tricubicValue(c, 0.5f * CORNER_REL[i] + v3(0.5f)));
                    }
                }
            }
            return mn;
        };
        const float cell = 1.0f / (float)out.startGridSize;
        float mn = INFINITY;
        for (int k = 0; k < out.startGridSize; k++) for (int j = 0; j < out.startGridSize; j++) for (int i = 0; i < out.startGridSize; i++) {
            const uint32_t idx = k * out.startGridXY + j * out.startGridSize + i;
            const V3 pos = V3{((float)i + 0.5f) * cell, ((float)j + 0.5f) * cell, ((float)k + 0.5f) * cell};
            mn = gmin(mn, rec(idx, pos, 0.5f * cell));
        }
        out.minBorderValue = mn;
    }
}

static inline uint32_t roundFloatGE(float a) { return (a >= 0.5f) ? 1 : 0; }

// OctreeSdf::getDistance; returns also the leaf's coefficient offset through outLeaf (for tests).
static inline float octreeDistance(const OctreeSdfData& o, V3 p, V3* grad = nullptr) {
    V3 f = (p - o.box.min) / o.startGridCellSize;
    const int ix = (int)std::floor(f.x), iy = (int)std::floor(f.y), iz = (int)std::floor(f.z);
    f = gfract(f);
    if (ix < 0 || ix >= o.startGridSize || iy < 0 || iy >= o.startGridSize || iz < 0 || iz >= o.startGridSize) {
        if (grad) return boxDistanceGrad(o.box, p, *grad) + o.minBorderValue;
        return boxDistance(o.box, p) + o.minBorderValue;
    }
    uint32_t w = o.data[iz * o.startGridXY + iy * o.startGridSize + ix];
    while (!(w & LEAF_BIT)) {
        const uint32_t child = (roundFloatGE(f.z) << 2) + (roundFloatGE(f.y) << 1) + roundFloatGE(f.x);
        w = o.data[(w & INDEX_MASK) + child];
        f = gfract(2.0f * f);
    }
    const float* c = reinterpret_cast<const float*>(&o.data[w & INDEX_MASK]);
    if (grad) *grad = normalize(tricubicGradient(c, f));
    return tricubicValue(c, f);
}

}  // namespace orc
