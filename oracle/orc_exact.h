// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_math.h header).
//
// Restates the ExactOctreeSdf construction and query (reference file:line):
//   GJK::IsNearMinimize (8-sphere-hull variant) + support functions   src/utils/GJK.cpp:830-866, 715-738, 644-652
//   PerNodeRegionTrianglesInfluence::calculateVerticesInfo            include/SdfLib/TrianglesInfluence.h:692-765
//   PerNodeRegionTrianglesInfluence::filterTriangles                  include/SdfLib/TrianglesInfluence.h:767-860
//   ExactOctreeSdf::ExactOctreeSdf                                    src/sdf/ExactOctreeSdf.cpp:7-31
//   ExactOctreeSdf::initOctree + processNode (numThreads < 2 branch)  include/SdfLib/ExactOctreeSdfDepthFirst.h:28-510
//   ExactOctreeSdf::getDistance (value / value+gradient)              src/sdf/ExactOctreeSdf.cpp:38-178, 180-320
// The reference walks an explicit stack and visits nodes of the last two levels twice; the same order is
// obtained here by recursion (children 7..0, then the parent's merge step).  The multi-thread branch of the
// reference is not restated: it is incorrect for startDepth >= 2 (ExactOctreeSdfDepthFirst.h:613 passes
// startOctreeDepth as the subtree root depth) — parity for this structure is defined single-threaded.
#pragma once
#include "orc_octree.h"
#include <memory>

namespace orc {

static inline V3 furthestOnHull(float half, const float radius[8], V3 dir) {
    float best = dot(v3(-half), dir) + radius[0];
    int bi = 0;
    for (int i = 1; i < 8; i++) {
        const float v = dot(CORNER_REL[i] * half, dir) + radius[i];
        if (v > best) { best = v; bi = i; }
    }
    return CORNER_REL[bi] * half + radius[bi] * dir;
}
static inline V3 furthestOnTriangle(const V3 tri[3], V3 dir) {
    const float d1 = dot(tri[0], dir), d2 = dot(tri[1], dir), d3 = dot(tri[2], dir);
    if (d1 > d2) return (d1 > d3) ? tri[0] : tri[2];
    return (d2 > d3) ? tri[1] : tri[2];
}

// Frank-Wolfe on (hull of the 8 corner spheres) (-) triangle; at most 15 iterations.
static inline bool isNearMinimize(float half, const float radius[8], const V3 tri[3], float thr, uint32_t* iters = nullptr) {
    const uint32_t MAX_ITER = 15;
    uint32_t iter = 0;
    float distToP, distToO;
    bool isNear = false;
    const float sqThr = thr * thr;
    V3 cur = -tri[0];
    do {
        const V3 g = normalize(-cur);
        const V3 p = furthestOnHull(half, radius, g) - furthestOnTriangle(tri, -g);
        distToP = dot(g, p - cur);
        distToO = dot(g, -cur);
        const V3 dir = p - cur;
        const float d = dot(dir, -cur);
        if (d < 1.0e-5) { if (iters) *iters = iter; return distToO <= distToP + thr; }
        cur += dir * gmin(d / dot(dir, dir), 1.0f);
        isNear = dot(cur, cur) < sqThr;
    } while (!isNear && distToO <= distToP + thr && ++iter < MAX_ITER);
    if (iters) *iters = iter;
    return isNear || iter >= MAX_ITER;
}

struct ExactOctreeData {
    Box box;
    int startGridSize = 0, startGridXY = 0; uint32_t startDepth = 0; float startGridCellSize = 0.f;
    uint32_t maxDepth = 0, bitEncodingStartDepth = 0, bitsPerIndex = 0;
    uint32_t minTrianglesInLeafs = 0, maxTrianglesInLeafs = 0, maxTrianglesEncodedInLeafs = 0;
    std::vector<uint32_t> nodes;        // 2 words per node: childrenIndex (bit31 = leaf), trianglesArrayIndex
    std::vector<uint8_t> nodeHasTriIdx; // oracle-only: 1 where trianglesArrayIndex was written (the reference leaves the rest uninitialised)
    std::vector<uint32_t> sets;
    std::vector<uint8_t> masks;
    std::vector<TriangleData> triangles;
};

struct ExactBuilder {
    const MeshView mesh;
    ExactOctreeData& out;
    uint32_t startDepth, maxDepth, minTriangles;
    bool useCache;
    struct Entry { uint32_t x, y, z, info; };
    std::vector<Entry> cache;
    V3 coordToId, minPoint;
    uint64_t cullTests = 0;

    const std::vector<TriangleData>* meshTd = nullptr;      // the mesh's TriangleData (calculateMeshTriangleData(mesh): welded when the mesh carries its box)
    const std::vector<TriangleData>* tris = nullptr;        // what the searches read: out.triangles, or the main builder's for a per-subtree builder
    ExactBuilder(const MeshView& m, ExactOctreeData& o, const std::vector<TriangleData>* td = nullptr) : mesh(m), out(o), meshTd(td) {}

    struct Node { uint32_t nodeIndex; uint32_t depth; V3 center; float size; uint32_t vi[8]; };
    // Parallel canonical build (test speed-up only, see runParallel): start cells collected in emission order instead of being processed
    struct Task { Node node; std::shared_ptr<const std::vector<uint32_t>> parentList; };
    std::vector<Task>* collect = nullptr;

    uint32_t bruteNearest(V3 p, const std::vector<uint32_t>& list) {
        if (useCache) {
            const V3 q = (p - minPoint) * coordToId;
            const uint32_t ix = (uint32_t)std::round(q.x), iy = (uint32_t)std::round(q.y), iz = (uint32_t)std::round(q.z);
            Entry& e = cache[((iz & 31u) << 10) | ((iy & 31u) << 5) | (ix & 31u)];
            if (e.x == ix && e.y == iy && e.z == iz) return e.info;
            uint32_t best = 0; float bd = INFINITY;
            for (uint32_t t : list) { const float d = sqDistPointTriangle(p, (*tris)[t]); if (d < bd) { best = t; bd = d; } }
            e = Entry{ix, iy, iz, best};
            return best;
        }
        uint32_t best = 0; float bd = INFINITY;
        for (uint32_t t : list) { const float d = sqDistPointTriangle(p, (*tris)[t]); if (d < bd) { best = t; bd = d; } }
        return best;
    }

    void filterTriangles(const Node& n, const std::vector<uint32_t>& in, std::vector<uint32_t>& outList) {
        outList.clear();
        float region[8][8], minDist[8];
        for (int i = 0; i < 8; i++) {
            minDist[i] = INFINITY;
            for (int c = 0; c < 8; c++) {
                region[i][c] = std::sqrt(sqDistPointTriangle(n.center + CORNER_REL[c] * n.size, (*tris)[n.vi[i]]));
                minDist[i] = gmin(minDist[i], region[i][c]);
            }
            for (int c = 0; c < 8; c++) region[i][c] -= minDist[i];
        }
        for (uint32_t idx : in) {
            V3 tri[3];
            for (int k = 0; k < 3; k++) tri[k] = mesh.vertices[mesh.indices[3 * idx + k]] - n.center;
            const V3 pt = 0.3333333f * (tri[0] + tri[1] + tri[2]);
            const uint32_t vId = ((pt.z > 0) ? 4 : 0) + ((pt.y > 0) ? 2 : 0) + ((pt.x > 0) ? 1 : 0);
            bool keep = true;
            if (n.vi[vId] != idx) cullTests++;
            if (n.vi[vId] != idx && !isNearMinimize(n.size, region[vId], tri, minDist[vId])) keep = false;
            if (keep) outList.push_back(idx);
        }
    }

    void emitSet(uint32_t nodeIndex, const std::vector<uint32_t>& list) {
        uint32_t at = (uint32_t)out.sets.size();
        const uint32_t n = (uint32_t)list.size();
        const uint32_t words = (n * out.bitsPerIndex + 31) / 32;
        out.sets.resize(out.sets.size() + words + 2, 0u);
        out.nodes[2 * nodeIndex + 1] = at; out.nodeHasTriIdx[nodeIndex] = 1;
        const uint32_t inv = 32 - out.bitsPerIndex;
        out.sets[at++] = n;
        uint32_t bIdx = 0;
        for (uint32_t t = 0; t < n; t++, bIdx += out.bitsPerIndex) {
            const uint32_t index = list[t], w = bIdx >> 5, bit = bIdx & 31;
            out.sets[at + w] |= (index << inv) >> bit;
            out.sets[at + w + 1] |= (static_cast<uint64_t>(index) << (64 - (bit + out.bitsPerIndex)));      // (the 64-bit value is narrowed by the compound assignment, as in the reference)
        }
    }

    void growNodes(size_t n) { out.nodes.resize(2 * n, 0u); out.nodeHasTriIdx.resize(n, 0); }
    size_t numNodes() const { return out.nodes.size() / 2; }

    // Returns through nodeList the node's final triangle list (after the merge step where it applies).
    void process(const Node& n, const std::vector<uint32_t>& parentList, std::vector<uint32_t>& nodeList) {
        const Stencil& st = stencil();
        filterTriangles(n, parentList, nodeList);
        const bool hasNode = n.nodeIndex != std::numeric_limits<uint32_t>::max();
        bool terminal = false;
        if (n.depth >= startDepth) terminal = nodeList.size() <= minTriangles;
        if (!terminal && n.depth < maxDepth) {
            uint32_t midInfo[19];
            for (int m = 0; m < 19; m++) midInfo[m] = bruteNearest(n.center + st.midRel[m] * n.size, nodeList);
            const float ns = 0.5f * n.size;
            const bool alloc = n.depth >= startDepth;
            const uint32_t childIndex = alloc ? (uint32_t)numNodes() : std::numeric_limits<uint32_t>::max();
            if (hasNode) out.nodes[2 * n.nodeIndex] = childIndex & 0x7FFFFFFFu;
            if (alloc) growNodes(numNodes() + 8);
            Node ch[8];
            for (int c = 0; c < 8; c++) {
                ch[c].nodeIndex = alloc ? childIndex + c : childIndex;
                ch[c].depth = n.depth + 1;
                ch[c].center = n.center + V3{(c & 1) ? ns : -ns, (c & 2) ? ns : -ns, (c & 4) ? ns : -ns};
                ch[c].size = ns;
                for (int j = 0; j < 8; j++) { const int s = st.childSrc[c][j]; ch[c].vi[j] = s >= 0 ? midInfo[s] : n.vi[-s - 1]; }
            }
            std::array<std::vector<uint32_t>, 8> chLists;
            std::shared_ptr<const std::vector<uint32_t>> shared;
            for (int c = 7; c >= 0; c--) {
                Node cn = ch[c];
                if (cn.depth == startDepth) {
                    V3 f = (cn.center - out.box.min) / out.startGridCellSize;
                    cn.nodeIndex = (uint32_t)((int)std::floor(f.z) * out.startGridXY + (int)std::floor(f.y) * out.startGridSize + (int)std::floor(f.x));
                    if (collect) {      // parallel build: the subtree is processed later by its own builder (nothing above the start depth emits)
                        if (!shared) shared = std::make_shared<const std::vector<uint32_t>>(nodeList);
                        collect->push_back(Task{cn, shared});
                        continue;
                    }
                }
                process(cn, nodeList, chLists[c]);
            }
            if (n.depth >= out.bitEncodingStartDepth) {
                // merge step: node list := sorted union of the children's lists; one MSB-first byte mask per child
                const size_t oldSize = nodeList.size();
                std::array<std::vector<uint8_t>, 8> maskBuf;
                for (int c = 0; c < 8; c++) maskBuf[c].assign((oldSize + 7) / 8, 0);
                nodeList.clear();
                uint32_t pos[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (;;) {
                    uint32_t mn = std::numeric_limits<uint32_t>::max(); bool any = false;
                    for (int c = 0; c < 8; c++) if (pos[c] < chLists[c].size()) { any = true; if (chLists[c][pos[c]] < mn) mn = chLists[c][pos[c]]; }
                    if (!any) break;
                    const uint32_t mIdx = (uint32_t)nodeList.size() / 8; const uint8_t mBit = (uint8_t)(1u << (7 - (nodeList.size() & 7)));
                    nodeList.push_back(mn);
                    for (int c = 0; c < 8; c++) if (pos[c] < chLists[c].size() && chLists[c][pos[c]] == mn) { pos[c]++; maskBuf[c][mIdx] |= mBit; }
                }
                const uint32_t childrenAt = out.nodes[2 * n.nodeIndex] & 0x7FFFFFFFu;
                const uint32_t nb = ((uint32_t)nodeList.size() + 7) / 8;
                for (int c = 0; c < 8; c++) {
                    const uint32_t at = (uint32_t)out.masks.size();
                    out.masks.resize(out.masks.size() + nb);
                    out.nodes[2 * (childrenAt + c) + 1] = at; out.nodeHasTriIdx[childrenAt + c] = 2;     // 2 = offset into masks (1 = into sets); exported as 1
                    std::memcpy(out.masks.data() + at, maskBuf[c].data(), nb);
                }
                if (n.depth == out.bitEncodingStartDepth) {
                    emitSet(n.nodeIndex, nodeList);
                    if (nodeList.size() > out.maxTrianglesEncodedInLeafs) out.maxTrianglesEncodedInLeafs = (uint32_t)nodeList.size();
                }
            }
        } else {
            out.nodes[2 * n.nodeIndex] = 0xFFFFFFFFu;     // setValues(true, max)
            if (n.depth <= out.bitEncodingStartDepth) emitSet(n.nodeIndex, nodeList);
            if (nodeList.size() > out.maxTrianglesInLeafs) out.maxTrianglesInLeafs = (uint32_t)nodeList.size();
        }
    }

    // threads > 1: canonical mode only (no lattice cache).  The reference's own multi-thread branch is NOT what runs here (it is
    // incorrect for startDepth >= 2, see the header): the start cells' subtrees are built by independent builders — nothing but
    // the three output arrays and the leaf statistics couples them — and concatenated in the single-thread emission order with
    // their indices rebased, which reproduces the sequential arrays bit for bit (tests/test_oracle_kats.py checks that).
    void run(const Box& inBox, uint32_t depth, uint32_t startDepth_, uint32_t minTri, bool vertexCache, int threads = 1) {
        startDepth = startDepth_; maxDepth = depth; minTriangles = minTri; useCache = vertexCache;
        tris = &out.triangles;
        std::vector<Task> tasks;
        const bool parallel = threads > 1 && !vertexCache && depth >= 2 && depth - 2 >= startDepth_;
        if (parallel) collect = &tasks;
        out.maxDepth = depth;
        const V3 bs = inBox.size();
        const float maxSize = gmax(gmax(bs.x, bs.y), bs.z);
        out.box.min = inBox.center() - 0.5f * maxSize;
        out.box.max = inBox.center() + 0.5f * maxSize;
        out.startGridSize = 1 << startDepth; out.startGridXY = out.startGridSize * out.startGridSize; out.startDepth = startDepth;
        out.startGridCellSize = maxSize / (float)out.startGridSize;
        // ExactOctreeSdf.cpp:25: calculateMeshTriangleData(mesh) — the same call OctreeSdf makes, seam welding included
        out.triangles = meshTd ? *meshTd : meshTriangleData(mesh.vertices, mesh.numVertices, mesh.indices, mesh.numTriangles);
        out.minTrianglesInLeafs = minTri;
        const uint32_t sod = startDepth < 1u ? startDepth : 1u;
        out.bitEncodingStartDepth = depth - 2;
        out.bitsPerIndex = (uint32_t)(int32_t)std::ceil(std::log2((float)out.triangles.size()));
        if (useCache) {
            const uint32_t inval = (1u << depth) + 1u;
            cache.assign(32 * 32 * 32, Entry{inval, inval, inval, 0});
        }
        const float s = (float)(1 << depth); const V3 sz = out.box.size();
        coordToId = V3{s / sz.x, s / sz.y, s / sz.z}; minPoint = out.box.min;

        std::vector<uint32_t> all;
        for (uint32_t t = 0; t < mesh.numTriangles; t++) { V3 nrm = out.triangles[t].normal(); if (dot(nrm, nrm) > 1e-3f) all.push_back(t); }

        const uint32_t G = (uint32_t)out.startGridSize;
        out.nodes.clear(); out.nodeHasTriIdx.clear(); growNodes((size_t)G * G * G);
        const float newSize = (float)(0.5f * out.box.size().x * std::pow(0.5f, sod));
        const V3 startCenter = out.box.min + newSize;
        const uint32_t vpa = 1u << sod;
        std::vector<Node> roots;
        for (uint32_t k = 0; k < vpa; k++) for (uint32_t j = 0; j < vpa; j++) for (uint32_t i = 0; i < vpa; i++) {
            Node n; n.nodeIndex = std::numeric_limits<uint32_t>::max(); n.depth = sod;
            n.center = startCenter + V3{(float)i, (float)j, (float)k} * 2.0f * newSize; n.size = newSize;
            for (int c = 0; c < 8; c++) n.vi[c] = bruteNearest(n.center + CORNER_REL[c] * n.size, all);
            roots.push_back(n);
        }
        std::shared_ptr<const std::vector<uint32_t>> sharedAll;
        for (int r = (int)roots.size() - 1; r >= 0; r--) {
            Node n = roots[r];
            if (n.depth == startDepth) {
                V3 f = (n.center - out.box.min) / out.startGridCellSize;
                n.nodeIndex = (uint32_t)((int)std::floor(f.z) * out.startGridXY + (int)std::floor(f.y) * out.startGridSize + (int)std::floor(f.x));
                if (collect) {
                    if (!sharedAll) sharedAll = std::make_shared<const std::vector<uint32_t>>(all);
                    collect->push_back(Task{n, sharedAll});
                    continue;
                }
            }
            std::vector<uint32_t> list;
            process(n, all, list);
        }
        if (parallel) { collect = nullptr; runTasks(tasks, threads); }
    }

    void runTasks(const std::vector<Task>& tasks, int threads) {
        const size_t nt = tasks.size();
        std::vector<ExactOctreeData> local(nt);
        std::vector<uint64_t> cull(nt, 0);
        #pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
        for (int64_t k = 0; k < (int64_t)nt; k++) {
            ExactOctreeData& L = local[k];
            L.box = out.box; L.startGridSize = out.startGridSize; L.startGridXY = out.startGridXY; L.startDepth = out.startDepth;
            L.startGridCellSize = out.startGridCellSize; L.maxDepth = out.maxDepth; L.bitEncodingStartDepth = out.bitEncodingStartDepth;
            L.bitsPerIndex = out.bitsPerIndex; L.minTrianglesInLeafs = out.minTrianglesInLeafs;
            ExactBuilder b(mesh, L);
            b.startDepth = startDepth; b.maxDepth = maxDepth; b.minTriangles = minTriangles; b.useCache = false;
            b.coordToId = coordToId; b.minPoint = minPoint; b.tris = tris;
            b.growNodes(1);                       // local node 0 = the start cell; its body follows from local index 1
            Node n = tasks[k].node; n.nodeIndex = 0;
            std::vector<uint32_t> list;
            b.process(n, *tasks[k].parentList, list);
            cull[k] = b.cullTests;
        }
        for (size_t k = 0; k < nt; k++) {
            const ExactOctreeData& L = local[k];
            const uint32_t bodyBase = (uint32_t)numNodes(), setBase = (uint32_t)out.sets.size(), maskBase = (uint32_t)out.masks.size();
            const size_t ln = L.nodes.size() / 2;
            growNodes(numNodes() + (ln - 1));
            for (size_t i = 0; i < ln; i++) {
                const uint32_t g = i == 0 ? tasks[k].node.nodeIndex : bodyBase + (uint32_t)(i - 1);
                uint32_t w0 = L.nodes[2 * i], w1 = L.nodes[2 * i + 1];
                if (w0 != 0xFFFFFFFFu) w0 = (w0 & 0x7FFFFFFFu) + (bodyBase - 1u);          // local child block >= 1 -> global
                const uint8_t kind = L.nodeHasTriIdx[i];
                if (kind == 1) w1 += setBase; else if (kind == 2) w1 += maskBase;
                out.nodes[2 * g] = w0; out.nodes[2 * g + 1] = w1; out.nodeHasTriIdx[g] = kind;
            }
            out.sets.insert(out.sets.end(), L.sets.begin(), L.sets.end());
            out.masks.insert(out.masks.end(), L.masks.begin(), L.masks.end());
            if (L.maxTrianglesInLeafs > out.maxTrianglesInLeafs) out.maxTrianglesInLeafs = L.maxTrianglesInLeafs;
            if (L.maxTrianglesEncodedInLeafs > out.maxTrianglesEncodedInLeafs) out.maxTrianglesEncodedInLeafs = L.maxTrianglesEncodedInLeafs;
            cullTests += cull[k];
        }
    }
};

static inline uint32_t roundFloatGT(float a) { return (a > 0.5f) ? 1u : 0u; }

static inline uint32_t unpackIndex(const uint32_t* set, uint32_t bIdx, uint32_t bits) {
    const uint32_t w = bIdx >> 5, bit = bIdx & 31;
    return ((set[w] << bit) >> (32 - bits)) | (uint32_t)((uint64_t)set[w + 1] >> (64 - (bit + bits)));
}

static inline float exactDistance(const ExactOctreeData& o, V3 p, V3* grad = nullptr, uint32_t* outTri = nullptr) {
    V3 f = (p - o.box.min) / o.startGridCellSize;
    const int ix = (int)std::floor(f.x), iy = (int)std::floor(f.y), iz = (int)std::floor(f.z);
    f = gfract(f);
    if (ix < 0 || ix >= o.startGridSize || iy < 0 || iy >= o.startGridSize || iz < 0 || iz >= o.startGridSize)
        return boxDistance(o.box, p) + std::sqrt(3.0f) * o.box.size().x;
    uint32_t node = (uint32_t)(iz * o.startGridXY + iy * o.startGridSize + ix);
    auto isLeaf = [&](uint32_t n) { return (o.nodes[2 * n] & LEAF_BIT) != 0; };
    auto childOf = [&](uint32_t n) {
        const uint32_t c = (roundFloatGT(f.z) << 2) + (roundFloatGT(f.y) << 1) + roundFloatGT(f.x);
        f = gfract(2.0f * f);
        return (o.nodes[2 * n] & 0x7FFFFFFFu) + c;
    };
    float minDist = INFINITY; uint32_t minIndex = 0;
    uint32_t depth = o.startDepth;
    while (!isLeaf(node) && depth < o.bitEncodingStartDepth) { node = childOf(node); depth++; }
    auto finish = [&]() {
        if (outTri) *outTri = minIndex;
        if (grad) return signedDistPointTriangleGradLocal(p, o.triangles[minIndex], *grad);
        return signedDistPointTriangle(p, o.triangles[minIndex]);
    };
    if (isLeaf(node)) {
        const uint32_t* set = &o.sets[o.nodes[2 * node + 1]];
        const uint32_t n = set[0];
        for (uint32_t t = 0, b = 0; t < n; t++, b += o.bitsPerIndex) {
            const uint32_t ti = unpackIndex(set + 1, b, o.bitsPerIndex);
            const float d = sqDistPointTriangle(p, o.triangles[ti]);
            if (d < minDist) { minIndex = ti; minDist = d; }
        }
        return finish();
    }
    const uint32_t* set = &o.sets[o.nodes[2 * node + 1]];
    node = childOf(node);
    std::vector<uint32_t> cur, nxt;
    {
        const uint32_t n = set[0];
        const uint8_t* mask = o.masks.data() + o.nodes[2 * node + 1];
        for (uint32_t t = 0; t < n; t++) if (mask[t >> 3] & (0x80u >> (t & 7))) cur.push_back(unpackIndex(set + 1, t * o.bitsPerIndex, o.bitsPerIndex));
    }
    while (!isLeaf(node)) {
        node = childOf(node);
        const uint8_t* mask = o.masks.data() + o.nodes[2 * node + 1];
        nxt.clear();
        for (uint32_t t = 0; t < cur.size(); t++) if (mask[t >> 3] & (0x80u >> (t & 7))) nxt.push_back(cur[t]);
        cur.swap(nxt);
    }
    for (uint32_t ti : cur) {
        const float d = sqDistPointTriangle(p, o.triangles[ti]);
        if (d < minDist) { minIndex = ti; minDist = d; }
    }
    return finish();
}

}  // namespace orc
