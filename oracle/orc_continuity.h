// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_math.h header).
//
// Restates OctreeSdf::initOctreeWithContinuityNoDelay<VHQueries<TriCubic>> — the CONTINUITY builder, SdfExporter's and
// the Unity plugin's default (reference src/sdf/OctreeSdfBreadthFirstNoDelay.h:84-1224, helpers
// src/sdf/OctreeSdfBreadthFirst.h:35-89) — in "canonical" mode (lattice-point cache disabled).
//
// Structure of the reference, kept here: per depth (a) Iter 1 over all live nodes: refresh the six outward neighbour
// words, fit, sample the 19 mid-points exactly, termination rule; (b) Iter 2 in node order: for non-terminal nodes find
// which of the 18 face/edge neighbours are LEAVES (coarser or equal), replace the shared mid-point samples by the node's
// own polynomial when it is within the threshold there (children then agree with the coarse neighbour) or else schedule
// that neighbour for subdivision, then append 8 children / 64 coefficients; (c) post-pass over the scheduled leaves: local
// breadth-first re-subdivision while a neighbour is subdivided-and-unmarked, recycling the leaf's old coefficient slot
// once.  Faithfully reproduced quirks of the reference: the node constructor takes the child path as uint8_t, so only the
// low 8 bits of `childIndices` survive (:21-22); leaves re-created by the post-pass are registered under the key
// parentChildrenIndex + childId even at the start depth (:1173-1175); mValueRange is never initialised there (0 here).
// The 24-entry neighbour-mask table (:139-176) is DERIVED from the stencil geometry (mid-points lying on the shared
// face/edge); tools/check_ref_expressions.py compares it with the reference's literals when the reference is present.
#pragma once
#include "orc_octree.h"
#include <map>

namespace orc {

// mask of mid-points (bit 18-i <-> mid-point i) on the face/edge in direction `dir` (axis bits) with sign code `sign`
// (bit k of sign = positive side on the k-th axis of dir, axes in x,y,z order)
static inline uint32_t neighbourMask(uint32_t dir, uint32_t sign) {
    const Stencil& st = stencil();
    int axes[3], na = 0;
    for (int a = 0; a < 3; a++) if (dir & (1u << a)) axes[na++] = a;
    if (sign >= (1u << na)) return 0;
    uint32_t m = 0;
    for (int i = 0; i < 19; i++) {
        const float rel[3] = {st.midRel[i].x, st.midRel[i].y, st.midRel[i].z};
        bool on = true;
        for (int k = 0; k < na; k++) on = on && rel[axes[k]] == (((sign >> k) & 1u) ? 1.f : -1.f);
        if (na == 2) for (int a = 0; a < 3; a++) if (!(dir & (1u << a))) on = on && rel[a] == 0.f;   // edge: only its mid-point
        if (on) m |= 1u << (18 - i);
    }
    return m;
}
struct NeighbourMasks { uint32_t m[24]; NeighbourMasks() { for (uint32_t d = 1; d <= 6; d++) for (uint32_t s = 0; s < 4; s++) m[4 * (d - 1) + s] = neighbourMask(d, s); } };
static inline const NeighbourMasks& neighbourMasks() { static const NeighbourMasks n; return n; }

struct CNode {
    uint64_t childIndices; uint32_t parentChildrenIndex; bool isTerminal, ignore;
    uint8_t nDepth[6]; uint32_t nIdx[6];
    V3 center; float size;
    float vv[8][8]; uint32_t vi[8];
    float coeff[64]; float mid[19][8]; uint32_t midInfo[19];
};

static const uint32_t B31 = 1u << 31, B30 = 1u << 30;

struct ContinuityBuilder {
    const MeshView mesh; const std::vector<TriangleData>& td; const SphereBvh& bvh; OctreeSdfData& out;
    uint32_t startDepth, maxDepth; int rule; float sqThr, param1;
    std::vector<uint32_t>& oc;            // mOctreeData
    uint64_t numQueries = 0, numResubdivided = 0;        // numQueries is updated atomically: Iter 1 runs under OpenMP
    ContinuityBuilder(const MeshView& m, const std::vector<TriangleData>& t, const SphereBvh& b, OctreeSdfData& o) : mesh(m), td(t), bvh(b), out(o), oc(o.data) {}

    bool isLeaf(uint32_t at) const { return (oc[at] & LEAF_BIT) != 0; }
    bool isMarked(uint32_t at) const { return (oc[at] & MARK_BIT) != 0; }
    uint32_t childrenIndex(uint32_t at) const { return oc[at] & INDEX_MASK; }
    void setValues(uint32_t at, bool leaf, uint32_t index) { oc[at] = (index & INDEX_MASK) | (leaf ? LEAF_BIT : 0u); }

    void sample(V3 p, float outv[8], uint32_t& info) {
        #pragma omp atomic
        numQueries++;
        info = bvh.nearestTriangle(p); pointValues(p, info, mesh, td, outv);
    }
    // calculateVerticesInfo<19>: bit (18-i) of mask set -> interpolate from coeff, else exact sample
    void midPoints(CNode& n, uint32_t mask) {
        const Stencil& st = stencil();
        for (int i = 0; i < 19; i++) {
            if (mask & (1u << (18 - i))) tricubicVertexValues(n.coeff, 0.5f * st.midRel[i] + 0.5f, 2.0f * n.size, n.mid[i]);
            else sample(n.center + st.midRel[i] * n.size, n.mid[i], n.midInfo[i]);
        }
    }
    uint32_t gridIndexOf(const CNode& n, int* gx = nullptr) const {
        V3 f = (n.center - out.box.min) / out.startGridCellSize;
        const int x = (int)std::floor(f.x), y = (int)std::floor(f.y), z = (int)std::floor(f.z);
        if (gx) { gx[0] = x; gx[1] = y; gx[2] = z; }
        return (uint32_t)(z * out.startGridXY + y * out.startGridSize + x);
    }
    static void neighboursVector(uint32_t o, uint32_t childId, uint32_t pci, uint32_t depth, const uint32_t* pN, const uint8_t* pD, uint32_t* oN, uint8_t* oD) {
        for (uint32_t n = 1; n <= 6; n++) {
            const uint32_t k = (~(o ^ childId)) & n;
            oN[n - 1] = k != 0 ? pN[k - 1] + (n ^ childId) * (1u - (pN[k - 1] >> 31)) : pci + (n ^ childId);
            oD[n - 1] = k != 0 ? pD[k - 1] : (uint8_t)depth;
        }
    }
    void neighboursInGrid(uint32_t o, const int g[3], uint32_t* oN) const {
        const int G = out.startGridSize;
        for (uint32_t n = 1; n <= 6; n++) {
            const int x = g[0] + ((n & 1) ? ((o & 1) ? 1 : -1) : 0), y = g[1] + ((n & 2) ? ((o & 2) ? 1 : -1) : 0), z = g[2] + ((n & 4) ? ((o & 4) ? 1 : -1) : 0);
            oN[n - 1] = (x >= 0 && x < G && y >= 0 && y < G && z >= 0 && z < G) ? (uint32_t)(z * G * G + y * G + x) : B30;
        }
    }
    // 8 children of `node` (its block starts at childIndex); depth = node's depth
    void makeChildren(const CNode& node, uint32_t depth, uint32_t childIndex, std::vector<CNode>& dst, std::vector<uint32_t>* depthDst) {
        const Stencil& st = stencil();
        const float ns = 0.5f * node.size;
        int g[3] = {0, 0, 0};
        if (depth == startDepth) gridIndexOf(node, g);
        for (uint32_t c = 0; c < 8; c++) {
            CNode ch;
            std::memset(&ch, 0, sizeof(CNode));
            ch.parentChildrenIndex = childIndex;
            ch.childIndices = (uint8_t)((node.childIndices << 3) | c);       // uint8_t constructor parameter in the reference
            ch.center = node.center + V3{(c & 1) ? ns : -ns, (c & 2) ? ns : -ns, (c & 4) ? ns : -ns};
            ch.size = ns; ch.isTerminal = false; ch.ignore = false;
            for (int j = 0; j < 8; j++) {
                const int s = st.childSrc[c][j];
                if (s >= 0) { std::memcpy(ch.vv[j], node.mid[s], 32); ch.vi[j] = node.midInfo[s]; }
                else { std::memcpy(ch.vv[j], node.vv[-s - 1], 32); ch.vi[j] = node.vi[-s - 1]; }
            }
            if (depth == startDepth) { neighboursInGrid(c, g, ch.nIdx); for (int k = 0; k < 6; k++) ch.nDepth[k] = (uint8_t)depth; }
            else neighboursVector(c, (uint32_t)(node.childIndices & 7), node.parentChildrenIndex, depth, node.nIdx, node.nDepth, ch.nIdx, ch.nDepth);
            dst.push_back(ch);
            if (depthDst) depthDst->push_back(depth + 1);
        }
    }
    // side code of the outward neighbour `n` for a node with child id c (post-pass formula, :801-803)
    static uint32_t outwardSign(uint32_t n, uint32_t c) {
        return ((((n & c) >> 2) & 1u) << ((n & 1u) | ((n & 2u) >> 1))) + ((((n & c) >> 1) & 1u) << (n & 1u)) + (n & c & 1u);
    }
    // the 18 face/edge neighbours of a node below the start depth: (block word, dir, sign)
    template <typename F> static void forEach18(const CNode& node, F f) {
        const uint32_t c = (uint32_t)(node.childIndices & 7), nc = ~c;
        const uint32_t pci = node.parentChildrenIndex; const uint32_t* N = node.nIdx;
        f(N[0], 1u, c & 1u); f(N[0], 3u, 2u ^ (c & 3u)); f(N[0], 5u, ((nc >> 1) & 2u) + (c & 1u));
        f(N[1], 2u, (c >> 1) & 1u); f(N[1], 3u, 1u ^ (c & 3u)); f(N[1], 6u, 2u ^ ((c >> 1) & 3u));
        f(N[2], 3u, c & 3u);
        f(N[3], 4u, (c >> 2) & 1u); f(N[3], 5u, ((c >> 1) & 2u) + (nc & 1u)); f(N[3], 6u, 1u ^ ((c >> 1) & 3u));
        f(N[4], 5u, ((c >> 1) & 2u) + (c & 1u));
        f(N[5], 6u, (c >> 1) & 3u);
        f(pci, 1u, nc & 1u); f(pci, 2u, (nc >> 1) & 1u); f(pci, 4u, (nc >> 2) & 1u);
        f(pci, 3u, nc & 3u); f(pci, 5u, ((nc >> 1) & 2u) + (nc & 1u)); f(pci, 6u, (nc >> 1) & 3u);
    }
    // the 18 neighbours of a start-grid node: (dx,dy,dz, dir, sign)
    template <typename F> static void forEach18Grid(F f) {
        f(-1, 0, 0, 1u, 0u); f(1, 0, 0, 1u, 1u); f(0, -1, 0, 2u, 0u); f(0, 1, 0, 2u, 1u);
        f(-1, -1, 0, 3u, 0u); f(1, -1, 0, 3u, 1u); f(-1, 1, 0, 3u, 2u); f(1, 1, 0, 3u, 3u);
        f(0, 0, -1, 4u, 0u); f(0, 0, 1, 4u, 1u);
        f(-1, 0, -1, 5u, 0u); f(1, 0, -1, 5u, 1u); f(-1, 0, 1, 5u, 2u); f(1, 0, 1, 5u, 3u);
        f(0, -1, -1, 6u, 0u); f(0, 1, -1, 6u, 1u); f(0, -1, 1, 6u, 2u); f(0, 1, 1, 6u, 3u);
    }

    void run(const Box& inBox, uint32_t depth, uint32_t startDepth_, int rule_, float p0, float p1) {
        startDepth = startDepth_; maxDepth = depth; rule = rule_; sqThr = p0 * p0; param1 = p1;
        const NeighbourMasks& NM = neighbourMasks();
        const Stencil& st = stencil();
        out.maxDepth = depth;
        const V3 bs = inBox.size();
        const float maxSize = gmax(gmax(bs.x, bs.y), bs.z);
        out.box.min = inBox.center() - 0.5f * maxSize; out.box.max = inBox.center() + 0.5f * maxSize;
        out.startGridSize = 1 << startDepth; out.startGridXY = out.startGridSize * out.startGridSize;
        out.startGridCellSize = maxSize / (float)out.startGridSize;
        const uint32_t sod = startDepth < 1u ? startDepth : 1u;
        const int G = out.startGridSize;
        oc.assign((size_t)G * G * G, 0u);
        std::vector<std::vector<CNode>> buf(maxDepth + 1);
        {
            const float newSize = (float)(0.5f * out.box.size().x * std::pow(0.5f, sod));
            const V3 startCenter = out.box.min + newSize;
            const uint32_t vpa = 1u << sod;
            for (uint32_t k = 0; k < vpa; k++) for (uint32_t j = 0; j < vpa; j++) for (uint32_t i = 0; i < vpa; i++) {
                CNode n; std::memset(&n, 0, sizeof(CNode));
                n.parentChildrenIndex = 0xFFFFFFFFu; n.childIndices = 0;
                n.center = startCenter + V3{(float)i, (float)j, (float)k} * 2.0f * newSize; n.size = newSize;
                for (int c = 0; c < 8; c++) sample(n.center + CORNER_REL[c] * n.size, n.vv[c], n.vi[c]);
                buf[sod].push_back(n);
            }
        }
        std::vector<uint32_t> toSubdivide;
        std::map<uint32_t, std::pair<uint32_t, uint32_t>> leaves;
        float valueRange = 0.f;

        for (uint32_t cd = sod; cd <= maxDepth; cd++) {
            // ---------------- Iter 1
            // The reference runs this loop under OpenMP too (OctreeSdfBreadthFirstNoDelay.h:246-370): a node reads only words of
            // shallower levels (final since the previous level) and writes its own word; with the lattice cache off every sample is a
            // pure function of its position, so the result does not depend on the thread count.
            if (cd < maxDepth) {
                const int64_t cnt = (int64_t)buf[cd].size();
                #pragma omp parallel for schedule(dynamic, 16)
                for (int64_t id = 0; id < cnt; id++) {
                    CNode& node = buf[cd][id];
                    if (node.ignore) continue;
                    uint32_t word = 0xFFFFFFFFu;
                    if (cd > startDepth) word = node.parentChildrenIndex + (uint32_t)(node.childIndices & 7);
                    else if (cd == startDepth) word = gridIndexOf(node);
                    if (cd > startDepth) {
                        for (uint32_t nb = 1; nb <= 6; nb++) {
                            uint32_t& ix = node.nIdx[nb - 1];
                            if (((ix >> 30) & 1u) != 0) continue;
                            if (isLeaf(ix & ~B31)) ix = B31 | ix;
                            else {
                                ix = childrenIndex(ix & ~B31); node.nDepth[nb - 1]++;
                                while (node.nDepth[nb - 1] < cd) {
                                    const uint32_t dd = cd - node.nDepth[nb - 1];
                                    const uint32_t cid = (uint32_t)((node.childIndices >> (3 * dd)) & 7);
                                    ix += (nb ^ cid);
                                    if (isLeaf(ix & ~B31)) { ix = B31 | ix; break; }
                                    ix = childrenIndex(ix & ~B31); node.nDepth[nb - 1]++;
                                }
                            }
                        }
                    }
                    if (cd >= startDepth) tricubicFit(node.vv, 2.0f * node.size, node.coeff);
                    midPoints(node, 0u);
                    bool terminal = false;
                    if (cd >= startDepth && rule != RULE_NONE) terminal = ruleValue(rule, node.coeff, node.mid, param1) < sqThr;
                    node.isTerminal = terminal;
                    if (word != 0xFFFFFFFFu) setValues(word, terminal, 0xFFFFFFFFu);
                }
            }
            // ---------------- Iter 2
            toSubdivide.clear();
            for (size_t id = 0; id < buf[cd].size(); id++) {
                CNode& node = buf[cd][id];         // buf[cd + 1] grows below, buf[cd] does not (the post-pass runs after this loop)
                if (node.ignore) continue;
                uint32_t word = 0xFFFFFFFFu; int g[3] = {0, 0, 0};
                if (cd > startDepth) word = node.parentChildrenIndex + (uint32_t)(node.childIndices & 7);
                else if (cd == startDepth) word = gridIndexOf(node, g);
                if (!node.isTerminal && cd < maxDepth) {
                    uint32_t samplesMask = 0; uint32_t nbIds[24];
                    for (int k = 0; k < 24; k++) nbIds[k] = 0xFFFFFFFFu;
                    if (cd > startDepth) {
                        const uint32_t c = (uint32_t)(node.childIndices & 7);
                        forEach18(node, [&](uint32_t nodeId, uint32_t dir, uint32_t sign) {
                            if ((nodeId >> 31) || (!(nodeId >> 30) && isLeaf(nodeId + (dir ^ c)))) {
                                nbIds[4 * (dir - 1) + sign] = (nodeId >> 31) ? (nodeId & ~B31) : nodeId + (dir ^ c);
                                samplesMask |= NM.m[4 * (dir - 1) + sign];
                            }
                        });
                    } else if (cd == startDepth) {
                        forEach18Grid([&](int dx, int dy, int dz, uint32_t dir, uint32_t sign) {
                            const int x = g[0] + dx, y = g[1] + dy, z = g[2] + dz;
                            if (x >= 0 && x < G && y >= 0 && y < G && z >= 0 && z < G) {
                                const uint32_t at = (uint32_t)(z * G * G + y * G + x);
                                if (isLeaf(at)) { nbIds[4 * (dir - 1) + sign] = at; samplesMask |= NM.m[4 * (dir - 1) + sign]; }
                            }
                        });
                    }
                    uint32_t subdivisionMask = 0;
                    for (int i = 0; i < 19; i++) {
                        if (!(samplesMask & (1u << (18 - i)))) continue;
                        const V3 f = 0.5f * st.midRel[i] + 0.5f;
                        const float iv = tricubicValue(node.coeff, f);
                        const float e = node.mid[i][0] - iv;
                        if (e * e > sqThr) subdivisionMask |= (samplesMask & (1u << (18 - i)));
                        else tricubicVertexValues(node.coeff, f, 2.0f * node.size, node.mid[i]);
                    }
                    for (int k = 0; k < 24; k++) if ((subdivisionMask & NM.m[k]) && !(nbIds[k] >> 30)) toSubdivide.push_back(nbIds[k]);
                    uint32_t childIndex = 0xFFFFFFFFu;
                    if (cd >= startDepth) {
                        childIndex = (uint32_t)oc.size();
                        setValues(word, false, childIndex);
                        oc.resize(oc.size() + 8, ~(7u << 29));
                    }
                    makeChildren(node, cd, childIndex, buf[cd + 1], nullptr);
                } else {
                    const uint32_t at = (uint32_t)oc.size();
                    setValues(word, true, at);
                    oc.resize(oc.size() + 64);
                    if (cd >= maxDepth) tricubicFit(node.vv, 2.0f * node.size, node.coeff);
                    std::memcpy(&oc[at], node.coeff, 256);
                    for (int i = 0; i < 8; i++) valueRange = gmax(valueRange, std::fabs(node.vv[i][0]));
                    leaves.insert(std::make_pair(word, std::make_pair(cd, (uint32_t)id)));
                }
            }
            numResubdivided += toSubdivide.size();
            // ---------------- post-pass: re-subdivide the scheduled leaves
            for (size_t si = 0; si < toSubdivide.size(); si++) {
                const uint32_t nodeId = toSubdivide[si];
                auto it = leaves.find(nodeId);
                if (it == leaves.end()) continue;     // (the reference prints a message and then dereferences end(): undefined; never observed)
                std::vector<CNode> cache; std::vector<uint32_t> dcache;
                cache.push_back(buf[it->second.first][it->second.second]); dcache.push_back(it->second.first);
                uint32_t pword;
                if (it->second.first > startDepth) pword = cache[0].parentChildrenIndex + (uint32_t)(cache[0].childIndices & 7);
                else pword = gridIndexOf(cache[0]);
                if (!isLeaf(pword)) continue;
                bool recycled = false;
                const uint32_t oldCoeffIndex = childrenIndex(pword);
                bool first = true;
                size_t ci = 0;
                while (ci < cache.size()) {
                    CNode node = cache[ci]; const uint32_t depthN = dcache[ci]; ci++;
                    uint32_t word = 0xFFFFFFFFu; int g[3] = {0, 0, 0};
                    if (depthN > startDepth) word = node.parentChildrenIndex + (uint32_t)(node.childIndices & 7);
                    else if (depthN == startDepth) word = gridIndexOf(node, g);
                    uint32_t samplesMask = 0, subdividedMask = 0;
                    const uint32_t c = (uint32_t)(node.childIndices & 7);
                    if (depthN > startDepth) {
                        for (uint32_t nb = 1; nb <= 6; nb++) {
                            const uint32_t sign = outwardSign(nb, c);
                            uint32_t& ix = node.nIdx[nb - 1];
                            if (((ix >> 30) & 1u) != 0) continue;
                            if ((!first || (ix >> 31)) && isLeaf(ix & ~B31)) { ix = B31 | ix; samplesMask |= NM.m[4 * (nb - 1) + sign]; }
                            else {
                                if (!first || (ix >> 31)) { ix = childrenIndex(ix & ~B31); node.nDepth[nb - 1]++; }
                                while (node.nDepth[nb - 1] < depthN && node.nDepth[nb - 1] < cd) {
                                    const uint32_t dd = depthN - node.nDepth[nb - 1];
                                    const uint32_t cid = (uint32_t)((node.childIndices >> (3 * dd)) & 7);
                                    ix += (nb ^ cid);
                                    if (isLeaf(ix & ~B31)) { ix = B31 | ix; samplesMask |= NM.m[4 * (nb - 1) + sign]; break; }
                                    ix = childrenIndex(ix & ~B31); node.nDepth[nb - 1]++;
                                }
                                if (cd >= depthN && !(ix >> 31)) {
                                    const uint32_t next = (ix & ~B31) + (nb ^ c);
                                    subdividedMask |= (isLeaf(next) || isMarked(next)) ? 0u : NM.m[4 * (nb - 1) + sign];
                                }
                            }
                        }
                    }
                    if (cd >= depthN) {
                        if (depthN > startDepth) {
                            const uint32_t nc = ~c, pci = node.parentChildrenIndex; const uint32_t* N = node.nIdx;
                            auto upd = [&](uint32_t nid, uint32_t dir, uint32_t sign) {
                                const bool leafish = (nid >> 31) || (nid >> 30) || isLeaf(nid + (dir ^ c)) || isMarked(nid + (dir ^ c));
                                subdividedMask |= leafish ? 0u : NM.m[4 * (dir - 1) + sign];
                            };
                            upd(pci, 1u, nc & 1u); upd(pci, 2u, (nc >> 1) & 1u); upd(pci, 4u, (nc >> 2) & 1u);
                            upd(pci, 3u, nc & 3u); upd(pci, 5u, ((nc >> 1) & 2u) + (nc & 1u)); upd(pci, 6u, (nc >> 1) & 3u);
                            upd(N[0], 3u, 2u ^ (c & 3u)); upd(N[0], 5u, ((nc >> 1) & 2u) + (c & 1u));
                            upd(N[1], 3u, 1u ^ (c & 3u)); upd(N[1], 6u, 2u ^ ((c >> 1) & 3u));
                            upd(N[3], 5u, ((c >> 1) & 2u) + (nc & 1u)); upd(N[3], 6u, 1u ^ ((c >> 1) & 3u));
                        } else if (depthN == startDepth) {
                            forEach18Grid([&](int dx, int dy, int dz, uint32_t dir, uint32_t sign) {
                                const int x = g[0] + dx, y = g[1] + dy, z = g[2] + dz;
                                if (x >= 0 && x < G && y >= 0 && y < G && z >= 0 && z < G) {
                                    const uint32_t at = (uint32_t)(z * G * G + y * G + x);
                                    subdividedMask |= (isLeaf(at) || isMarked(at)) ? 0u : NM.m[4 * (dir - 1) + sign];
                                }
                            });
                        }
                        samplesMask = ~subdividedMask;
                    }
                    if (cd >= depthN && samplesMask != 0xFFFFFFFFu) {
                        const bool recycleMid = first && !node.ignore;
                        if (!recycleMid) { tricubicFit(node.vv, 2.0f * node.size, node.coeff); midPoints(node, samplesMask); }
                        for (int i = 0; i < 19; i++) {
                            const V3 f = 0.5f * st.midRel[i] + 0.5f;
                            if ((samplesMask & (1u << (18 - i))) == 0) {
                                const float iv = tricubicValue(node.coeff, f);
                                const float e = node.mid[i][0] - iv;
                                if (e * e < sqThr) tricubicVertexValues(node.coeff, f, 2.0f * node.size, node.mid[i]);
                            } else if (recycleMid) tricubicVertexValues(node.coeff, f, 2.0f * node.size, node.mid[i]);
                        }
                        uint32_t childIndex = 0xFFFFFFFFu;
                        if (depthN >= startDepth) {
                            childIndex = (uint32_t)oc.size();
                            setValues(word, false, childIndex);
                            oc[word] |= MARK_BIT;
                            oc.resize(oc.size() + 8, LEAF_BIT);        // setValues(true, 0)
                        }
                        makeChildren(node, depthN, childIndex, cache, &dcache);
                    } else {
                        uint32_t at = (uint32_t)oc.size();
                        if (recycled) { setValues(word, true, at); oc.resize(oc.size() + 64); }
                        else { at = oldCoeffIndex; setValues(word, true, at); recycled = true; }
                        tricubicFit(node.vv, 2.0f * node.size, node.coeff);
                        std::memcpy(&oc[at], node.coeff, 256);
                        node.isTerminal = true; node.ignore = true;
                        buf[depthN].push_back(node);
                        leaves.insert(std::make_pair(node.parentChildrenIndex + (uint32_t)(node.childIndices & 7), std::make_pair(depthN, (uint32_t)(buf[depthN].size() - 1))));
                    }
                    first = false;
                }
            }
        }
        // clear the mark bits by walking the tree from every start cell (:1191-1217)
        std::function<void(uint32_t)> unmark = [&](uint32_t at) {
            oc[at] &= ~MARK_BIT;
            if (!(oc[at] & LEAF_BIT)) for (uint32_t c = 0; c < 8; c++) unmark((oc[at] & INDEX_MASK) + c);
        };
        for (uint32_t at = 0; at < (uint32_t)(G * G * G); at++) unmark(at);
        out.valueRange = valueRange;
        out.numBvhQueries = numQueries;
    }
};

}  // namespace orc
