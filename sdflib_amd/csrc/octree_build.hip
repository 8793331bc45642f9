// OctreeSdf construction on the device (NO_CONTINUITY algorithm).  PRODUCT code — independent of oracle/.
//
// Reference behaviour reproduced: OctreeSdf::buildOctree (src/sdf/OctreeSdf.cpp:37-86), initOctree<VHQueries<TriCubic>>
// and its processNode (src/sdf/OctreeSdfDepthFirst.h:32-558), VHQueries::calculateVerticesInfo with the lattice
// cache disabled ("canonical" mode, include/SdfLib/TrianglesInfluence.h:951-996), computeMinBorderValue
// (src/sdf/OctreeSdf.cpp:155-230).
//
// MI355X formulation.  The reference walks the tree depth-first, one node at a time; here the tree is grown
// LEVEL-SYNCHRONOUSLY (all nodes of a depth at once = 10^2..10^5.5 nodes x 19 samples of parallelism):
//   sampleBatch (octree_sampler.h)     : exact samples of whole levels: dedup by lattice point, fp64 BVH nearest triangle per
//                                        unique point (dev_bvh.h), fp32 Hermite datum per sample
//   k_decide          : one lane per node: 64x64 fit (reference summation order), 19-point error rule, leaf/inner
//   prefix sum + k_scatter_children : child slots; the 27-point stencil is handed down to the 8 children
// and afterwards the breadth-first arrays are relabelled into the reference's array layout:
//   k_alloc (bottom-up) : words needed by every subtree ; k_emit (top-down) : pre-order offsets, children 7..0,
// which is exactly the order in which the reference's DFS stack appends blocks (OctreeSdfDepthFirst.h:213-336).
// Results do not depend on the breadth-first order, only on the tree, so they equal the reference's array.
// All fp32 arithmetic follows the reference's operation order; compile with -ffp-contract=off.
#include "octree_internal.h"
#include "dev_bvh.h"
#include "dev_tricubic.h"
#include "dev_fit_mfma.h"
#include "octree_sampler.h"
#include "dev_prims.h"
#include <cmath>
#include <cstring>

namespace sdfhip {


SDF_DEV uint32_t floatOrderKey(float f) { const uint32_t b = __float_as_uint(f); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
SDF_HD float floatFromOrderKey(uint32_t k) { const uint32_t b = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k; float f; memcpy(&f, &b, 4); return f; }

struct DecideArgs {
    const float* corner; const float* mid; const uint32_t* coord;
    uint32_t n, depth, maxDepth; float half; int rule; float sqThreshold, param1;
    uint32_t* flag; uint32_t* inner; float* coeff;
    uint32_t* valueRangeBits;   // atomicMax over |corner value| bits
    uint32_t* minBorderKey;     // atomicMin over order keys
    uint32_t* recheckCount;     // FIT_MFMA: decisions re-evaluated with the exact fit
};

// One lane per node: fit, error rule, leaf/inner decision, leaf payload.
// MFMA = false: the 64x64 fit is computed here in the reference's summation order (defines the topology).
// MFMA = true : the coefficients were produced by k_fit_mfma; decisions closer to the threshold than the fit's rounding
//               uncertainty are re-evaluated with the reference-ordered fit, so the topology is the same as with MFMA = false.
template <bool MFMA>
__global__ void __launch_bounds__(128) k_decide(DecideArgs a) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    float s[64], c[64];
    const float4* cr = reinterpret_cast<const float4*>(a.corner) + 8 * (size_t)i;
    float absMax = 0.f;
    auto loadCorners = [&]() {
#pragma unroll
        for (int v = 0; v < 8; v++) {
            const float4 q = cr[v];
            s[8 * v] = q.x; s[8 * v + 1] = q.y; s[8 * v + 2] = q.z; s[8 * v + 3] = q.w;
            s[8 * v + 4] = 0.f; s[8 * v + 5] = 0.f; s[8 * v + 6] = 0.f; s[8 * v + 7] = 0.f;
        }
    };
#pragma unroll
    for (int v = 0; v < 8; v++) absMax = gmax(absMax, fabsf(cr[v].x));
    if (MFMA) {
        const float4* src = reinterpret_cast<const float4*>(a.coeff) + 16 * (size_t)i;
#pragma unroll
        for (int q = 0; q < 16; q++) { const float4 v4 = src[q]; c[4 * q] = v4.x; c[4 * q + 1] = v4.y; c[4 * q + 2] = v4.z; c[4 * q + 3] = v4.w; }
    } else {
        loadCorners();
        tricubicFit(s, 2.0f * a.half, c);
    }
    bool terminal = true;
    bool rechecked = false;
    if (a.depth < a.maxDepth) {
        const float4* md = reinterpret_cast<const float4*>(a.mid) + 19 * (size_t)i;
        float v = ruleValue(a.rule, [&](int n) { return c[n]; }, [&](int m) { return md[m].x; }, a.param1);
        if (MFMA) {
            float cmax = 0.f;
#pragma unroll
            for (int n = 0; n < 64; n++) cmax = gmax(cmax, fabsf(c[n]));
            const float eps = 1e-5f * cmax;
            const float tol = 8.0f * sqrtf(gmax(v, a.sqThreshold)) * eps + eps * eps;
            if (!(fabsf(v - a.sqThreshold) > tol)) {          // close to the threshold (or NaN): decide with the reference-ordered fit
                loadCorners();
                tricubicFit(s, 2.0f * a.half, c);
                v = ruleValue(a.rule, [&](int n) { return c[n]; }, [&](int m) { return md[m].x; }, a.param1);
                rechecked = true;
                atomicAdd(a.recheckCount, 1u);
            }
        }
        terminal = v < a.sqThreshold;
    }
    a.flag[i] = terminal ? 1u : 0u;
    a.inner[i] = terminal ? 0u : 1u;
    if (!terminal) return;
    if (!MFMA || rechecked) {
        float4* dst = reinterpret_cast<float4*>(a.coeff) + 16 * (size_t)i;
#pragma unroll
        for (int q = 0; q < 16; q++) dst[q] = make_float4(c[4 * q], c[4 * q + 1], c[4 * q + 2], c[4 * q + 3]);
    }
    if (__float_as_uint(absMax) > *reinterpret_cast<volatile uint32_t*>(a.valueRangeBits)) atomicMax(a.valueRangeBits, __float_as_uint(absMax));      // (the maximum only grows: most leaves need no atomic)
    // border corners (computeMinBorderValue): a leaf corner lying on the box boundary contributes P(corner)
    const uint32_t co = a.coord[i];
    const uint32_t x = co & 1023u, y = (co >> 10) & 1023u, z = co >> 20;
    const uint32_t last = (1u << a.depth) - 1u;
    if (x == 0 || y == 0 || z == 0 || x == last || y == last || z == last) {
        float mn = INFINITY;
#pragma unroll 1
        for (uint32_t k = 0; k < 8; k++) {
            const uint32_t bx = k & 1u, by = (k >> 1) & 1u, bz = k >> 2;
            const bool onBorder = (x + bx == 0) || (y + by == 0) || (z + bz == 0) || (x + bx == last + 1) || (y + by == last + 1) || (z + bz == last + 1);
            if (!onBorder) continue;
            const float v = tricubicValueExact([&](int n) { return c[n]; }, F3{(float)bx, (float)by, (float)bz});
            mn = gmin(mn, v);
        }
        if (mn < INFINITY) atomicMin(a.minBorderKey, floatOrderKey(mn));
    }
}

__global__ void k_fill_all_inner(uint32_t n, uint32_t* __restrict__ flag, uint32_t* __restrict__ inner, uint32_t* __restrict__ childBase) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    flag[i] = 0u; inner[i] = 1u; childBase[i] = 8u * i;
}

__global__ void k_scale8(uint32_t n, uint32_t* __restrict__ v) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] *= 8u;
}

// The speculative levels descend from THIS build's start cells only, in cell order (z-major, the shard's range): a shard of an N-GPU build
// samples its own eighth of them, not the whole level (until round 5 every shard sampled all of it: 6.9 ms per shard at world 8 against
// 11.9 ms for the whole build).  Geometry of the start cells [cellBegin, cellEnd) out of the complete start level:
__global__ void k_spec_base(const float* __restrict__ center, const uint32_t* __restrict__ coord, uint32_t n, uint32_t G, uint32_t cellBegin, uint32_t cellEnd,
                            float* __restrict__ ocenter, uint32_t* __restrict__ ocoord) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t co = coord[i];
    const uint32_t cell = (co >> 20) * G * G + ((co >> 10) & 1023u) * G + (co & 1023u);
    if (cell < cellBegin || cell >= cellEnd) return;
    const uint32_t d = cell - cellBegin;
    ocenter[3 * (size_t)d] = center[3 * (size_t)i]; ocenter[3 * (size_t)d + 1] = center[3 * (size_t)i + 1]; ocenter[3 * (size_t)d + 2] = center[3 * (size_t)i + 2];
    ocoord[d] = co;
}
// index of a node in a speculative level: its start cell's place in the range, then one octal digit per depth below the start depth
// (digit = x_bit | y_bit << 1 | z_bit << 2, the order k_expand_geometry writes children in)
SDF_DEV uint32_t specLevelIndex(uint32_t co, uint32_t depth, uint32_t startDepth, uint32_t G, uint32_t cellBegin) {
    const uint32_t s = depth - startDepth, x = co & 1023u, y = (co >> 10) & 1023u, z = co >> 20;
    uint32_t idx = (z >> s) * G * G + (y >> s) * G + (x >> s) - cellBegin;
    for (int b = (int)s - 1; b >= 0; b--) idx = (idx << 3) | ((x >> b) & 1u) | (((y >> b) & 1u) << 1) | (((z >> b) & 1u) << 2);
    return idx;
}
// mid-points of the nodes that turned out to exist, taken from the speculatively sampled level
__global__ void k_gather_spec_mids(const uint32_t* __restrict__ coord, uint32_t n, uint32_t depth, uint32_t startDepth, uint32_t G, uint32_t cellBegin,
                                   const float* __restrict__ specMid, float* __restrict__ mid) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t i = gid / 19u, m = gid - 19u * i;
    if (i >= n) return;
    reinterpret_cast<float4*>(mid)[gid] = reinterpret_cast<const float4*>(specMid)[19 * (size_t)specLevelIndex(coord[i], depth, startDepth, G, cellBegin) + m];
}

struct ScatterArgs {
    const float* center; const uint32_t* coord; const float* corner; const float* mid; const uint32_t* inner; const uint32_t* childBase;
    uint32_t n; float half;
    float* ncenter; uint32_t* ncoord; float* ncorner;
};

// One lane per (node, child, corner): hand the 27-point stencil down (OctreeSdfDepthFirst.h:225-336 tables).
__global__ void __launch_bounds__(256) k_scatter_children(ScatterArgs a) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t i = gid >> 6;
    if (i >= a.n || !a.inner[i]) return;
    const uint32_t c = (gid >> 3) & 7u, j = gid & 7u;
    const uint32_t child = a.childBase[i] + c;
    const int src = kStencilDev.src[c][j];
    const float4 v = (src >= 0) ? reinterpret_cast<const float4*>(a.mid)[19 * (size_t)i + src]
                                : reinterpret_cast<const float4*>(a.corner)[8 * (size_t)i + (-src - 1)];
    reinterpret_cast<float4*>(a.ncorner)[8 * (size_t)child + j] = v;
    if (j == 0) {
        const float ns = 0.5f * a.half;
        a.ncenter[3 * (size_t)child] = a.center[3 * (size_t)i] + ((c & 1u) ? ns : -ns);
        a.ncenter[3 * (size_t)child + 1] = a.center[3 * (size_t)i + 1] + ((c & 2u) ? ns : -ns);
        a.ncenter[3 * (size_t)child + 2] = a.center[3 * (size_t)i + 2] + ((c & 4u) ? ns : -ns);
        const uint32_t co = a.coord[i];
        const uint32_t x = 2u * (co & 1023u) + (c & 1u), y = 2u * ((co >> 10) & 1023u) + ((c >> 1) & 1u), z = 2u * (co >> 20) + (c >> 2);
        a.ncoord[child] = x | (y << 10) | (z << 20);
    }
}

// Move the start-depth level into start-grid cell order (z-major) and keep only cells [cellBegin, cellEnd).
__global__ void k_to_cell_order(const float* __restrict__ center, const uint32_t* __restrict__ coord, const float* __restrict__ corner, uint32_t n,
                                uint32_t G, uint32_t cellBegin, uint32_t cellEnd, float* __restrict__ ocenter, uint32_t* __restrict__ ocoord, float* __restrict__ ocorner,
                                const float* __restrict__ mid, float* __restrict__ omid) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t i = gid >> 3, j = gid & 7u;
    if (i >= n) return;
    const uint32_t co = coord[i];
    const uint32_t cell = (co >> 20) * G * G + ((co >> 10) & 1023u) * G + (co & 1023u);
    if (cell < cellBegin || cell >= cellEnd) return;
    const uint32_t d = cell - cellBegin;
    reinterpret_cast<float4*>(ocorner)[8 * (size_t)d + j] = reinterpret_cast<const float4*>(corner)[8 * (size_t)i + j];
    if (mid) for (uint32_t m = j; m < 19u; m += 8u) reinterpret_cast<float4*>(omid)[19 * (size_t)d + m] = reinterpret_cast<const float4*>(mid)[19 * (size_t)i + m];
    if (j == 0) {
        ocenter[3 * (size_t)d] = center[3 * (size_t)i]; ocenter[3 * (size_t)d + 1] = center[3 * (size_t)i + 1]; ocenter[3 * (size_t)d + 2] = center[3 * (size_t)i + 2];
        ocoord[d] = co;
    }
}

// bottom-up: words of a node's block plus all descendants' blocks
__global__ void k_alloc(uint32_t n, const uint32_t* __restrict__ flag, const uint32_t* __restrict__ childBase, const uint32_t* __restrict__ nextAlloc, uint32_t* __restrict__ alloc) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (flag[i]) { alloc[i] = 64u; return; }
    uint32_t s = 8u;
    const uint32_t cb = childBase[i];
#pragma unroll
    for (int c = 0; c < 8; c++) s += nextAlloc[cb + c];
    alloc[i] = s;
}

struct EmitArgs {
    uint32_t n; const uint32_t* flag; const uint32_t* childBase; const float* coeff; const uint32_t* pos; const uint32_t* blk;
    const uint32_t* nextAlloc; uint32_t* nextPos; uint32_t* nextBlk;
    // destination: absolute index a < gridEnd lives in grid[a - gridBegin], otherwise in body[a - bodyOffset]
    uint32_t* grid; uint32_t gridBegin, gridEnd; uint32_t* body; uint64_t bodyOffset;
    // LAYOUT mode (k_emit<true>): the level's slice of the query layout instead — the node's word (as the reference array would hold it) into
    // qOrig, its layout word into qTopo (inner: index of its child block in the next level's slice; leaf: LEAF_BIT | block id), a leaf's
    // coefficients into its 256-byte-aligned block of qCoef.  Leaves take their block ids in level order: id = leafBase + (i - inner nodes before i).
    uint32_t* qTopo; uint32_t* qOrig; float* qCoef; uint32_t nextLevelBase, leafBase;
};

// top-down: write node words and leaf payloads at their reference positions; children are laid out 7..0
template <bool LAYOUT>
__global__ void __launch_bounds__(256) k_emit(EmitArgs a) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t i = gid >> 4, q = gid & 15u;     // 16 lanes per node: each copies one float4 of a leaf payload
    if (i >= a.n) return;
    const uint32_t pos = a.pos[i], blk = a.blk[i];
    const bool leaf = a.flag[i] != 0u;
    const uint32_t cbase = a.childBase[i];           // 8 x (inner nodes before i); for an inner node: its first child in the next level
    if (q == 0) {
        const uint32_t word = (blk & INDEX_MASK) | (leaf ? LEAF_BIT : 0u);
        if (LAYOUT) { a.qOrig[i] = word; a.qTopo[i] = leaf ? (LEAF_BIT | (a.leafBase + i - (cbase >> 3))) : a.nextLevelBase + cbase; }
        else if (pos < a.gridEnd) a.grid[pos - a.gridBegin] = word; else a.body[pos - a.bodyOffset] = word;
    }
    if (leaf) {
        const float4 v = reinterpret_cast<const float4*>(a.coeff)[16 * (size_t)i + q];
        if (LAYOUT) reinterpret_cast<float4*>(a.qCoef)[16 * (size_t)(a.leafBase + i - (cbase >> 3)) + q] = v;
        else {
            uint32_t* d = a.body + (blk - a.bodyOffset) + 4 * q;      // blocks always live in the body
            d[0] = __float_as_uint(v.x); d[1] = __float_as_uint(v.y); d[2] = __float_as_uint(v.z); d[3] = __float_as_uint(v.w);
        }
    } else if (q < 8) {
        const uint32_t cb = a.childBase[i];
        uint32_t off = blk + 8u;
        for (uint32_t c = 7; c > q; c--) off += a.nextAlloc[cb + c];
        a.nextPos[cb + q] = blk + q;
        a.nextBlk[cb + q] = off;
    }
}

__global__ void k_init_start_pos(uint32_t n, uint32_t cellBegin, const uint32_t* __restrict__ blkIn, uint32_t* __restrict__ pos, uint32_t* __restrict__ blk) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    pos[i] = cellBegin + i; blk[i] = blkIn[i];
}

int continuityBuildImpl(sdfhip_ctx* ctx, sdfhip_mesh* mesh, const sdfhip_octree_params* P, sdfhip_octree** out);   // octree_continuity.hip

static int allocLevelCommon(BuildLevel& L) {
    SDF_TRY(L.center.reserve(3ull * L.n));
    SDF_TRY(L.coord.reserve(L.n));
    SDF_TRY(L.corner.reserve(32ull * L.n));
    return SDFHIP_OK;
}

// Rank of a start-grid cell in the reference's single-thread DFS (roots and children are popped 7..0).
static uint32_t dfsRankOfCell(uint32_t x, uint32_t y, uint32_t z, uint32_t startDepth, uint32_t sod) {
    uint32_t rank = 0;
    for (uint32_t l = 0; l < startDepth; l++) {          // l = 0 is the most significant level
        const uint32_t sh = startDepth - 1 - l;
        const uint32_t c = ((x >> sh) & 1u) | (((y >> sh) & 1u) << 1) | (((z >> sh) & 1u) << 2);
        rank = rank * 8u + (7u - c);
    }
    (void)sod;
    return rank;
}

// The finished levels written out top-down (the caller is inside an AllocScope on the context's stream): as the reference's array — grid
// words of this shard's cells into dGrid, its subtree bodies (first word = absolute index body_offset) into dBody — or, layout = true
// (complete builds only), as the tree's query layout (sdfhip_octree::qTopo / qOrig / qCoef; see EmitArgs).
static int emitLevels(sdfhip_octree* T, uint64_t body_offset, uint32_t* dGrid, uint32_t* dBody, bool layout) {
    hipStream_t st = T->ctx->stream;
    const uint32_t sod = T->startOctreeDepth, startDepth = T->params.start_depth, maxDepth = T->params.depth;
    const uint32_t G = 1u << startDepth, G3 = G * G * G;
    const uint32_t cellBegin = T->info.cell_begin, cellEnd = T->info.cell_end, nCells = cellEnd - cellBegin;
    BuildLevel* S = T->levels[startDepth - sod].get();
    SDF_REQUIRE(S && S->n == nCells, "internal: start level missing");
    // body offsets of the cells: prefix sums of the root subtree sizes in body order
    std::vector<uint32_t> rootAlloc(nCells), rootBlk(nCells);
    SDF_HIP_CHECK(hipMemcpyAsync(rootAlloc.data(), S->alloc.p, 4ull * nCells, hipMemcpyDeviceToHost, st));
    SDF_HIP_CHECK(hipStreamSynchronize(st));
    if (T->params.layout == SDFHIP_LAYOUT_SUBTREES) {
        uint64_t off = body_offset;
        for (uint32_t c = 0; c < nCells; c++) { rootBlk[c] = (uint32_t)off; off += rootAlloc[c]; }
    } else {
        std::vector<uint32_t> cellOfRank(G3);
        for (uint32_t z = 0; z < G; z++) for (uint32_t y = 0; y < G; y++) for (uint32_t x = 0; x < G; x++)
            cellOfRank[dfsRankOfCell(x, y, z, startDepth, sod)] = z * G * G + y * G + x;
        uint64_t off = body_offset;
        for (uint32_t r = 0; r < G3; r++) { const uint32_t c = cellOfRank[r]; rootBlk[c] = (uint32_t)off; off += rootAlloc[c]; }
    }
    DevBuf<uint32_t> dRootBlk; SDF_TRY(dRootBlk.reserve(nCells));
    SDF_HIP_CHECK(hipMemcpyAsync(dRootBlk.p, rootBlk.data(), 4ull * nCells, hipMemcpyHostToDevice, st));
    k_init_start_pos<<<gridFor(nCells, 256), 256, 0, st>>>(nCells, cellBegin, dRootBlk.p, S->pos.p, S->blk.p);
    uint64_t totalNodes = 0, totalLeaves = 0;
    if (layout) {
        SDF_REQUIRE(nCells == G3, "internal: the query layout is emitted by complete builds only");
        std::vector<uint32_t> levelNodes, levelLeafBase(1, 0u);
        for (uint32_t d = startDepth; d <= maxDepth; d++) {
            const BuildLevel* L = T->levels[d - sod].get();
            if (!L || L->n == 0) break;
            totalNodes += L->n; totalLeaves += L->numLeaves;
            levelNodes.push_back(L->n); levelLeafBase.push_back((uint32_t)totalLeaves);
        }
        SDF_REQUIRE(totalNodes < (1ull << 30), "tree too large for the query layout");
        SDF_TRY(T->qTopo.reserve(totalNodes)); SDF_TRY(T->qOrig.reserve(totalNodes)); SDF_TRY(T->qCoef.reserve(64ull * (totalLeaves ? totalLeaves : 1)));
        T->qNodes = totalNodes; T->qLeaves = totalLeaves; T->qLevelNodes = levelNodes; T->qLevelLeafBase = levelLeafBase;
    }
    uint64_t levelBase = 0, leafBase = 0;
    for (uint32_t d = startDepth; d <= maxDepth; d++) {
        BuildLevel* L = T->levels[d - sod].get();
        if (!L || L->n == 0) break;
        BuildLevel* N = (d < maxDepth) ? T->levels[d + 1 - sod].get() : nullptr;
        EmitArgs ea{L->n, L->flag.p, L->childBase.p, L->coeff.p, L->pos.p, L->blk.p,
                    N ? N->alloc.p : nullptr, N ? N->pos.p : nullptr, N ? N->blk.p : nullptr,
                    dGrid, cellBegin, G3, dBody, body_offset,
                    layout ? T->qTopo.p + levelBase : nullptr, layout ? T->qOrig.p + levelBase : nullptr, layout ? T->qCoef.p : nullptr, (uint32_t)(levelBase + L->n), (uint32_t)leafBase};
        // grid words exist only for the start level; deeper node words always live in bodies
        if (d > startDepth) { ea.gridBegin = 0; ea.gridEnd = 0; }
        if (layout) k_emit<true><<<gridFor(16ull * L->n, 256), 256, 0, st>>>(ea);
        else k_emit<false><<<gridFor(16ull * L->n, 256), 256, 0, st>>>(ea);
        levelBase += L->n; leafBase += L->numLeaves;
    }
    SDF_HIP_CHECK(hipGetLastError());
    SDF_HIP_CHECK(hipStreamSynchronize(st));
    if (layout) T->qReady = true;
    return SDFHIP_OK;
}

static int buildImpl(sdfhip_ctx* ctx, sdfhip_mesh* mesh, const sdfhip_octree_params* P, bool shardOnly, sdfhip_octree** out) {
    SDF_REQUIRE(ctx && mesh && P && out, "NULL argument");
    SDF_REQUIRE(mesh->ctx == ctx, "mesh belongs to another context");
    std::lock_guard<std::recursive_mutex> building(ctx->buildLock);
    NearScratchMark nearMark(ctx);
    if (P->algorithm == SDFHIP_ALG_CONTINUITY) {
        SDF_REQUIRE(!shardOnly, "the CONTINUITY builder is not sharded (Iter 2 couples neighbouring start cells)");
        return continuityBuildImpl(ctx, mesh, P, out);
    }
    if (P->algorithm != SDFHIP_ALG_NO_CONTINUITY) { setError("algorithm %d is not provided (UNIFORM is test-only in the reference)", P->algorithm); return SDFHIP_E_UNSUPPORTED; }
    SDF_REQUIRE(P->depth >= 1, "depth must be at least 1");
    if (P->depth > 10) { setError("depth %u is above this build's limit of 10: node coordinates are packed 10 bits per axis (the reference's own limit is its 30-bit word index, OctreeSdf.h:53-55, which a depth-11 tree of a real surface exceeds anyway)", (unsigned)P->depth); return SDFHIP_E_UNSUPPORTED; }
    SDF_REQUIRE(P->start_depth <= P->depth, "start_depth > depth");
    SDF_REQUIRE(P->rule >= SDFHIP_RULE_NONE && P->rule <= SDFHIP_RULE_BY_DISTANCE, "unknown termination rule");
    SDF_REQUIRE(P->layout == SDFHIP_LAYOUT_GLOBAL_DFS || P->layout == SDFHIP_LAYOUT_SUBTREES, "unknown layout");
    SDF_REQUIRE(P->fit_mode == SDFHIP_FIT_EXACT || P->fit_mode == SDFHIP_FIT_MFMA, "unknown fit_mode");
    const uint32_t maxDepth = P->depth, startDepth = P->start_depth;
    const uint32_t G = 1u << startDepth, G3 = G * G * G;
    uint32_t cellBegin = P->cell_begin, cellEnd = P->cell_end;
    if (cellBegin == 0 && cellEnd == 0) cellEnd = G3;
    SDF_REQUIRE(cellBegin < cellEnd && cellEnd <= G3, "bad cell range");
    const bool partial = !(cellBegin == 0 && cellEnd == G3);
    SDF_REQUIRE(!partial || P->layout == SDFHIP_LAYOUT_SUBTREES, "sharded builds use the SUBTREES layout");
    SDF_TRY(sdfhip_mesh_ensure_bvh(mesh));
    SDF_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    AllocScope allocScope(st);       // device buffers of this call come from the stream-ordered pool
    const double tStart = nowSeconds();

    std::unique_ptr<sdfhip_octree> T(new sdfhip_octree());
    T->ctx = ctx; T->params = *P; T->params.cell_begin = cellBegin; T->params.cell_end = cellEnd;
    // cube-ify the box (OctreeSdf.cpp:43-46)
    const float sx = P->box_max[0] - P->box_min[0], sy = P->box_max[1] - P->box_min[1], sz = P->box_max[2] - P->box_min[2];
    SDF_REQUIRE(sx > 0 && sy > 0 && sz > 0 && std::isfinite(sx) && std::isfinite(sy) && std::isfinite(sz), "empty or non-finite box");
    const float maxSize = gmax(gmax(sx, sy), sz);
    const float cx = P->box_min[0] + 0.5f * sx, cy = P->box_min[1] + 0.5f * sy, cz = P->box_min[2] + 0.5f * sz;
    float bmin[3] = {cx - 0.5f * maxSize, cy - 0.5f * maxSize, cz - 0.5f * maxSize};
    float bmax[3] = {cx + 0.5f * maxSize, cy + 0.5f * maxSize, cz + 0.5f * maxSize};
    memcpy(T->info.box_min, bmin, 12); memcpy(T->info.box_max, bmax, 12);
    T->info.start_grid_size = (int32_t)G; T->info.max_depth = maxDepth;
    T->cellSize = maxSize / (float)G; T->info.start_grid_cell_size = T->cellSize;
    const uint32_t sod = startDepth < 1u ? startDepth : 1u;
    T->startOctreeDepth = sod;

    MeshDev md{meshBvh(mesh), mesh->dVerts.p, mesh->dIdx.p, mesh->dTri.p};
    DevBuf<uint32_t> stats;            // [0] valueRange bits, [1] minBorder key, [2] MFMA re-checks
    SDF_TRY(stats.reserve(3));
    { const uint32_t init[3] = {0u, 0xFFFFFFFFu, 0u}; SDF_HIP_CHECK(hipMemcpyAsync(stats.p, init, 12, hipMemcpyHostToDevice, st)); }
    DevBuf<unsigned char> scanTmp; size_t scanTmpBytes = 0;

    size_t stackBytes;          // LDS traversal stack of a 128-lane block: BVH depth + 2 entries per lane
    { int depth = 1; while ((1ull << (depth - 1)) < mesh->numTriangles) depth++; stackBytes = (size_t)(depth + 2) * 128 * sizeof(uint32_t); }

    // root level
    T->levels.resize(maxDepth - sod + 1);
    {
        std::unique_ptr<BuildLevel> L(new BuildLevel());
        L->depth = sod; L->n = 1u << (3 * sod);
        const float newSize = (float)(0.5f * (bmax[0] - bmin[0]) * std::pow(0.5f, sod));
        L->half = newSize;
        SDF_TRY(allocLevelCommon(*L));
        std::vector<float> hc(3 * L->n); std::vector<uint32_t> hco(L->n);
        const float scx = bmin[0] + newSize, scy = bmin[1] + newSize, scz = bmin[2] + newSize;
        const uint32_t vpa = 1u << sod;
        for (uint32_t k = 0; k < vpa; k++) for (uint32_t j = 0; j < vpa; j++) for (uint32_t i = 0; i < vpa; i++) {
            const uint32_t r = i + vpa * j + vpa * vpa * k;
            hc[3 * r] = scx + ((float)i * 2.0f) * newSize; hc[3 * r + 1] = scy + ((float)j * 2.0f) * newSize; hc[3 * r + 2] = scz + ((float)k * 2.0f) * newSize;
            hco[r] = i | (j << 10) | (k << 20);
        }
        SDF_HIP_CHECK(hipMemcpyAsync(L->center.p, hc.data(), hc.size() * 4, hipMemcpyHostToDevice, st));
        SDF_HIP_CHECK(hipMemcpyAsync(L->coord.p, hco.data(), hco.size() * 4, hipMemcpyHostToDevice, st));
        SDF_HIP_CHECK(hipStreamSynchronize(st));   // hc/hco go out of scope
        T->levels[0] = std::move(L);
    }

    double tSamples = 0, tDecide = 0;
    SampleScratch SS;
    SS.near = &ctx->nearScratch; ctx->nearScratch.counterReady = false;      // (builds on one context are serialised by its buildLock)
    std::unique_ptr<BuildLevel> spec[2];            // complete levels startDepth+1 and +2, sampled speculatively (may stay empty)
    const uint64_t SPEC_SAMPLE_LIMIT = 1500000;
    {   // Levels down to the start depth exist a priori: create their geometry now and take ALL their samples (the 8 corners of the
        // root level, 19 mid-points per node of every level) in one deduplicated batch instead of one latency-bound launch each.
        const double t0 = nowSeconds();
        SampleBatch B;
        BuildLevel* R0 = T->levels[0].get();
        B.add(R0->center.p, R0->coord.p, R0->half, R0->n, 8, R0->corner.p, 4);
        T->info.num_samples += 8ull * R0->n;
        for (uint32_t d = sod; d <= startDepth && d <= maxDepth; d++) {
            BuildLevel* L = T->levels[d - sod].get();
            if (d < maxDepth) {
                SDF_TRY(L->mid.reserve(76ull * L->n));
                B.add(L->center.p, L->coord.p, L->half, L->n, 19, L->mid.p, 4);
                T->info.num_samples += 19ull * L->n;
                L->presampled = true;
            }
            if (d < startDepth && d < maxDepth) {
                std::unique_ptr<BuildLevel> N(new BuildLevel());
                N->depth = d + 1; N->n = 8u * L->n; N->half = 0.5f * L->half;
                SDF_TRY(allocLevelCommon(*N));
                k_expand_geometry<<<gridFor(8ull * L->n, 256), 256, 0, st>>>(L->center.p, L->coord.p, L->half, L->n, N->center.p, N->coord.p);
                T->levels[d + 1 - sod] = std::move(N);
            }
        }
        // Speculation: the next two levels are sampled as COMPLETE levels too while that costs no extra latency (few, long
        // traversals: the batch is latency bound up to ~1e6 samples); the nodes that really exist pick their samples out of them
        // (k_gather_spec_mids) and two more dependent launches disappear from the critical path.  Identical values: a sample
        // depends only on its position, and the complete level computes the node centres with the same operations.
        std::unique_ptr<BuildLevel> specBase;          // this build's start cells, in cell order
        if (startDepth + 1 < maxDepth && T->levels[startDepth - sod]) {
            const BuildLevel* full = T->levels[startDepth - sod].get();
            specBase.reset(new BuildLevel());
            specBase->depth = startDepth; specBase->n = cellEnd - cellBegin; specBase->half = full->half;
            SDF_TRY(specBase->center.reserve(3ull * specBase->n)); SDF_TRY(specBase->coord.reserve(specBase->n));
            k_spec_base<<<gridFor(full->n, 256), 256, 0, st>>>(full->center.p, full->coord.p, full->n, G, cellBegin, cellEnd, specBase->center.p, specBase->coord.p);
        }
        for (uint32_t d = startDepth + 1; d <= startDepth + 2 && d < maxDepth; d++) {
            const BuildLevel* prev = (d == startDepth + 1) ? specBase.get() : spec[d - startDepth - 2].get();
            if (!prev || 19ull * 8ull * prev->n > SPEC_SAMPLE_LIMIT) break;
            std::unique_ptr<BuildLevel> N(new BuildLevel());
            N->depth = d; N->n = 8u * prev->n; N->half = 0.5f * prev->half;
            SDF_TRY(N->center.reserve(3ull * N->n)); SDF_TRY(N->coord.reserve(N->n)); SDF_TRY(N->mid.reserve(76ull * N->n));
            k_expand_geometry<<<gridFor(8ull * prev->n, 256), 256, 0, st>>>(prev->center.p, prev->coord.p, prev->half, prev->n, N->center.p, N->coord.p);
            B.add(N->center.p, N->coord.p, N->half, N->n, 19, N->mid.p, 4);
            spec[d - startDepth - 1] = std::move(N);
        }
        SDF_TRY(sampleBatch(st, md, B, SS, stackBytes, T->info.num_traversals));
        SDF_HIP_CHECK(hipStreamSynchronize(st));
        tSamples += nowSeconds() - t0;
    }
    for (uint32_t d = sod; d <= maxDepth; d++) {
        BuildLevel* L = T->levels[d - sod].get();
        if (!L || L->n == 0) break;
        if (d == startDepth) {
            // bring the level into cell order and restrict it to this shard's cells
            std::unique_ptr<BuildLevel> R(new BuildLevel());
            R->depth = d; R->n = cellEnd - cellBegin; R->half = L->half; R->presampled = L->presampled;
            SDF_TRY(allocLevelCommon(*R));
            if (L->presampled) SDF_TRY(R->mid.reserve(76ull * R->n));
            k_to_cell_order<<<gridFor(8ull * L->n, 256), 256, 0, st>>>(L->center.p, L->coord.p, L->corner.p, L->n, G, cellBegin, cellEnd, R->center.p, R->coord.p, R->corner.p,
                                                                        L->presampled ? L->mid.p : nullptr, L->presampled ? R->mid.p : nullptr);
            SDF_HIP_CHECK(hipStreamSynchronize(st));      // the unordered level is released by the assignment below
            T->levels[d - sod] = std::move(R);
            L = T->levels[d - sod].get();
        }
        SDF_TRY(L->flag.reserve(L->n)); SDF_TRY(L->inner.reserve(L->n)); SDF_TRY(L->childBase.reserve(L->n));
        if (d < maxDepth && !L->presampled) {
            SDF_TRY(L->mid.reserve(76ull * L->n));
            const double t0 = nowSeconds();
            SDF_TRY(sampleMidPoints(st, md, L->coord.p, L->center.p, L->half, L->n, L->mid.p, 4, SS, stackBytes, T->info.num_traversals));
            SDF_HIP_CHECK(hipStreamSynchronize(st));
            tSamples += nowSeconds() - t0;
            T->info.num_samples += 19ull * L->n;
        }
        if (d >= startDepth) {
            SDF_TRY(L->coeff.reserve(64ull * L->n));
            DecideArgs a{L->corner.p, L->mid.p, L->coord.p, L->n, d, maxDepth, L->half, P->rule, P->rule_params[0] * P->rule_params[0], P->rule_params[1],
                         L->flag.p, L->inner.p, L->coeff.p, stats.p, stats.p + 1, stats.p + 2};
            const double t0 = nowSeconds();
            if (P->fit_mode == SDFHIP_FIT_MFMA) {
                k_fit_mfma<4><<<gridFor(L->n, 128), 256, 0, st>>>(L->corner.p, nullptr, 2.0f * L->half, L->n, L->coeff.p);
                k_decide<true><<<gridFor(L->n, 128), 128, 0, st>>>(a);
            } else k_decide<false><<<gridFor(L->n, 128), 128, 0, st>>>(a);
            if (d < maxDepth) {
                size_t need = 0;
                SDF_HIP_CHECK(devExclusiveSum(nullptr, need, L->inner.p, L->childBase.p, (size_t)L->n, st));
                if (need > scanTmpBytes) { SDF_TRY(scanTmp.reserve(need)); scanTmpBytes = need; }
                SDF_HIP_CHECK(devExclusiveSum(scanTmp.p, need, L->inner.p, L->childBase.p, (size_t)L->n, st));
                SDF_TRY(readBackWords(st, L->childBase.p + (L->n - 1), L->inner.p + (L->n - 1), 1, &L->numInner));
                k_scale8<<<gridFor(L->n, 256), 256, 0, st>>>(L->n, L->childBase.p);
            } else { SDF_HIP_CHECK(hipMemsetAsync(L->childBase.p, 0, 4ull * L->n, st)); SDF_HIP_CHECK(hipStreamSynchronize(st)); L->numInner = 0; }      // (no inner node before any node: k_emit ranks the leaves by it)
            tDecide += nowSeconds() - t0;
            L->numLeaves = L->n - L->numInner;
        } else {
            k_fill_all_inner<<<gridFor(L->n, 256), 256, 0, st>>>(L->n, L->flag.p, L->inner.p, L->childBase.p);
            L->numInner = L->n; L->numLeaves = 0;
        }
        SDF_HIP_CHECK(hipGetLastError());
        if (d >= startDepth) { T->info.num_nodes += L->n; T->info.num_leaves += L->numLeaves; T->info.leaves_per_depth[d] = L->numLeaves; }
        if (d < maxDepth && L->numInner > 0) {
            SDF_REQUIRE(L->numInner <= (1u << 24), "level too large");
            const bool precreated = T->levels[d + 1 - sod] != nullptr;      // levels down to the start depth already have their geometry (same values)
            std::unique_ptr<BuildLevel> fresh;
            if (!precreated) {
                fresh.reset(new BuildLevel());
                fresh->depth = d + 1; fresh->n = 8u * L->numInner; fresh->half = 0.5f * L->half;
                SDF_TRY(allocLevelCommon(*fresh));
            }
            BuildLevel* N = precreated ? T->levels[d + 1 - sod].get() : fresh.get();
            SDF_REQUIRE(N->n == 8u * L->numInner, "internal: pre-created level has the wrong size");
            ScatterArgs sa{L->center.p, L->coord.p, L->corner.p, L->mid.p, L->inner.p, L->childBase.p, L->n, L->half, N->center.p, N->coord.p, N->corner.p};
            k_scatter_children<<<gridFor(64ull * L->n, 256), 256, 0, st>>>(sa);
            SDF_HIP_CHECK(hipGetLastError());
            if (!precreated) {
                const uint32_t si = d + 1 - startDepth - 1;      // 0 or 1 for the two speculative levels
                if (d + 1 > startDepth && si < 2 && spec[si] && d + 1 < maxDepth) {
                    SDF_TRY(fresh->mid.reserve(76ull * fresh->n));
                    k_gather_spec_mids<<<gridFor(19ull * fresh->n, 256), 256, 0, st>>>(fresh->coord.p, fresh->n, d + 1, startDepth, G, cellBegin, spec[si]->mid.p, fresh->mid.p);
                    SDF_HIP_CHECK(hipGetLastError());
                    fresh->presampled = true;
                    T->info.num_samples += 19ull * fresh->n;
                }
                T->levels[d + 1 - sod] = std::move(fresh);
            }
        }
        // this level's mid-points are no longer needed once the children exist
        SDF_HIP_CHECK(hipStreamSynchronize(st));
        L->mid.release();
        if (d < startDepth) { L->corner.release(); }
    }

    // subtree sizes, bottom-up
    uint64_t bodyWords = 0;
    {
        const uint32_t* nextAlloc = nullptr;
        uint64_t total = 0;
        for (int d = (int)maxDepth; d >= (int)startDepth; d--) {
            BuildLevel* L = T->levels[d - sod].get();
            if (!L || L->n == 0) continue;
            total += 64ull * L->numLeaves + 8ull * L->numInner;
            SDF_TRY(L->alloc.reserve(L->n)); SDF_TRY(L->pos.reserve(L->n)); SDF_TRY(L->blk.reserve(L->n));
            k_alloc<<<gridFor(L->n, 256), 256, 0, st>>>(L->n, L->flag.p, L->childBase.p, nextAlloc, L->alloc.p);
            nextAlloc = L->alloc.p;
        }
        bodyWords = total;
        SDF_HIP_CHECK(hipGetLastError());
    }
    if ((uint64_t)G3 + bodyWords > (uint64_t)INDEX_MASK) { setError("octree needs %llu words: exceeds the 30-bit node index of the reference layout", (unsigned long long)(G3 + bodyWords)); return SDFHIP_E_TOO_LARGE; }

    uint32_t stat[3];
    SDF_HIP_CHECK(hipMemcpyAsync(stat, stats.p, 12, hipMemcpyDeviceToHost, st));
    SDF_HIP_CHECK(hipStreamSynchronize(st));
    memcpy(&T->info.value_range, &stat[0], 4);
    T->info.min_border_value = (stat[1] == 0xFFFFFFFFu) ? INFINITY : floatFromOrderKey(stat[1]);
    T->info.cell_begin = cellBegin; T->info.cell_end = cellEnd;
    T->info.fit_rechecks = stat[2];
    if (getenv("SDFHIP_TIMING")) fprintf(stderr, "[sdfhip] octree build: hipMalloc/hipFree so far on this thread: %ld calls, %.4f s\n", g_allocCalls(), g_allocSeconds());
    T->info.body_words = bodyWords;
    T->info.body_offset = G3;     // provisional (single shard); emit_shard overrides it
    T->info.num_words = partial ? 0 : (uint64_t)G3 + bodyWords;
    T->info.seconds_samples = tSamples; T->info.seconds_decide = tDecide;
    T->built = true;

    if (!shardOnly) {
        // A complete build is born with its QUERY layout (node words packed breadth-first + 256-byte-aligned coefficient blocks) and without
        // a resident copy of the reference's array: the first query costs nothing extra (round 4: 7.2 ms of re-walking the array at C2, and
        // both copies resident), and download / device_words / .bin rebuild the array from the layout bit for bit (k_ql_restore).
        SDF_TRY(emitLevels(T.get(), G3, nullptr, nullptr, true));
        T->hasData = true;
        T->levels.clear();
    }
    SDF_TRY(sampleFallbacks(st, SS, T->info));
    T->info.seconds_total = nowSeconds() - tStart;
    *out = T.release();
    return SDFHIP_OK;
}

}  // namespace sdfhip

using namespace sdfhip;

extern "C" {

int sdfhip_octree_build(sdfhip_ctx* ctx, sdfhip_mesh* mesh, const sdfhip_octree_params* params, sdfhip_octree** out) {
    SDF_API_BEGIN
    SDF_REQUIRE(params != nullptr, "params is NULL");
    SDF_REQUIRE(params->cell_begin == 0 && (params->cell_end == 0 || params->cell_end == (1u << (3 * params->start_depth))), "sdfhip_octree_build builds all cells; use sdfhip_octree_build_shard");
    return buildImpl(ctx, mesh, params, false, out);
    SDF_API_END
}

int sdfhip_octree_build_shard(sdfhip_ctx* ctx, sdfhip_mesh* mesh, const sdfhip_octree_params* params, sdfhip_octree** out) {
    SDF_API_BEGIN
    return buildImpl(ctx, mesh, params, true, out);
    SDF_API_END
}

int sdfhip_octree_emit_shard(sdfhip_octree* T, uint64_t body_offset, uint32_t* dst_grid, uint32_t* dst_body, int where) {
    SDF_API_BEGIN
    SDF_REQUIRE(T && dst_grid && dst_body, "NULL argument");
    SDF_REQUIRE(T->built && !T->levels.empty(), "construction state is no longer available");
    sdfhip_ctx* ctx = T->ctx;
    SDF_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    AllocScope allocScope(st);       // device buffers of this call come from the stream-ordered pool
    const uint32_t startDepth = T->params.start_depth;
    const uint32_t G = 1u << startDepth, G3 = G * G * G;
    const uint32_t nCells = T->info.cell_end - T->info.cell_begin;
    SDF_REQUIRE(body_offset >= G3 && body_offset + T->info.body_words <= (uint64_t)INDEX_MASK, "body_offset out of range");
    DevBuf<uint32_t> tmpGrid, tmpBody;
    uint32_t* dGrid = dst_grid; uint32_t* dBody = dst_body;
    if (where == SDFHIP_HOST) {
        SDF_TRY(tmpGrid.reserve(nCells)); SDF_TRY(tmpBody.reserve(T->info.body_words));
        dGrid = tmpGrid.p; dBody = tmpBody.p;
    }
    SDF_TRY(emitLevels(T, body_offset, dGrid, dBody, false));
    if (where == SDFHIP_HOST) {
        SDF_HIP_CHECK(hipMemcpyAsync(dst_grid, dGrid, 4ull * nCells, hipMemcpyDeviceToHost, st));
        SDF_HIP_CHECK(hipMemcpyAsync(dst_body, dBody, 4ull * T->info.body_words, hipMemcpyDeviceToHost, st));
    }
    SDF_HIP_CHECK(hipStreamSynchronize(st));
    T->info.body_offset = body_offset;
    return SDFHIP_OK;
    SDF_API_END
}

int sdfhip_octree_from_data(sdfhip_ctx* ctx, const uint32_t* words, uint64_t num_words, int where, const float box_min[3], const float box_max[3],
                            int32_t start_grid_size, uint32_t max_depth, float value_range, float min_border_value, sdfhip_octree** out) {
    SDF_API_BEGIN
    SDF_REQUIRE(ctx && words && box_min && box_max && out, "NULL argument");
    SDF_REQUIRE(start_grid_size >= 1 && (uint64_t)start_grid_size * start_grid_size * start_grid_size <= num_words, "start grid does not fit");
    SDF_HIP_CHECK(hipSetDevice(ctx->device));
    std::unique_ptr<sdfhip_octree> T(new sdfhip_octree());
    T->ctx = ctx;
    memcpy(T->info.box_min, box_min, 12); memcpy(T->info.box_max, box_max, 12);
    T->info.start_grid_size = start_grid_size; T->info.max_depth = max_depth;
    T->info.value_range = value_range; T->info.min_border_value = min_border_value; T->info.num_words = num_words;
    T->cellSize = (box_max[0] - box_min[0]) / (float)start_grid_size;       // load(): mBox.getSize().x / mStartGridSize (OctreeSdf.h:232)
    T->info.start_grid_cell_size = T->cellSize;
    SDF_TRY(T->data.reserve(num_words));
    SDF_HIP_CHECK(hipMemcpyAsync(T->data.p, words, 4ull * num_words, where == SDFHIP_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, ctx->stream));
    SDF_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    T->hasData = true;
    *out = T.release();
    return SDFHIP_OK;
    SDF_API_END
}

int sdfhip_octree_set_start_grid_cell_size(sdfhip_octree* tree, float cell_size) {
    SDF_API_BEGIN
    SDF_REQUIRE(tree && cell_size > 0.f, "bad argument");
    tree->cellSize = cell_size; tree->info.start_grid_cell_size = cell_size;
    return SDFHIP_OK;
    SDF_API_END
}

int sdfhip_octree_destroy(sdfhip_octree* tree) { delete tree; return SDFHIP_OK; }

int sdfhip_octree_get_info(sdfhip_octree* tree, sdfhip_octree_info* out) {
    SDF_API_BEGIN
    SDF_REQUIRE(tree && out, "NULL argument");
    *out = tree->info;
    return SDFHIP_OK;
    SDF_API_END
}

int sdfhip_octree_download(sdfhip_octree* tree, uint32_t* out_words, int where) {
    SDF_API_BEGIN
    SDF_REQUIRE(tree && out_words, "NULL argument");
    SDF_REQUIRE(tree->hasData, "tree has no assembled node array (sharded build: emit + from_data first)");
    SDF_HIP_CHECK(hipSetDevice(tree->ctx->device));
    return octreeDownload(tree, out_words, where);
    SDF_API_END
}

const uint32_t* sdfhip_octree_device_words(sdfhip_octree* tree) {
    if (!tree || !tree->hasData) return nullptr;
    if (hipSetDevice(tree->ctx->device) != hipSuccess || octreeMaterialize(tree) != SDFHIP_OK) return nullptr;       // (takes the tree's lock; a no-op when the array is resident)
    std::lock_guard<std::mutex> own(tree->qLock);
    if (!tree->data.p) return nullptr;                  // compacted by another thread in between
    tree->dataPinned = true;                            // no automatic release behind the caller's back from now on
    return tree->data.p;
}

}  // extern "C"

// (sdfhip_ctx_create: the runtime loads a translation unit's code object on the first use of one of its kernels — milliseconds that would
// otherwise land in the first build or the first query of a process)
namespace sdfhip { void loadKernelsOctreeBuild() { hipFuncAttributes a; (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_scale8)); (void)hipGetLastError(); } }
