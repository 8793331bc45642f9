// Exact distance samples shared by the NO_CONTINUITY and CONTINUITY builders (device + host helpers).  PRODUCT code.
// Included by octree_build.hip and octree_continuity.hip; everything here has internal linkage.
#pragma once
#include "dev_bvh_fast.h"
#include "dev_tricubic.h"
#include "dev_prims.h"

namespace sdfhip {
namespace {

struct MeshDev { BvhDev bvh; const float* verts; const uint32_t* idx; const float* td; };

SDF_DEV F3 cornerRel(uint32_t c) { return F3{(c & 1u) ? 1.f : -1.f, (c & 2u) ? 1.f : -1.f, (c & 4u) ? 1.f : -1.f}; }

SDF_DEV void valuesAt(const MeshDev& m, F3 p, uint32_t t, float* __restrict__ out4) {
    const uint32_t a = m.idx[3 * t], b = m.idx[3 * t + 1], c = m.idx[3 * t + 2];
    F3 g;
    const float d = signedDistPointTriangleGrad(p, m.td + (size_t)TD_FLOATS * t,
                                                F3{m.verts[3 * a], m.verts[3 * a + 1], m.verts[3 * a + 2]},
                                                F3{m.verts[3 * b], m.verts[3 * b + 1], m.verts[3 * b + 2]},
                                                F3{m.verts[3 * c], m.verts[3 * c + 1], m.verts[3 * c + 2]}, g);
    *reinterpret_cast<float4*>(out4) = make_float4(d, g.x, g.y, g.z);
}

// Geometry of ALL children of a level (levels above the start depth subdivide unconditionally): child c of node i is 8 i + c.
__global__ void k_expand_geometry(const float* __restrict__ center, const uint32_t* __restrict__ coord, float half, uint32_t n,
                                  float* __restrict__ ncenter, uint32_t* __restrict__ ncoord) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t i = gid >> 3, c = gid & 7u;
    if (i >= n) return;
    const float ns = 0.5f * half;
    ncenter[3 * (size_t)gid] = center[3 * (size_t)i] + ((c & 1u) ? ns : -ns);
    ncenter[3 * (size_t)gid + 1] = center[3 * (size_t)i + 1] + ((c & 2u) ? ns : -ns);
    ncenter[3 * (size_t)gid + 2] = center[3 * (size_t)i + 2] + ((c & 4u) ? ns : -ns);
    const uint32_t co = coord[i];
    const uint32_t x = 2u * (co & 1023u) + (c & 1u), y = 2u * ((co >> 10) & 1023u) + ((c >> 1) & 1u), z = 2u * (co >> 20) + (c >> 2);
    ncoord[gid] = x | (y << 10) | (z << 20);
}

// ---- exact samples of whole levels: the build's hot path -------------------------------------------------------------
// Neighbouring nodes share mid-points (a face centre belongs to 2 nodes, an edge mid-point to up to 4), and the reference
// answers each of them with the same deterministic query.  The samples of a batch are therefore grouped by lattice point
// (radix sort of a 39-bit key), one traversal is run per group of samples whose fp32 POSITION BITS are identical — the
// positions come from different node centres, and only equal bits guarantee the same answer — and every sample then
// computes its Hermite datum from the shared nearest-triangle id.  About 1.6x fewer traversals, issued in lattice order.
// A batch is up to MAX_SEGS segments = (level, kind) pairs: the 19 mid-points or the 8 corners of every node of a level.
// Levels whose nodes are known a priori (everything above the start depth subdivides) go into ONE batch: their traversals
// are few and long (points far from the surface), i.e. latency bound, and gain nothing from being launched one after another.
constexpr int MAX_SEGS = 8;
struct SampleSeg { const float* center; const uint32_t* coord; float* out; float half; uint32_t n, begin; int points, stride; };
struct SampleBatch {
    SampleSeg seg[MAX_SEGS]; int count = 0; uint32_t total = 0;
    void add(const float* center, const uint32_t* coord, float half, uint32_t n, int points, float* out, int stride) {
        seg[count] = SampleSeg{center, coord, out, half, n, total, points, stride}; count++;
        const uint64_t t = (uint64_t)total + (uint64_t)points * n;
        total = t < 0xFFFFFFFFull ? (uint32_t)t : 0xFFFFFFFFu;      // saturates; sampleBatch rejects anything >= 2^31
    }
};
struct SampleRef { int seg; uint32_t node, k; };
SDF_DEV SampleRef sampleRef(const SampleBatch& B, uint32_t q) {
    int s = 0;
    for (int i = 1; i < B.count; i++) s = (q >= B.seg[i].begin) ? i : s;
    const uint32_t local = q - B.seg[s].begin, pts = (uint32_t)B.seg[s].points;
    return SampleRef{s, local / pts, local % pts};
}
SDF_DEV F3 sampleRel(const SampleBatch& B, const SampleRef& r) { return B.seg[r.seg].points == 8 ? cornerRel(r.k) : midRel((int)r.k); }
SDF_DEV F3 samplePosition(const SampleBatch& B, uint32_t q) {
    const SampleRef r = sampleRef(B, q);
    const float* c = B.seg[r.seg].center + 3 * (size_t)r.node;
    return F3{c[0], c[1], c[2]} + sampleRel(B, r) * B.seg[r.seg].half;
}
// 12-bit coordinate -> every third bit (Morton / Z-order): the sorted order then walks the lattice in small cubes, so the 64
// traversals of a wave start from neighbouring points and visit largely the same BVH nodes
SDF_DEV uint64_t spreadBits3(uint32_t v) {
    uint64_t r = 0;
#pragma unroll
    for (int b = 0; b < 12; b++) r |= (uint64_t)((v >> b) & 1u) << (3 * b);
    return r;
}
__global__ void k_sample_keys(SampleBatch B, uint64_t* __restrict__ key, uint32_t* __restrict__ val) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= B.total) return;
    const SampleRef r = sampleRef(B, q);
    const uint32_t co = B.seg[r.seg].coord[r.node];
    const F3 rel = sampleRel(B, r);
    const uint32_t lx = 2u * (co & 1023u) + (uint32_t)(rel.x + 1.f), ly = 2u * ((co >> 10) & 1023u) + (uint32_t)(rel.y + 1.f), lz = 2u * (co >> 20) + (uint32_t)(rel.z + 1.f);
    key[q] = spreadBits3(lx) | (spreadBits3(ly) << 1) | (spreadBits3(lz) << 2) | ((uint64_t)r.seg << 36);
    val[q] = q;
}
// sorted entry j starts a new traversal unless it is the same lattice point AND the same position bits as entry j-1
__global__ void k_sample_mark(SampleBatch B, const uint64_t* __restrict__ key, const uint32_t* __restrict__ val, uint32_t* __restrict__ isRep) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= B.total) return;
    bool rep = true;
    if (j > 0 && key[j] == key[j - 1]) {
        const F3 a = samplePosition(B, val[j]), b = samplePosition(B, val[j - 1]);
        rep = !(__float_as_uint(a.x) == __float_as_uint(b.x) && __float_as_uint(a.y) == __float_as_uint(b.y) && __float_as_uint(a.z) == __float_as_uint(b.z));
    }
    isRep[j] = rep ? 1u : 0u;
}
__global__ void k_sample_rep_list(SampleBatch B, const uint32_t* __restrict__ isRep, const uint32_t* __restrict__ scan, const uint32_t* __restrict__ val, uint32_t total,
                                  uint32_t* __restrict__ repSample, float* __restrict__ repPos) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < total && isRep[j]) {
        const uint32_t slot = scan[j];
        repSample[slot] = val[j];
        const F3 p = samplePosition(B, val[j]);
        repPos[3 * (size_t)slot] = p.x; repPos[3 * (size_t)slot + 1] = p.y; repPos[3 * (size_t)slot + 2] = p.z;
    }
}
// With `world` > 1 this launch covers only the 128-representative blocks b with b % world == rank (dealt round-robin so that
// every rank gets the same mix of short and long traversals); the other blocks' ids arrive through the exchange.
template <int BLOCK>
__global__ void __launch_bounds__(BLOCK) k_sample_nearest(BvhDev b, SampleBatch B, const uint32_t* __restrict__ repSample, uint32_t numReps, uint32_t* __restrict__ repTri,
                                                          uint32_t rank, uint32_t world) {
    extern __shared__ uint32_t s_stack[];        // [stackDepth][BLOCK], stackDepth = BVH depth + 2 (smaller stack -> more waves per CU)
    const uint32_t lb = xcdLogicalBlock();            // the representatives are in Morton order: a contiguous eighth of them per XCD
    const uint64_t r = ((uint64_t)lb * world + rank) * blockDim.x + threadIdx.x;
    if (r >= numReps) return;
    repTri[r] = bvhNearest<BLOCK>(b, samplePosition(B, repSample[r]), s_stack + threadIdx.x);
}
// stride = floats per sample in the segment's output: 4 ([f, fx, fy, fz]) or 8 (the CONTINUITY builder's Hermite slots, mixed derivatives 0)
__global__ void k_sample_values(MeshDev m, SampleBatch B, const uint32_t* __restrict__ val, const uint32_t* __restrict__ isRep, const uint32_t* __restrict__ scan,
                                const uint32_t* __restrict__ repTri) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= B.total) return;
    const uint32_t q = val[j];
    const uint32_t slot = scan[j] + isRep[j] - 1u;          // exclusive scan: the group's representative is the last flagged entry at or before j
    const SampleRef r = sampleRef(B, q);
    const SampleSeg& S = B.seg[r.seg];
    float* out = S.out + (size_t)S.stride * ((size_t)S.points * r.node + r.k);
    valuesAt(m, samplePosition(B, q), repTri[slot], out);
    if (S.stride == 8) *reinterpret_cast<float4*>(out + 4) = make_float4(0.f, 0.f, 0.f, 0.f);
}

// scratch of the sampler, reused across levels (grows only)
struct SampleScratch {
    DevBuf<uint64_t> key, keyS; DevBuf<uint32_t> val, valS, isRep, scan, repSample, repTri; DevBuf<float> repPos; DevBuf<unsigned char> tmp; size_t tmpBytes = 0;
    NearScratch* near = nullptr;                    // the context's scratch of the two-phase nearest search (dev_bvh_fast.h); set by the builders
    const sdfhip_exchange* exchange = nullptr;      // set by the CONTINUITY builder when the context has one (world > 1)
    bool pending = false; SampleBatch pendingBatch; uint32_t pendingReps = 0; uint32_t* pendingTri = nullptr;
};
// A batch runs in two halves so that a caller can put host work between them: Begin enqueues everything up to the traversals,
// End (the exchange, if any, and) the per-sample Hermite data.  One batch may be pending per scratch.
static int sampleBatchBegin(hipStream_t st, const MeshDev& md, const SampleBatch& B, SampleScratch& S, size_t stackBytes, uint64_t& traversals) {
    const uint32_t total = B.total;
    SDF_REQUIRE(!S.pending, "internal: sample batch begun while another is pending");
    if (total == 0) return SDFHIP_OK;
    SDF_REQUIRE(total < (1u << 31), "level too large for one sample batch (2^31 samples)");
    SDF_TRY(S.key.reserve(total)); SDF_TRY(S.keyS.reserve(total)); SDF_TRY(S.val.reserve(total)); SDF_TRY(S.valS.reserve(total));
    SDF_TRY(S.isRep.reserve(total)); SDF_TRY(S.scan.reserve(total));
    k_sample_keys<<<gridFor(total, 256), 256, 0, st>>>(B, S.key.p, S.val.p);
    size_t b1 = 0, b2 = 0;
    SDF_HIP_CHECK(devSortPairs(nullptr, b1, S.key.p, S.keyS.p, S.val.p, S.valS.p, (size_t)total, 0, (unsigned)39, st));
    SDF_HIP_CHECK(devExclusiveSum(nullptr, b2, S.isRep.p, S.scan.p, (size_t)total, st));
    const size_t need = b1 > b2 ? b1 : b2;
    if (need > S.tmpBytes) { SDF_TRY(S.tmp.reserve(need)); S.tmpBytes = need; }
    SDF_HIP_CHECK(devSortPairs(S.tmp.p, b1, S.key.p, S.keyS.p, S.val.p, S.valS.p, (size_t)total, 0, (unsigned)39, st));
    k_sample_mark<<<gridFor(total, 256), 256, 0, st>>>(B, S.keyS.p, S.valS.p, S.isRep.p);
    SDF_HIP_CHECK(devExclusiveSum(S.tmp.p, b2, S.isRep.p, S.scan.p, (size_t)total, st));
    uint32_t numReps = 0;
    SDF_TRY(readBackWords(st, S.scan.p + (total - 1), S.isRep.p + (total - 1), 1, &numReps));
    SDF_TRY(S.repSample.reserve(numReps)); SDF_TRY(S.repPos.reserve(3 * (size_t)numReps));
    k_sample_rep_list<<<gridFor(total, 256), 256, 0, st>>>(B, S.isRep.p, S.scan.p, S.valS.p, total, S.repSample.p, S.repPos.p);
    const uint32_t blocks = gridFor(numReps, 128);
    const bool exactOnly = nearestExactOnly();
    const int stackDepth = (int)(stackBytes / (128 * sizeof(uint32_t)));
    if (S.exchange) {
        const uint32_t rank = (uint32_t)S.exchange->rank, world = (uint32_t)S.exchange->world;
        S.pendingTri = S.exchange->acquire(S.exchange->user, numReps);
        SDF_REQUIRE(S.pendingTri, "exchange: acquire failed");
        const uint32_t mine = blocks > rank ? (blocks - rank + world - 1) / world : 0;
        if (!exactOnly) SDF_TRY(nearestTwoPhase(st, md.bvh, S.repPos.p, numReps, S.pendingTri, *S.near, stackDepth, rank, world));
        else if (mine) k_sample_nearest<128><<<xcdGrid(mine), 128, stackBytes, st>>>(md.bvh, B, S.repSample.p, numReps, S.pendingTri, rank, world);
    } else {
        SDF_TRY(S.repTri.reserve(numReps));
        S.pendingTri = S.repTri.p;
        if (!exactOnly) SDF_TRY(nearestTwoPhase(st, md.bvh, S.repPos.p, numReps, S.pendingTri, *S.near, stackDepth, 0u, 1u));
        else k_sample_nearest<128><<<xcdGrid(blocks), 128, stackBytes, st>>>(md.bvh, B, S.repSample.p, numReps, S.pendingTri, 0u, 1u);   // 64 / 256 lanes per block measured the same
    }
    SDF_HIP_CHECK(hipGetLastError());
    S.pending = true; S.pendingBatch = B; S.pendingReps = numReps;
    traversals += numReps;
    return SDFHIP_OK;
}
static int sampleBatchEnd(hipStream_t st, const MeshDev& md, SampleScratch& S) {
    if (!S.pending) return SDFHIP_OK;
    S.pending = false;
    if (S.exchange) {
        SDF_HIP_CHECK(hipStreamSynchronize(st));
        SDF_REQUIRE(S.exchange->all_reduce_sum(S.exchange->user, S.pendingReps) == 0, "exchange: all_reduce_sum failed");
    }
    const uint32_t total = S.pendingBatch.total;
    k_sample_values<<<gridFor(total, 256), 256, 0, st>>>(md, S.pendingBatch, S.valS.p, S.isRep.p, S.scan.p, S.pendingTri);
    SDF_HIP_CHECK(hipGetLastError());
    return SDFHIP_OK;
}
static int sampleBatch(hipStream_t st, const MeshDev& md, const SampleBatch& B, SampleScratch& S, size_t stackBytes, uint64_t& traversals) {
    SDF_TRY(sampleBatchBegin(st, md, B, S, stackBytes, traversals));
    return sampleBatchEnd(st, md, S);
}
// the search's totals of this build into the tree's info (synchronises the stream)
static int sampleFallbacks(hipStream_t st, SampleScratch& S, sdfhip_octree_info& info) {
    info.num_nearest_fallbacks = 0;
    if (!S.near) return SDFHIP_OK;
    NearTotals t;
    SDF_TRY(nearTotals(st, *S.near, t));
    info.num_nearest_fallbacks = t.fallbacks; info.near_expansions = t.expansions; info.near_triangle_tests = t.triangleTests;
    info.seconds_near_candidates = t.candidateSeconds; info.seconds_near_search = t.searchSeconds;
    return SDFHIP_OK;
}
// the 19 mid-points of one level
static int sampleMidPoints(hipStream_t st, const MeshDev& md, const uint32_t* coord, const float* center, float half, uint32_t n, float* mid, int stride,
                           SampleScratch& S, size_t stackBytes, uint64_t& traversals) {
    SampleBatch B;
    B.add(center, coord, half, n, 19, mid, stride);
    return sampleBatch(st, md, B, S, stackBytes, traversals);
}

}  // namespace
}  // namespace sdfhip
