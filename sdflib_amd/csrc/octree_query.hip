// Batched OctreeSdf::getDistance on the device.  PRODUCT code — independent of oracle/.
//
// Reference behaviour reproduced: OctreeSdf::getDistance(vec3) / (vec3, vec3&) (src/sdf/OctreeSdf.cpp:93-152),
// roundFloat '>= 0.5' (:88-91), BoundingBox::getDistance for points outside the start grid
// (include/SdfLib/utils/Mesh.h:42-63, reproduced as written including the gradient overload's quirks).
//
// One lane per query: start-grid cell -> dependent 4-byte loads down the node array -> 256-byte coefficient
// block (16 x dwordx4) -> 64-term polynomial.  HBM/L2-gather bound: ~292 B of algorithmic traffic per query.
// EVAL_EXACT evaluates in the reference's literal order without FMA (bit-identical results); EVAL_FAST uses a
// separable Horner scheme with FMA.  Compile with -ffp-contract=off.
#include <sys/mman.h>
#include "octree_internal.h"
#include "dev_tricubic.h"
#include <string.h>
#include <cmath>
#include "dev_prims.h"

namespace sdfhip {

struct QueryTree {
    const uint32_t* topo;       // packed node words (sdfhip_octree::qTopo)
    const float* coef;          // 256-byte aligned coefficient blocks (sdfhip_octree::qCoef)
    float bminx, bminy, bminz, bmaxx, bmaxy, bmaxz;
    float cellSize, minBorder;
    int G;
};

SDF_HD float boxDistance(const QueryTree& t, F3 p) {
    const F3 size = F3{t.bmaxx - t.bminx, t.bmaxy - t.bminy, t.bmaxz - t.bminz};
    const F3 center = F3{t.bminx, t.bminy, t.bminz} + 0.5f * size;
    const F3 d = p - center;
    const F3 q = F3{fabsf(d.x), fabsf(d.y), fabsf(d.z)} - 0.5f * size;
    const F3 qm = F3{gmax(q.x, 0.f), gmax(q.y, 0.f), gmax(q.z, 0.f)};
    return length(qm) + gmin(gmax(q.x, gmax(q.y, q.z)), 0.0f);
}

// BoundingBox::getDistance(point, outGradient) as written in the reference (full size, uncentred point; only
// the selected component is written in the 'inside' branch).
SDF_HD float boxDistanceGrad(const QueryTree& t, F3 p, float* g) {
    const float size[3] = {t.bmaxx - t.bminx, t.bmaxy - t.bminy, t.bmaxz - t.bminz};
    const float pt[3] = {p.x, p.y, p.z};
    float a[3];
    for (int i = 0; i < 3; i++) a[i] = fabsf(pt[i]) - size[i];
    const int k = a[0] > a[1] ? 0 : 1;
    const int l = a[2] > a[k] ? 2 : k;
    if (a[l] < 0) g[l] = pt[l] / fabsf(pt[l]);
    else {
        float b[3];
        for (int i = 0; i < 3; i++) b[i] = gmax(a[i], 0.0f);
        const float c = sqrtf(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]);
        for (int i = 0; i < 3; i++) g[i] = a[i] > 0 ? b[i] / c * pt[i] / fabsf(pt[i]) : 0.0f;
    }
    return boxDistance(t, p);
}

// Walk from the start grid to the leaf holding p: index of its 64 coefficients and the local coordinates in [0,1)^3.
// Returns false for points outside the start grid.
SDF_HD bool locateLeaf(const QueryTree& t, F3 p, uint32_t& at, F3& f) {
    f = F3{(p.x - t.bminx) / t.cellSize, (p.y - t.bminy) / t.cellSize, (p.z - t.bminz) / t.cellSize};
    const float flx = floorf(f.x), fly = floorf(f.y), flz = floorf(f.z);
    const int ix = (int)flx, iy = (int)fly, iz = (int)flz;
    f = F3{f.x - flx, f.y - fly, f.z - flz};
    if (ix < 0 || ix >= t.G || iy < 0 || iy >= t.G || iz < 0 || iz >= t.G) return false;
    uint32_t w = t.topo[(iz * t.G + iy) * t.G + ix];
    while (!(w & LEAF_BIT)) {
        const uint32_t child = ((f.z >= 0.5f) ? 4u : 0u) + ((f.y >= 0.5f) ? 2u : 0u) + ((f.x >= 0.5f) ? 1u : 0u);
        w = t.topo[(w & INDEX_MASK) + child];
        f = F3{gfract(2.0f * f.x), gfract(2.0f * f.y), gfract(2.0f * f.z)};
    }
    at = w & INDEX_MASK;
    return true;
}

// the polynomial (and its gradient) of a leaf whose 64 coefficients are in c[]
template <int EVAL, bool GRAD>
SDF_HD float evalLeaf(const float (&c)[64], F3 f, float* grad) {
    auto cf = [&](int n) { return c[n]; };
    if (EVAL == SDFHIP_EVAL_EXACT) {
        if (GRAD) {
            const F3 g = normalize(F3{tricubicDerivExact<1, 0, 0>(cf, f), tricubicDerivExact<0, 1, 0>(cf, f), tricubicDerivExact<0, 0, 1>(cf, f)});
            grad[0] = g.x; grad[1] = g.y; grad[2] = g.z;
        }
        return tricubicValueExact(cf, f);
    } else {
        if (GRAD) {
            F3 g;
            const float v = tricubicValueGradFast(cf, f, g);
            g = normalize(g);
            grad[0] = g.x; grad[1] = g.y; grad[2] = g.z;
            return v;
        }
        return tricubicValueFast(cf, f);
    }
}
template <int EVAL, bool GRAD>
SDF_HD float queryOne(const QueryTree& t, F3 p, float* grad) {
    uint32_t at; F3 f;
    if (!locateLeaf(t, p, at, f)) {
        if (GRAD) return boxDistanceGrad(t, p, grad) + t.minBorder;
        return boxDistance(t, p) + t.minBorder;
    }
    float c[64];
    const float4* src = reinterpret_cast<const float4*>(t.coef + 64ull * at);        // `at` = block id: 256-byte aligned, two cache lines
#pragma unroll
    for (int q = 0; q < 16; q++) { const float4 v = src[q]; c[4 * q] = v.x; c[4 * q + 1] = v.y; c[4 * q + 2] = v.z; c[4 * q + 3] = v.w; }
    return evalLeaf<EVAL, GRAD>(c, f, grad);
}

// The batched query.  The coefficient blocks are fetched cooperatively: a lane-private 256-byte block (round 1) costs sixteen load
// instructions that each touch 64 different cache lines (one per lane), and the texture path handles a line per cycle; here sixteen
// lanes read one lane's block as ONE contiguous segment, so an instruction touches 8 lines, and the rows go through LDS to their
// owners, sixteen queries at a time (4.3 KB of LDS per wave).  Measured, 10 M queries: C2 0.446 -> 0.331 ms, the HBM-resident
// depth-9 tree 0.734 -> 0.544 ms, the bare 256-byte gather of the calibration kernel 4.9 -> 6.1 TB/s; same bits.
constexpr int QROW = 68;            // floats per LDS row: 64 + 4 of padding
template <int EVAL, bool GRAD>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) k_octree_query_coop(QueryTree t, const float* __restrict__ pts, uint64_t n, float* __restrict__ dist, float* __restrict__ grad) {
    __shared__ float s_rows[4][16 * QROW];
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, grp = lane >> 4, sub = lane & 15;
    const bool live = i < n;
    const F3 p = live ? F3{pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]} : F3{0.f, 0.f, 0.f};
    uint32_t at = 0; F3 f = F3{0.f, 0.f, 0.f};
    const bool inside = live && locateLeaf(t, p, at, f);
    const uint32_t mine = inside ? at : 0u;
    float c[64];
#pragma unroll
    for (int ch = 0; ch < 4; ch++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = 4 * r + grp;
            const uint32_t blk = __shfl(mine, 16 * ch + row);
            const float4 v = reinterpret_cast<const float4*>(t.coef + 64ull * blk)[sub];
            *reinterpret_cast<float4*>(&s_rows[w][row * QROW + 4 * sub]) = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (grp == ch) {
            const float4* rowp = reinterpret_cast<const float4*>(&s_rows[w][sub * QROW]);
#pragma unroll
            for (int q = 0; q < 16; q++) { const float4 v = rowp[q]; c[4 * q] = v.x; c[4 * q + 1] = v.y; c[4 * q + 2] = v.z; c[4 * q + 3] = v.w; }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    if (!live) return;
    float g[3] = {0.f, 0.f, 0.f};
    float d;
    if (!inside) d = (GRAD ? boxDistanceGrad(t, p, g) : boxDistance(t, p)) + t.minBorder;
    else d = evalLeaf<EVAL, GRAD>(c, f, g);
    dist[i] = d;
    if (GRAD) { grad[3 * i] = g[0]; grad[3 * i + 1] = g[1]; grad[3 * i + 2] = g[2]; }
}

template <int EVAL, bool GRAD>
__global__ void __launch_bounds__(256) k_octree_query_grid(QueryTree t, F3 origin, F3 step, uint32_t nx, uint32_t ny, uint32_t nz,
                                                           float* __restrict__ dist, float* __restrict__ grad) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;      // (an XCD-contiguous slab order was measured: slower here)
    const uint64_t n = (uint64_t)nx * ny * nz;
    if (i >= n) return;
    const uint32_t x = (uint32_t)(i % nx), y = (uint32_t)((i / nx) % ny), z = (uint32_t)(i / ((uint64_t)nx * ny));
    const F3 p = F3{origin.x + (float)x * step.x, origin.y + (float)y * step.y, origin.z + (float)z * step.z};
    float g[3] = {0.f, 0.f, 0.f};
    const float d = queryOne<EVAL, GRAD>(t, p, g);
    dist[i] = d;
    if (GRAD) { grad[3 * i] = g[0]; grad[3 * i + 1] = g[1]; grad[3 * i + 2] = g[2]; }
}

// ---- query-side layout -------------------------------------------------------------------------------------------------
// One breadth-first sweep over `data` per level.  Ranks are handed out per wave (ballot + one atomic per wave), so that the children
// of neighbouring nodes — and the coefficient blocks of neighbouring leaves — stay neighbours in the packed arrays.
SDF_DEV uint32_t waveRank(bool take, uint32_t* counter) {
    const uint64_t m = __ballot(take);
    const uint32_t lane = __lane_id();
    uint32_t base = 0;
    if (m != 0ull) {
        const int leader = __ffsll((unsigned long long)m) - 1;
        if ((int)lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(m));
        base = __shfl(base, leader);
    }
    return base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
}
// counters: [0] inner nodes of this level, [1] leaves so far (all levels), [2] malformed words
__global__ void k_ql_level(const uint32_t* __restrict__ data, uint64_t numWords, const uint32_t* __restrict__ src, uint32_t count, uint32_t* __restrict__ rank,
                           uint32_t* __restrict__ nextSrc, uint32_t* __restrict__ counters) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = j < count;
    const uint32_t w = live ? data[src ? src[j] : j] : LEAF_BIT;
    const bool leaf = (w & LEAF_BIT) != 0u;
    const uint64_t idx = w & INDEX_MASK;
    const bool bad = live && (leaf ? idx + 64u > numWords : (idx + 8u > numWords || idx == 0u));
    const uint32_t rl = waveRank(live && leaf && !bad, counters + 1);
    const uint32_t ri = waveRank(live && !leaf && !bad, counters);
    if (!live) return;
    if (bad) { atomicAdd(counters + 2, 1u); rank[j] = LEAF_BIT; return; }
    if (leaf) rank[j] = LEAF_BIT | rl;
    else {
        rank[j] = ri;
#pragma unroll
        for (uint32_t c = 0; c < 8u; c++) nextSrc[8u * ri + c] = (uint32_t)idx + c;
    }
}
__global__ void k_ql_fill(const uint32_t* __restrict__ data, const uint32_t* __restrict__ src, const uint32_t* __restrict__ rank, uint32_t count,
                          uint32_t* __restrict__ topoLevel, uint32_t* __restrict__ origLevel, uint32_t nextLevelBase, float* __restrict__ coef) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    const uint32_t r = rank[j];
    const uint32_t word = data[src ? src[j] : j];
    origLevel[j] = word;               // the node's word as the reference's array holds it: what octreeMaterialize rebuilds that array from
    if (!(r & LEAF_BIT)) { topoLevel[j] = nextLevelBase + 8u * r; return; }
    topoLevel[j] = r;
    const uint32_t* from = data + (word & INDEX_MASK);
    float4* to = reinterpret_cast<float4*>(coef + 64ull * (r & INDEX_MASK));
#pragma unroll
    for (int q = 0; q < 16; q++) to[q] = make_float4(__uint_as_float(from[4 * q]), __uint_as_float(from[4 * q + 1]), __uint_as_float(from[4 * q + 2]), __uint_as_float(from[4 * q + 3]));
}

// The layout of the tree whose reference array is `data` (device, info.num_words words; need not be T->data: the CONTINUITY builder
// converts its working array and never makes a resident copy).  Caller holds T->qLock or owns T exclusively.
int octreeLayoutFromArray(sdfhip_octree* T, const uint32_t* data) {
    sdfhip_ctx* ctx = T->ctx;
    hipStream_t st = ctx->stream;
    const uint64_t numWords = T->info.num_words;
    const uint64_t G = (uint64_t)T->info.start_grid_size, G3 = G * G * G;
    SDF_REQUIRE(G3 >= 1 && G3 <= numWords && G3 < (1ull << 30), "start grid does not fit the node array");
    struct Level { DevBuf<uint32_t> src, rank; uint32_t count = 0; };
    std::vector<std::unique_ptr<Level>> levels;
    DevBuf<uint32_t> counters;
    SDF_TRY(counters.reserve(3));
    SDF_HIP_CHECK(hipMemsetAsync(counters.p, 0, 12, st));
    std::unique_ptr<Level> cur(new Level());
    cur->count = (uint32_t)G3;                                     // level 0 = the start grid, read in place (src == nullptr)
    uint64_t total = 0;
    uint32_t h[3] = {0, 0, 0};
    std::vector<uint32_t> levelNodes, levelLeafBase(1, 0u);
    for (;;) {
        total += cur->count;
        SDF_REQUIRE(levels.size() < 32 && total <= numWords, "node array is not a valid octree (too deep or cyclic)");
        SDF_TRY(cur->rank.reserve(cur->count));
        std::unique_ptr<Level> next(new Level());
        SDF_TRY(next->src.reserve(8ull * cur->count));
        SDF_HIP_CHECK(hipMemsetAsync(counters.p, 0, 4, st));
        k_ql_level<<<gridFor(cur->count, 256), 256, 0, st>>>(data, numWords, levels.empty() ? nullptr : cur->src.p, cur->count, cur->rank.p, next->src.p, counters.p);
        SDF_TRY(readBackWords(st, counters.p, nullptr, 3, h));
        SDF_REQUIRE(h[2] == 0, "node array is not a valid octree (index out of range)");
        levelNodes.push_back(cur->count); levelLeafBase.push_back(h[1]);
        levels.push_back(std::move(cur));
        if (h[0] == 0) break;
        SDF_REQUIRE(8ull * h[0] < (1ull << 30), "tree too large for the query layout");
        next->count = 8u * h[0];
        cur = std::move(next);
    }
    SDF_REQUIRE(total < (1ull << 30), "tree too large for the query layout");
    const uint64_t leaves = h[1];
    SDF_TRY(T->qTopo.reserve(total)); SDF_TRY(T->qOrig.reserve(total)); SDF_TRY(T->qCoef.reserve(64ull * (leaves ? leaves : 1)));
    uint64_t base = 0;
    for (size_t i = 0; i < levels.size(); i++) {
        Level& L = *levels[i];
        k_ql_fill<<<gridFor(L.count, 256), 256, 0, st>>>(data, i == 0 ? nullptr : L.src.p, L.rank.p, L.count, T->qTopo.p + base, T->qOrig.p + base, (uint32_t)(base + L.count), T->qCoef.p);
        base += L.count;
    }
    SDF_HIP_CHECK(hipGetLastError());
    SDF_HIP_CHECK(hipStreamSynchronize(st));           // the level buffers are released below
    T->qNodes = total; T->qLeaves = leaves; T->qLevelNodes = levelNodes; T->qLevelLeafBase = levelLeafBase; T->qReady = true;
    return SDFHIP_OK;
}

// Trees that arrive as arrays (sdfhip_octree_from_data: loaded files, reassembled shards, broadcasts) get their layout on the first query;
// trees built here are born with it (octree_build.hip: emitQueryLayout; octree_continuity.hip) and never pass through this.
static int ensureQueryLayout(sdfhip_octree* T) {
    std::lock_guard<std::mutex> own(T->qLock);
    if (T->qReady) return SDFHIP_OK;
    SDF_REQUIRE(T->data.p, "tree has neither its node array nor a query layout");
    SDF_TRY(octreeLayoutFromArray(T, T->data.p));
    sdfhip_ctx* ctx = T->ctx;
    hipStream_t st = ctx->stream;
    const uint64_t numWords = T->info.num_words, total = T->qNodes, leaves = T->qLeaves;
    // The layout holds everything the array holds (node words + coefficient blocks), so a LARGE array need not stay on the device beside
    // it: above SDFHIP_COMPACT_ABOVE_MB (default 1024) it is released here and rebuilt on demand (octreeMaterialize).
    static const uint64_t compactAbove = (getenv("SDFHIP_COMPACT_ABOVE_MB") ? strtoull(getenv("SDFHIP_COMPACT_ABOVE_MB"), nullptr, 10) : 1024ull) << 20;
    // (not once sdfhip_octree_device_words has handed its address out; and the block goes back to the device, not to the stream's cache:
    // a query runs outside any allocation scope, so nothing else would apply the cache's high-water mark)
    if (4ull * numWords >= compactAbove && total + 64ull * leaves == numWords && !T->dataPinned) {
        T->data.release();
        BigBlockCache::get().trimTo(ctx->device, st, BigBlockCache::get().keepFor(ctx->device, st));
    }
    return SDFHIP_OK;
}

// One level of the inverse of k_ql_fill: node words back to their positions, coefficient blocks back behind their leaves' indices.
__global__ void k_ql_restore(const uint32_t* __restrict__ topoLevel, const uint32_t* __restrict__ origLevel, const uint32_t* __restrict__ posLevel, uint32_t count,
                             uint32_t nextLevelBase, uint32_t* __restrict__ posNext, const float* __restrict__ coef, uint32_t* __restrict__ data) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    const uint32_t word = origLevel[j], t = topoLevel[j];
    data[posLevel ? posLevel[j] : j] = word;
    const uint32_t idx = word & INDEX_MASK;
    if (t & LEAF_BIT) {
        const float4* from = reinterpret_cast<const float4*>(coef + 64ull * (t & INDEX_MASK));
        uint32_t* to = data + idx;
#pragma unroll
        for (int q = 0; q < 16; q++) { const float4 x = from[q]; to[4 * q] = __float_as_uint(x.x); to[4 * q + 1] = __float_as_uint(x.y); to[4 * q + 2] = __float_as_uint(x.z); to[4 * q + 3] = __float_as_uint(x.w); }
    } else {
        const uint32_t child = t - nextLevelBase;
#pragma unroll
        for (uint32_t c = 0; c < 8u; c++) posNext[child + c] = idx + c;
    }
}

// The reference's node array into `out` (num_words words on the device), from the resident copy or, for a compacted tree, from the layout.
static int octreeWordsInto(sdfhip_octree* T, uint32_t* out) {
    hipStream_t st = T->ctx->stream;
    if (T->data.p) { SDF_HIP_CHECK(hipMemcpyAsync(out, T->data.p, 4ull * T->info.num_words, hipMemcpyDeviceToDevice, st)); return SDFHIP_OK; }
    SDF_REQUIRE(T->qReady, "tree has neither its node array nor a query layout");
    DevBuf<uint32_t> posA, posB;
    uint32_t widest = 0; for (uint32_t c : T->qLevelNodes) widest = c > widest ? c : widest;
    SDF_TRY(posA.reserve(widest)); SDF_TRY(posB.reserve(widest));
    uint64_t base = 0;
    for (size_t i = 0; i < T->qLevelNodes.size(); i++) {
        const uint32_t count = T->qLevelNodes[i];
        k_ql_restore<<<gridFor(count, 256), 256, 0, st>>>(T->qTopo.p + base, T->qOrig.p + base, i == 0 ? nullptr : posA.p, count, (uint32_t)(base + count), posB.p, T->qCoef.p, out);
        std::swap(posA, posB);
        base += count;
    }
    SDF_HIP_CHECK(hipGetLastError());
    SDF_HIP_CHECK(hipStreamSynchronize(st));           // posA / posB are released below
    return SDFHIP_OK;
}

int octreeMaterialize(sdfhip_octree* T) {
    std::lock_guard<std::mutex> own(T->qLock);
    if (T->data.p) return SDFHIP_OK;
    DevBuf<uint32_t> d;
    SDF_TRY(d.reserve(T->info.num_words));
    SDF_TRY(octreeWordsInto(T, d.p));
    T->data = std::move(d);
    return SDFHIP_OK;
}

int octreeDownload(sdfhip_octree* T, uint32_t* out_words, int where) {
    hipStream_t st = T->ctx->stream;
    std::lock_guard<std::mutex> own(T->qLock);
    AllocScope allocScope(st);       // the transient array comes from the context's block cache (what every built tree's download goes through)
    if (where == SDFHIP_DEVICE) { SDF_TRY(octreeWordsInto(T, out_words)); SDF_HIP_CHECK(hipStreamSynchronize(st)); return SDFHIP_OK; }
    DevBuf<uint32_t> tmp;                              // a compacted tree is rebuilt in a transient block: its footprint stays what it was
    const uint32_t* src = T->data.p;
    if (!src) { SDF_TRY(tmp.reserve(T->info.num_words)); SDF_TRY(octreeWordsInto(T, tmp.p)); src = tmp.p; }
    SDF_HIP_CHECK(hipMemcpyAsync(out_words, src, 4ull * T->info.num_words, hipMemcpyDeviceToHost, st));
    SDF_HIP_CHECK(hipStreamSynchronize(st));
    return SDFHIP_OK;
}

// host copies of the query layout for the scalar / few-point entry: the same queryOne, compiled for the host
constexpr uint64_t kOctreeHostScalarMax = 32;
static int ensureHostLayout(sdfhip_octree* T) {
    SDF_TRY(ensureQueryLayout(T));
    std::lock_guard<std::mutex> own(T->qLock);
    if (T->hReady) return SDFHIP_OK;
    hipStream_t st = T->ctx->stream;
    T->hTopo.resize(T->qNodes); T->hCoef.resize(64 * (T->qLeaves ? T->qLeaves : 1));
    SDF_HIP_CHECK(hipMemcpyAsync(T->hTopo.data(), T->qTopo.p, 4 * T->qNodes, hipMemcpyDeviceToHost, st));
    if (T->qLeaves) SDF_HIP_CHECK(hipMemcpyAsync(T->hCoef.data(), T->qCoef.p, 256 * T->qLeaves, hipMemcpyDeviceToHost, st));
    SDF_HIP_CHECK(hipStreamSynchronize(st));
    T->hReady = true;
    return SDFHIP_OK;
}

// ---- leaf-driven lattice evaluation ------------------------------------------------------------------------------------
// A lattice query in the point kernel pays, per point, three IEEE divisions, the walk and sixteen coefficient loads before its ~190
// flop of separable Horner: 300 instructions, a third of them spent finding out what the neighbouring points found out too.  Here the
// LEAVES are streamed and each emits the lattice points it contains:
//   * the walk's first value f0(i) = ((origin + i * step) - boxMin) / cellSize depends on the axis index alone: three tables of
//     n floats (k_lat_tables) replace 3 n^3 divisions.  f0 is monotone in i (every operation is), so the indices a leaf [a, b) owns on
//     an axis are the range [lower_bound(a), lower_bound(b)) of that table (k_lat_ranges) — by construction the leaf the point walk
//     reaches — and the walk's final local coordinate, fract(2 fract(2 ... fract(f0))), equals fract(2^l f0): every step is exact;
//   * one lane takes one (x, y) COLUMN of a leaf's points: the x- and y-contractions of tricubicValueGradFast (176 of its 192 flop)
//     do not depend on z and are done once, each point of the column then costs 16 FMAs — in the very order of the point kernel's
//     EVAL_FAST code, so both paths give identical bits;
//   * leaves of one level have (nearly) the same number of points: a wave only holds leaves of one level, so its lanes stay in step.
// Points outside the start grid (box distance) are written by k_lattice_outside.  The plan (tables, ranges, per-level launch shapes)
// is kept with the tree for the next call with the same lattice.  EVAL_EXACT lattices use the same plan (k_lattice_columns_exact).
__global__ void k_lc_root(uint32_t G, uint32_t count, uint32_t* __restrict__ cell) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    const uint32_t x = j % G, y = (j / G) % G, z = j / (G * G);
    cell[2 * j] = x | (y << 16); cell[2 * j + 1] = z;
}
__global__ void k_lc_level(const uint32_t* __restrict__ topoLevel, uint32_t count, uint32_t nextLevelBase, const uint32_t* __restrict__ cell, uint32_t* __restrict__ nextCell,
                           uint32_t* __restrict__ leafCell) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    const uint32_t w = topoLevel[j], xy = cell[2 * j], z = cell[2 * j + 1];
    if (w & LEAF_BIT) { leafCell[2 * (size_t)(w & INDEX_MASK)] = xy; leafCell[2 * (size_t)(w & INDEX_MASK) + 1] = z; return; }
    const uint32_t to = w - nextLevelBase, x2 = (xy & 0xFFFFu) << 1, y2 = (xy >> 16) << 1, z2 = z << 1;
#pragma unroll
    for (uint32_t c = 0; c < 8u; c++) { nextCell[2 * (size_t)(to + c)] = (x2 | (c & 1u)) | ((y2 | ((c >> 1) & 1u)) << 16); nextCell[2 * (size_t)(to + c) + 1] = z2 | (c >> 2); }
}
// caller holds T->qLock
static int ensureLeafCells(sdfhip_octree* T) {
    if (T->cellsReady) return SDFHIP_OK;
    hipStream_t st = T->ctx->stream;
    const uint32_t G = (uint32_t)T->info.start_grid_size;
    uint32_t widest = 0;
    for (uint32_t c : T->qLevelNodes) widest = c > widest ? c : widest;
    DevBuf<uint32_t> a, b;
    SDF_TRY(a.reserve(2ull * widest)); SDF_TRY(b.reserve(2ull * widest));
    SDF_TRY(T->qLeafCell.reserve(2ull * (T->qLeaves ? T->qLeaves : 1)));
    k_lc_root<<<gridFor(T->qLevelNodes[0], 256), 256, 0, st>>>(G, T->qLevelNodes[0], a.p);
    uint64_t base = 0;
    uint32_t* cur = a.p; uint32_t* nxt = b.p;
    for (size_t l = 0; l < T->qLevelNodes.size(); l++) {
        const uint32_t count = T->qLevelNodes[l];
        k_lc_level<<<gridFor(count, 256), 256, 0, st>>>(T->qTopo.p + base, count, (uint32_t)(base + count), cur, nxt, T->qLeafCell.p);
        base += count;
        uint32_t* t = cur; cur = nxt; nxt = t;
    }
    SDF_HIP_CHECK(hipGetLastError());
    SDF_HIP_CHECK(hipStreamSynchronize(st));            // a and b are released here
    T->cellsReady = true;
    return SDFHIP_OK;
}

struct LatLevels { uint32_t leafBase[kLatMaxLevels + 1]; int levels; };

__global__ void k_lat_tables(float* __restrict__ F, F3 origin, F3 step, F3 bmin, float cellSize, uint32_t nx, uint32_t ny, uint32_t nz) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nx + ny + nz) return;
    // the operations of k_octree_query_grid + locateLeaf, per axis
    if (j < nx) F[j] = ((origin.x + (float)j * step.x) - bmin.x) / cellSize;
    else if (j < nx + ny) F[j] = ((origin.y + (float)(j - nx) * step.y) - bmin.y) / cellSize;
    else F[j] = ((origin.z + (float)(j - nx - ny) * step.z) - bmin.z) / cellSize;
}
SDF_DEV uint32_t latLowerBound(const float* __restrict__ F, uint32_t n, float a) {      // first i with F[i] >= a  (F non-decreasing)
    uint32_t lo = 0, hi = n;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (F[mid] >= a) hi = mid; else lo = mid + 1; }
    return lo;
}
// dims: 3 words per level = the largest number of lattice indices a leaf of the level owns per axis
__global__ void k_lat_ranges(const uint32_t* __restrict__ leafCell, uint32_t leaves, LatLevels L, const float* __restrict__ F, uint32_t nx, uint32_t ny, uint32_t nz,
                             uint16_t* __restrict__ ranges, uint32_t* __restrict__ dims) {
    __shared__ uint32_t blockDims[3 * kLatMaxLevels];
    for (uint32_t k = threadIdx.x; k < 3u * kLatMaxLevels; k += blockDim.x) blockDims[k] = 0u;
    __syncthreads();
    const uint32_t i0 = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t i = i0 < leaves ? i0 : leaves - 1u;            // (idle lanes repeat the last leaf: same writes, same maxima)
    int l = 0;
    while (l + 1 < L.levels && i >= L.leafBase[l + 1]) l++;
    const float inv = __uint_as_float((uint32_t)(127 - l) << 23);          // 2^-l
    const uint32_t xy = leafCell[2 * (size_t)i], cz = leafCell[2 * (size_t)i + 1], cx = xy & 0xFFFFu, cy = xy >> 16;
    const uint32_t x0 = latLowerBound(F, nx, (float)cx * inv), x1 = latLowerBound(F, nx, (float)(cx + 1u) * inv);
    const uint32_t y0 = latLowerBound(F + nx, ny, (float)cy * inv), y1 = latLowerBound(F + nx, ny, (float)(cy + 1u) * inv);
    const uint32_t z0 = latLowerBound(F + nx + ny, nz, (float)cz * inv), z1 = latLowerBound(F + nx + ny, nz, (float)(cz + 1u) * inv);
    uint16_t* r = ranges + 6 * (size_t)i;
    r[0] = (uint16_t)x0; r[1] = (uint16_t)x1; r[2] = (uint16_t)y0; r[3] = (uint16_t)y1; r[4] = (uint16_t)z0; r[5] = (uint16_t)z1;
    // per block first (a few hundred thousand atomics on the same dozen addresses took 9.6 ms; this takes microseconds)
    if (x1 > x0 && y1 > y0 && z1 > z0) { atomicMax(&blockDims[3 * l], x1 - x0); atomicMax(&blockDims[3 * l + 1], y1 - y0); atomicMax(&blockDims[3 * l + 2], z1 - z0); }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < 3u * kLatMaxLevels; k += blockDim.x) if (blockDims[k]) atomicMax(dims + k, blockDims[k]);
}

// Work order.  A leaf narrower than a cache line writes partial lines; they are completed by its x-neighbours, which may be leaves of
// other levels.  If those run much later (a launch per level, or level after level within a launch) the partial lines are written
// back as they are: a third of the write requests at the memory interface were 32-byte ones.  So the waves are dealt START CELL by
// start cell, all levels of a cell together (k_lat_groups .. k_lat_waves build the list once per plan): what one cell's leaves leave
// incomplete, the same cell's other leaves complete while the lines are still in the L2 / the Infinity Cache.  Measured (value +
// gradient, kernel time): 512^3 789 -> 689 us, 256^3 121 -> 116 us (one launch per level: 187 us; point kernel: 264 us).  What is
// left is the stores themselves: the arithmetic and the loads alone take 37 us at 256^3, and a 200^3 lattice, whose 128 MB of results
// stay in the Infinity Cache, is written at twice the rate per byte.  A column marches through z, i.e. through planes 256 KB apart:
// what reaches HBM is a stream of 64-128-byte pieces scattered over many DRAM rows (about 3 TB/s; a sequential fill of the same
// arrays runs at 6.7 TB/s).  Fuller requests do not help (x-fastest lanes across adjoining leaves, below: same time), nor do
// non-temporal stores; the z-scatter is what leaf-driven evaluation costs, and it is still 2.2x faster than the point kernel.
// Group g = startCell * levels + level; a group's leaves take groupCount * cols lanes, rounded up to whole waves.

SDF_DEV uint32_t latGroupOf(const uint32_t* __restrict__ leafCell, uint32_t leaf, int l, const LatCols& C) {
    const uint32_t xy = leafCell[2 * (size_t)leaf], cz = leafCell[2 * (size_t)leaf + 1];
    const uint32_t sx = (xy & 0xFFFFu) >> l, sy = (xy >> 16) >> l, sz = cz >> l;
    return ((sz * C.G + sy) * C.G + sx) * (uint32_t)C.levels + (uint32_t)l;
}
__global__ void k_lat_groups(const uint32_t* __restrict__ leafCell, uint32_t leaves, LatLevels L, LatCols C, const uint16_t* __restrict__ ranges, uint32_t* __restrict__ groupCount) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= leaves) return;
    const uint16_t* r = ranges + 6 * (size_t)i;
    if (r[1] <= r[0] || r[3] <= r[2] || r[5] <= r[4]) return;               // owns no lattice point
    int l = 0;
    while (l + 1 < L.levels && i >= L.leafBase[l + 1]) l++;
    atomicAdd(groupCount + latGroupOf(leafCell, i, l, C), 1u);
}
// per group: waves it needs (scan input)
__global__ void k_lat_group_waves(const uint32_t* __restrict__ groupCount, uint32_t groups, LatCols C, uint32_t* __restrict__ groupWaves) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= groups) return;
    const uint32_t l = g % (uint32_t)C.levels, cols = C.mx[l] * C.my[l];
    if (cols != 0u && cols <= 32u) { const uint32_t k = 64u / cols; groupWaves[g] = (groupCount[g] + k - 1u) / k; }      // k whole leaves per wave
    else groupWaves[g] = (uint32_t)(((uint64_t)groupCount[g] * cols + 63ull) >> 6);
}
// sort key of a leaf: its group, then its cell within the start cell in (z, y, x) order — x-neighbours become list neighbours, so the
// leaves a wave takes together write adjoining pieces of the same rows.  Leaves without lattice points go to the end.
__global__ void k_lat_keys(const uint32_t* __restrict__ leafCell, uint32_t leaves, LatLevels L, LatCols C, const uint16_t* __restrict__ ranges,
                           uint64_t* __restrict__ key, uint32_t* __restrict__ val) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= leaves) return;
    val[i] = i;
    const uint16_t* r = ranges + 6 * (size_t)i;
    if (r[1] <= r[0] || r[3] <= r[2] || r[5] <= r[4]) { key[i] = ~0ull; return; }
    int l = 0;
    while (l + 1 < L.levels && i >= L.leafBase[l + 1]) l++;
    const uint32_t xy = leafCell[2 * (size_t)i], cz = leafCell[2 * (size_t)i + 1], m = (1u << l) - 1u;
    const uint64_t local = ((uint64_t)(cz & m) << (2 * l)) | ((uint64_t)((xy >> 16) & m) << l) | (uint64_t)(xy & 0xFFFFu & m);
    key[i] = ((uint64_t)latGroupOf(leafCell, i, l, C) << 39) | local;
}
// wave descriptor: {first sorted leaf of the group, leaves in the group, level, index of the wave within the group}
__global__ void k_lat_waves(const uint32_t* __restrict__ groupWaveBase, const uint32_t* __restrict__ groupLeafBase, const uint32_t* __restrict__ groupCount, uint32_t groups,
                            uint32_t levels, uint32_t waves, uint4* __restrict__ desc) {
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= waves) return;
    uint32_t lo = 0, hi = groups;                   // last group whose base is <= w  (groupWaveBase is non-decreasing, [groups] = waves)
    while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (groupWaveBase[mid] <= w) lo = mid; else hi = mid; }
    while (lo + 1 < groups && groupWaveBase[lo + 1] <= w) lo++;           // skip empty groups sharing the base
    desc[w] = make_uint4(groupLeafBase[lo], groupCount[lo], lo % levels, w - groupWaveBase[lo]);
}

// (k_lattice_columns / k_lattice_columns_exact live in octree_lattice.hip: that translation unit is compiled WITH the SLP vectoriser, whose
// packed v_pk_fma_f32 the separable Horner form of the fast lattice wants, this one without: Makefile)

// lattice points outside the start grid: the point kernel's box-distance branch
template <bool GRAD>
__global__ void __launch_bounds__(256) k_lattice_outside(QueryTree t, const float* __restrict__ F, F3 origin, F3 step, uint32_t nx, uint32_t ny, float* __restrict__ dist,
                                                         float* __restrict__ grad) {
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, z = blockIdx.z;
    if (x >= nx) return;
    const float G = (float)t.G;
    const float a = floorf(F[x]), b = floorf(F[nx + y]), c = floorf(F[nx + ny + z]);
    if (a >= 0.f && a < G && b >= 0.f && b < G && c >= 0.f && c < G) return;
    const F3 p = F3{origin.x + (float)x * step.x, origin.y + (float)y * step.y, origin.z + (float)z * step.z};
    const size_t at = ((size_t)z * ny + y) * nx + x;
    float g[3] = {0.f, 0.f, 0.f};
    dist[at] = (GRAD ? boxDistanceGrad(t, p, g) : boxDistance(t, p)) + t.minBorder;
    if (GRAD) { grad[3 * at] = g[0]; grad[3 * at + 1] = g[1]; grad[3 * at + 2] = g[2]; }
}

// caller holds T->qLock.  Leaves the plan with leafDriven == false when the point kernel is the better (or only) choice.
static int ensureLatticePlan(sdfhip_octree* T, const float origin[3], const float step[3], uint32_t nx, uint32_t ny, uint32_t nz) {
    sdfhip_octree::LatticePlan& P = T->lattice;
    if (P.valid && !memcmp(P.origin, origin, 12) && !memcmp(P.step, step, 12) && P.n[0] == nx && P.n[1] == ny && P.n[2] == nz) return SDFHIP_OK;
    P.valid = false; P.leafDriven = false; P.outside = false; P.classes.clear();
    memcpy(P.origin, origin, 12); memcpy(P.step, step, 12); P.n[0] = nx; P.n[1] = ny; P.n[2] = nz;
    const int levels = (int)T->qLevelNodes.size();
    const uint64_t G = (uint64_t)T->info.start_grid_size;
    bool ok = levels >= 1 && levels <= kLatMaxLevels && (G << (levels - 1)) <= 65536ull && nx <= 65535u && ny <= 65535u && nz <= 65535u && T->qLeaves > 0;
    for (int a = 0; a < 3; a++) ok = ok && std::isfinite(origin[a]) && std::isfinite(step[a]) && step[a] > 0.f;      // the tables must be non-decreasing
    if (!ok) { P.valid = true; return SDFHIP_OK; }
    hipStream_t st = T->ctx->stream;
    SDF_TRY(ensureLeafCells(T));
    const uint32_t nt = nx + ny + nz;
    SDF_TRY(P.F.reserve(nt + 1)); SDF_TRY(P.ranges.reserve(6ull * T->qLeaves));      // + 1: the column kernel reads one entry ahead
    DevBuf<uint32_t> dims;
    SDF_TRY(dims.reserve(3 * kLatMaxLevels));
    SDF_HIP_CHECK(hipMemsetAsync(dims.p, 0, 12 * kLatMaxLevels, st));
    LatLevels L{};
    L.levels = levels;
    for (int l = 0; l <= levels; l++) L.leafBase[l] = T->qLevelLeafBase[(size_t)l];
    k_lat_tables<<<gridFor(nt, 256), 256, 0, st>>>(P.F.p, F3{origin[0], origin[1], origin[2]}, F3{step[0], step[1], step[2]},
                                                   F3{T->info.box_min[0], T->info.box_min[1], T->info.box_min[2]}, T->cellSize, nx, ny, nz);
    k_lat_ranges<<<gridFor(T->qLeaves, 256), 256, 0, st>>>(T->qLeafCell.p, (uint32_t)T->qLeaves, L, P.F.p, nx, ny, nz, P.ranges.p, dims.p);
    uint32_t hd[3 * kLatMaxLevels];
    float ends[6];
    SDF_HIP_CHECK(hipMemcpyAsync(hd, dims.p, sizeof(hd), hipMemcpyDeviceToHost, st));
    SDF_HIP_CHECK(hipMemcpyAsync(ends + 0, P.F.p, 4, hipMemcpyDeviceToHost, st));
    SDF_HIP_CHECK(hipMemcpyAsync(ends + 1, P.F.p + nx - 1, 4, hipMemcpyDeviceToHost, st));
    SDF_HIP_CHECK(hipMemcpyAsync(ends + 2, P.F.p + nx, 4, hipMemcpyDeviceToHost, st));
    SDF_HIP_CHECK(hipMemcpyAsync(ends + 3, P.F.p + nx + ny - 1, 4, hipMemcpyDeviceToHost, st));
    SDF_HIP_CHECK(hipMemcpyAsync(ends + 4, P.F.p + nx + ny, 4, hipMemcpyDeviceToHost, st));
    SDF_HIP_CHECK(hipMemcpyAsync(ends + 5, P.F.p + nt - 1, 4, hipMemcpyDeviceToHost, st));
    SDF_HIP_CHECK(hipStreamSynchronize(st));
    for (int a = 0; a < 6; a++) {
        if (!std::isfinite(ends[a]) || std::fabs(ends[a]) >= 16777216.f) { P.valid = true; return SDFHIP_OK; }       // the point kernel answers
        if (std::floor(ends[a]) < 0.f || std::floor(ends[a]) >= (float)G) P.outside = true;
    }
    // lanes launched vs points: a lattice much coarser than the leaves would launch mostly empty columns
    double slots = 0;
    for (int l = 0; l < levels; l++) {
        const uint32_t count = T->qLevelLeafBase[(size_t)l + 1] - T->qLevelLeafBase[(size_t)l];
        const uint32_t mx = hd[3 * l], my = hd[3 * l + 1], mz = hd[3 * l + 2];
        if (!count || !mx || !my || !mz) continue;
        if ((double)count * mx * my >= 2147483648.0) { P.valid = true; return SDFHIP_OK; }
        slots += (double)count * mx * my * mz;
        P.classes.push_back(sdfhip_octree::LatticeClass{(uint32_t)l, T->qLevelLeafBase[(size_t)l], count, mx, my, mz});
    }
    P.leafDriven = slots <= 2.0 * (double)nx * ny * nz + 1e6;
    const uint64_t groups = G * G * G * (uint64_t)levels;
    if (P.leafDriven && groups > (1ull << 24)) P.leafDriven = false;
    if (P.leafDriven) {
        // the wave list: start cell by start cell, all levels of a cell together (see k_lattice_columns)
        LatCols C{};
        C.levels = levels; C.G = (uint32_t)G;
        for (const sdfhip_octree::LatticeClass& c : P.classes) { C.mx[c.level] = c.mx; C.my[c.level] = c.my; }
        DevBuf<uint32_t> count, wavesOf, val;
        DevBuf<uint64_t> key, keyS;
        SDF_TRY(count.reserve(groups)); SDF_TRY(wavesOf.reserve(groups + 1)); SDF_TRY(val.reserve(T->qLeaves)); SDF_TRY(key.reserve(T->qLeaves)); SDF_TRY(keyS.reserve(T->qLeaves));
        SDF_TRY(P.groupWaveBase.reserve(groups + 1)); SDF_TRY(P.groupLeafBase.reserve(groups + 1)); SDF_TRY(P.sortedLeaf.reserve(T->qLeaves));
        SDF_HIP_CHECK(hipMemsetAsync(count.p, 0, 4 * groups, st));
        k_lat_groups<<<gridFor(T->qLeaves, 256), 256, 0, st>>>(T->qLeafCell.p, (uint32_t)T->qLeaves, L, C, P.ranges.p, count.p);
        k_lat_group_waves<<<gridFor(groups, 256), 256, 0, st>>>(count.p, (uint32_t)groups, C, wavesOf.p);
        size_t need = 0;
        SDF_HIP_CHECK(devExclusiveSum(nullptr, need, wavesOf.p, P.groupWaveBase.p, (size_t)groups, st));
        DevBuf<uint8_t> tmp;
        SDF_TRY(tmp.reserve(need + 16));
        SDF_HIP_CHECK(devExclusiveSum(tmp.p, need, wavesOf.p, P.groupWaveBase.p, (size_t)groups, st));
        SDF_HIP_CHECK(devExclusiveSum(tmp.p, need, count.p, P.groupLeafBase.p, (size_t)groups, st));
        uint32_t lastBase = 0, lastWaves = 0;
        SDF_HIP_CHECK(hipMemcpyAsync(&lastBase, P.groupWaveBase.p + groups - 1, 4, hipMemcpyDeviceToHost, st));
        SDF_HIP_CHECK(hipMemcpyAsync(&lastWaves, wavesOf.p + groups - 1, 4, hipMemcpyDeviceToHost, st));
        SDF_HIP_CHECK(hipStreamSynchronize(st));
        P.waves = lastBase + lastWaves;
        if (P.waves == 0 || P.waves >= (1u << 30)) P.leafDriven = false;
        else {
            SDF_TRY(P.waveDesc.reserve(4ull * P.waves));
            k_lat_keys<<<gridFor(T->qLeaves, 256), 256, 0, st>>>(T->qLeafCell.p, (uint32_t)T->qLeaves, L, C, P.ranges.p, key.p, val.p);
            size_t sortNeed = 0;
            SDF_HIP_CHECK(devSortPairs(nullptr, sortNeed, key.p, keyS.p, val.p, P.sortedLeaf.p, (size_t)T->qLeaves, 0, (unsigned)64, st));
            DevBuf<uint8_t> sortTmp;
            SDF_TRY(sortTmp.reserve(sortNeed + 16));
            SDF_HIP_CHECK(devSortPairs(sortTmp.p, sortNeed, key.p, keyS.p, val.p, P.sortedLeaf.p, (size_t)T->qLeaves, 0, (unsigned)64, st));
            k_lat_waves<<<gridFor(P.waves, 256), 256, 0, st>>>(P.groupWaveBase.p, P.groupLeafBase.p, count.p, (uint32_t)groups, (uint32_t)levels, P.waves,
                                                            reinterpret_cast<uint4*>(P.waveDesc.p));
            SDF_HIP_CHECK(hipGetLastError());
            SDF_HIP_CHECK(hipStreamSynchronize(st));        // the temporaries are released here
        }
    }
    P.valid = true;
    return SDFHIP_OK;
}

static QueryTree makeQueryTree(const sdfhip_octree* T) {
    QueryTree q;
    q.topo = T->qTopo.p; q.coef = T->qCoef.p;
    q.bminx = T->info.box_min[0]; q.bminy = T->info.box_min[1]; q.bminz = T->info.box_min[2];
    q.bmaxx = T->info.box_max[0]; q.bmaxy = T->info.box_max[1]; q.bmaxz = T->info.box_max[2];
    q.cellSize = T->cellSize; q.minBorder = T->info.min_border_value; q.G = T->info.start_grid_size;
    return q;
}

// host-pointer batches of this many points and more are overlapped (SDFHIP_HOST_OVERLAP=0: one upload, one launch, one download, as until round 5)
constexpr uint64_t kOverlapMin = 1ull << 21, kOverlapPiece = 1ull << 20;
static inline bool hostOverlap() { static const bool v = !(getenv("SDFHIP_HOST_OVERLAP") && !strcmp(getenv("SDFHIP_HOST_OVERLAP"), "0")); return v; }

template <typename... A>
static void launchQuery(int eval_mode, bool grad, unsigned blocks, hipStream_t st, A... a) {
    if (eval_mode == SDFHIP_EVAL_EXACT) {
        if (grad) k_octree_query_coop<SDFHIP_EVAL_EXACT, true><<<blocks, 256, 0, st>>>(a...);
        else k_octree_query_coop<SDFHIP_EVAL_EXACT, false><<<blocks, 256, 0, st>>>(a...);
    } else {
        if (grad) k_octree_query_coop<SDFHIP_EVAL_FAST, true><<<blocks, 256, 0, st>>>(a...);
        else k_octree_query_coop<SDFHIP_EVAL_FAST, false><<<blocks, 256, 0, st>>>(a...);
    }
}

}  // namespace sdfhip

using namespace sdfhip;

extern "C" {

int sdfhip_octree_query(sdfhip_octree* T, const float* xyz, uint64_t n, float* out_dist, float* out_grad, int where, int eval_mode) {
    SDF_API_BEGIN
    SDF_REQUIRE(T && xyz && out_dist, "NULL argument");
    SDF_REQUIRE(T->hasData, "tree has no assembled node array");
    SDF_REQUIRE(eval_mode == SDFHIP_EVAL_EXACT || eval_mode == SDFHIP_EVAL_FAST, "unknown eval_mode");
    if (n == 0) return SDFHIP_OK;
    const uint64_t chunk = queryChunk(where == SDFHIP_HOST);
    if (n > chunk) {
        for (uint64_t off = 0; off < n; off += chunk) {
            const uint64_t m = n - off < chunk ? n - off : chunk;
            SDF_TRY(sdfhip_octree_query(T, xyz + 3 * off, m, out_dist + off, out_grad ? out_grad + 3 * off : nullptr, where, eval_mode));
        }
        return SDFHIP_OK;
    }
    sdfhip_ctx* ctx = T->ctx;
    SDF_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    if (where == SDFHIP_HOST && n <= kOctreeHostScalarMax) {
        // the scalar getDistance of the reference's API (and any handful of points): a kernel launch, two PCIe hops and a stream
        // synchronisation cost tens of microseconds; the same code on host copies of the arrays answers in about one
        SDF_TRY(ensureHostLayout(T));
        QueryTree hq = makeQueryTree(T);
        hq.topo = T->hTopo.data(); hq.coef = T->hCoef.data();
        for (uint64_t i = 0; i < n; i++) {
            const F3 p = F3{xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
            float g[3] = {0.f, 0.f, 0.f};
            if (eval_mode == SDFHIP_EVAL_EXACT) out_dist[i] = out_grad ? queryOne<SDFHIP_EVAL_EXACT, true>(hq, p, g) : queryOne<SDFHIP_EVAL_EXACT, false>(hq, p, g);
            else out_dist[i] = out_grad ? queryOne<SDFHIP_EVAL_FAST, true>(hq, p, g) : queryOne<SDFHIP_EVAL_FAST, false>(hq, p, g);
            if (out_grad) { out_grad[3 * i] = g[0]; out_grad[3 * i + 1] = g[1]; out_grad[3 * i + 2] = g[2]; }
        }
        return SDFHIP_OK;
    }
    SDF_TRY(ensureQueryLayout(T));
    std::unique_lock<std::mutex> own(ctx->stage.lock, std::defer_lock);
    if (where == SDFHIP_HOST && 12 * n <= sdfhip_stage::kStageKeepBytes) own.try_lock();
    DevBuf<float> pp, pd, pg;
    DevBuf<float>& dp = own.owns_lock() ? ctx->stage.pts : pp; DevBuf<float>& dd = own.owns_lock() ? ctx->stage.dist : pd; DevBuf<float>& dg = own.owns_lock() ? ctx->stage.grad : pg;
    const float* p = xyz; float* d = out_dist; float* g = out_grad;
    if (where == SDFHIP_HOST) {
        SDF_TRY(dp.reserve(3 * n)); SDF_TRY(dd.reserve(n));
        if (out_grad) SDF_TRY(dg.reserve(3 * n));
        // Plain copies between the caller's (pageable) arrays and the context's buffers around the kernel: 10 M points in 3.2 ms = 0.90 of
        // the box's PCIe bound.  (Rounds 2-3 had a pipeline that pinned the caller's pages in place piece by piece and overlapped upload,
        // kernel and download on two streams — 2.93-3.07 ms — but ended in a GPU memory access fault after mixed registration failures
        // over the same arrays, cause never found; it was removed in round 4: a legal call must not be able to take the process down.)
        p = dp.p; d = dd.p; g = out_grad ? dg.p : nullptr;
    }
    const QueryTree q = makeQueryTree(T);
    if (where == SDFHIP_HOST && hostOverlap() && n >= kOverlapMin) {
        // Large host batches, overlapped (round 6): the batch is cut into pieces; this thread uploads piece k + 1 and evaluates it on the
        // context's stream while a second host thread sends piece k's results back on `downStream` - PCIe is full duplex, so the call costs
        // about max(upload, download) + one piece instead of their sum (10 M points: 3.19 -> 2.64 ms, with gradients 5.45 -> 3.86 ms).  Plain copies between the caller's PAGEABLE arrays and device
        // buffers, as before: nothing of the caller's memory is registered with the runtime (the in-place-pinning pipeline of rounds 2-3,
        // removed in round 4 after a GPU memory access fault under forced registration failures, is not coming back).  A pageable
        // hipMemcpyAsync occupies its calling thread until the data are staged, hence the second thread.
        const unsigned U = out_grad ? 1u : 2u;      // uploader threads (measured below)
        {
            std::lock_guard<std::mutex> side(ctx->sideLock);
            if (!ctx->downStream) SDF_HIP_CHECK(hipStreamCreateWithFlags(&ctx->downStream, hipStreamNonBlocking));
            for (unsigned u = 1; u < U; u++) if (!ctx->upSide[u - 1]) SDF_HIP_CHECK(hipStreamCreateWithFlags(&ctx->upSide[u - 1], hipStreamNonBlocking));
        }
        hipStream_t down = ctx->downStream;
        const uint64_t piece = kOverlapPiece, pieces = (n + piece - 1) / piece;
        std::vector<hipEvent_t> ev(pieces, nullptr);
        struct Events { std::vector<hipEvent_t>& e; ~Events() { for (hipEvent_t x : e) if (x) (void)hipEventDestroy(x); } } evGuard{ev};
        for (hipEvent_t& e : ev) SDF_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        std::atomic<int> failed{0};
        std::vector<std::atomic<uint8_t>> ready(pieces);
        for (auto& r : ready) r.store(0);
        const int device = ctx->device;
        float* const dDist = d; float* const dGrad = g;
        std::thread downloader([&, device, down] {
            if (hipSetDevice(device) != hipSuccess) { failed.store(1); return; }
            for (uint64_t k = 0; k < pieces; k++) {
                while (!ready[k].load(std::memory_order_acquire)) { if (failed.load()) return; std::this_thread::yield(); }
                const uint64_t off = k * piece, m = n - off < piece ? n - off : piece;
                if (hipStreamWaitEvent(down, ev[k], 0) != hipSuccess || hipMemcpyAsync(out_dist + off, dDist + off, 4 * m, hipMemcpyDeviceToHost, down) != hipSuccess ||
                    (out_grad && hipMemcpyAsync(out_grad + 3 * off, dGrad + 3 * off, 12 * m, hipMemcpyDeviceToHost, down) != hipSuccess)) { failed.store(1); return; }
            }
            if (hipStreamSynchronize(down) != hipSuccess) failed.store(1);
        });
        // uploaders: thread u takes pieces u, u + U, ... on a stream of its own (the runtime stages a pageable copy on the calling thread:
        // two threads stage twice as fast as one until the link is full)
        // measured (profiles/r06o_host_overlap_ab.txt, 10 M points, medians of 30): value only - 1 uploader 3.03 ms, 2 uploaders 2.64 ms, 3 worse; value +
        // gradient (more bytes come back than go up) - 1 uploader 3.86 ms, 2 uploaders 4.55 ms; pieces of 2^20 points (2^19: +0.3 ms, 2^22: +0.2 ms)
        hipStream_t upStreams[3] = {st, nullptr, nullptr};
        for (unsigned u = 1; u < U; u++) upStreams[u] = ctx->upSide[u - 1];
        const float* dpts = dp.p;
        auto upload = [&, device](unsigned u, bool setDevice) {
            if (setDevice && hipSetDevice(device) != hipSuccess) { failed.store(1); return; }
            hipStream_t s = upStreams[u];
            for (uint64_t k = u; k < pieces && !failed.load(); k += U) {
                const uint64_t off = k * piece, m = n - off < piece ? n - off : piece;
                if (hipMemcpyAsync(const_cast<float*>(dpts) + 3 * off, xyz + 3 * off, 12 * m, hipMemcpyHostToDevice, s) != hipSuccess) { failed.store(1); return; }
                launchQuery(eval_mode, dGrad != nullptr, gridFor(m, 256), s, q, dpts + 3 * off, m, dDist + off, dGrad ? dGrad + 3 * off : nullptr);
                if (hipGetLastError() != hipSuccess || hipEventRecord(ev[k], s) != hipSuccess) { failed.store(1); return; }
                ready[k].store(1, std::memory_order_release);
            }
        };
        std::vector<std::thread> ups;
        if (!failed.load()) for (unsigned u = 1; u < U; u++) ups.emplace_back(upload, u, true);
        if (!failed.load()) upload(0, false);
        for (std::thread& t : ups) t.join();
        downloader.join();
        for (unsigned u = 0; u < U; u++) if (upStreams[u]) (void)hipStreamSynchronize(upStreams[u]);
        if (failed.load()) { setError("HIP error in the overlapped host-pointer query: %s", hipGetErrorString(hipGetLastError())); return SDFHIP_E_HIP; }
        return SDFHIP_OK;
    }
    if (where == SDFHIP_HOST) SDF_HIP_CHECK(hipMemcpyAsync(dp.p, xyz, 12 * n, hipMemcpyHostToDevice, st));
    const unsigned blocks = gridFor(n, 256);
    launchQuery(eval_mode, g != nullptr, blocks, st, q, (const float*)p, n, d, g);
    SDF_HIP_CHECK(hipGetLastError());
    if (where == SDFHIP_HOST) {
        SDF_HIP_CHECK(hipMemcpyAsync(out_dist, d, 4 * n, hipMemcpyDeviceToHost, st));
        if (out_grad) SDF_HIP_CHECK(hipMemcpyAsync(out_grad, g, 12 * n, hipMemcpyDeviceToHost, st));
        SDF_HIP_CHECK(hipStreamSynchronize(st));
    }
    return SDFHIP_OK;
    SDF_API_END
}

int sdfhip_octree_compact(sdfhip_octree* T) {
    SDF_API_BEGIN
    SDF_REQUIRE(T && T->hasData, "NULL argument or tree without an assembled node array");
    SDF_HIP_CHECK(hipSetDevice(T->ctx->device));
    SDF_TRY(ensureQueryLayout(T));
    std::lock_guard<std::mutex> own(T->qLock);
    // the layout reproduces the array only if every word of it is a node word or a coefficient of exactly one leaf (true for every array
    // a builder of this library or the reference emits); anything else keeps its array
    if (T->data.p && T->qNodes + 64ull * T->qLeaves == T->info.num_words) {
        SDF_HIP_CHECK(hipStreamSynchronize(T->ctx->stream));
        T->data.release(); T->dataPinned = false;          // an explicit request: pointers from sdfhip_octree_device_words are invalid from here on (sdfhip.h)
        BigBlockCache::get().trimTo(T->ctx->device, T->ctx->stream, BigBlockCache::get().keepFor(T->ctx->device, T->ctx->stream));
    }
    return SDFHIP_OK;
    SDF_API_END
}

int sdfhip_octree_device_bytes(sdfhip_octree* T, uint64_t* out_bytes) {
    SDF_API_BEGIN
    SDF_REQUIRE(T && out_bytes, "NULL argument");
    std::lock_guard<std::mutex> own(T->qLock);
    const sdfhip_octree::LatticePlan& P = T->lattice;
    *out_bytes = 4ull * (T->data.n + T->qTopo.n + T->qOrig.n + T->qCoef.n + T->qLeafCell.n + P.F.n + P.groupWaveBase.n + P.groupLeafBase.n + P.sortedLeaf.n + P.waveDesc.n) + 2ull * P.ranges.n;
    return SDFHIP_OK;
    SDF_API_END
}

int sdfhip_octree_query_grid(sdfhip_octree* T, const float origin[3], const float step[3], uint32_t nx, uint32_t ny, uint32_t nz,
                             float* out_dist, float* out_grad, int where, int eval_mode) {
    SDF_API_BEGIN
    SDF_REQUIRE(T && origin && step && out_dist, "NULL argument");
    SDF_REQUIRE(T->hasData, "tree has no assembled node array");
    SDF_REQUIRE(eval_mode == SDFHIP_EVAL_EXACT || eval_mode == SDFHIP_EVAL_FAST, "unknown eval_mode");
    const uint64_t n = (uint64_t)nx * ny * nz;
    if (n == 0) return SDFHIP_OK;
    SDF_REQUIRE((double)nx * ny * nz <= (double)(1ull << 38), "lattice larger than 2^38 points");
    sdfhip_ctx* ctx = T->ctx;
    SDF_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    SDF_TRY(ensureQueryLayout(T));
    DevBuf<float> dd, dg;
    float* d = out_dist; float* g = out_grad;
    if (where == SDFHIP_HOST) {
        SDF_TRY(dd.reserve(n));
        if (out_grad) SDF_TRY(dg.reserve(3 * n));
        d = dd.p; g = out_grad ? dg.p : nullptr;
    }
    const QueryTree q = makeQueryTree(T);
    const F3 o = F3{origin[0], origin[1], origin[2]}, s = F3{step[0], step[1], step[2]};
    const unsigned blocks = gridFor(n, 256);
    static const bool noLeafDriven = getenv("SDFHIP_LATTICE_POINTS") != nullptr;      // A/B switch: always the point kernel
    bool answered = false;
    if (!noLeafDriven) {
        std::lock_guard<std::mutex> own(T->qLock);
        SDF_TRY(ensureLatticePlan(T, origin, step, nx, ny, nz));
        const sdfhip_octree::LatticePlan& P = T->lattice;
        if (P.leafDriven) {
            LatCols C{};
            C.levels = (int)T->qLevelNodes.size(); C.G = (uint32_t)T->info.start_grid_size;
            for (const sdfhip_octree::LatticeClass& c : P.classes) { C.mx[c.level] = c.mx; C.my[c.level] = c.my; }
            const uint4* desc = reinterpret_cast<const uint4*>(P.waveDesc.p);
            if (eval_mode == SDFHIP_EVAL_FAST) {
                latticeColumnsLaunch(st, false, T->qCoef.p, P.F.p, P.ranges.p, desc, P.sortedLeaf.p, P.waves, C, nx, ny, d, g);
            } else {
                latticeColumnsLaunch(st, true, T->qCoef.p, P.F.p, P.ranges.p, desc, P.sortedLeaf.p, P.waves, C, nx, ny, d, g);
            }
            if (P.outside) {
                const dim3 og(gridFor(nx, 256), ny, nz);
                if (g) k_lattice_outside<true><<<og, 256, 0, st>>>(q, P.F.p, o, s, nx, ny, d, g);
                else k_lattice_outside<false><<<og, 256, 0, st>>>(q, P.F.p, o, s, nx, ny, d, nullptr);
            }
            answered = true;
        }
    }
    if (answered) {
    } else if (eval_mode == SDFHIP_EVAL_EXACT) {
        if (g) k_octree_query_grid<SDFHIP_EVAL_EXACT, true><<<blocks, 256, 0, st>>>(q, o, s, nx, ny, nz, d, g);
        else k_octree_query_grid<SDFHIP_EVAL_EXACT, false><<<blocks, 256, 0, st>>>(q, o, s, nx, ny, nz, d, nullptr);
    } else {
        if (g) k_octree_query_grid<SDFHIP_EVAL_FAST, true><<<blocks, 256, 0, st>>>(q, o, s, nx, ny, nz, d, g);
        else k_octree_query_grid<SDFHIP_EVAL_FAST, false><<<blocks, 256, 0, st>>>(q, o, s, nx, ny, nz, d, nullptr);
    }
    SDF_HIP_CHECK(hipGetLastError());
    if (where == SDFHIP_HOST) {
        SDF_HIP_CHECK(hipMemcpyAsync(out_dist, d, 4 * n, hipMemcpyDeviceToHost, st));
        if (out_grad) SDF_HIP_CHECK(hipMemcpyAsync(out_grad, g, 12 * n, hipMemcpyDeviceToHost, st));
        SDF_HIP_CHECK(hipStreamSynchronize(st));
    }
    return SDFHIP_OK;
    SDF_API_END
}

}  // extern "C"

// (sdfhip_ctx_create: the runtime loads a translation unit's code object on the first use of one of its kernels — milliseconds that would
// otherwise land in the first build or the first query of a process)
namespace sdfhip { void loadKernelsOctreeQuery() { hipFuncAttributes a; (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_octree_query_coop<SDFHIP_EVAL_EXACT, false>)); (void)hipGetLastError(); } }
