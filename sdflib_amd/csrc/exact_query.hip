// Batched ExactOctreeSdf::getDistance on the device.  PRODUCT code — independent of oracle/.
//
// Reference behaviour reproduced: ExactOctreeSdf::getDistance(vec3) / (vec3, vec3&) (src/sdf/ExactOctreeSdf.cpp:38-178,
// 180-320): descend to the bit-encoding depth, decode the bit-packed triangle set, filter it through the byte masks of
// the last two levels, brute-force the nearest triangle (first minimum in ascending id), sign by the pseudonormal;
// roundFloat is '> 0.5' here (:33-36).  The reference decodes into mutable scratch vectors (not re-entrant); this kernel
// streams the set through the mask chain with two running rank counters instead, so it needs no scratch at all.
// Compile with -ffp-contract=off.
#include "exact_internal.h"
#include "dev_prims.h"
#include <memory>
#include <cstring>
#include <cstdlib>

namespace sdfhip {

struct ExactView {
    const uint32_t* nodes; const uint32_t* sets; const uint8_t* masks; const float* td; const float* frames;
    float bminx, bminy, bminz, bmaxx, bmaxy, bmaxz, cellSize;
    int G; uint32_t startDepth, bitEnc, bits;
    uint32_t numNodes;                     // also the sort key of a query outside the grid (sorts behind every node)
    uint64_t maskBytes, setWords;          // array lengths (both arrays carry padding behind them): the batched decode clamps its look-ahead loads
};

SDF_HD uint32_t unpackIndex(const uint32_t* __restrict__ set, uint32_t bIdx, uint32_t bits) {
    const uint32_t w = bIdx >> 5, bit = bIdx & 31u;
    return ((set[w] << bit) >> (32u - bits)) | (uint32_t)((unsigned long long)set[w + 1] >> (64u - (bit + bits)));
}
SDF_HD bool maskBit(const uint8_t* __restrict__ m, uint32_t k) { return (m[k >> 3] & (0x80u >> (k & 7u))) != 0; }

// One query, as the reference answers it.  HOSTSIDE = true: the same code compiled for the host, reading host copies of the arrays (the
// 37-float TriangleData instead of the packed frames: the same numbers) — what sdfhip_exact_query runs for a handful of points, where a
// kernel launch + two PCIe hops + a stream synchronisation (tens of microseconds) would dwarf the microseconds of work.
template <bool GRAD, bool HOSTSIDE>
SDF_HD float exactOne(const ExactView& v, F3 p, float* grad, uint32_t* tri) {
    F3 f = F3{(p.x - v.bminx) / v.cellSize, (p.y - v.bminy) / v.cellSize, (p.z - v.bminz) / v.cellSize};
    const float flx = floorf(f.x), fly = floorf(f.y), flz = floorf(f.z);
    const int ix = (int)flx, iy = (int)fly, iz = (int)flz;
    f = F3{f.x - flx, f.y - fly, f.z - flz};
    if (ix < 0 || ix >= v.G || iy < 0 || iy >= v.G || iz < 0 || iz >= v.G) {
        const F3 size = F3{v.bmaxx - v.bminx, v.bmaxy - v.bminy, v.bmaxz - v.bminz};
        const F3 center = F3{v.bminx, v.bminy, v.bminz} + 0.5f * size;
        const F3 d = p - center;
        const F3 q = F3{fabsf(d.x), fabsf(d.y), fabsf(d.z)} - 0.5f * size;
        const F3 qm = F3{gmax(q.x, 0.f), gmax(q.y, 0.f), gmax(q.z, 0.f)};
        if (tri) *tri = 0;
        return (length(qm) + gmin(gmax(q.x, gmax(q.y, q.z)), 0.0f)) + sqrtf(3.0f) * size.x;
    }
    uint32_t node = (uint32_t)((iz * v.G + iy) * v.G + ix);
    auto isLeaf = [&](uint32_t nd) { return (v.nodes[2 * (size_t)nd] & 0x80000000u) != 0u; };
    auto descend = [&](uint32_t nd) {
        const uint32_t c = ((f.z > 0.5f) ? 4u : 0u) + ((f.y > 0.5f) ? 2u : 0u) + ((f.x > 0.5f) ? 1u : 0u);
        f = F3{gfract(2.0f * f.x), gfract(2.0f * f.y), gfract(2.0f * f.z)};
        return (v.nodes[2 * (size_t)nd] & 0x7FFFFFFFu) + c;
    };
    uint32_t depth = v.startDepth;
    while (!isLeaf(node) && depth < v.bitEnc) { node = descend(node); depth++; }
    const uint32_t* set = v.sets + v.nodes[2 * (size_t)node + 1];
    const uint32_t cnt = set[0];
    const uint8_t* m1 = nullptr; const uint8_t* m2 = nullptr;
    if (!isLeaf(node)) {
        node = descend(node);
        m1 = v.masks + v.nodes[2 * (size_t)node + 1];
        if (!isLeaf(node)) { node = descend(node); m2 = v.masks + v.nodes[2 * (size_t)node + 1]; }
    }
    float best = INFINITY; uint32_t bestTri = 0;
    uint32_t r1 = 0;                       // rank among the entries that passed the first mask
    for (uint32_t t = 0; t < cnt; t++) {
        if (m1) {
            if (!maskBit(m1, t)) continue;
            const uint32_t k = r1++;
            if (m2 && !maskBit(m2, k)) continue;
        }
        const uint32_t ti = unpackIndex(set + 1, t * v.bits, v.bits);
        TriFrame fr;
        if constexpr (HOSTSIDE) loadFrame(v.td + (size_t)TD_FLOATS * ti, fr); else loadFramePacked(v.frames, ti, fr);
        const float d = sqDistPointTriangle(p, fr);
        if (d < best) { best = d; bestTri = ti; }
    }
    if (tri) *tri = bestTri;
    if (GRAD) {
        F3 g;
        const float d = signedDistPointTriangleGradLocal(p, v.td + (size_t)TD_FLOATS * bestTri, g);
        grad[0] = g.x; grad[1] = g.y; grad[2] = g.z;
        return d;
    }
    return signedDistPointTriangle(p, v.td + (size_t)TD_FLOATS * bestTri);
}

template <bool GRAD>
__global__ void __launch_bounds__(256) k_exact_query(ExactView v, const float* __restrict__ pts, uint64_t n, float* __restrict__ dist, float* __restrict__ grad,
                                                     uint32_t* __restrict__ tri) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t t = 0;
    dist[i] = exactOne<GRAD, false>(v, F3{pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]}, GRAD ? grad + 3 * i : nullptr, tri ? &t : nullptr);
    if (tri) tri[i] = t;
}


constexpr uint64_t kHostScalarMax = 32;
static int ensureHostCopy(sdfhip_exact* T) {
    std::lock_guard<std::mutex> own(T->hostLock);
    if (T->hostReady) return SDFHIP_OK;
    hipStream_t st = T->ctx->stream;
    const sdfhip_exact_info& I = T->info;
    T->hNodes.resize(2 * I.num_nodes); T->hSets.resize(I.num_set_words + 2, 0u); T->hMasks.resize(I.num_mask_bytes + 1, 0); T->hTri.resize((size_t)TD_FLOATS * I.num_triangles);
    SDF_HIP_CHECK(hipMemcpyAsync(T->hNodes.data(), T->nodes.p, 8 * I.num_nodes, hipMemcpyDeviceToHost, st));
    if (I.num_set_words) SDF_HIP_CHECK(hipMemcpyAsync(T->hSets.data(), T->sets.p, 4 * I.num_set_words, hipMemcpyDeviceToHost, st));
    if (I.num_mask_bytes) SDF_HIP_CHECK(hipMemcpyAsync(T->hMasks.data(), T->masks.p, I.num_mask_bytes, hipMemcpyDeviceToHost, st));
    SDF_HIP_CHECK(hipMemcpyAsync(T->hTri.data(), T->tri(), sizeof(float) * TD_FLOATS * I.num_triangles, hipMemcpyDeviceToHost, st));
    SDF_HIP_CHECK(hipStreamSynchronize(st));
    T->hostReady = true;
    return SDFHIP_OK;
}

// ---- leaf-sorted, wave-cooperative path (large batches) ------------------------------------------------------------------
// k_exact_locate walks every query down to the node it ends in and records that node's id; queries are then radix-sorted by it so that
// the queries of one leaf sit next to each other, and k_exact_tiles (below) gives each wave 64 consecutive sorted queries: for every
// distinct leaf in them the wave decodes the leaf's triangle list ONCE and all of the leaf's queries share the staged triangles.
constexpr uint32_t QNONE = 0xFFFFFFFFu;

// (A one-pass counting sort on the node id — rank = old value of a per-node counter, position = rank + scan — was built and measured in
// round 4: 10 M returning atomics cost 0.33 ms and the scattered 4-byte writes 0.38 ms against 0.30 ms for the three radix passes; dropped.)
__global__ void __launch_bounds__(256) k_exact_locate(ExactView v, const float* __restrict__ pts, uint64_t n, float* __restrict__ dist, float* __restrict__ grad,
                                                      uint32_t* __restrict__ tri, uint32_t* __restrict__ key, uint32_t* __restrict__ qidx) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    qidx[i] = (uint32_t)i;
    const F3 p = F3{pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
    F3 f = F3{(p.x - v.bminx) / v.cellSize, (p.y - v.bminy) / v.cellSize, (p.z - v.bminz) / v.cellSize};
    const float flx = floorf(f.x), fly = floorf(f.y), flz = floorf(f.z);
    const int ix = (int)flx, iy = (int)fly, iz = (int)flz;
    f = F3{f.x - flx, f.y - fly, f.z - flz};
    if (ix < 0 || ix >= v.G || iy < 0 || iy >= v.G || iz < 0 || iz >= v.G) {
        const F3 size = F3{v.bmaxx - v.bminx, v.bmaxy - v.bminy, v.bmaxz - v.bminz};
        const F3 center = F3{v.bminx, v.bminy, v.bminz} + 0.5f * size;
        const F3 d = p - center;
        const F3 q = F3{fabsf(d.x), fabsf(d.y), fabsf(d.z)} - 0.5f * size;
        const F3 qm = F3{gmax(q.x, 0.f), gmax(q.y, 0.f), gmax(q.z, 0.f)};
        dist[i] = (length(qm) + gmin(gmax(q.x, gmax(q.y, q.z)), 0.0f)) + sqrtf(3.0f) * size.x;
        if (tri) tri[i] = 0;
        key[i] = v.numNodes;
        return;
    }
    uint32_t node = (uint32_t)((iz * v.G + iy) * v.G + ix);
    auto isLeaf = [&](uint32_t nd) { return (v.nodes[2 * (size_t)nd] & 0x80000000u) != 0u; };
    auto descend = [&](uint32_t nd) {
        const uint32_t c = ((f.z > 0.5f) ? 4u : 0u) + ((f.y > 0.5f) ? 2u : 0u) + ((f.x > 0.5f) ? 1u : 0u);
        f = F3{gfract(2.0f * f.x), gfract(2.0f * f.y), gfract(2.0f * f.z)};
        return (v.nodes[2 * (size_t)nd] & 0x7FFFFFFFu) + c;
    };
    uint32_t depth = v.startDepth;
    while (!isLeaf(node) && depth < v.bitEnc) { node = descend(node); depth++; }
    if (!isLeaf(node)) {
        node = descend(node);
        if (!isLeaf(node)) node = descend(node);
    }
    key[i] = node;          // what the node means for the query (set, masks) is in the tree's leafCtx table
}

// leafCtx (see exact_internal.h): the tree is walked level by level from the start grid, exactly as a query walks it — down to the
// bit-encoding depth (or a leaf above it) for the packed set, then at most two more levels for the byte masks.
struct CtxItem { uint32_t node, stage, set, m1; };
__global__ void __launch_bounds__(256) k_exact_ctx_level(ExactView v, const CtxItem* __restrict__ in, uint32_t nIn, uint32_t depth, CtxItem* __restrict__ out,
                                                         uint32_t* __restrict__ outCount, uint32_t* __restrict__ ctx, uint32_t capacity) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nIn) return;
    CtxItem it = in[i];
    const uint32_t w0 = v.nodes[2 * (size_t)it.node], aux = v.nodes[2 * (size_t)it.node + 1];
    const bool leaf = (w0 & 0x80000000u) != 0u;
    uint32_t m2 = QNONE;
    bool final = false;
    if (it.stage == 0) {
        if (leaf || depth >= v.bitEnc) { it.set = aux; it.m1 = QNONE; it.stage = 1; final = leaf; }      // the node whose packed set a query decodes
    } else if (it.stage == 1) { it.m1 = aux; it.stage = 2; final = leaf; }                                // first mask level
    else { m2 = aux; final = true; }                                                                      // second mask level: the walk ends here
    if (final) {
        uint32_t* c = ctx + 4 * (size_t)it.node;
        c[0] = it.set; c[1] = it.m1; c[2] = m2; c[3] = v.sets[it.set];
        return;
    }
    const uint32_t base = atomicAdd(outCount, 8u);
    const uint32_t child = w0 & 0x7FFFFFFFu;
    // a node array that is not a tree (shared or cyclic child links in imported data) grows the frontier beyond the node count: nothing is
    // written past the buffer or read past the array, the count alone tells the host
    if ((uint64_t)base + 8u > capacity || (uint64_t)child + 8u > capacity) return;
#pragma unroll
    for (uint32_t k = 0; k < 8; k++) out[base + k] = CtxItem{child + k, it.stage, it.set, it.m1};
}



// ---- k_exact_tiles ------------------------------------------------------------------------------------------------------------------
// A chunk of 64 set entries keeps a quarter of them on average, so survivors are compacted over the WHOLE set before any distance is
// evaluated (evaluating chunk by chunk filled half the wave: 47 % of the lanes per VALU instruction in round 2's kernel).  Per run of
// queries of one leaf:
//   1. DECODE: the whole set streams through the two mask levels (ballot ranks) and the surviving triangle ids are COMPACTED into a
//      per-wave LDS list (EX_IDS entries per round; longer lists take several rounds);
//   2. EVALUATE: the list is walked in FULL tiles of 64 frames (80-byte packed frames gathered into LDS, one triangle per lane); lane
//      (g, i) works for query i of the run on the tile's triangles g, g + G, ...; runs of more than 21 queries are cut into equal parts
//      of 11..21 queries so that G x part fills at least 80 % of the wave; the distance itself is branch-free
//      (sqDistPointTriangleSelect).  A variant evaluating two triangles per step with packed fp32 instructions was measured and dropped:
//      v_pk_mul_f32 / v_pk_add_f32 issue at half rate on gfx950 (no arithmetic gain) and its 107 registers cost occupancy (3.5 vs 3.2 ms);
//   3. REDUCE: the partial minima are 64-bit keys (distance bits << 32 | position in the leaf's list): the smallest key is the first
//      minimum in list order, the reference's `d < best` rule, whatever the order of the ids — merged over g by a shuffle tree and handed
//      to the query's own lane, which looks the triangle id up in the LDS list;
//   4. once per WAVE, after its last run, all 64 lanes compute the signed distance (and gradient) of their own query's triangle.
// One wave per block (64 threads): no block-level barriers, LDS per wave = ids + one frame tile.
constexpr int EX_BATCH = 8;                   // chunks of 64 set entries decoded per round
constexpr int EX_IDS = 64 * EX_BATCH;        // survivors of a round (LDS list)
constexpr int EX_WORDS = (64 * EX_BATCH * 32 / 32 + 2 + 63) / 64;        // packed-set words a lane fetches per round at 32 bits per index (9)
constexpr unsigned long long EX_KEY_NONE = ((unsigned long long)0x7F800000u << 32);     // +inf at position 0: no candidate is smaller unless its distance is (`best = INFINITY`)

SDF_DEV unsigned long long shflKey(unsigned long long k, int src) {
    const uint32_t lo = __shfl((uint32_t)k, src), hi = __shfl((uint32_t)(k >> 32), src);
    return ((unsigned long long)hi << 32) | lo;
}

template <bool GRAD>
__global__ void __launch_bounds__(64) k_exact_tiles(ExactView v, const float* __restrict__ pts, uint64_t n, const uint32_t* __restrict__ skey, const uint32_t* __restrict__ sidx,
                                                    const uint4* __restrict__ leafCtx, float* __restrict__ dist, float* __restrict__ grad, uint32_t* __restrict__ tri) {
    __shared__ float4 s_tile[64 * 5];                                  // one tile: 64 packed frames of 80 bytes
    __shared__ uint32_t s_ids[EX_IDS];
    uint32_t* s_words = reinterpret_cast<uint32_t*>(s_tile);           // the round's packed-set words live in the frame tile's space while it is idle
    static_assert(64 * EX_WORDS <= 64 * 20, "the frame tile must hold a round's set words");
    const int lane = threadIdx.x;
    const uint64_t e = (uint64_t)xcdLogicalBlock() * 64u + (uint64_t)lane;       // queries are sorted by leaf: a contiguous range of leaves per XCD
    uint32_t myKey = v.numNodes, q = 0;
    if (e < n) { myKey = skey[e]; q = sidx[e]; }
    const bool active = myKey < v.numNodes;
    F3 p = F3{0.f, 0.f, 0.f};
    uint32_t cSet = 0, cM1 = QNONE, cM2 = QNONE;
    uint32_t cCnt = 0;
    if (active) {          // the lanes of a run read the same 16 bytes of the table: one request per run, not one per query
        const uint4 c = leafCtx[myKey];
        cSet = c.x; cM1 = c.y; cM2 = c.z; cCnt = c.w;
        p = F3{pts[3 * (size_t)q], pts[3 * (size_t)q + 1], pts[3 * (size_t)q + 2]};
    }
    bool done = !active;
    unsigned long long acc = EX_KEY_NONE;                // this lane's own query: (distance bits, list position) of the nearest so far ...
    uint32_t accTri = 0;                                 // ... and its triangle (0 while nothing was nearer than +inf, as in the reference)
    const unsigned long long ltMask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (;;) {
        const unsigned long long pending = __ballot(!done);
        if (pending == 0ull) break;
        const int leader = __ffsll((long long)pending) - 1;
        const uint32_t k = __shfl(myKey, leader);
        const bool inRun = !done && myKey == k;          // sorted: the run's queries sit in lanes [leader, leader + r)
        const uint32_t r = (uint32_t)__popcll(__ballot(inRun));
        const uint32_t setIdx = __shfl(cSet, leader), m1i = __shfl(cM1, leader), m2i = __shfl(cM2, leader);
        const uint32_t cnt = __shfl(cCnt, leader);
        const bool m1 = m1i != QNONE, m2 = m2i != QNONE;
        const uint32_t nsub = (r + 20u) / 21u, part = (r + nsub - 1u) / nsub;
        uint32_t base = 0, r1 = 0, posBase = 0;
        while (base < cnt) {
            // 1. decode a round of survivors into s_ids: up to EX_BATCH chunks of 64 set entries per round.  Everything the round needs
            // from memory is fetched by ONE set of independent, coalesced loads — lane l takes byte l of the first mask from the round's
            // first entry on, bytes l and l + 1 of the second mask from the current rank on, and the packed-set words 64 j + l — so the
            // round costs one memory latency (it was three per chunk: mask byte, second mask byte at the rank the first gives, set words;
            // the waves spent most of their time waiting for that chain).  The words go through LDS, the mask bytes through shuffles.
            uint32_t nIds = 0;
            {
                const uint32_t left = cnt - base;
                const uint32_t nch = (left + 63u) / 64u < (uint32_t)EX_BATCH ? (left + 63u) / 64u : (uint32_t)EX_BATCH;
                const uint32_t endEntry = (base + 64u * nch < cnt) ? base + 64u * nch : cnt;
                const uint32_t firstWord = (base * v.bits) >> 5;
                const uint32_t nWords = ((endEntry * v.bits + 31u) >> 5) + 1u - firstWord;          // + 1: unpacking reads word w + 1 as well
                const uint32_t r1start = r1;
                uint32_t B1 = 0xFFu, B2 = 0xFFFFu;
                if (m1) { const uint64_t at = (uint64_t)m1i + (base >> 3) + (uint32_t)lane; B1 = v.masks[at < v.maskBytes ? at : v.maskBytes]; }
                if (m2) {
                    const uint64_t at = (uint64_t)m2i + (r1start >> 3) + (uint32_t)lane;
                    B2 = (uint32_t)v.masks[at < v.maskBytes ? at : v.maskBytes] | ((uint32_t)v.masks[at + 1 < v.maskBytes ? at + 1 : v.maskBytes] << 8);
                }
                uint32_t W[EX_WORDS];
#pragma unroll
                for (int j = 0; j < EX_WORDS; j++) {
                    const uint64_t at = (uint64_t)setIdx + 1u + firstWord + 64u * j + (uint32_t)lane;
                    W[j] = (64u * j < nWords) ? v.sets[at <= v.setWords ? at : v.setWords] : 0u;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
                for (int j = 0; j < EX_WORDS; j++) if (64u * j < nWords) s_words[64 * j + lane] = W[j];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                for (uint32_t c = 0; c < nch; c++) {
                    const uint32_t t = base + 64u * c + (uint32_t)lane;
                    const uint32_t b1 = (uint32_t)__shfl((int)B1, (int)(8u * c) + (lane >> 3));
                    const bool pass1 = t < cnt && (b1 & (0x80u >> (lane & 7))) != 0u;
                    const unsigned long long bal1 = __ballot(pass1);
                    const uint32_t rel = (r1start & 7u) + (r1 - r1start) + (uint32_t)__popcll(bal1 & ltMask);      // bit offset from the first second-mask byte fetched
                    r1 += (uint32_t)__popcll(bal1);
                    const uint32_t two = (uint32_t)__shfl((int)B2, (int)((rel >> 3) < 63u ? (rel >> 3) : 63u));
                    const uint32_t b2 = ((rel >> 3) == 64u) ? (two >> 8) : (two & 0xFFu);
                    const bool pass = pass1 && (m2 ? (b2 & (0x80u >> (rel & 7u))) != 0u : true);
                    const unsigned long long bal = __ballot(pass);
                    if (pass) {
                        const uint32_t bIdx = t * v.bits - (firstWord << 5), w = bIdx >> 5, bit = bIdx & 31u;
                        const uint32_t w0 = s_words[w], w1 = s_words[w + 1];
                        s_ids[nIds + (uint32_t)__popcll(bal & ltMask)] = ((w0 << bit) >> (32u - v.bits)) | (uint32_t)((unsigned long long)w1 >> (64u - (bit + v.bits)));
                    }
                    nIds += (uint32_t)__popcll(bal);
                }
                base += 64u * nch;
            }
            if (nIds == 0) continue;
            // 2. evaluate: part by part of the run, tile by tile of the list
            for (uint32_t sub = 0; sub < nsub; sub++) {
                const uint32_t lo = sub * part, sz = (r - lo < part) ? r - lo : part;
                const uint32_t G = 64u / sz, g = (uint32_t)lane / sz, i = (uint32_t)lane - g * sz;
                const int owner = leader + (int)(lo + i);
                const F3 po = F3{__shfl(p.x, owner), __shfl(p.y, owner), __shfl(p.z, owner)};
                unsigned long long best = EX_KEY_NONE;
                for (uint32_t tb = 0; tb < nIds; tb += 64) {
                    const uint32_t nk = (nIds - tb < 64u) ? nIds - tb : 64u;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    if ((uint32_t)lane < nk) {
                        const float4* src = reinterpret_cast<const float4*>(v.frames) + 5 * (size_t)s_ids[tb + (uint32_t)lane];
                        float4* dst = s_tile + 5 * lane;
#pragma unroll
                        for (int c = 0; c < 5; c++) dst[c] = src[c];
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    if (g < G) {
                        for (uint32_t sIdx = g; sIdx < nk; sIdx += G) {
                            const float4* fp = s_tile + 5 * sIdx;
                            const float4 a = fp[0], bq = fp[1], c = fp[2], d4 = fp[3], e4 = fp[4];
                            TriFrame fr;
                            fr.origin = F3{a.x, a.y, a.z};
                            fr.m[0] = a.w; fr.m[1] = bq.x; fr.m[2] = bq.y; fr.m[3] = bq.z; fr.m[4] = bq.w; fr.m[5] = c.x; fr.m[6] = c.y; fr.m[7] = c.z; fr.m[8] = c.w;
                            fr.b = F2{d4.x, d4.y}; fr.c = F2{d4.z, d4.w}; fr.v2 = e4.x; fr.v3 = F2{e4.y, e4.z};
                            const float d = sqDistPointTriangleSelect(po, fr);
                            const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (posBase + tb + sIdx);
                            best = key < best ? key : best;
                        }
                    }
                }
                // 3. reduce over g (lanes i, i + sz, i + 2 sz, ...), result in lanes [0, sz); hand it to the queries' own lanes
                if (g >= G) best = EX_KEY_NONE;
                for (uint32_t span = 1; span < G; span <<= 1) {
                    const int src = lane + (int)(span * sz);
                    const unsigned long long other = shflKey(best, src < 64 ? src : lane);
                    if (src < 64 && other < best) best = other;
                }
                const int from = lane - leader - (int)lo;              // lane of query (lo + from) holds it after the reduction
                const unsigned long long mine = shflKey(best, (from >= 0 && from < (int)sz) ? from : lane);
                if (inRun && from >= 0 && from < (int)sz && mine < acc) { acc = mine; accTri = s_ids[(uint32_t)mine - posBase]; }
            }
            posBase += nIds;
        }
        if (inRun) done = true;
    }
    // 4. every lane finishes its own query
    if (active) {
        const uint32_t bestTri = accTri;
        if (GRAD) {
            F3 gr;
            dist[q] = signedDistPointTriangleGradLocal(p, v.td + (size_t)TD_FLOATS * bestTri, gr);
            grad[3 * (size_t)q] = gr.x; grad[3 * (size_t)q + 1] = gr.y; grad[3 * (size_t)q + 2] = gr.z;
        } else dist[q] = signedDistPointTriangle(p, v.td + (size_t)TD_FLOATS * bestTri);
        if (tri) tri[q] = bestTri;
    }
}

// ---- decoded leaf lists + k_exact_lists (round 4) -------------------------------------------------------------------------------------
// k_exact_tiles spends a quarter of its time decoding — the same packed set through the same two masks for every wave that meets the
// leaf — and, measured per wave, most of the rest WAITING: a decode round and every 64-frame tile is a dependent memory round trip with
// nothing issued behind it (VALU busy a third of a wave's lifetime).  So the survivors of every node a query can end in are decoded ONCE
// per tree into one array of triangle ids (`leafLists`, node -> {offset, count} in `leafList`; made by the first batched query like
// leafCtx; 4 B per surviving entry: 2.3 x 10^8 B at C3) and the query kernel becomes a flat software pipeline over the tiles of all
// runs of a wave: while tile t is evaluated out of LDS the 80-byte frames of tile t + 1 are in flight to registers and the ids of tile
// t + 2 behind them.  Same comparisons in the same list order as k_exact_tiles, i.e. the reference's `d < best` scan: same bits.
// Trees whose lists would exceed SDFHIP_EXACT_LISTS_MB (default 4096) keep the decoding kernel.
template <bool WRITE>
__global__ void __launch_bounds__(64) k_exact_leaf_lists(ExactView v, const uint4* __restrict__ leafCtx, uint32_t* __restrict__ counts, const uint64_t* __restrict__ offsets,
                                                        uint32_t* __restrict__ lists) {
    __shared__ uint32_t s_words[64 * EX_WORDS];
    __shared__ uint32_t s_ids[EX_IDS];
    const int lane = threadIdx.x;
    const uint32_t node = blockIdx.x;
    const uint4 ctx = leafCtx[node];
    if (ctx.x == QNONE) { if (!WRITE && lane == 0) counts[node] = 0u; return; }
    const uint32_t setIdx = ctx.x, m1i = ctx.y, m2i = ctx.z, cnt = ctx.w;
    const bool m1 = m1i != QNONE, m2 = m2i != QNONE;
    const unsigned long long ltMask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const uint64_t outBase = WRITE ? offsets[node] : 0ull;
    uint32_t base = 0, r1 = 0, total = 0;
    while (base < cnt) {
        // one round of k_exact_tiles' decode: the same loads, ranks and unpacking
        uint32_t nIds = 0;
        const uint32_t left = cnt - base;
        const uint32_t nch = (left + 63u) / 64u < (uint32_t)EX_BATCH ? (left + 63u) / 64u : (uint32_t)EX_BATCH;
        const uint32_t endEntry = (base + 64u * nch < cnt) ? base + 64u * nch : cnt;
        const uint32_t firstWord = (base * v.bits) >> 5;
        const uint32_t nWords = ((endEntry * v.bits + 31u) >> 5) + 1u - firstWord;
        const uint32_t r1start = r1;
        uint32_t B1 = 0xFFu, B2 = 0xFFFFu;
        if (m1) { const uint64_t at = (uint64_t)m1i + (base >> 3) + (uint32_t)lane; B1 = v.masks[at < v.maskBytes ? at : v.maskBytes]; }
        if (m2) {
            const uint64_t at = (uint64_t)m2i + (r1start >> 3) + (uint32_t)lane;
            B2 = (uint32_t)v.masks[at < v.maskBytes ? at : v.maskBytes] | ((uint32_t)v.masks[at + 1 < v.maskBytes ? at + 1 : v.maskBytes] << 8);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
        for (int j = 0; j < EX_WORDS; j++) {
            const uint64_t at = (uint64_t)setIdx + 1u + firstWord + 64u * j + (uint32_t)lane;
            if (64u * j < nWords) s_words[64 * j + lane] = v.sets[at <= v.setWords ? at : v.setWords];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        for (uint32_t c = 0; c < nch; c++) {
            const uint32_t t = base + 64u * c + (uint32_t)lane;
            const uint32_t b1 = (uint32_t)__shfl((int)B1, (int)(8u * c) + (lane >> 3));
            const bool pass1 = t < cnt && (b1 & (0x80u >> (lane & 7))) != 0u;
            const unsigned long long bal1 = __ballot(pass1);
            const uint32_t rel = (r1start & 7u) + (r1 - r1start) + (uint32_t)__popcll(bal1 & ltMask);
            r1 += (uint32_t)__popcll(bal1);
            const uint32_t two = (uint32_t)__shfl((int)B2, (int)((rel >> 3) < 63u ? (rel >> 3) : 63u));
            const uint32_t b2 = ((rel >> 3) == 64u) ? (two >> 8) : (two & 0xFFu);
            const bool pass = pass1 && (m2 ? (b2 & (0x80u >> (rel & 7u))) != 0u : true);
            const unsigned long long bal = __ballot(pass);
            if (WRITE && pass) {
                const uint32_t bIdx = t * v.bits - (firstWord << 5), w = bIdx >> 5, bit = bIdx & 31u;
                const uint32_t w0 = s_words[w], w1 = s_words[w + 1];
                s_ids[nIds + (uint32_t)__popcll(bal & ltMask)] = ((w0 << bit) >> (32u - v.bits)) | (uint32_t)((unsigned long long)w1 >> (64u - (bit + v.bits)));
            }
            nIds += (uint32_t)__popcll(bal);
        }
        base += 64u * nch;
        if (WRITE) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            for (uint32_t k = (uint32_t)lane; k < nIds; k += 64u) lists[outBase + total + k] = s_ids[k];
        }
        total += nIds;
    }
    if (!WRITE && lane == 0) counts[node] = total;
}
__global__ void k_exact_list_table(const uint32_t* __restrict__ counts, const uint64_t* __restrict__ offsets, uint32_t n, uint2* __restrict__ leafList) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) leafList[i] = make_uint2((uint32_t)offsets[i], counts[i]);
}
__global__ void k_widen(const uint32_t* __restrict__ in, uint32_t n, uint64_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i];
}

template <bool GRAD>
__global__ void __launch_bounds__(64) k_exact_lists(ExactView v, const float* __restrict__ pts, uint64_t n, const uint32_t* __restrict__ skey, const uint32_t* __restrict__ sidx,
                                                    const uint2* __restrict__ leafList, const uint32_t* __restrict__ lists, float* __restrict__ dist, float* __restrict__ grad,
                                                    uint32_t* __restrict__ tri) {
    __shared__ float4 s_tile[64 * 5];                                  // one tile: 64 packed frames of 80 bytes
    const int lane = threadIdx.x;
    const uint64_t e = (uint64_t)xcdLogicalBlock() * 64u + (uint64_t)lane;       // queries are sorted by leaf: a contiguous range of leaves per XCD
    uint32_t myKey = v.numNodes, q = 0;
    if (e < n) { myKey = skey[e]; q = sidx[e]; }
    const bool active = myKey < v.numNodes;                             // sorted: the active lanes are a prefix of the wave
    F3 p = F3{0.f, 0.f, 0.f};
    uint32_t lOff = 0, lCnt = 0;
    if (active) {
        const uint2 c = leafList[myKey];
        lOff = c.x; lCnt = c.y;
        p = F3{pts[3 * (size_t)q], pts[3 * (size_t)q + 1], pts[3 * (size_t)q + 2]};
    }
    // runs: a leader per distinct leaf; tiles of all runs numbered through
    const uint32_t prevKey = __shfl_up(myKey, 1);
    const bool isLeader = active && (lane == 0 || prevKey != myKey);
    const unsigned long long leaders = __ballot(isLeader), act = __ballot(active);
    const int nActive = __popcll(act);
    uint32_t rlen = 0;
    if (isLeader) {
        const unsigned long long after = (lane == 63) ? 0ull : (leaders >> (lane + 1));
        const int next = after ? lane + 1 + (__ffsll((long long)after) - 1) : nActive;
        rlen = (uint32_t)(next - lane);
    }
    const uint32_t nT = isLeader ? (lCnt + 63u) / 64u : 0u;
    uint32_t tileBase = nT;                                              // inclusive scan over the lanes, made exclusive below
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(tileBase, d); if (lane >= d) tileBase += o; }
    const uint32_t T = __shfl(tileBase, 63);
    tileBase -= nT;
    unsigned long long acc = EX_KEY_NONE;                // this lane's own query: (distance bits, position in its leaf's list) of the nearest so far
    uint32_t accTri = 0;
    // tile t -> its run (leader lane, list offset, entries, queries) and its first entry
    struct TileRef { int L; uint32_t off, cnt, r, tb; };
    auto tileOf = [&](uint32_t t) {
        const unsigned long long m = __ballot(isLeader && t >= tileBase && t < tileBase + nT);
        TileRef R; R.L = __ffsll((long long)m) - 1;
        R.off = __shfl(lOff, R.L); R.cnt = __shfl(lCnt, R.L); R.r = __shfl(rlen, R.L); R.tb = (t - __shfl(tileBase, R.L)) * 64u;
        return R;
    };
    auto loadId = [&](const TileRef& R) { return (R.tb + (uint32_t)lane < R.cnt) ? lists[(size_t)R.off + R.tb + (uint32_t)lane] : 0u; };
    // the tile in flight: five registers of 16 bytes per lane (named, not an array: the compiler kept an array in scratch)
    float4 fr0 = make_float4(0.f, 0.f, 0.f, 0.f), fr1 = fr0, fr2 = fr0, fr3 = fr0, fr4 = fr0;
#define SDF_EX_GATHER(R, id)                                                                                     \
    if ((R).tb + (uint32_t)lane < (R).cnt) {                                                                     \
        const float4* src_ = reinterpret_cast<const float4*>(v.frames) + 5 * (size_t)(id);                       \
        fr0 = src_[0]; fr1 = src_[1]; fr2 = src_[2]; fr3 = src_[3]; fr4 = src_[4];                                    \
    }
    TileRef cur{}, nxt{}, nx2{};
    uint32_t idCur = 0, idNxt = 0, idNx2 = 0;
    if (T > 0) { cur = tileOf(0); idCur = loadId(cur); SDF_EX_GATHER(cur, idCur) }
    if (T > 1) { nxt = tileOf(1); idNxt = loadId(nxt); }
    for (uint32_t t = 0; t < T; t++) {
        const uint32_t nk = (cur.cnt - cur.tb < 64u) ? cur.cnt - cur.tb : 64u;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if ((uint32_t)lane < nk) {
            float4* dst = s_tile + 5 * lane;
            dst[0] = fr0; dst[1] = fr1; dst[2] = fr2; dst[3] = fr3; dst[4] = fr4;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        // behind the evaluation: the frames of the next tile, the ids of the one after it
        if (t + 1 < T) { SDF_EX_GATHER(nxt, idNxt) }
        if (t + 2 < T) { nx2 = tileOf(t + 2); idNx2 = loadId(nx2); }
        const int leader = cur.L;
        const uint32_t r = cur.r;
        const bool inRun = lane >= leader && lane < leader + (int)r;
        const uint32_t nsub = (r + 20u) / 21u, part = (r + nsub - 1u) / nsub;
        for (uint32_t sub = 0; sub < nsub; sub++) {
            const uint32_t lo = sub * part, sz = (r - lo < part) ? r - lo : part;
            const uint32_t G = 64u / sz, g = (uint32_t)lane / sz, i = (uint32_t)lane - g * sz;
            const int owner = leader + (int)(lo + i);
            const F3 po = F3{__shfl(p.x, owner & 63), __shfl(p.y, owner & 63), __shfl(p.z, owner & 63)};
            unsigned long long best = EX_KEY_NONE;
            if (g < G) {
                for (uint32_t sIdx = g; sIdx < nk; sIdx += G) {
                    const float4* fp = s_tile + 5 * sIdx;
                    const float4 a = fp[0], bq = fp[1], c = fp[2], d4 = fp[3], e4 = fp[4];
                    TriFrame fr;
                    fr.origin = F3{a.x, a.y, a.z};
                    fr.m[0] = a.w; fr.m[1] = bq.x; fr.m[2] = bq.y; fr.m[3] = bq.z; fr.m[4] = bq.w; fr.m[5] = c.x; fr.m[6] = c.y; fr.m[7] = c.z; fr.m[8] = c.w;
                    fr.b = F2{d4.x, d4.y}; fr.c = F2{d4.z, d4.w}; fr.v2 = e4.x; fr.v3 = F2{e4.y, e4.z};
                    const float d = sqDistPointTriangleSelect(po, fr);
                    const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (cur.tb + sIdx);
                    best = key < best ? key : best;
                }
            }
            // reduce over g (lanes i, i + sz, i + 2 sz, ...), result in lanes [0, sz); hand it to the queries' own lanes
            for (uint32_t span = 1; span < G; span <<= 1) {
                const int src = lane + (int)(span * sz);
                const unsigned long long other = shflKey(best, src < 64 ? src : lane);
                if (src < 64 && other < best) best = other;
            }
            const int from = lane - leader - (int)lo;              // lane of query (lo + from) holds it after the reduction
            const bool mineValid = inRun && from >= 0 && from < (int)sz;
            const unsigned long long mine = shflKey(best, mineValid ? from : lane);
            const uint32_t cand = __shfl(idCur, (int)(((uint32_t)mine - cur.tb) & 63u));
            if (mineValid && mine < acc) { acc = mine; accTri = cand; }
        }
        cur = nxt; idCur = idNxt; nxt = nx2; idNxt = idNx2;
    }
#undef SDF_EX_GATHER
    if (active) {
        const uint32_t bestTri = accTri;
        if (GRAD) {
            F3 gr;
            dist[q] = signedDistPointTriangleGradLocal(p, v.td + (size_t)TD_FLOATS * bestTri, gr);
            grad[3 * (size_t)q] = gr.x; grad[3 * (size_t)q + 1] = gr.y; grad[3 * (size_t)q + 2] = gr.z;
        } else dist[q] = signedDistPointTriangle(p, v.td + (size_t)TD_FLOATS * bestTri);
        if (tri) tri[q] = bestTri;
    }
}

}  // namespace sdfhip

using namespace sdfhip;

// The per-node table of the batched query (exact_internal.h): one level-synchronous walk of the tree, once per tree.
static int ensureLeafCtx(sdfhip_exact* T, const ExactView& v) {
    std::lock_guard<std::mutex> own(T->leafCtxLock);
    if (T->leafCtxReady) return SDFHIP_OK;
    hipStream_t st = T->ctx->stream;
    const sdfhip_exact_info& I = T->info;
    const uint64_t nn = I.num_nodes;
    SDF_REQUIRE(nn < (1ull << 31), "tree too large for the batched query");
    SDF_TRY(T->leafCtx.reserve(4 * nn));
    SDF_HIP_CHECK(hipMemsetAsync(T->leafCtx.p, 0xFF, 16 * nn, st));
    DevBuf<uint32_t> fa, fb, cnt;                      // two frontiers of {node, stage, set, first mask} and the size of the next one
    SDF_TRY(fa.reserve(4 * nn)); SDF_TRY(fb.reserve(4 * nn)); SDF_TRY(cnt.reserve(1));
    const uint32_t G = (uint32_t)I.start_grid_size, g3 = G * G * G;
    std::vector<uint32_t> seed(4 * (size_t)g3);
    for (uint32_t i = 0; i < g3; i++) { seed[4 * i] = i; seed[4 * i + 1] = 0; seed[4 * i + 2] = QNONE; seed[4 * i + 3] = QNONE; }
    SDF_HIP_CHECK(hipMemcpyAsync(fa.p, seed.data(), 16 * (size_t)g3, hipMemcpyHostToDevice, st));
    uint32_t nIn = g3;
    for (uint32_t depth = I.start_depth; nIn > 0; depth++) {
        SDF_HIP_CHECK(hipMemsetAsync(cnt.p, 0, 4, st));
        k_exact_ctx_level<<<gridFor(nIn, 256), 256, 0, st>>>(v, reinterpret_cast<const CtxItem*>(fa.p), nIn, depth, reinterpret_cast<CtxItem*>(fb.p), cnt.p, T->leafCtx.p, (uint32_t)nn);
        SDF_HIP_CHECK(hipGetLastError());
        uint32_t nOut = 0;
        SDF_TRY(readBackWords(st, cnt.p, nullptr, 1, &nOut));
        SDF_REQUIRE(nOut <= nn, "node array is not a tree");
        std::swap(fa, fb);
        nIn = nOut;
        SDF_REQUIRE(depth < 64, "node array is not a tree");
    }
    T->leafCtxReady = true;
    return SDFHIP_OK;
}

// The decoded lists (k_exact_leaf_lists): count, scan, write — once per tree, after leafCtx.  The lists are an ACCELERATION: when they
// do not fit (more than SDFHIP_EXACT_LISTS_MB, default 4096, or more than half of what the device has free) or an allocation for them
// fails (a co-tenanted or nearly full device), the tree answers through the decoding kernel as it did before they existed.
static int makeLeafLists(sdfhip_exact* T, const ExactView& v, hipStream_t st) {
    const uint32_t nn = (uint32_t)T->info.num_nodes;
    static const uint64_t capMB = [] { const char* e = getenv("SDFHIP_EXACT_LISTS_MB"); return e ? (uint64_t)strtoull(e, nullptr, 10) : 4096ull; }();
    const uint4* lc = reinterpret_cast<const uint4*>(T->leafCtx.p);
    DevBuf<uint32_t> counts; DevBuf<uint64_t> wide, offsets; DevBuf<unsigned char> tmp;
    SDF_TRY(counts.reserve(nn)); SDF_TRY(wide.reserve(nn)); SDF_TRY(offsets.reserve(nn));
    k_exact_leaf_lists<false><<<nn, 64, 0, st>>>(v, lc, counts.p, nullptr, nullptr);
    k_widen<<<gridFor(nn, 256), 256, 0, st>>>(counts.p, nn, wide.p);
    size_t tb = 0;
    SDF_HIP_CHECK(devExclusiveSum(nullptr, tb, wide.p, offsets.p, (size_t)nn, st));
    SDF_TRY(tmp.reserve(tb));
    SDF_HIP_CHECK(devExclusiveSum(tmp.p, tb, wide.p, offsets.p, (size_t)nn, st));
    uint64_t lastOff = 0; uint32_t lastCnt = 0;
    SDF_HIP_CHECK(hipMemcpyAsync(&lastOff, offsets.p + (nn - 1), 8, hipMemcpyDeviceToHost, st));
    SDF_HIP_CHECK(hipMemcpyAsync(&lastCnt, counts.p + (nn - 1), 4, hipMemcpyDeviceToHost, st));
    SDF_HIP_CHECK(hipStreamSynchronize(st));
    const uint64_t total = lastOff + lastCnt;
    size_t freeB = 0, totalB = 0;
    if (hipMemGetInfo(&freeB, &totalB) != hipSuccess) { (void)hipGetLastError(); freeB = ~(size_t)0; }
    if (total >= (1ull << 32) || 4 * total > (capMB << 20) || 4 * total > freeB / 2) { T->listsState = 2; return SDFHIP_OK; }
    SDF_TRY(T->leafLists.reserve(total + 64)); SDF_TRY(T->leafList.reserve(2ull * nn));
    k_exact_leaf_lists<true><<<nn, 64, 0, st>>>(v, lc, nullptr, offsets.p, T->leafLists.p);
    k_exact_list_table<<<gridFor(nn, 256), 256, 0, st>>>(counts.p, offsets.p, nn, reinterpret_cast<uint2*>(T->leafList.p));
    SDF_HIP_CHECK(hipGetLastError());
    SDF_HIP_CHECK(hipStreamSynchronize(st));          // the temporaries die with this scope
    T->listEntries = total; T->listsState = 1;
    return SDFHIP_OK;
}
static int ensureLeafLists(sdfhip_exact* T, const ExactView& v) {
    std::lock_guard<std::mutex> own(T->leafCtxLock);
    if (T->listsState != 0) return SDFHIP_OK;
    if (makeLeafLists(T, v, T->ctx->stream) != SDFHIP_OK) {
        // no room for the lists: what was reserved goes back, the decoding kernel answers (and the failed allocation is not the caller's error)
        (void)hipGetLastError();
        T->leafLists.release(); T->leafList.release();
        T->listEntries = 0; T->listsState = 2;
        SDF_HIP_CHECK(hipStreamSynchronize(T->ctx->stream));
    }
    return SDFHIP_OK;
}

extern "C" {

int sdfhip_exact_query(sdfhip_exact* T, const float* xyz, uint64_t n, float* out_dist, float* out_grad, uint32_t* out_tri, int where) {
    SDF_API_BEGIN
    SDF_REQUIRE(T && xyz && out_dist, "NULL argument");
    SDF_REQUIRE(T->built, "tree is not built");
    if (n == 0) return SDFHIP_OK;
    const uint64_t chunk = queryChunk(where == SDFHIP_HOST);
    if (n > chunk) {
        for (uint64_t off = 0; off < n; off += chunk) {
            const uint64_t m = n - off < chunk ? n - off : chunk;
            SDF_TRY(sdfhip_exact_query(T, xyz + 3 * off, m, out_dist + off, out_grad ? out_grad + 3 * off : nullptr, out_tri ? out_tri + off : nullptr, where));
        }
        return SDFHIP_OK;
    }
    sdfhip_ctx* ctx = T->ctx;
    SDF_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    if (where == SDFHIP_HOST && n <= kHostScalarMax) {
        // the scalar getDistance of the reference's API (and any handful of points): answered by the same code on host copies of the arrays
        SDF_TRY(ensureHostCopy(T));
        const sdfhip_exact_info& I = T->info;
        const ExactView hv{T->hNodes.data(), T->hSets.data(), T->hMasks.data(), T->hTri.data(), nullptr, I.box_min[0], I.box_min[1], I.box_min[2], I.box_max[0], I.box_max[1], I.box_max[2],
                           T->cellSize, I.start_grid_size, I.start_depth, I.bit_encoding_start_depth, I.bits_per_index, (uint32_t)I.num_nodes, I.num_mask_bytes, I.num_set_words};
        for (uint64_t i = 0; i < n; i++) {
            const F3 p = F3{xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
            uint32_t t = 0;
            if (out_grad) { out_grad[3 * i] = 0.f; out_grad[3 * i + 1] = 0.f; out_grad[3 * i + 2] = 0.f; out_dist[i] = exactOne<true, true>(hv, p, out_grad + 3 * i, &t); }
            else out_dist[i] = exactOne<false, true>(hv, p, nullptr, &t);
            if (out_tri) out_tri[i] = t;
        }
        return SDFHIP_OK;
    }
    std::unique_lock<std::mutex> stageLock(ctx->stage.lock, std::defer_lock);
    if (where == SDFHIP_HOST && 12 * n <= sdfhip_stage::kStageKeepBytes) stageLock.try_lock();
    DevBuf<float> pp, pd, pg; DevBuf<uint32_t> pt;
    DevBuf<float>& dp = stageLock.owns_lock() ? ctx->stage.pts : pp; DevBuf<float>& dd = stageLock.owns_lock() ? ctx->stage.dist : pd;
    DevBuf<float>& dg = stageLock.owns_lock() ? ctx->stage.grad : pg; DevBuf<uint32_t>& dt = stageLock.owns_lock() ? ctx->stage.ids : pt;
    const float* p = xyz; float* d = out_dist; float* g = out_grad; uint32_t* t = out_tri;
    if (where == SDFHIP_HOST) {
        SDF_TRY(dp.reserve(3 * n)); SDF_TRY(dd.reserve(n));
        if (out_grad) SDF_TRY(dg.reserve(3 * n));
        if (out_tri) SDF_TRY(dt.reserve(n));
        SDF_HIP_CHECK(hipMemcpyAsync(dp.p, xyz, 12 * n, hipMemcpyHostToDevice, st));
        p = dp.p; d = dd.p; g = out_grad ? dg.p : nullptr; t = out_tri ? dt.p : nullptr;
        if (out_grad) SDF_HIP_CHECK(hipMemsetAsync(dg.p, 0, 12 * n, st));
    }
    const sdfhip_exact_info& I = T->info;
    ExactView v{T->nodes.p, T->sets.p, T->masks.p, T->tri(), T->frames(), I.box_min[0], I.box_min[1], I.box_min[2], I.box_max[0], I.box_max[1], I.box_max[2],
                T->cellSize, I.start_grid_size, I.start_depth, I.bit_encoding_start_depth, I.bits_per_index, (uint32_t)I.num_nodes, I.num_mask_bytes, I.num_set_words};
    if (n < 16384) {
        if (g) k_exact_query<true><<<gridFor(n, 256), 256, 0, st>>>(v, p, n, d, g, t);
        else k_exact_query<false><<<gridFor(n, 256), 256, 0, st>>>(v, p, n, d, nullptr, t);
    } else {
        // the tree's own scratch if no other host thread is using it (all work is ordered on the context's stream, so it can be
        // handed on as soon as this call has enqueued its kernels); a private one otherwise
        std::unique_lock<std::mutex> own(T->scratch.lock, std::try_to_lock);
        sdfhip_exact_scratch priv;
        sdfhip_exact_scratch& S = own.owns_lock() ? T->scratch : priv;
        DevBuf<uint32_t>&key = S.key, &keyS = S.keyS, &qi = S.qi, &qiS = S.qiS; DevBuf<unsigned char>& tmp = S.tmp;
        SDF_TRY(ensureLeafCtx(T, v));
        SDF_TRY(ensureLeafLists(T, v));
        SDF_TRY(key.reserve(n)); SDF_TRY(keyS.reserve(n)); SDF_TRY(qi.reserve(n)); SDF_TRY(qiS.reserve(n));
        k_exact_locate<<<gridFor(n, 256), 256, 0, st>>>(v, p, n, d, g, t, key.p, qi.p);
        // keys are node ids, num_nodes for a query outside the grid (sorted last, skipped): only the bits such keys have are sorted on
        // (21 instead of 32 at a million nodes: three radix passes instead of four)
        int keyBits = 1; while (keyBits < 32 && (1ull << keyBits) <= T->info.num_nodes) keyBits++;
        size_t tb = 0;
        SDF_HIP_CHECK(devSortPairs(nullptr, tb, key.p, keyS.p, qi.p, qiS.p, (size_t)n, 0, (unsigned)keyBits, st));
        SDF_TRY(tmp.reserve(tb));
        SDF_HIP_CHECK(devSortPairs(tmp.p, tb, key.p, keyS.p, qi.p, qiS.p, (size_t)n, 0, (unsigned)keyBits, st));
        const uint4* lc = reinterpret_cast<const uint4*>(T->leafCtx.p);
        if (T->listsState == 1) {
            const uint2* ll = reinterpret_cast<const uint2*>(T->leafList.p);
            if (g) k_exact_lists<true><<<xcdGrid(gridFor(n, 64)), 64, 0, st>>>(v, p, n, keyS.p, qiS.p, ll, T->leafLists.p, d, g, t);
            else k_exact_lists<false><<<xcdGrid(gridFor(n, 64)), 64, 0, st>>>(v, p, n, keyS.p, qiS.p, ll, T->leafLists.p, d, nullptr, t);
        }
        else if (g) k_exact_tiles<true><<<xcdGrid(gridFor(n, 64)), 64, 0, st>>>(v, p, n, keyS.p, qiS.p, lc, d, g, t);
        else k_exact_tiles<false><<<xcdGrid(gridFor(n, 64)), 64, 0, st>>>(v, p, n, keyS.p, qiS.p, lc, d, nullptr, t);
        SDF_HIP_CHECK(hipGetLastError());
        if (!own.owns_lock()) SDF_HIP_CHECK(hipStreamSynchronize(st));      // the private scratch dies with this scope
    }
    SDF_HIP_CHECK(hipGetLastError());
    if (where == SDFHIP_HOST) {
        SDF_HIP_CHECK(hipMemcpyAsync(out_dist, d, 4 * n, hipMemcpyDeviceToHost, st));
        if (out_grad) SDF_HIP_CHECK(hipMemcpyAsync(out_grad, g, 12 * n, hipMemcpyDeviceToHost, st));
        if (out_tri) SDF_HIP_CHECK(hipMemcpyAsync(out_tri, t, 4 * n, hipMemcpyDeviceToHost, st));
        SDF_HIP_CHECK(hipStreamSynchronize(st));
    }
    return SDFHIP_OK;
    SDF_API_END
}

int sdfhip_exact_set_start_grid_cell_size(sdfhip_exact* tree, float cell_size) {
    SDF_API_BEGIN
    SDF_REQUIRE(tree && cell_size > 0.f, "bad argument");
    std::lock_guard<std::mutex> own(tree->leafCtxLock);
    SDF_REQUIRE(!tree->leafCtxReady, "the cell size must be set before the first batched query");
    tree->cellSize = cell_size; tree->info.start_grid_cell_size = cell_size;
    return SDFHIP_OK;
    SDF_API_END
}

int sdfhip_exact_from_data(sdfhip_ctx* ctx, const sdfhip_exact_info* info, const uint32_t* nodes, const uint32_t* sets, const uint8_t* masks,
                           const float* triangle_data, sdfhip_exact** out) {
    SDF_API_BEGIN
    SDF_REQUIRE(ctx && info && nodes && sets && masks && triangle_data && out, "NULL argument");
    SDF_REQUIRE(info->start_grid_size >= 1 && info->num_nodes >= (uint64_t)info->start_grid_size * info->start_grid_size * info->start_grid_size, "start grid does not fit");
    SDF_REQUIRE(info->bits_per_index >= 1 && info->bits_per_index <= 32 && info->num_triangles >= 1, "bad header");
    SDF_REQUIRE(info->max_depth >= 2 && info->max_depth - info->bit_encoding_start_depth == 2, "unsupported file: the query decodes exactly two mask levels (bitEncodingStartDepth must be maxDepth - 2)");
    SDF_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    std::unique_ptr<sdfhip_exact> E(new sdfhip_exact());
    E->ctx = ctx; E->info = *info;
    E->cellSize = (info->box_max[0] - info->box_min[0]) / (float)info->start_grid_size;     // load(): mBox.getSize().x / mStartGridSize
    E->info.start_grid_cell_size = E->cellSize;
    SDF_TRY(E->nodes.reserve(2 * info->num_nodes)); SDF_TRY(E->hasTri.reserve(info->num_nodes));
    SDF_TRY(E->sets.reserve(info->num_set_words + 2)); SDF_TRY(E->masks.reserve(info->num_mask_bytes + 1)); SDF_TRY(E->ownTri.reserve((size_t)TD_FLOATS * info->num_triangles));
    SDF_HIP_CHECK(hipMemsetAsync(E->hasTri.p, 1, info->num_nodes, st));
    SDF_HIP_CHECK(hipMemsetAsync(E->sets.p, 0, 4 * (info->num_set_words + 2), st));
    SDF_HIP_CHECK(hipMemcpyAsync(E->nodes.p, nodes, 8 * info->num_nodes, hipMemcpyHostToDevice, st));
    if (info->num_set_words) SDF_HIP_CHECK(hipMemcpyAsync(E->sets.p, sets, 4 * info->num_set_words, hipMemcpyHostToDevice, st));
    if (info->num_mask_bytes) SDF_HIP_CHECK(hipMemcpyAsync(E->masks.p, masks, info->num_mask_bytes, hipMemcpyHostToDevice, st));
    SDF_HIP_CHECK(hipMemcpyAsync(E->ownTri.p, triangle_data, sizeof(float) * TD_FLOATS * info->num_triangles, hipMemcpyHostToDevice, st));
    SDF_TRY(E->ownFrames.reserve((size_t)FRAME_FLOATS * info->num_triangles));
    SDF_TRY(packFrames(st, E->ownTri.p, (uint32_t)info->num_triangles, E->ownFrames.p));
    SDF_HIP_CHECK(hipStreamSynchronize(st));
    E->built = true;
    *out = E.release();
    return SDFHIP_OK;
    SDF_API_END
}

int sdfhip_exact_from_parts(sdfhip_ctx* ctx, sdfhip_mesh* mesh, const sdfhip_exact_info* info, const uint32_t* nodes, const uint8_t* has, const uint32_t* sets,
                            const uint8_t* masks, int where, sdfhip_exact** out) {
    SDF_API_BEGIN
    SDF_REQUIRE(ctx && mesh && info && nodes && sets && masks && out, "NULL argument");
    SDF_REQUIRE(mesh->ctx == ctx && info->num_triangles == mesh->numTriangles, "mesh does not match the header");
    SDF_REQUIRE(info->start_grid_size >= 1 && info->num_nodes >= (uint64_t)info->start_grid_size * info->start_grid_size * info->start_grid_size, "start grid does not fit");
    SDF_REQUIRE(info->bits_per_index >= 1 && info->bits_per_index <= 32, "bad header");
    SDF_REQUIRE(info->max_depth >= 2 && info->max_depth - info->bit_encoding_start_depth == 2, "unsupported header: the query decodes exactly two mask levels (bitEncodingStartDepth must be maxDepth - 2)");
    SDF_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const hipMemcpyKind kind = where == SDFHIP_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
    std::unique_ptr<sdfhip_exact> E(new sdfhip_exact());
    E->ctx = ctx; E->mesh = mesh; E->info = *info;
    // parts of a BUILD carry the build's cell size; without it the loaded tree's (ExactOctreeSdf.h load())
    E->cellSize = info->start_grid_cell_size > 0.f ? info->start_grid_cell_size : (info->box_max[0] - info->box_min[0]) / (float)info->start_grid_size;
    E->info.start_grid_cell_size = E->cellSize;
    SDF_TRY(E->nodes.reserve(2 * info->num_nodes)); SDF_TRY(E->hasTri.reserve(info->num_nodes));
    SDF_TRY(E->sets.reserve(info->num_set_words + 2)); SDF_TRY(E->masks.reserve(info->num_mask_bytes + 1));
    SDF_HIP_CHECK(hipMemsetAsync(E->sets.p, 0, 4 * (info->num_set_words + 2), st));
    SDF_HIP_CHECK(hipMemcpyAsync(E->nodes.p, nodes, 8 * info->num_nodes, kind, st));
    if (has) SDF_HIP_CHECK(hipMemcpyAsync(E->hasTri.p, has, info->num_nodes, kind, st));
    else SDF_HIP_CHECK(hipMemsetAsync(E->hasTri.p, 1, info->num_nodes, st));
    if (info->num_set_words) SDF_HIP_CHECK(hipMemcpyAsync(E->sets.p, sets, 4 * info->num_set_words, kind, st));
    if (info->num_mask_bytes) SDF_HIP_CHECK(hipMemcpyAsync(E->masks.p, masks, info->num_mask_bytes, kind, st));
    SDF_HIP_CHECK(hipStreamSynchronize(st));
    E->built = true;
    *out = E.release();
    return SDFHIP_OK;
    SDF_API_END
}

int sdfhip_exact_triangle_data(sdfhip_exact* tree, float* out_host) {
    SDF_API_BEGIN
    SDF_REQUIRE(tree && tree->built && out_host, "NULL argument or tree not built");
    SDF_HIP_CHECK(hipMemcpyAsync(out_host, tree->tri(), sizeof(float) * TD_FLOATS * tree->info.num_triangles, hipMemcpyDeviceToHost, tree->ctx->stream));
    SDF_HIP_CHECK(hipStreamSynchronize(tree->ctx->stream));
    return SDFHIP_OK;
    SDF_API_END
}

}  // extern "C"

// (sdfhip_ctx_create: the runtime loads a translation unit's code object on the first use of one of its kernels — milliseconds that would
// otherwise land in the first build or the first query of a process)
namespace sdfhip { void loadKernelsExactQuery() { hipFuncAttributes a; (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_widen)); (void)hipGetLastError(); } }
