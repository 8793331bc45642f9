// Batched ExactOctreeSdf::getDistance on the device.  PRODUCT code — independent of oracle/.
//
// Reference behaviour reproduced: ExactOctreeSdf::getDistance(vec3) / (vec3, vec3&) (src/sdf/ExactOctreeSdf.cpp:38-178,
// 180-320): descend to the bit-encoding depth, decode the bit-packed triangle set, filter it through the byte masks of
// the last two levels, brute-force the nearest triangle (first minimum in ascending id), sign by the pseudonormal;
// roundFloat is '> 0.5' here (:33-36).  The reference decodes into mutable scratch vectors (not re-entrant); this kernel
// streams the set through the mask chain with two running rank counters instead, so it needs no scratch at all.
// Compile with -ffp-contract=off.
#include "exact_internal.h"
#include <hipcub/hipcub.hpp>
#include <memory>

namespace sdfhip {

struct ExactView {
    const uint32_t* nodes; const uint32_t* sets; const uint8_t* masks; const float* td; const float* frames;
    float bminx, bminy, bminz, bmaxx, bmaxy, bmaxz, cellSize;
    int G; uint32_t startDepth, bitEnc, bits;
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
// k_exact_locate walks every query down to its leaf and records (leaf id, set / mask positions); queries are then
// radix-sorted by leaf id so that the queries of one leaf sit next to each other.  k_exact_sorted gives each wave 64
// consecutive sorted queries: for every distinct leaf in them the wave decodes the leaf's triangle list ONCE (bit-packed
// set filtered through the byte masks with __ballot/popcount ranks), stages the surviving triangles' 80-byte frames in LDS
// (each triangle fetched once per wave instead of once per query lane) and all lanes of that leaf scan the staged tile
// in list order — so the first-minimum rule of the reference is preserved.  The lanes of the wave that do not hold a query of
// the current leaf help with its triangles (see the comment in the kernel).
constexpr uint32_t QNONE = 0xFFFFFFFFu;

__global__ void __launch_bounds__(256) k_exact_locate(ExactView v, const float* __restrict__ pts, uint64_t n, float* __restrict__ dist, float* __restrict__ grad,
                                                      uint32_t* __restrict__ tri, uint32_t* __restrict__ key, uint32_t* __restrict__ qidx, uint32_t* __restrict__ qctx) {
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
        key[i] = QNONE;
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
    const uint32_t setIdx = v.nodes[2 * (size_t)node + 1];
    uint32_t m1 = QNONE, m2 = QNONE;
    if (!isLeaf(node)) {
        node = descend(node); m1 = v.nodes[2 * (size_t)node + 1];
        if (!isLeaf(node)) { node = descend(node); m2 = v.nodes[2 * (size_t)node + 1]; }
    }
    key[i] = node;
    qctx[3 * i] = setIdx; qctx[3 * i + 1] = m1; qctx[3 * i + 2] = m2;
}

template <bool GRAD>
__global__ void __launch_bounds__(256) k_exact_sorted(ExactView v, const float* __restrict__ pts, uint64_t n, const uint32_t* __restrict__ skey, const uint32_t* __restrict__ sidx,
                                                      const uint32_t* __restrict__ qctx, float* __restrict__ dist, float* __restrict__ grad, uint32_t* __restrict__ tri) {
    __shared__ float s_frames[4][64 * FRAME_FLOATS];
    __shared__ uint32_t s_tri[4][64];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint64_t e = (uint64_t)xcdLogicalBlock() * blockDim.x + threadIdx.x;       // queries are sorted by leaf: a contiguous range of leaves per XCD
    uint32_t myKey = QNONE, q = 0;
    if (e < n) { myKey = skey[e]; q = sidx[e]; }
    const bool active = myKey != QNONE;
    F3 p = F3{0.f, 0.f, 0.f};
    uint32_t cSet = 0, cM1 = QNONE, cM2 = QNONE;
    if (active) { p = F3{pts[3 * (size_t)q], pts[3 * (size_t)q + 1], pts[3 * (size_t)q + 2]}; cSet = qctx[3 * (size_t)q]; cM1 = qctx[3 * (size_t)q + 1]; cM2 = qctx[3 * (size_t)q + 2]; }
    bool done = !active;
    const unsigned long long ltMask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (;;) {
        const unsigned long long pending = __ballot(!done);
        if (pending == 0ull) break;
        const int leader = __ffsll((long long)pending) - 1;
        const uint32_t k = __shfl(myKey, leader);
        const bool inRun = !done && myKey == k;
        const uint32_t setIdx = __shfl(cSet, leader), m1i = __shfl(cM1, leader), m2i = __shfl(cM2, leader);
        const uint32_t* set = v.sets + setIdx;
        const uint32_t cnt = set[0];
        const uint8_t* m1 = (m1i != QNONE) ? v.masks + m1i : nullptr;
        const uint8_t* m2 = (m2i != QNONE) ? v.masks + m2i : nullptr;
        // The run's queries sit in consecutive lanes [leader, leader + r).  Runs are short (about 10 queries per leaf at 10 M
        // queries), so the (query, triangle) pairs of a tile are spread over the WHOLE wave: lane j works for the query of lane
        // leader + j % r on the staged triangles g, g + G, ... with g = j / r, G = 64 / r, and the partial minima are merged
        // afterwards by (distance, list position) — the first minimum in list order wins, exactly like the sequential scan.
        const uint32_t r = (uint32_t)__popcll(__ballot(inRun));
        const uint32_t G = 64u / r, g = (uint32_t)lane / r;
        const int owner = leader + (int)((uint32_t)lane % r);
        const F3 po = F3{__shfl(p.x, owner), __shfl(p.y, owner), __shfl(p.z, owner)};
        float best = INFINITY; uint32_t bestTri = 0, bestPos = 0xFFFFFFFFu;
        uint32_t r1 = 0, staged = 0;
        for (uint32_t base = 0; base < cnt; base += 64) {
            const uint32_t t = base + (uint32_t)lane;
            const bool valid = t < cnt;
            const bool pass1 = valid && (m1 ? maskBit(m1, t) : true);
            const unsigned long long b1 = __ballot(pass1);
            const uint32_t k1 = r1 + (uint32_t)__popcll(b1 & ltMask);
            const bool pass = pass1 && (m2 ? maskBit(m2, k1) : true);
            r1 += (uint32_t)__popcll(b1);
            const unsigned long long b = __ballot(pass);
            const uint32_t nk = (uint32_t)__popcll(b);
            if (nk == 0) continue;
            if (pass) {
                const uint32_t slot = (uint32_t)__popcll(b & ltMask);
                const uint32_t ti = unpackIndex(set + 1, t * v.bits, v.bits);
                s_tri[w][slot] = ti;
                const float4* src = reinterpret_cast<const float4*>(v.frames) + 5 * (size_t)ti;
                float4* dst = reinterpret_cast<float4*>(&s_frames[w][slot * FRAME_FLOATS]);
#pragma unroll
                for (int c = 0; c < 5; c++) dst[c] = src[c];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            if (g < G) {
                for (uint32_t sIdx = g; sIdx < nk; sIdx += G) {
                    const float4* fp = reinterpret_cast<const float4*>(&s_frames[w][sIdx * FRAME_FLOATS]);
                    const float4 a = fp[0], bq = fp[1], c = fp[2], d4 = fp[3], e4 = fp[4];
                    TriFrame fr;
                    fr.origin = F3{a.x, a.y, a.z};
                    fr.m[0] = a.w; fr.m[1] = bq.x; fr.m[2] = bq.y; fr.m[3] = bq.z; fr.m[4] = bq.w; fr.m[5] = c.x; fr.m[6] = c.y; fr.m[7] = c.z; fr.m[8] = c.w;
                    fr.b = F2{d4.x, d4.y}; fr.c = F2{d4.z, d4.w}; fr.v2 = e4.x; fr.v3 = F2{e4.y, e4.z};
                    const float d = sqDistPointTriangle(po, fr);
                    if (d < best) { best = d; bestTri = s_tri[w][sIdx]; bestPos = staged + sIdx; }
                }
            }
            staged += nk;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        {   // merge the G partial results of every query into its own lane
            const uint32_t i = inRun ? (uint32_t)(lane - leader) : 0u;
            float mb = INFINITY; uint32_t mt = 0, mp = 0xFFFFFFFFu;
            for (uint32_t gg = 0; gg < G; gg++) {
                const int src = (int)(i + r * gg);
                const float cd = __shfl(best, src); const uint32_t ct = __shfl(bestTri, src), cp = __shfl(bestPos, src);
                if (cd < mb || (cd == mb && cp < mp)) { mb = cd; mt = ct; mp = cp; }
            }
            best = mb; bestTri = mt;
        }
        if (inRun) {
            if (GRAD) {
                F3 g;
                dist[q] = signedDistPointTriangleGradLocal(p, v.td + (size_t)TD_FLOATS * bestTri, g);
                grad[3 * (size_t)q] = g.x; grad[3 * (size_t)q + 1] = g.y; grad[3 * (size_t)q + 2] = g.z;
            } else dist[q] = signedDistPointTriangle(p, v.td + (size_t)TD_FLOATS * bestTri);
            if (tri) tri[q] = bestTri;
            done = true;
        }
    }
}

}  // namespace sdfhip

using namespace sdfhip;

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
                           T->cellSize, I.start_grid_size, I.start_depth, I.bit_encoding_start_depth, I.bits_per_index};
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
                T->cellSize, I.start_grid_size, I.start_depth, I.bit_encoding_start_depth, I.bits_per_index};
    if (n < 16384) {
        if (g) k_exact_query<true><<<gridFor(n, 256), 256, 0, st>>>(v, p, n, d, g, t);
        else k_exact_query<false><<<gridFor(n, 256), 256, 0, st>>>(v, p, n, d, nullptr, t);
    } else {
        // the tree's own scratch if no other host thread is using it (all work is ordered on the context's stream, so it can be
        // handed on as soon as this call has enqueued its kernels); a private one otherwise
        std::unique_lock<std::mutex> own(T->scratch.lock, std::try_to_lock);
        sdfhip_exact_scratch priv;
        sdfhip_exact_scratch& S = own.owns_lock() ? T->scratch : priv;
        DevBuf<uint32_t>&key = S.key, &keyS = S.keyS, &qi = S.qi, &qiS = S.qiS, &qctx = S.qctx; DevBuf<unsigned char>& tmp = S.tmp;
        SDF_TRY(key.reserve(n)); SDF_TRY(keyS.reserve(n)); SDF_TRY(qi.reserve(n)); SDF_TRY(qiS.reserve(n)); SDF_TRY(qctx.reserve(3 * n));
        k_exact_locate<<<gridFor(n, 256), 256, 0, st>>>(v, p, n, d, g, t, key.p, qi.p, qctx.p);
        int keyBits = 1; while (keyBits < 32 && (1ull << keyBits) <= T->info.num_nodes) keyBits++;
        size_t tb = 0;
        SDF_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, key.p, keyS.p, qi.p, qiS.p, (int)n, 0, 32, st));
        SDF_TRY(tmp.reserve(tb));
        (void)keyBits;   // outside-the-grid queries carry key 0xFFFFFFFF: sort on all 32 bits so that they end up last
        SDF_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(tmp.p, tb, key.p, keyS.p, qi.p, qiS.p, (int)n, 0, 32, st));
        if (g) k_exact_sorted<true><<<xcdGrid(gridFor(n, 256)), 256, 0, st>>>(v, p, n, keyS.p, qiS.p, qctx.p, d, g, t);
        else k_exact_sorted<false><<<xcdGrid(gridFor(n, 256)), 256, 0, st>>>(v, p, n, keyS.p, qiS.p, qctx.p, d, nullptr, t);
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
