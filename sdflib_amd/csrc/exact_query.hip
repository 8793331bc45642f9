// Batched ExactOctreeSdf::getDistance on the device.  PRODUCT code — independent of oracle/.
//
// Reference behaviour reproduced: ExactOctreeSdf::getDistance(vec3) / (vec3, vec3&) (src/sdf/ExactOctreeSdf.cpp:38-178,
// 180-320): descend to the bit-encoding depth, decode the bit-packed triangle set, filter it through the byte masks of
// the last two levels, brute-force the nearest triangle (first minimum in ascending id), sign by the pseudonormal;
// roundFloat is '> 0.5' here (:33-36).  The reference decodes into mutable scratch vectors (not re-entrant); this kernel
// streams the set through the mask chain with two running rank counters instead, so it needs no scratch at all.
// Compile with -ffp-contract=off.
#include "exact_internal.h"
#include <memory>

namespace sdfhip {

struct ExactView {
    const uint32_t* nodes; const uint32_t* sets; const uint8_t* masks; const float* td; const float* frames;
    float bminx, bminy, bminz, bmaxx, bmaxy, bmaxz, cellSize;
    int G; uint32_t startDepth, bitEnc, bits;
};

SDF_DEV uint32_t unpackIndex(const uint32_t* __restrict__ set, uint32_t bIdx, uint32_t bits) {
    const uint32_t w = bIdx >> 5, bit = bIdx & 31u;
    return ((set[w] << bit) >> (32u - bits)) | (uint32_t)((unsigned long long)set[w + 1] >> (64u - (bit + bits)));
}
SDF_DEV bool maskBit(const uint8_t* __restrict__ m, uint32_t k) { return (m[k >> 3] & (0x80u >> (k & 7u))) != 0; }

template <bool GRAD>
__global__ void __launch_bounds__(256) k_exact_query(ExactView v, const float* __restrict__ pts, uint64_t n, float* __restrict__ dist, float* __restrict__ grad,
                                                     uint32_t* __restrict__ tri) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
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
        TriFrame fr; loadFramePacked(v.frames, ti, fr);
        const float d = sqDistPointTriangle(p, fr);
        if (d < best) { best = d; bestTri = ti; }
    }
    if (GRAD) {
        F3 g;
        dist[i] = signedDistPointTriangleGradLocal(p, v.td + (size_t)TD_FLOATS * bestTri, g);
        grad[3 * i] = g.x; grad[3 * i + 1] = g.y; grad[3 * i + 2] = g.z;
    } else dist[i] = signedDistPointTriangle(p, v.td + (size_t)TD_FLOATS * bestTri);
    if (tri) tri[i] = bestTri;
}

}  // namespace sdfhip

using namespace sdfhip;

extern "C" {

int sdfhip_exact_query(sdfhip_exact* T, const float* xyz, uint64_t n, float* out_dist, float* out_grad, uint32_t* out_tri, int where) {
    SDF_REQUIRE(T && xyz && out_dist, "NULL argument");
    SDF_REQUIRE(T->built, "tree is not built");
    if (n == 0) return SDFHIP_OK;
    sdfhip_ctx* ctx = T->ctx;
    SDF_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    DevBuf<float> dp, dd, dg; DevBuf<uint32_t> dt;
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
    if (g) k_exact_query<true><<<gridFor(n, 256), 256, 0, st>>>(v, p, n, d, g, t);
    else k_exact_query<false><<<gridFor(n, 256), 256, 0, st>>>(v, p, n, d, nullptr, t);
    SDF_HIP_CHECK(hipGetLastError());
    if (where == SDFHIP_HOST) {
        SDF_HIP_CHECK(hipMemcpyAsync(out_dist, d, 4 * n, hipMemcpyDeviceToHost, st));
        if (out_grad) SDF_HIP_CHECK(hipMemcpyAsync(out_grad, g, 12 * n, hipMemcpyDeviceToHost, st));
        if (out_tri) SDF_HIP_CHECK(hipMemcpyAsync(out_tri, t, 4 * n, hipMemcpyDeviceToHost, st));
        SDF_HIP_CHECK(hipStreamSynchronize(st));
    }
    return SDFHIP_OK;
}

int sdfhip_exact_from_data(sdfhip_ctx* ctx, const sdfhip_exact_info* info, const uint32_t* nodes, const uint32_t* sets, const uint8_t* masks,
                           const float* triangle_data, sdfhip_exact** out) {
    SDF_REQUIRE(ctx && info && nodes && sets && masks && triangle_data && out, "NULL argument");
    SDF_REQUIRE(info->start_grid_size >= 1 && info->num_nodes >= (uint64_t)info->start_grid_size * info->start_grid_size * info->start_grid_size, "start grid does not fit");
    SDF_REQUIRE(info->bits_per_index >= 1 && info->bits_per_index <= 32 && info->num_triangles >= 1, "bad header");
    SDF_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    std::unique_ptr<sdfhip_exact> E(new sdfhip_exact());
    E->ctx = ctx; E->info = *info;
    E->cellSize = (info->box_max[0] - info->box_min[0]) / (float)info->start_grid_size;     // load(): mBox.getSize().x / mStartGridSize
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
}

}  // extern "C"
