// Exact distance samples shared by the NO_CONTINUITY and CONTINUITY builders (device + host helpers).  PRODUCT code.
// Included by octree_build.hip and octree_continuity.hip; everything here has internal linkage.
#pragma once
#include "dev_bvh.h"
#include "dev_tricubic.h"
#include <hipcub/hipcub.hpp>

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

// ---- 19 mid-points of every node of a level: the build's hot path --------------------------------------------------
// Neighbouring nodes share mid-points (a face centre belongs to 2 nodes, an edge mid-point to up to 4), and the reference
// answers each of them with the same deterministic query.  The samples of a level are therefore grouped by lattice point
// (radix sort of a 36-bit key), one traversal is run per group of samples whose fp32 POSITION BITS are identical — the
// positions come from different node centres, and only equal bits guarantee the same answer — and every sample then
// computes its Hermite datum from the shared nearest-triangle id.  About 1.6x fewer traversals, issued in lattice order.
SDF_DEV F3 midPosition(const float* __restrict__ center, float half, uint32_t q) {
    const uint32_t node = q / 19u;
    return F3{center[3 * node], center[3 * node + 1], center[3 * node + 2]} + midRel((int)(q - 19u * node)) * half;
}
__global__ void k_mid_keys(const uint32_t* __restrict__ coord, uint32_t n, uint64_t* __restrict__ key, uint32_t* __restrict__ val) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= 19u * n) return;
    const uint32_t node = q / 19u, co = coord[node];
    const F3 r = midRel((int)(q - 19u * node));
    const uint64_t lx = 2u * (co & 1023u) + (uint32_t)(r.x + 1.f), ly = 2u * ((co >> 10) & 1023u) + (uint32_t)(r.y + 1.f), lz = 2u * (co >> 20) + (uint32_t)(r.z + 1.f);
    key[q] = lx | (ly << 12) | (lz << 24);
    val[q] = q;
}
// sorted entry j starts a new traversal unless it is the same lattice point AND the same position bits as entry j-1
__global__ void k_mid_mark(const uint64_t* __restrict__ key, const uint32_t* __restrict__ val, const float* __restrict__ center, float half, uint32_t total,
                           uint32_t* __restrict__ isRep) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= total) return;
    bool rep = true;
    if (j > 0 && key[j] == key[j - 1]) {
        const F3 a = midPosition(center, half, val[j]), b = midPosition(center, half, val[j - 1]);
        rep = !(__float_as_uint(a.x) == __float_as_uint(b.x) && __float_as_uint(a.y) == __float_as_uint(b.y) && __float_as_uint(a.z) == __float_as_uint(b.z));
    }
    isRep[j] = rep ? 1u : 0u;
}
__global__ void k_mid_rep_list(const uint32_t* __restrict__ isRep, const uint32_t* __restrict__ scan, const uint32_t* __restrict__ val, uint32_t total,
                               uint32_t* __restrict__ repSample) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < total && isRep[j]) repSample[scan[j]] = val[j];
}
__global__ void __launch_bounds__(128) k_mid_nearest(BvhDev b, const float* __restrict__ center, float half, const uint32_t* __restrict__ repSample, uint32_t numReps,
                                                     uint32_t* __restrict__ repTri) {
    extern __shared__ uint32_t s_stack[];        // [stackDepth][128], stackDepth = BVH depth + 2 (smaller stack -> more waves per CU)
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= numReps) return;
    repTri[r] = bvhNearest<128>(b, midPosition(center, half, repSample[r]), s_stack + threadIdx.x);
}
// stride = floats per sample in `mid`: 4 ([f, fx, fy, fz]) or 8 (the CONTINUITY builder's Hermite slots, mixed derivatives 0)
__global__ void k_mid_values(MeshDev m, const float* __restrict__ center, float half, const uint32_t* __restrict__ val, const uint32_t* __restrict__ isRep,
                             const uint32_t* __restrict__ scan, const uint32_t* __restrict__ repTri, uint32_t total, float* __restrict__ mid, int stride) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= total) return;
    const uint32_t q = val[j];
    const uint32_t slot = scan[j] + isRep[j] - 1u;          // exclusive scan: the group's representative is the last flagged entry at or before j
    valuesAt(m, midPosition(center, half, q), repTri[slot], mid + (size_t)stride * q);
    if (stride == 8) *reinterpret_cast<float4*>(mid + 8 * (size_t)q + 4) = make_float4(0.f, 0.f, 0.f, 0.f);
}

// scratch of the mid-point sampler, reused across levels (grows only)
struct SampleScratch {
    DevBuf<uint64_t> key, keyS; DevBuf<uint32_t> val, valS, isRep, scan, repSample, repTri; DevBuf<unsigned char> tmp; size_t tmpBytes = 0;
};
static int sampleMidPoints(hipStream_t st, const MeshDev& md, const uint32_t* coord, const float* center, float half, uint32_t n, float* mid, int stride,
                           SampleScratch& S, size_t stackBytes, uint64_t& traversals) {
    const uint32_t total = 19u * n;
    SDF_TRY(S.key.reserve(total)); SDF_TRY(S.keyS.reserve(total)); SDF_TRY(S.val.reserve(total)); SDF_TRY(S.valS.reserve(total));
    SDF_TRY(S.isRep.reserve(total)); SDF_TRY(S.scan.reserve(total));
    k_mid_keys<<<gridFor(total, 256), 256, 0, st>>>(coord, n, S.key.p, S.val.p);
    size_t b1 = 0, b2 = 0;
    SDF_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, b1, S.key.p, S.keyS.p, S.val.p, S.valS.p, (int)total, 0, 36, st));
    SDF_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, b2, S.isRep.p, S.scan.p, (int)total, st));
    const size_t need = b1 > b2 ? b1 : b2;
    if (need > S.tmpBytes) { SDF_TRY(S.tmp.reserve(need)); S.tmpBytes = need; }
    SDF_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(S.tmp.p, b1, S.key.p, S.keyS.p, S.val.p, S.valS.p, (int)total, 0, 36, st));
    k_mid_mark<<<gridFor(total, 256), 256, 0, st>>>(S.keyS.p, S.valS.p, center, half, total, S.isRep.p);
    SDF_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(S.tmp.p, b2, S.isRep.p, S.scan.p, (int)total, st));
    uint32_t lastScan = 0, lastFlag = 0;
    SDF_HIP_CHECK(hipMemcpyAsync(&lastScan, S.scan.p + (total - 1), 4, hipMemcpyDeviceToHost, st));
    SDF_HIP_CHECK(hipMemcpyAsync(&lastFlag, S.isRep.p + (total - 1), 4, hipMemcpyDeviceToHost, st));
    SDF_HIP_CHECK(hipStreamSynchronize(st));
    const uint32_t numReps = lastScan + lastFlag;
    SDF_TRY(S.repSample.reserve(numReps)); SDF_TRY(S.repTri.reserve(numReps));
    k_mid_rep_list<<<gridFor(total, 256), 256, 0, st>>>(S.isRep.p, S.scan.p, S.valS.p, total, S.repSample.p);
    k_mid_nearest<<<gridFor(numReps, 128), 128, stackBytes, st>>>(md.bvh, center, half, S.repSample.p, numReps, S.repTri.p);
    k_mid_values<<<gridFor(total, 256), 256, 0, st>>>(md, center, half, S.valS.p, S.isRep.p, S.scan.p, S.repTri.p, total, mid, stride);
    SDF_HIP_CHECK(hipGetLastError());
    traversals += numReps;
    return SDFHIP_OK;
}


}  // namespace
}  // namespace sdfhip
