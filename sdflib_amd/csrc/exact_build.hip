// ExactOctreeSdf construction on the device.  PRODUCT code — independent of oracle/.
//
// Reference behaviour reproduced (single-thread semantics, lattice cache disabled):
//   ExactOctreeSdf::ExactOctreeSdf                               src/sdf/ExactOctreeSdf.cpp:7-31
//   initOctree<PerNodeRegionTrianglesInfluence<None>> + processNode   include/SdfLib/ExactOctreeSdfDepthFirst.h:28-510
//   PerNodeRegionTrianglesInfluence::calculateVerticesInfo / filterTriangles   include/SdfLib/TrianglesInfluence.h:692-860
//   GJK::IsNearMinimize                                           src/utils/GJK.cpp:830-866
//
// MI355X formulation: the reference's depth-first walk becomes level-synchronous passes.
//   k_node_regions  : one wave per node, lane (i,c) -> distance from corner c to the triangle nearest to corner i
//   k_cull          : the hot kernel. Work item = 1024-entry chunk of a node's PARENT list; one wave per chunk, lanes
//                     stride the chunk, gather 3 indices + 3 vertices per triangle, run the Frank-Wolfe test, and
//                     compact the survivors IN ORDER with __ballot + mbcnt prefix (lists stay ascending by id)
//   prefix sum + k_compact : chunk counts -> packed per-node lists
//   k_brute_nearest / k_brute_nearest_mids : first-minimum nearest triangle of a node's sample points over the node's list
//                     (block per (node, corner) at the root; block per node with LDS-staged frames for the 19 mid-points)
//   k_merge_*       : the "second visit" of the last two levels: sorted union of the children's lists + per-child
//                     MSB-first byte masks (ExactOctreeSdfDepthFirst.h:195-259), as binary-search membership tests
//   k_ex_sizes / k_ex_offsets / k_ex_emit_* : pre/post-order offsets of the reference's three arrays (nodes, bit-packed
//                     sets, masks) in its DFS order (children 7..0), then the bit-packing itself (:261-283, 448-468)
// Compile with -ffp-contract=off.
#include "exact_internal.h"
#include "dev_gjk.h"
#include "dev_prims.h"
#include <cmath>
#include <cstring>
#include <memory>

namespace sdfhip {

constexpr uint32_t CHUNK = 1024;
constexpr uint32_t NONE = 0xFFFFFFFFu;

struct ExMesh { const float* verts; const uint32_t* idx; const float* td; const float* frames; };

SDF_DEV F3 ldv(const float* __restrict__ v, uint32_t i) { return F3{v[3 * i], v[3 * i + 1], v[3 * i + 2]}; }

// triangles whose frame normal is usable (ExactOctreeSdfDepthFirst.h:104-108)
__global__ void k_valid_triangles(const float* __restrict__ td, uint32_t T, uint32_t* __restrict__ valid) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const F3 n = triNormal(td + (size_t)TD_FLOATS * t + 3);
    valid[t] = (dot(n, n) > 1e-3f) ? 1u : 0u;
}
__global__ void k_compact_valid(const uint32_t* __restrict__ valid, const uint32_t* __restrict__ scan, uint32_t T, uint32_t* __restrict__ out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < T && valid[t]) out[scan[t]] = t;
}

// wave-wide first-minimum: smallest distance, ties -> smallest list position
SDF_DEV void waveArgMin(float& d, uint32_t& pos) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float od = __shfl_xor(d, off);
        const uint32_t op = __shfl_xor(pos, off);
        if (od < d || (od == d && op < pos)) { d = od; pos = op; }
    }
}

// Nearest triangle (first minimum in list order) for sample points of nodes.  One block of TPB threads per (node, point).
// pointsPerNode = 8 (corners) or 19 (mid-points).  skipFlag: nodes with skip[i] != 0 are not evaluated.
template <int TPB>
__global__ void __launch_bounds__(TPB) k_brute_nearest(ExMesh m, const float* __restrict__ center, float half, uint32_t n, int pointsPerNode,
                                                       const uint32_t* __restrict__ list, const uint32_t* __restrict__ listOff, const uint32_t* __restrict__ listLen,
                                                       const uint32_t* __restrict__ skip, uint32_t* __restrict__ outTri) {
    const uint32_t item = blockIdx.x;
    const uint32_t node = item / (uint32_t)pointsPerNode, pi = item - node * (uint32_t)pointsPerNode;
    if (node >= n) return;
    if (skip && skip[node]) return;
    const F3 ce = ldv(center, node);
    const F3 rel = (pointsPerNode == 8) ? cornerRel((int)pi) : midRel((int)pi);
    const F3 p = ce + rel * half;
    const uint32_t off = listOff[node], len = listLen[node];
    float best = INFINITY; uint32_t bestPos = NONE;
    for (uint32_t k = threadIdx.x; k < len; k += TPB) {
        const uint32_t t = list[off + k];
        TriFrame f; loadFramePacked(m.frames, t, f);
        const float d = sqDistPointTriangleSelect(p, f);
        if (d < best) { best = d; bestPos = k; }
    }
    waveArgMin(best, bestPos);
    if (TPB > 64) {
        __shared__ float sd[TPB / 64]; __shared__ uint32_t sp[TPB / 64];
        const int w = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) { sd[w] = best; sp[w] = bestPos; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int k = 1; k < TPB / 64; k++) if (sd[k] < best || (sd[k] == best && sp[k] < bestPos)) { best = sd[k]; bestPos = sp[k]; }
        }
    }
    if (threadIdx.x == 0) outTri[(size_t)node * pointsPerNode + pi] = (bestPos == NONE) ? (len ? list[off] : 0u) : list[off + bestPos];
}

// Near the root there are few (node, point) items with LONG lists (the root's 19 mid-points against all triangles: 19 waves marching
// through 327 680 frames each was 4 ms of latency): the list of an item is cut into `slices` pieces, a block per (item, piece), and the
// pieces' minima meet in a 64-bit atomic minimum on (distance bits << 32 | list position) — the first minimum in list order, as the
// sequential scan finds it (an infinite or NaN distance never wins: `d < best` with best = +inf); k_brute_finish turns keys into ids.
template <int TPB>
__global__ void __launch_bounds__(TPB) k_brute_nearest_sliced(ExMesh m, const float* __restrict__ center, float half, uint32_t n, int pointsPerNode,
                                                              const uint32_t* __restrict__ list, const uint32_t* __restrict__ listOff, const uint32_t* __restrict__ listLen,
                                                              const uint32_t* __restrict__ skip, uint32_t slices, unsigned long long* __restrict__ keys) {
    const uint32_t item = blockIdx.x / slices, piece = blockIdx.x - item * slices;
    const uint32_t node = item / (uint32_t)pointsPerNode, pi = item - node * (uint32_t)pointsPerNode;
    if (node >= n) return;
    if (skip && skip[node]) return;
    const F3 p = ldv(center, node) + ((pointsPerNode == 8) ? cornerRel((int)pi) : midRel((int)pi)) * half;
    const uint32_t off = listOff[node], len = listLen[node];
    const uint32_t per = (len + slices - 1u) / slices, k0 = piece * per, k1 = (k0 + per < len) ? k0 + per : len;
    float best = INFINITY; uint32_t bestPos = NONE;
    for (uint32_t k = k0 + threadIdx.x; k < k1; k += TPB) {
        TriFrame f; loadFramePacked(m.frames, list[off + k], f);
        const float d = sqDistPointTriangleSelect(p, f);
        if (d < best) { best = d; bestPos = k; }
    }
    waveArgMin(best, bestPos);
    if ((threadIdx.x & 63) == 0 && bestPos != NONE) atomicMin(keys + item, ((unsigned long long)__float_as_uint(best) << 32) | bestPos);
}
__global__ void k_brute_finish(uint32_t n, int pointsPerNode, const uint32_t* __restrict__ list, const uint32_t* __restrict__ listOff, const uint32_t* __restrict__ listLen,
                               const uint32_t* __restrict__ skip, const unsigned long long* __restrict__ keys, uint32_t* __restrict__ outTri) {
    const uint32_t item = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t node = item / (uint32_t)pointsPerNode;
    if (node >= n || (skip && skip[node])) return;
    const uint32_t pos = (uint32_t)keys[item], off = listOff[node], len = listLen[node];
    outTri[item] = (pos == NONE) ? (len ? list[off] : 0u) : list[off + pos];
}

// The 19 mid-points of a node against the node's own list, ONE block per node: the list's 80-byte frames are staged through LDS
// once (64 per tile) and shared by the 19 points, instead of 19 blocks each streaming the list from L2 (the first version's
// profile: 4.9 GB x2 of FETCH per launch).  Thread = (point m, slice s): the tile's triangles s, s + 13, ... ; partial minima
// are merged by (distance, list position), i.e. the first minimum in list order like the sequential scan.
constexpr int BN_SLICES = 13;                     // 19 points x 13 slices = 247 of 256 threads
constexpr uint32_t BN_PIECE = 4096;               // entries of a LONG list one workgroup takes (k_brute_mids_long)
// thread (point, slice) over the list entries [k0, k1) of a node: tiles of 64 frames through s_fr
SDF_DEV void midsRange(const ExMesh& m, F3 p, bool worker, int sl, int tid, const uint32_t* __restrict__ list, uint32_t off, uint32_t k0, uint32_t k1,
                       float4* s_fr, float& best, uint32_t& bestPos) {
    for (uint32_t base = k0; base < k1; base += 64) {
        const uint32_t cnt = (k1 - base < 64u) ? k1 - base : 64u;
        __syncthreads();                                              // previous tile fully consumed
        for (uint32_t e = (uint32_t)tid; e < 5u * cnt; e += 256u) {    // 5 x 16 B per triangle, coalesced within a frame
            const uint32_t k = e / 5u, c = e - 5u * k;
            s_fr[5 * k + c] = reinterpret_cast<const float4*>(m.frames)[5 * (size_t)list[off + base + k] + c];
        }
        __syncthreads();
        if (worker) {
            for (uint32_t k = (uint32_t)sl; k < cnt; k += BN_SLICES) {
                const float4 a = s_fr[5 * k], b = s_fr[5 * k + 1], c = s_fr[5 * k + 2], d4 = s_fr[5 * k + 3], e4 = s_fr[5 * k + 4];
                TriFrame f;
                f.origin = F3{a.x, a.y, a.z};
                f.m[0] = a.w; f.m[1] = b.x; f.m[2] = b.y; f.m[3] = b.z; f.m[4] = b.w; f.m[5] = c.x; f.m[6] = c.y; f.m[7] = c.z; f.m[8] = c.w;
                f.b = F2{d4.x, d4.y}; f.c = F2{d4.z, d4.w}; f.v2 = e4.x; f.v3 = F2{e4.y, e4.z};
                const float d = sqDistPointTriangleSelect(p, f);
                if (d < best) { best = d; bestPos = base + k; }
            }
        }
    }
}
// longLen: nodes whose list is longer are left to k_brute_mids_long
__global__ void __launch_bounds__(256) k_brute_nearest_mids(ExMesh m, const float* __restrict__ center, float half, uint32_t n, const uint32_t* __restrict__ list,
                                                            const uint32_t* __restrict__ listOff, const uint32_t* __restrict__ listLen,
                                                            const uint32_t* __restrict__ skip, uint32_t* __restrict__ outTri, uint32_t longLen) {
    __shared__ float4 s_fr[64 * 5];
    __shared__ float s_d[19 * BN_SLICES]; __shared__ uint32_t s_p[19 * BN_SLICES];
    const uint32_t node = blockIdx.x;
    if (node >= n) return;
    if (skip && skip[node]) return;                                   // whole block
    const int tid = threadIdx.x;
    const int mi = tid / BN_SLICES, sl = tid - BN_SLICES * mi;
    const bool worker = mi < 19;
    const F3 p = ldv(center, node) + (worker ? midRel(mi) : F3{0.f, 0.f, 0.f}) * half;
    const uint32_t off = listOff[node], len = listLen[node];
    if (len > longLen) return;                                        // whole block
    float best = INFINITY; uint32_t bestPos = NONE;
    midsRange(m, p, worker, sl, tid, list, off, 0u, len, s_fr, best, bestPos);
    if (worker) { s_d[tid] = best; s_p[tid] = bestPos; }
    __syncthreads();
    if (worker && sl == 0) {
        for (int q = 1; q < BN_SLICES; q++) {
            const float od = s_d[tid + q]; const uint32_t op = s_p[tid + q];
            if (od < best || (od == best && op < bestPos)) { best = od; bestPos = op; }
        }
        outTri[(size_t)node * 19 + mi] = (bestPos == NONE) ? (len ? list[off] : 0u) : list[off + bestPos];
    }
}
// The few nodes with LONG lists (near the centre of a round shape a node's list is most of the mesh: 300 000 entries x 19 points in
// ONE workgroup was the whole duration of a level's launch): listed here, cut into pieces of BN_PIECE entries, a workgroup per piece
// (k_brute_mids_long: persistent over the (node, piece) items, the pieces' minima meet in a 64-bit atomic minimum on
// (distance bits << 32 | list position), the first minimum in list order as the sequential scan finds it), ids by k_brute_long_finish.
__global__ void k_long_nodes(uint32_t n, const uint32_t* __restrict__ listLen, const uint32_t* __restrict__ skip, uint32_t longLen, uint2* __restrict__ items,
                             uint32_t* __restrict__ itemCount, unsigned long long* __restrict__ keys) {
    const uint32_t node = blockIdx.x * blockDim.x + threadIdx.x;
    if (node >= n || (skip && skip[node]) || listLen[node] <= longLen) return;
    const uint32_t np = (listLen[node] + BN_PIECE - 1u) / BN_PIECE;
    const uint32_t base = atomicAdd(itemCount, np);
    for (uint32_t q = 0; q < np; q++) items[base + q] = make_uint2(node, q);
    for (int q = 0; q < 19; q++) keys[19 * (size_t)node + q] = ~0ull;
}
__global__ void __launch_bounds__(256) k_brute_mids_long(ExMesh m, const float* __restrict__ center, float half, const uint32_t* __restrict__ list,
                                                         const uint32_t* __restrict__ listOff, const uint32_t* __restrict__ listLen, const uint2* __restrict__ items,
                                                         const uint32_t* __restrict__ itemCount, unsigned long long* __restrict__ keys) {
    __shared__ float4 s_fr[64 * 5];
    __shared__ float s_d[19 * BN_SLICES]; __shared__ uint32_t s_p[19 * BN_SLICES];
    const int tid = threadIdx.x;
    const int mi = tid / BN_SLICES, sl = tid - BN_SLICES * mi;
    const bool worker = mi < 19;
    const uint32_t count = *itemCount;
    for (uint32_t w = blockIdx.x; w < count; w += gridDim.x) {
        const uint2 it = items[w];
        const uint32_t node = it.x;
        const uint32_t off = listOff[node], len = listLen[node];
        const uint32_t k0 = it.y * BN_PIECE;
        const uint32_t k1 = (k0 + BN_PIECE < len) ? k0 + BN_PIECE : len;
        const F3 p = ldv(center, node) + (worker ? midRel(mi) : F3{0.f, 0.f, 0.f}) * half;
        float best = INFINITY; uint32_t bestPos = NONE;
        midsRange(m, p, worker, sl, tid, list, off, k0, k1, s_fr, best, bestPos);
        __syncthreads();                                              // (s_d / s_p of the previous item are consumed)
        if (worker) { s_d[tid] = best; s_p[tid] = bestPos; }
        __syncthreads();
        if (worker && sl == 0) {
            for (int q = 1; q < BN_SLICES; q++) {
                const float od = s_d[tid + q]; const uint32_t op = s_p[tid + q];
                if (od < best || (od == best && op < bestPos)) { best = od; bestPos = op; }
            }
            if (bestPos != NONE) atomicMin(keys + 19 * (size_t)node + mi, ((unsigned long long)__float_as_uint(best) << 32) | bestPos);
        }
    }
}
__global__ void k_brute_long_finish(uint32_t n, const uint32_t* __restrict__ list, const uint32_t* __restrict__ listOff, const uint32_t* __restrict__ listLen, const uint32_t* __restrict__ skip,
                                    uint32_t longLen, const unsigned long long* __restrict__ keys, uint32_t* __restrict__ outTri) {
    const uint32_t item = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t node = item / 19u;
    if (node >= n || (skip && skip[node])) return;
    const uint32_t len = listLen[node];
    if (len <= longLen) return;
    const uint32_t pos = (uint32_t)keys[item], off = listOff[node];
    outTri[item] = (pos == NONE) ? list[off] : list[off + pos];
}

// 8x8 corner-sphere radii of a node: lane l = 8*i + c  ->  dist(corner c, nearest triangle of corner i) - min_c
__global__ void __launch_bounds__(256) k_node_regions(ExMesh m, const float* __restrict__ center, float half, uint32_t n,
                                                      const uint32_t* __restrict__ cornerTri, float* __restrict__ region /*64 n*/, float* __restrict__ minDist /*8 n*/) {
    const uint32_t node = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (node >= n) return;
    const int i = lane >> 3, c = lane & 7;
    const uint32_t t = cornerTri[8 * (size_t)node + i];
    TriFrame f; loadFramePacked(m.frames, t, f);
    const F3 p = ldv(center, node) + cornerRel(c) * half;
    const float r = sqrtf(sqDistPointTriangle(p, f));
    // min over the 8 lanes of group i in ascending c with glm::min(a,b) = (b<a)?b:a semantics (order-independent for non-NaN)
    float mn = r;
#pragma unroll
    for (int off = 4; off >= 1; off >>= 1) { const float o = __shfl_xor(mn, off); mn = (o < mn) ? o : mn; }
    region[64 * (size_t)node + lane] = r - mn;
    if (c == 0) minDist[8 * (size_t)node + i] = mn;
}

// chunk table: chunkNode[chunkBase[i] + k] = i
__global__ void k_chunk_counts(uint32_t n, const uint32_t* __restrict__ pLen, uint32_t* __restrict__ nChunks) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) nChunks[i] = (pLen[i] + CHUNK - 1) / CHUNK;
}
__global__ void k_chunk_fill(uint32_t n, const uint32_t* __restrict__ nChunks, const uint32_t* __restrict__ chunkBase, uint32_t* __restrict__ chunkNode) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t b = chunkBase[i], c = nChunks[i];
    for (uint32_t k = 0; k < c; k++) chunkNode[b + k] = i;
}

struct CullArgs {
    ExMesh m; const float* center; float half;
    const uint32_t* cornerTri; const float* region; const float* minDist;
    const uint32_t* plist; const uint32_t* pOff; const uint32_t* pLen;      // parent list of every node
    const uint32_t* chunkNode; const uint32_t* chunkBase; uint32_t numChunks;
    uint32_t* tmp;            // survivors of chunk q, compacted at tmp[q * CHUNK ...]
    uint32_t* chunkCount;     // survivors per chunk
    unsigned long long* cullTests;
    uint32_t perWave;         // consecutive chunks a wave streams through
};

// THE hot kernel of the Exact build: one wave per chunk of a node's parent list.
// The Frank-Wolfe test takes 1..15 iterations depending on the triangle (the first version's counters: 23 % of the lanes busy per
// VALU instruction), so a lane does not wait for its neighbours: as soon as its triangle is decided it takes the chunk's next
// undecided entry (rank within the ballot of idle lanes), and every wave-loop iteration is one Frank-Wolfe step for the lanes
// that hold a triangle.  Decisions land in a per-wave bit mask in LDS; the survivors are then written out in list order.
// Measured in round 4 and NOT kept (profiles/r04_cull_experiments.txt): staging the chunk through LDS so that a refill costs no
// dependent gathers, and holding refills back until 4 .. 24 lanes are idle - the kernel's time did not move (19.4 ms of a C3 build
// either way).  What the per-launch trace shows instead: 10.2 of those 19.4 ms are the LAST level, 824 000 chunks of a hundred or
// two entries each (one node per wave), i.e. two or three triangles per lane and then a tail in which the lanes whose Frank-Wolfe
// runs are short wait for the long ones - the 0.64 lanes per VALU instruction.  The remedy, below: a wave streams through several
// chunks with two chunk contexts in LDS (idle lanes start the next chunk while the last triangles of the current one finish):
// k_cull 19.4 -> 15.0 ms per C3 build (the last level 10.3 -> 5.5 ms), build 41 -> 35.5 ms.
// A wave STREAMS through `perWave` consecutive chunks with two chunk
// contexts in LDS.  When the chunk handing out entries is exhausted, the lanes that fall idle start on the next chunk in the other
// context while the last triangles of the old one finish; a context whose chunk has no running lane left writes its survivors out and
// is free again.  The tail is paid once per wave instead of once per chunk; chunk bookkeeping (tmp, chunkCount) is unchanged.
__global__ void __launch_bounds__(256) k_cull(CullArgs a) {
    __shared__ float s_region[4][2][64];
    __shared__ float s_min[4][2][8];
    __shared__ uint32_t s_corner[4][2][8];
    __shared__ uint32_t s_keep[4][2][CHUNK / 32];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t wave = blockIdx.x * 4u + (uint32_t)w;
    const uint32_t qFirst = wave * a.perWave;
    if (qFirst >= a.numChunks) return;
    const uint32_t qLast = (qFirst + a.perWave < a.numChunks) ? qFirst + a.perWave : a.numChunks;      // one past
    const unsigned long long ltMask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    // per context (wave-uniform): state 0 = empty, 1 = handing out entries, 2 = draining
    int state[2] = {0, 0};
    uint32_t cq[2] = {0, 0}, coff[2] = {0, 0}, cbegin[2] = {0, 0}, cend[2] = {0, 0}, cnext[2] = {0, 0};
    F3 cce[2] = {F3{0.f, 0.f, 0.f}, F3{0.f, 0.f, 0.f}};
    uint32_t qNext = qFirst;
    unsigned long long tests = 0;
    NearMinimizeState fw; int vId = 0, slot = 0; uint32_t myEntry = 0;
    bool busy = false;
    for (;;) {
        // a chunk for an empty context, if nobody is handing out entries
        if (state[0] != 1 && state[1] != 1 && qNext < qLast && (state[0] == 0 || state[1] == 0)) {
            const int s = state[0] == 0 ? 0 : 1;
            const uint32_t q = qNext++;
            const uint32_t node = a.chunkNode[q], ck = q - a.chunkBase[node];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            s_region[w][s][lane] = a.region[64 * (size_t)node + lane];
            if (lane < 8) { s_min[w][s][lane] = a.minDist[8 * (size_t)node + lane]; s_corner[w][s][lane] = a.cornerTri[8 * (size_t)node + lane]; }
            if (lane < (int)(CHUNK / 32)) s_keep[w][s][lane] = 0u;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            const uint32_t len = a.pLen[node];
            cq[s] = q; cce[s] = ldv(a.center, node); coff[s] = a.pOff[node];
            cbegin[s] = ck * CHUNK; cend[s] = (cbegin[s] + CHUNK < len) ? cbegin[s] + CHUNK : len; cnext[s] = cbegin[s];
            state[s] = 1;
        }
        const int feed = state[0] == 1 ? 0 : (state[1] == 1 ? 1 : -1);
        // idle lanes take the next entries of the chunk that hands them out
        if (feed >= 0) {
            const unsigned long long idle = __ballot(!busy);
            if (idle != 0ull) {
                if (!busy) {
                    const uint32_t k = cnext[feed] + (uint32_t)__popcll(idle & ltMask);
                    if (k < cend[feed]) {
                        const uint32_t t = a.plist[coff[feed] + k];
                        const uint32_t i0 = a.m.idx[3 * t], i1 = a.m.idx[3 * t + 1], i2 = a.m.idx[3 * t + 2];
                        const F3 ce = cce[feed];
                        const F3 t0 = ldv(a.m.verts, i0) - ce, t1 = ldv(a.m.verts, i1) - ce, t2 = ldv(a.m.verts, i2) - ce;
                        const F3 pt = 0.3333333f * ((t0 + t1) + t2);
                        vId = ((pt.z > 0) ? 4 : 0) + ((pt.y > 0) ? 2 : 0) + ((pt.x > 0) ? 1 : 0);
                        myEntry = k - cbegin[feed]; slot = feed;
                        if (s_corner[w][feed][vId] == t) atomicOr(&s_keep[w][feed][myEntry >> 5], 1u << (myEntry & 31u));      // the corner's own nearest triangle is always kept
                        else { fw.start(t0, t1, t2); busy = true; }
                    }
                }
                cnext[feed] += (uint32_t)__popcll(idle);
                tests += (unsigned long long)__popcll(__ballot(busy) & idle);      // entries that entered the Frank-Wolfe test in this round
            }
            if (cnext[feed] >= cend[feed]) state[feed] = 2;          // everything handed out: the chunk drains
        }
        const unsigned long long running = __ballot(busy);
        // a draining chunk without a running lane: its survivors go out in list order, the context is free
#pragma unroll
        for (int s = 0; s < 2; s++) {
            if (state[s] == 2 && __ballot(busy && slot == s) == 0ull) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                uint32_t kept = 0;
                for (uint32_t base = 0; cbegin[s] + base < cend[s]; base += 64) {
                    const uint32_t e = base + (uint32_t)lane;
                    const bool keep = (cbegin[s] + e < cend[s]) && ((s_keep[w][s][e >> 5] >> (e & 31u)) & 1u);
                    const unsigned long long mask = __ballot(keep);
                    if (keep) a.tmp[(size_t)cq[s] * CHUNK + kept + (uint32_t)__popcll(mask & ltMask)] = a.plist[coff[s] + cbegin[s] + e];
                    kept += (uint32_t)__popcll(mask);
                }
                if (lane == 0) a.chunkCount[cq[s]] = kept;
                state[s] = 0;
            }
        }
        if (running == 0ull && state[0] == 0 && state[1] == 0 && qNext >= qLast) break;
        if (busy) {
            bool keep;
            if (fw.step(a.half, &s_region[w][slot][8 * vId], s_min[w][slot][vId], keep)) {
                if (keep) atomicOr(&s_keep[w][slot][myEntry >> 5], 1u << (myEntry & 31u));
                busy = false;
            }
        }
    }
    if (lane == 0 && tests) atomicAdd(a.cullTests, tests);
}

// packed lists: survivors of chunk q go to list[chunkScan[q] ...]; listOff[node] = chunkScan[first chunk of node]
__global__ void __launch_bounds__(256) k_compact(const uint32_t* __restrict__ tmp, const uint32_t* __restrict__ chunkCount, const uint32_t* __restrict__ chunkScan,
                                                 uint32_t numChunks, uint32_t* __restrict__ list) {
    const uint32_t q = blockIdx.x * 4u + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (q >= numChunks) return;
    const uint32_t c = chunkCount[q], o = chunkScan[q];
    for (uint32_t k = lane; k < c; k += 64) list[o + k] = tmp[(size_t)q * CHUNK + k];
}
__global__ void k_node_list_ranges(uint32_t n, const uint32_t* __restrict__ nChunks, const uint32_t* __restrict__ chunkBase, const uint32_t* __restrict__ chunkScan,
                                   const uint32_t* __restrict__ chunkCount, uint32_t numChunks, uint32_t* __restrict__ listOff, uint32_t* __restrict__ listLen) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t b = chunkBase[i], c = nChunks[i];
    if (c == 0) { listOff[i] = (b < numChunks) ? chunkScan[b] : (numChunks ? chunkScan[numChunks - 1] + chunkCount[numChunks - 1] : 0u); listLen[i] = 0; return; }
    const uint32_t last = b + c - 1;
    listOff[i] = chunkScan[b];
    listLen[i] = chunkScan[last] + chunkCount[last] - chunkScan[b];
}

__global__ void k_ex_decide(uint32_t n, uint32_t depth, uint32_t startDepth, uint32_t maxDepth, uint32_t minTri, const uint32_t* __restrict__ listLen,
                            uint32_t* __restrict__ flag, uint32_t* __restrict__ inner, uint32_t* __restrict__ maxLeaf) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    bool terminal = false;
    if (depth >= startDepth) terminal = listLen[i] <= minTri;
    const bool leaf = terminal || depth >= maxDepth;
    flag[i] = leaf ? 1u : 0u; inner[i] = leaf ? 0u : 1u;
    if (leaf) atomicMax(maxLeaf, listLen[i]);
}

struct ExChildArgs {
    const float* center; const uint32_t* coord; const uint32_t* cornerTri; const uint32_t* midTri; const uint32_t* inner; const uint32_t* childBase;
    const uint32_t* listOff; const uint32_t* listLen; uint32_t n; float half;
    float* ncenter; uint32_t* ncoord; uint32_t* ncornerTri; uint32_t* npOff; uint32_t* npLen;
};
__global__ void __launch_bounds__(256) k_ex_children(ExChildArgs a) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t i = gid >> 6;
    if (i >= a.n || !a.inner[i]) return;
    const uint32_t c = (gid >> 3) & 7u, j = gid & 7u;
    const uint32_t child = a.childBase[i] + c;
    const int src = kStencilDev.src[c][j];
    a.ncornerTri[8 * (size_t)child + j] = (src >= 0) ? a.midTri[19 * (size_t)i + src] : a.cornerTri[8 * (size_t)i + (-src - 1)];
    if (j == 0) {
        const float ns = 0.5f * a.half;
        a.ncenter[3 * (size_t)child] = a.center[3 * (size_t)i] + ((c & 1u) ? ns : -ns);
        a.ncenter[3 * (size_t)child + 1] = a.center[3 * (size_t)i + 1] + ((c & 2u) ? ns : -ns);
        a.ncenter[3 * (size_t)child + 2] = a.center[3 * (size_t)i + 2] + ((c & 4u) ? ns : -ns);
        const uint32_t co = a.coord[i];
        const uint32_t x = 2u * (co & 1023u) + (c & 1u), y = 2u * ((co >> 10) & 1023u) + ((c >> 1) & 1u), z = 2u * (co >> 20) + (c >> 2);
        a.ncoord[child] = x | (y << 10) | (z << 20);
        a.npOff[child] = a.listOff[i]; a.npLen[child] = a.listLen[i];
    }
}

__global__ void k_ex_to_cell_order(const float* __restrict__ center, const uint32_t* __restrict__ coord, const uint32_t* __restrict__ cornerTri,
                                   const uint32_t* __restrict__ pOff, const uint32_t* __restrict__ pLen, uint32_t n, uint32_t G,
                                   float* __restrict__ ocenter, uint32_t* __restrict__ ocoord, uint32_t* __restrict__ ocornerTri, uint32_t* __restrict__ opOff, uint32_t* __restrict__ opLen) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t i = gid >> 3, j = gid & 7u;
    if (i >= n) return;
    const uint32_t co = coord[i];
    const uint32_t d = (co >> 20) * G * G + ((co >> 10) & 1023u) * G + (co & 1023u);
    ocornerTri[8 * (size_t)d + j] = cornerTri[8 * (size_t)i + j];
    if (j == 0) {
        ocenter[3 * (size_t)d] = center[3 * (size_t)i]; ocenter[3 * (size_t)d + 1] = center[3 * (size_t)i + 1]; ocenter[3 * (size_t)d + 2] = center[3 * (size_t)i + 2];
        ocoord[d] = co; opOff[d] = pOff[i]; opLen[d] = pLen[i];
    }
}

// sharded build: keep only the start cells listed in `sel` (z-major cell ids, ascending)
__global__ void k_ex_select_cells(const float* __restrict__ center, const uint32_t* __restrict__ coord, const uint32_t* __restrict__ cornerTri,
                                  const uint32_t* __restrict__ pOff, const uint32_t* __restrict__ pLen, const uint32_t* __restrict__ sel, uint32_t nSel,
                                  float* __restrict__ ocenter, uint32_t* __restrict__ ocoord, uint32_t* __restrict__ ocornerTri, uint32_t* __restrict__ opOff, uint32_t* __restrict__ opLen) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t i = gid >> 3, j = gid & 7u;
    if (i >= nSel) return;
    const uint32_t s = sel[i];
    ocornerTri[8 * (size_t)i + j] = cornerTri[8 * (size_t)s + j];
    if (j == 0) {
        ocenter[3 * (size_t)i] = center[3 * (size_t)s]; ocenter[3 * (size_t)i + 1] = center[3 * (size_t)s + 1]; ocenter[3 * (size_t)i + 2] = center[3 * (size_t)s + 2];
        ocoord[i] = coord[s]; opOff[i] = pOff[s]; opLen[i] = pLen[s];
    }
}


// ---- merge step ("second visit") ---------------------------------------------------------------------------------
SDF_DEV bool sortedContains(const uint32_t* __restrict__ l, uint32_t len, uint32_t v) {
    uint32_t lo = 0, hi = len;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (l[mid] < v) lo = mid + 1; else hi = mid; }
    return lo < len && l[lo] == v;
}
// first position >= from with l[pos] >= v (l ascending): gallop from `from`, then bisect — cheap when the answer is near `from`
SDF_DEV uint32_t lowerBoundFrom(const uint32_t* __restrict__ l, uint32_t len, uint32_t from, uint32_t v) {
    if (from >= len || l[from] >= v) return from;
    uint32_t lo = from, step = 1;                    // invariant: l[lo] < v
    while (lo + step < len && l[lo + step] < v) { lo += step; step <<= 1; }
    uint32_t hi = (lo + step < len) ? lo + step : len;          // l[hi] >= v or hi == len
    lo++;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (l[mid] < v) lo = mid + 1; else hi = mid; }
    return lo;
}
struct MergeArgs {
    uint32_t n; const uint32_t* inner; const uint32_t* childBase;
    const uint32_t* list; const uint32_t* listOff; const uint32_t* listLen;                 // this level's filtered lists
    uint32_t* ulist; uint32_t* uLen;                                                         // union lists (same offsets as list)
    // children level
    const uint32_t* cflag; const uint32_t* clist; const uint32_t* clistOff; const uint32_t* clistLen; const uint32_t* culist; const uint32_t* cuLen;
    const uint32_t* maskOff; uint8_t* masks;                                                 // phase 2
    uint32_t* maxEncoded; int trackEncoded;
};
SDF_DEV void childFinal(const MergeArgs& a, uint32_t child, const uint32_t*& l, uint32_t& len) {
    if (a.cflag[child]) { l = a.clist + a.clistOff[child]; len = a.clistLen[child]; }
    else { l = a.culist + a.clistOff[child]; len = a.cuLen[child]; }
}
// Is v among the 64 ascending entries of win (LDS; the entries behind a list's end are 0xFFFFFFFF)?  Six fixed steps.
SDF_DEV bool windowContains(const uint32_t* win, uint32_t v) {
    uint32_t pos = 0;
#pragma unroll
    for (uint32_t step = 32; step >= 1; step >>= 1) pos += (win[pos + step - 1u] < v) ? step : 0u;
    return win[pos] == v;
}
// A child's final list is a SUBSET of its parent's filtered list (k_cull filters the parent's list) and both ascend: the values of 64
// consecutive parent entries can only be found among the NEXT 64 entries of the child's list behind a cursor that moves forward.  A chunk
// therefore costs one round trip — the parent's 64 values and the eight 64-entry windows, all coalesced — and its look-ups are six-step
// searches in LDS (before: galloping window searches and bisections in global memory, ~10 dependent fetches per element).
// phase 1: union = elements of the node's filtered list present in any child's final list (order kept); one wave per node
__global__ void __launch_bounds__(256) k_merge_union(MergeArgs a) {
    __shared__ uint32_t s_win[4][8][64];
    const uint32_t node = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (node >= a.n || !a.inner[node]) return;
    const uint32_t off = a.listOff[node], len = a.listLen[node], cb = a.childBase[node];
    const uint32_t* cl[8]; uint32_t clen[8], cur[8];
#pragma unroll
    for (int c = 0; c < 8; c++) { childFinal(a, cb + (uint32_t)c, cl[c], clen[c]); cur[c] = 0; }
    uint32_t kept = 0;
    for (uint32_t base = 0; base < len; base += 64) {
        const uint32_t k = base + lane;
        const uint32_t v = (k < len) ? a.list[off + k] : 0xFFFFFFFFu;
        uint32_t w[8];
#pragma unroll
        for (int c = 0; c < 8; c++) w[c] = (cur[c] + (uint32_t)lane < clen[c]) ? cl[c][cur[c] + (uint32_t)lane] : 0xFFFFFFFFu;
        const uint32_t vmax = (uint32_t)__shfl((int)v, (int)((len - base < 64u ? len - base : 64u) - 1u));
#pragma unroll
        for (int c = 0; c < 8; c++) s_win[wv][c][lane] = w[c];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        bool keep = false;
#pragma unroll
        for (int c = 0; c < 8; c++) {
            uint32_t cnt = (uint32_t)__popcll(__ballot(cur[c] + (uint32_t)lane < clen[c] && w[c] <= vmax));
            if (k < len && !keep && cnt) keep = windowContains(s_win[wv][c], v);
            if (cnt == 64u && cur[c] + 64u < clen[c] && cl[c][cur[c] + 64u] <= vmax) {       // (not a subset after all: the general search)
                if (k < len && !keep) keep = sortedContains(cl[c] + cur[c] + 64u, clen[c] - cur[c] - 64u, v);
                cnt = ((vmax == 0xFFFFFFFFu) ? clen[c] : lowerBoundFrom(cl[c], clen[c], cur[c] + 64u, vmax + 1u)) - cur[c];
            }
            cur[c] += cnt;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const unsigned long long mask = __ballot(keep);
        const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
        if (keep) a.ulist[off + kept + before] = v;
        kept += (uint32_t)__popcll(mask);
    }
    if (lane == 0) { a.uLen[node] = kept; if (a.trackEncoded) atomicMax(a.maxEncoded, kept); }
}
__global__ void k_mask_sizes(uint32_t n, const uint32_t* __restrict__ inner, const uint32_t* __restrict__ uLen, uint32_t* __restrict__ maskBytes) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) maskBytes[i] = inner[i] ? 8u * ((uLen[i] + 7u) / 8u) : 0u;
}
// phase 2: per-child MSB-first byte masks over the union list; one wave per (node, child), the child's list streamed like in phase 1
// (it is a subset of the union list)
__global__ void __launch_bounds__(256) k_merge_masks(MergeArgs a) {
    __shared__ uint32_t s_win[4][64];
    const uint32_t item = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t node = item >> 3, c = item & 7u;
    if (node >= a.n || !a.inner[node]) return;
    const uint32_t off = a.listOff[node], ul = a.uLen[node], nb = (ul + 7u) / 8u;
    const uint32_t* l; uint32_t ll; childFinal(a, a.childBase[node] + c, l, ll);
    uint8_t* dst = a.masks + a.maskOff[node] + (size_t)c * nb;
    uint32_t cur = 0;
    for (uint32_t base = 0; base < ul; base += 64) {
        const uint32_t k = base + lane;
        const uint32_t v = (k < ul) ? a.ulist[off + k] : 0xFFFFFFFFu;
        const uint32_t w = (cur + (uint32_t)lane < ll) ? l[cur + (uint32_t)lane] : 0xFFFFFFFFu;
        const uint32_t vmax = (uint32_t)__shfl((int)v, (int)((ul - base < 64u ? ul - base : 64u) - 1u));
        s_win[wv][lane] = w;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        uint32_t cnt = (uint32_t)__popcll(__ballot(cur + (uint32_t)lane < ll && w <= vmax));
        bool in = (k < ul) && cnt && windowContains(s_win[wv], v);
        if (cnt == 64u && cur + 64u < ll && l[cur + 64u] <= vmax) {       // (not a subset after all: the general search)
            if (k < ul && !in) in = sortedContains(l + cur + 64u, ll - cur - 64u, v);
            cnt = ((vmax == 0xFFFFFFFFu) ? ll : lowerBoundFrom(l, ll, cur + 64u, vmax + 1u)) - cur;
        }
        cur += cnt;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const unsigned long long m = __ballot(in);
        if (lane < 8) {
            const uint32_t byteIdx = (base >> 3) + (uint32_t)lane;
            if (byteIdx < nb) dst[byteIdx] = (uint8_t)(__brev((uint32_t)((m >> (8 * lane)) & 0xFFull)) >> 24);
        }
    }
}

// ---- sizes (bottom-up) and offsets (top-down) of the three output arrays -------------------------------------------
struct SizeArgs {
    uint32_t n, depth, bitEnc, bits; const uint32_t* flag; const uint32_t* childBase; const uint32_t* listLen; const uint32_t* uLen;
    const uint32_t* cNode; const uint32_t* cSet; const uint32_t* cMask;     // children level sizes (may be null)
    uint32_t* aNode; uint32_t* aSet; uint32_t* aMask;
};
SDF_DEV uint32_t setWords(uint32_t count, uint32_t bits) { return (count * bits + 31u) / 32u + 2u; }
__global__ void k_ex_sizes(SizeArgs a) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    if (a.flag[i]) { a.aNode[i] = 0; a.aMask[i] = 0; a.aSet[i] = (a.depth <= a.bitEnc) ? setWords(a.listLen[i], a.bits) : 0u; return; }
    uint32_t sn = 8, ss = 0, sm = 0;
    const uint32_t cb = a.childBase[i];
    for (int c = 0; c < 8; c++) { sn += a.cNode[cb + c]; ss += a.cSet[cb + c]; sm += a.cMask[cb + c]; }
    if (a.depth >= a.bitEnc) sm += 8u * ((a.uLen[i] + 7u) / 8u);
    if (a.depth == a.bitEnc) ss = setWords(a.uLen[i], a.bits);
    a.aNode[i] = sn; a.aSet[i] = ss; a.aMask[i] = sm;
}
struct OffArgs {
    uint32_t n, depth, bitEnc; const uint32_t* flag; const uint32_t* childBase; const uint32_t* uLen;
    const uint32_t* pos; const uint32_t* blk; const uint32_t* setPos; const uint32_t* maskPos;
    const uint32_t* cNode; const uint32_t* cSet; const uint32_t* cMask;
    uint32_t* cpos; uint32_t* cblk; uint32_t* csetPos; uint32_t* cmaskPos; uint32_t* cmaskIdx;   // children outputs
    // final node array (2 words per node).  `self*` receives the writes at `pos` (the node itself), `nodes/hasTri` the writes
    // at blk+c (its children block).  Stored VALUES are absolute; buffer index = absolute position - bias, so a shard can
    // emit its part of the array into local buffers (start level: self = the shard's start-grid slots, bias 0).
    uint32_t* selfNodes; uint8_t* selfHas; uint32_t selfBias;
    uint32_t* nodes; uint8_t* hasTri; uint32_t bodyBias;
};
__global__ void k_ex_offsets(OffArgs a) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t i = gid >> 3, c = gid & 7u;
    if (i >= a.n) return;
    const uint32_t pos = a.pos[i] - a.selfBias;
    if (a.flag[i]) {
        if (c == 0) {
            a.selfNodes[2 * (size_t)pos] = 0xFFFFFFFFu;
            if (a.depth <= a.bitEnc) { a.selfNodes[2 * (size_t)pos + 1] = a.setPos[i]; a.selfHas[pos] = 1; }
        }
        return;
    }
    const uint32_t blk = a.blk[i], cb = a.childBase[i];
    uint32_t nb = blk + 8u, sp = a.setPos[i], mp = a.maskPos[i], allMask = 0;
    for (uint32_t k = 7; k > c; k--) { nb += a.cNode[cb + k]; sp += a.cSet[cb + k]; mp += a.cMask[cb + k]; }
    for (uint32_t k = 0; k < 8; k++) allMask += a.cMask[cb + k];
    a.cpos[cb + c] = blk + c; a.cblk[cb + c] = nb; a.csetPos[cb + c] = sp; a.cmaskPos[cb + c] = mp;
    if (a.depth >= a.bitEnc) {
        const uint32_t nbytes = (a.uLen[i] + 7u) / 8u;
        const uint32_t own = a.maskPos[i] + allMask;                 // the node's own 8 masks follow its children's
        a.cmaskIdx[cb + c] = own + c * nbytes;
        a.nodes[2 * (size_t)(blk + c - a.bodyBias) + 1] = own + c * nbytes; a.hasTri[blk + c - a.bodyBias] = 1;
    }
    if (c == 0) {
        a.selfNodes[2 * (size_t)pos] = blk & 0x7FFFFFFFu;
        if (a.depth == a.bitEnc) { a.selfNodes[2 * (size_t)pos + 1] = a.setPos[i]; a.selfHas[pos] = 1; }
    }
}
// bit-packed set of a node: [count][indices, `bits` each, MSB-first across words][spare]; one wave per node
struct PackArgs { uint32_t n, depth, bitEnc, bits; const uint32_t* flag; const uint32_t* list; const uint32_t* ulist; const uint32_t* listOff; const uint32_t* listLen; const uint32_t* uLen; const uint32_t* setPos; uint32_t* sets; uint32_t setBias; };
__global__ void __launch_bounds__(256) k_ex_pack_sets(PackArgs a) {
    const uint32_t node = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (node >= a.n) return;
    const bool leaf = a.flag[node] != 0u;
    if (!((leaf && a.depth <= a.bitEnc) || (!leaf && a.depth == a.bitEnc))) return;
    const uint32_t* l = (leaf ? a.list : a.ulist) + a.listOff[node];
    const uint32_t cnt = leaf ? a.listLen[node] : a.uLen[node];
    uint32_t* dst = a.sets + (a.setPos[node] - a.setBias);
    if (lane == 0) dst[0] = cnt;
    const uint32_t inv = 32u - a.bits;
    for (uint32_t t = lane; t < cnt; t += 64) {
        const uint32_t index = l[t], bIdx = t * a.bits, w = bIdx >> 5, bit = bIdx & 31u;
        atomicOr(dst + 1 + w, (index << inv) >> bit);
        const uint32_t hi = (uint32_t)((unsigned long long)index << (64u - (bit + a.bits)));
        if (hi) atomicOr(dst + 2 + w, hi);
    }
}
__global__ void __launch_bounds__(256) k_ex_copy_masks(uint32_t n, uint32_t depth, uint32_t bitEnc, const uint32_t* __restrict__ inner, const uint32_t* __restrict__ uLen,
                                                       const uint32_t* __restrict__ maskOff, const uint8_t* __restrict__ masks, const uint32_t* __restrict__ maskPos,
                                                       const uint32_t* __restrict__ childBase, const uint32_t* __restrict__ cMask, uint8_t* __restrict__ out, uint32_t maskBias) {
    const uint32_t node = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (node >= n || !inner[node] || depth < bitEnc) return;
    uint32_t allMask = 0;
    for (uint32_t k = 0; k < 8; k++) allMask += cMask[childBase[node] + k];
    const uint32_t bytes = 8u * ((uLen[node] + 7u) / 8u);
    const uint8_t* src = masks + maskOff[node];
    uint8_t* dst = out + (maskPos[node] + allMask - maskBias);
    for (uint32_t k = lane; k < bytes; k += 64) dst[k] = src[k];
}
__global__ void k_ex_init_cells(uint32_t nCells, const uint32_t* __restrict__ blkIn, const uint32_t* __restrict__ setIn, const uint32_t* __restrict__ maskIn,
                                uint32_t* __restrict__ pos, uint32_t* __restrict__ blk, uint32_t* __restrict__ setPos, uint32_t* __restrict__ maskPos) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nCells) return;
    pos[i] = i; blk[i] = blkIn[i]; setPos[i] = setIn[i]; maskPos[i] = maskIn[i];      // pos = slot in the (local) start-grid buffer
}
__global__ void k_fill_inner(uint32_t n, uint32_t* __restrict__ flag, uint32_t* __restrict__ inner, uint32_t* __restrict__ childBase) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { flag[i] = 0u; inner[i] = 1u; childBase[i] = 8u * i; }
}
__global__ void k_mul8(uint32_t n, uint32_t* __restrict__ v) { const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) v[i] *= 8u; }

static uint32_t dfsRank(uint32_t x, uint32_t y, uint32_t z, uint32_t startDepth) {
    uint32_t r = 0;
    for (uint32_t l = 0; l < startDepth; l++) {
        const uint32_t sh = startDepth - 1 - l;
        const uint32_t c = ((x >> sh) & 1u) | (((y >> sh) & 1u) << 1) | (((z >> sh) & 1u) << 2);
        r = r * 8u + (7u - c);
    }
    return r;
}

struct ScanHelper {
    DevBuf<unsigned char> tmp; size_t bytes = 0; hipStream_t st;
    int exclusive(const uint32_t* in, uint32_t* out, uint32_t n) {
        if (n == 0) return SDFHIP_OK;
        size_t need = 0;
        SDF_HIP_CHECK(devExclusiveSum(nullptr, need, in, out, (size_t)n, st));
        if (need > bytes) { SDF_TRY(tmp.reserve(need)); bytes = need; }
        SDF_HIP_CHECK(devExclusiveSum(tmp.p, need, in, out, (size_t)n, st));
        return SDFHIP_OK;
    }
};

// two scan totals (last offset + last value; a null pair counts 0) side by side for one read-back
__global__ void k_ex_totals2(const uint32_t* __restrict__ s0, const uint32_t* __restrict__ v0, const uint32_t* __restrict__ s1, const uint32_t* __restrict__ v1, uint32_t* __restrict__ out2) {
    if (threadIdx.x == 0) { out2[0] = s0 ? *s0 + *v0 : 0u; out2[1] = s1 ? *s1 + *v1 : 0u; }
}
static int lastPlus(hipStream_t st, const uint32_t* scan, const uint32_t* val, uint32_t n, uint32_t& total) {
    total = 0;
    if (n == 0) return SDFHIP_OK;
    return readBackWords(st, scan + (n - 1), val + (n - 1), 1, &total);
}

}  // namespace sdfhip

using namespace sdfhip;

// Top-down pass: absolute positions of every node / set / mask block, then the payloads.  All pointers are device buffers
// that the caller zeroed; grid* = the start-grid slots of this build (all cells, or the shard's cells in ascending z-major
// order), body* = the nodes from absolute position nodeOff on, sets / masks from setOff / maskOff on.
static int exactEmit(sdfhip_exact* E, uint32_t nodeOff, uint32_t setOff, uint32_t maskOff, uint32_t* gridNodes, uint8_t* gridHas,
                     uint32_t* bodyNodes, uint8_t* bodyHas, uint32_t* sets, uint8_t* masks) {
    hipStream_t st = E->ctx->stream;
    AllocScope allocScope(st);       // device buffers of this call come from the stream-ordered pool
    const sdfhip_exact_info& I = E->info;
    const uint32_t sod = E->sod, startDepth = I.start_depth, maxDepth = I.max_depth, bitEnc = I.bit_encoding_start_depth, bits = I.bits_per_index;
    std::vector<std::unique_ptr<ExLevel>>& LV = E->levels;
    ExLevel* S = LV[startDepth - sod].get();
    const uint32_t nCells = S->n;
    std::vector<uint32_t> oB(nCells), oS(nCells), oM(nCells);
    for (uint32_t i = 0; i < nCells; i++) { oB[i] = nodeOff + E->relB[i]; oS[i] = setOff + E->relS[i]; oM[i] = maskOff + E->relM[i]; }
    DevBuf<uint32_t> dB, dS, dM;
    SDF_TRY(dB.reserve(nCells)); SDF_TRY(dS.reserve(nCells)); SDF_TRY(dM.reserve(nCells));
    SDF_HIP_CHECK(hipMemcpyAsync(dB.p, oB.data(), 4ull * nCells, hipMemcpyHostToDevice, st));
    SDF_HIP_CHECK(hipMemcpyAsync(dS.p, oS.data(), 4ull * nCells, hipMemcpyHostToDevice, st));
    SDF_HIP_CHECK(hipMemcpyAsync(dM.p, oM.data(), 4ull * nCells, hipMemcpyHostToDevice, st));
    k_ex_init_cells<<<gridFor(nCells, 256), 256, 0, st>>>(nCells, dB.p, dS.p, dM.p, S->pos.p, S->blk.p, S->setPos.p, S->maskPos.p);
    for (uint32_t d = startDepth; d <= maxDepth; d++) {
        ExLevel* L = LV[d - sod].get();
        if (!L || L->n == 0) break;
        ExLevel* C = (d < maxDepth) ? LV[d + 1 - sod].get() : nullptr;
        const bool top = d == startDepth;
        OffArgs oa{L->n, d, bitEnc, L->flag.p, L->childBase.p, L->uLen.p, L->pos.p, L->blk.p, L->setPos.p, L->maskPos.p,
                   C ? C->aNode.p : nullptr, C ? C->aSet.p : nullptr, C ? C->aMask.p : nullptr,
                   C ? C->pos.p : nullptr, C ? C->blk.p : nullptr, C ? C->setPos.p : nullptr, C ? C->maskPos.p : nullptr, C ? C->maskIdx.p : nullptr,
                   top ? gridNodes : bodyNodes, top ? gridHas : bodyHas, top ? 0u : nodeOff, bodyNodes, bodyHas, nodeOff};
        k_ex_offsets<<<gridFor(8ull * L->n, 256), 256, 0, st>>>(oa);
        if (d <= bitEnc) {
            PackArgs pa{L->n, d, bitEnc, bits, L->flag.p, L->list.p, L->ulist.p, L->listOff.p, L->listLen.p, L->uLen.p, L->setPos.p, sets, setOff};
            k_ex_pack_sets<<<gridFor(64ull * L->n, 256), 256, 0, st>>>(pa);
        }
        if (d >= bitEnc && C && L->numInner > 0)
            k_ex_copy_masks<<<gridFor(64ull * L->n, 256), 256, 0, st>>>(L->n, d, bitEnc, L->inner.p, L->uLen.p, L->maskOff.p, L->masks.p, L->maskPos.p, L->childBase.p, C->aMask.p, masks, maskOff);
    }
    SDF_HIP_CHECK(hipGetLastError());
    SDF_HIP_CHECK(hipStreamSynchronize(st));      // dB/dS/dM go out of scope
    return SDFHIP_OK;
}

// rankRange (nullable) = [begin, end) of start cells in the reference's DFS emission order (dfsRank) built by this shard
static int exactBuildImpl(sdfhip_ctx* ctx, sdfhip_mesh* mesh, const float box_min[3], const float box_max[3], uint32_t maxDepth, uint32_t startDepth,
                          uint32_t minTri, const uint32_t* rankRange, sdfhip_exact** out) {
    SDF_REQUIRE(ctx && mesh && box_min && box_max && out, "NULL argument");
    SDF_REQUIRE(mesh->ctx == ctx, "mesh belongs to another context");
    std::lock_guard<std::recursive_mutex> building(ctx->buildLock);
    SDF_REQUIRE(maxDepth >= 2, "depth must be at least 2");
    if (maxDepth > 10) { setError("depth %u is above this build's limit of 10: node coordinates are packed 10 bits per axis (the reference's own limit is its 30-bit word index, OctreeSdf.h:53-55, which a depth-11 tree of a real surface exceeds anyway)", (unsigned)maxDepth); return SDFHIP_E_UNSUPPORTED; }
    SDF_REQUIRE(startDepth + 2 <= maxDepth, "start_depth must be <= max_depth - 2 (the reference dereferences a null node otherwise)");
    SDF_REQUIRE(mesh->numTriangles >= 2, "at least 2 triangles are needed (bits per index)");
    SDF_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    AllocScope allocScope(st);       // device buffers of this call come from the stream-ordered pool
    const double tStart = nowSeconds();
    static const bool timing = getenv("SDFHIP_TIMING") != nullptr;      // per-level wall times up to the level's last read-back (kernels enqueued after it are billed to the next level) and the allocator's share
    const double alloc0 = g_allocSeconds(); const long allocCalls0 = g_allocCalls();
    std::vector<double> levelSeconds;
    std::unique_ptr<sdfhip_exact> E(new sdfhip_exact());
    E->ctx = ctx; E->mesh = mesh;
    const float sx = box_max[0] - box_min[0], sy = box_max[1] - box_min[1], sz = box_max[2] - box_min[2];
    SDF_REQUIRE(sx > 0 && sy > 0 && sz > 0 && std::isfinite(sx) && std::isfinite(sy) && std::isfinite(sz), "empty or non-finite box");
    const float maxSize = gmax(gmax(sx, sy), sz);
    const float cx = box_min[0] + 0.5f * sx, cy = box_min[1] + 0.5f * sy, cz = box_min[2] + 0.5f * sz;
    float bmin[3] = {cx - 0.5f * maxSize, cy - 0.5f * maxSize, cz - 0.5f * maxSize};
    float bmax[3] = {cx + 0.5f * maxSize, cy + 0.5f * maxSize, cz + 0.5f * maxSize};
    sdfhip_exact_info& I = E->info;
    memcpy(I.box_min, bmin, 12); memcpy(I.box_max, bmax, 12);
    const uint32_t G = 1u << startDepth, G3 = G * G * G;
    I.start_grid_size = (int32_t)G; I.start_depth = startDepth; I.max_depth = maxDepth;
    I.bit_encoding_start_depth = maxDepth - 2; I.min_triangles_in_leafs = minTri; I.num_triangles = mesh->numTriangles;
    I.bits_per_index = (uint32_t)(int32_t)std::ceil(std::log2((float)mesh->numTriangles));
    E->cellSize = maxSize / (float)G; I.start_grid_cell_size = E->cellSize;
    const uint32_t bitEnc = maxDepth - 2, bits = I.bits_per_index;
    const uint32_t sod = startDepth < 1u ? startDepth : 1u;
    const uint32_t T = mesh->numTriangles;
    ExMesh md{mesh->dVerts.p, mesh->dIdx.p, mesh->dTri.p, mesh->dFrames.p};
    ScanHelper scan; scan.st = st;

    DevBuf<uint32_t> stats;          // [0] maxLeaf, [1] maxEncoded
    DevBuf<unsigned long long> cullTests;
    SDF_TRY(stats.reserve(2)); SDF_TRY(cullTests.reserve(1));
    SDF_HIP_CHECK(hipMemsetAsync(stats.p, 0, 8, st)); SDF_HIP_CHECK(hipMemsetAsync(cullTests.p, 0, 8, st));

    // the root list: every triangle with a usable normal, ascending
    DevBuf<uint32_t> allList, valid, vscan;
    uint32_t allLen = 0;
    SDF_TRY(valid.reserve(T)); SDF_TRY(vscan.reserve(T)); SDF_TRY(allList.reserve(T));
    k_valid_triangles<<<gridFor(T, 256), 256, 0, st>>>(md.td, T, valid.p);
    SDF_TRY(scan.exclusive(valid.p, vscan.p, T));
    k_compact_valid<<<gridFor(T, 256), 256, 0, st>>>(valid.p, vscan.p, T, allList.p);
    SDF_TRY(lastPlus(st, vscan.p, valid.p, T, allLen));
    SDF_REQUIRE(allLen > 0, "mesh has no usable triangle");

    std::vector<std::unique_ptr<ExLevel>>& LV = E->levels;
    LV.resize(maxDepth - sod + 1);
    {   // root level
        std::unique_ptr<ExLevel> L(new ExLevel());
        L->depth = sod; L->n = 1u << (3 * sod);
        const float newSize = (float)(0.5f * (bmax[0] - bmin[0]) * std::pow(0.5f, sod));
        L->half = newSize;
        SDF_TRY(L->center.reserve(3ull * L->n)); SDF_TRY(L->coord.reserve(L->n)); SDF_TRY(L->cornerTri.reserve(8ull * L->n));
        SDF_TRY(L->pOff.reserve(L->n)); SDF_TRY(L->pLen.reserve(L->n));
        std::vector<float> hc(3 * L->n); std::vector<uint32_t> hco(L->n), hz(L->n, 0u), hl(L->n, allLen);
        const float scx = bmin[0] + newSize, scy = bmin[1] + newSize, scz = bmin[2] + newSize;
        const uint32_t vpa = 1u << sod;
        for (uint32_t k = 0; k < vpa; k++) for (uint32_t j = 0; j < vpa; j++) for (uint32_t i = 0; i < vpa; i++) {
            const uint32_t r = i + vpa * j + vpa * vpa * k;
            hc[3 * r] = scx + ((float)i * 2.0f) * newSize; hc[3 * r + 1] = scy + ((float)j * 2.0f) * newSize; hc[3 * r + 2] = scz + ((float)k * 2.0f) * newSize;
            hco[r] = i | (j << 10) | (k << 20);
        }
        SDF_HIP_CHECK(hipMemcpyAsync(L->center.p, hc.data(), hc.size() * 4, hipMemcpyHostToDevice, st));
        SDF_HIP_CHECK(hipMemcpyAsync(L->coord.p, hco.data(), hco.size() * 4, hipMemcpyHostToDevice, st));
        SDF_HIP_CHECK(hipMemcpyAsync(L->pOff.p, hz.data(), hz.size() * 4, hipMemcpyHostToDevice, st));
        SDF_HIP_CHECK(hipMemcpyAsync(L->pLen.p, hl.data(), hl.size() * 4, hipMemcpyHostToDevice, st));
        // 8 corners of every root: brute force over ALL usable triangles
        {
            const uint32_t items = 8 * L->n;
            uint32_t slices = allLen / (256u * 16u); if (slices < 1u) slices = 1u; if (slices > 256u) slices = 256u;
            DevBuf<unsigned long long> keys;
            SDF_TRY(keys.reserve(items));
            SDF_HIP_CHECK(hipMemsetAsync(keys.p, 0xFF, 8ull * items, st));
            k_brute_nearest_sliced<256><<<items * slices, 256, 0, st>>>(md, L->center.p, L->half, L->n, 8, allList.p, L->pOff.p, L->pLen.p, nullptr, slices, keys.p);
            k_brute_finish<<<gridFor(items, 256), 256, 0, st>>>(L->n, 8, allList.p, L->pOff.p, L->pLen.p, nullptr, keys.p, L->cornerTri.p);
            SDF_HIP_CHECK(hipStreamSynchronize(st));
        }
        SDF_HIP_CHECK(hipStreamSynchronize(st));
        LV[0] = std::move(L);
    }

    const uint32_t* prevList = allList.p;
    levelSeconds.push_back(nowSeconds() - tStart);
    for (uint32_t d = sod; d <= maxDepth; d++) {
        ExLevel* L = LV[d - sod].get();
        if (!L || L->n == 0) break;
        const double tLevel = nowSeconds();
        if (d == startDepth && d > sod) {
            std::unique_ptr<ExLevel> R(new ExLevel());
            R->depth = d; R->n = G3; R->half = L->half;
            SDF_REQUIRE(L->n == G3, "internal: start level is not a full grid");
            SDF_TRY(R->center.reserve(3ull * G3)); SDF_TRY(R->coord.reserve(G3)); SDF_TRY(R->cornerTri.reserve(8ull * G3));
            SDF_TRY(R->pOff.reserve(G3)); SDF_TRY(R->pLen.reserve(G3));
            k_ex_to_cell_order<<<gridFor(8ull * L->n, 256), 256, 0, st>>>(L->center.p, L->coord.p, L->cornerTri.p, L->pOff.p, L->pLen.p, L->n, G,
                                                                           R->center.p, R->coord.p, R->cornerTri.p, R->pOff.p, R->pLen.p);
            SDF_HIP_CHECK(hipStreamSynchronize(st));
            LV[d - sod] = std::move(R);
            L = LV[d - sod].get();
        }
        if (d == startDepth && rankRange) {      // sharded build: continue with this shard's start cells only (ascending z-major)
            SDF_REQUIRE(L->n == G3, "internal: start level is not a full grid");
            std::vector<uint32_t>& sel = E->shardCells;
            for (uint32_t z = 0; z < G; z++) for (uint32_t y = 0; y < G; y++) for (uint32_t x = 0; x < G; x++) {
                const uint32_t r = dfsRank(x, y, z, startDepth);
                if (r >= rankRange[0] && r < rankRange[1]) sel.push_back(z * G * G + y * G + x);
            }
            const uint32_t ns = (uint32_t)sel.size();
            DevBuf<uint32_t> dSel; SDF_TRY(dSel.reserve(ns));
            SDF_HIP_CHECK(hipMemcpyAsync(dSel.p, sel.data(), 4ull * ns, hipMemcpyHostToDevice, st));
            std::unique_ptr<ExLevel> Q(new ExLevel());
            Q->depth = d; Q->n = ns; Q->half = L->half;
            SDF_TRY(Q->center.reserve(3ull * ns)); SDF_TRY(Q->coord.reserve(ns)); SDF_TRY(Q->cornerTri.reserve(8ull * ns)); SDF_TRY(Q->pOff.reserve(ns)); SDF_TRY(Q->pLen.reserve(ns));
            k_ex_select_cells<<<gridFor(8ull * ns, 256), 256, 0, st>>>(L->center.p, L->coord.p, L->cornerTri.p, L->pOff.p, L->pLen.p, dSel.p, ns,
                                                                       Q->center.p, Q->coord.p, Q->cornerTri.p, Q->pOff.p, Q->pLen.p);
            SDF_HIP_CHECK(hipStreamSynchronize(st));
            LV[d - sod] = std::move(Q);
            L = LV[d - sod].get();
        }
        const uint32_t n = L->n;
        // ---- cull the parent lists into this level's lists
        DevBuf<float> region, minDist; DevBuf<uint32_t> nChunks, chunkBase, chunkNode, chunkCount, chunkScan, tmp, longCount, totals2; DevBuf<uint2> longItems; DevBuf<unsigned long long> longKeys;
        SDF_TRY(region.reserve(64ull * n)); SDF_TRY(minDist.reserve(8ull * n)); SDF_TRY(nChunks.reserve(n)); SDF_TRY(chunkBase.reserve(n));
        k_node_regions<<<gridFor(64ull * n, 256), 256, 0, st>>>(md, L->center.p, L->half, n, L->cornerTri.p, region.p, minDist.p);
        k_chunk_counts<<<gridFor(n, 256), 256, 0, st>>>(n, L->pLen.p, nChunks.p);
        SDF_TRY(scan.exclusive(nChunks.p, chunkBase.p, n));
        uint32_t numChunks = 0;
        SDF_TRY(lastPlus(st, chunkBase.p, nChunks.p, n, numChunks));
        SDF_TRY(chunkNode.reserve(numChunks)); SDF_TRY(chunkCount.reserve(numChunks)); SDF_TRY(chunkScan.reserve(numChunks)); SDF_TRY(tmp.reserve((size_t)numChunks * CHUNK));
        SDF_TRY(L->listOff.reserve(n)); SDF_TRY(L->listLen.reserve(n));
        uint32_t total = 0;
        if (numChunks) {
            k_chunk_fill<<<gridFor(n, 256), 256, 0, st>>>(n, nChunks.p, chunkBase.p, chunkNode.p);
            CullArgs ca{md, L->center.p, L->half, L->cornerTri.p, region.p, minDist.p, prevList, L->pOff.p, L->pLen.p, chunkNode.p, chunkBase.p, numChunks,
                        tmp.p, chunkCount.p, cullTests.p};
            // chunks per wave: enough waves to fill the chip several times over, then as long a stream per wave as that leaves
            { uint32_t pw = numChunks / 32768u; if (pw < 1u) pw = 1u; if (pw > 16u) pw = 16u; ca.perWave = pw; }      // (divisors 4096 .. 65536 and caps 16 .. 64 measured: a plateau, profiles/r04_cull_experiments.txt)
            k_cull<<<gridFor(gridFor(numChunks, ca.perWave), 4), 256, 0, st>>>(ca);
            SDF_TRY(scan.exclusive(chunkCount.p, chunkScan.p, numChunks));
        }
        // The lists' lengths follow from the chunk counts alone, so leaf / inner is decided and the inner nodes are counted BEFORE the host
        // asks how long the level's list is: one read-back brings both totals (two per level until round 5).
        k_node_list_ranges<<<gridFor(n, 256), 256, 0, st>>>(n, nChunks.p, chunkBase.p, chunkScan.p, chunkCount.p, numChunks, L->listOff.p, L->listLen.p);
        SDF_TRY(L->flag.reserve(n)); SDF_TRY(L->inner.reserve(n)); SDF_TRY(L->childBase.reserve(n)); SDF_TRY(totals2.reserve(2));
        k_ex_decide<<<gridFor(n, 256), 256, 0, st>>>(n, d, startDepth, maxDepth, minTri, L->listLen.p, L->flag.p, L->inner.p, stats.p);
        if (d < maxDepth) SDF_TRY(scan.exclusive(L->inner.p, L->childBase.p, n));
        k_ex_totals2<<<1, 64, 0, st>>>(numChunks ? chunkScan.p + (numChunks - 1) : nullptr, numChunks ? chunkCount.p + (numChunks - 1) : nullptr,
                                       d < maxDepth ? L->childBase.p + (n - 1) : nullptr, d < maxDepth ? L->inner.p + (n - 1) : nullptr, totals2.p);
        uint32_t h2[2] = {0, 0};
        SDF_TRY(readBackWords(st, totals2.p, nullptr, 2, h2));
        total = h2[0]; L->numInner = h2[1];
        if (numChunks) {
            SDF_TRY(L->list.reserve(total));
            k_compact<<<gridFor(numChunks, 4), 256, 0, st>>>(tmp.p, chunkCount.p, chunkScan.p, numChunks, L->list.p);
        } else SDF_TRY(L->list.reserve(1));
        L->listTotal = total;
        if (d < maxDepth) k_mul8<<<gridFor(n, 256), 256, 0, st>>>(n, L->childBase.p);
        SDF_HIP_CHECK(hipGetLastError());
        if (L->numInner > 0) {
            SDF_TRY(L->midTri.reserve(19ull * n));
            // few nodes with long lists near the root: a block per (node, point) keeps the chip busy; many nodes with short lists below:
            // a block per node shares the staged frames among the 19 points
            {
                // many nodes with short lists: a block per node shares the staged frames among the 19 points (k_brute_nearest_mids); the nodes
                // with long lists (all of them near the root) are cut into pieces of BN_PIECE entries (k_brute_mids_long)
                const uint32_t longLen = BN_PIECE;
                const size_t maxItems = (size_t)L->listTotal / BN_PIECE + n + 1;
                SDF_TRY(longItems.reserve(maxItems)); SDF_TRY(longCount.reserve(1)); SDF_TRY(longKeys.reserve(19ull * n));
                SDF_HIP_CHECK(hipMemsetAsync(longCount.p, 0, 4, st));
                k_long_nodes<<<gridFor(n, 256), 256, 0, st>>>(n, L->listLen.p, L->flag.p, longLen, longItems.p, longCount.p, longKeys.p);
                k_brute_nearest_mids<<<n, 256, 0, st>>>(md, L->center.p, L->half, n, L->list.p, L->listOff.p, L->listLen.p, L->flag.p, L->midTri.p, longLen);
                k_brute_mids_long<<<2048, 256, 0, st>>>(md, L->center.p, L->half, L->list.p, L->listOff.p, L->listLen.p, longItems.p, longCount.p, longKeys.p);
                k_brute_long_finish<<<gridFor(19ull * n, 256), 256, 0, st>>>(n, L->list.p, L->listOff.p, L->listLen.p, L->flag.p, longLen, longKeys.p, L->midTri.p);
            }
            std::unique_ptr<ExLevel> N(new ExLevel());
            N->depth = d + 1; N->n = 8u * L->numInner; N->half = 0.5f * L->half;
            SDF_TRY(N->center.reserve(3ull * N->n)); SDF_TRY(N->coord.reserve(N->n)); SDF_TRY(N->cornerTri.reserve(8ull * N->n));
            SDF_TRY(N->pOff.reserve(N->n)); SDF_TRY(N->pLen.reserve(N->n));
            ExChildArgs xa{L->center.p, L->coord.p, L->cornerTri.p, L->midTri.p, L->inner.p, L->childBase.p, L->listOff.p, L->listLen.p, n, L->half,
                           N->center.p, N->coord.p, N->cornerTri.p, N->pOff.p, N->pLen.p};
            k_ex_children<<<gridFor(64ull * n, 256), 256, 0, st>>>(xa);
            SDF_HIP_CHECK(hipGetLastError());
            LV[d + 1 - sod] = std::move(N);
        }
        // (no wait: the blocks go back to the stream's cache, whose next user is queued behind the kernels that still read them.  This holds
        // because the whole exact build enqueues on ctx->stream only - a side stream or an exchange consumer here would need an event first)
        L->midTri.release(); L->center.release(); L->cornerTri.release();
        prevList = L->list.p;
        levelSeconds.push_back(nowSeconds() - tLevel);
    }

    // ---- merge steps, deepest first: depth maxDepth-1 then maxDepth-2
    for (int d = (int)maxDepth - 1; d >= (int)bitEnc; d--) {
        ExLevel* L = LV[d - sod].get();
        ExLevel* C = ((uint32_t)d + 1 <= maxDepth) ? LV[d + 1 - sod].get() : nullptr;
        if (!L || L->n == 0) continue;
        SDF_TRY(L->ulist.reserve(L->listTotal ? L->listTotal : 1)); SDF_TRY(L->uLen.reserve(L->n)); SDF_TRY(L->maskBytes.reserve(L->n)); SDF_TRY(L->maskOff.reserve(L->n));
        SDF_HIP_CHECK(hipMemsetAsync(L->uLen.p, 0, 4ull * L->n, st));
        if (L->numInner == 0 || !C) { SDF_HIP_CHECK(hipMemsetAsync(L->maskBytes.p, 0, 4ull * L->n, st)); SDF_HIP_CHECK(hipMemsetAsync(L->maskOff.p, 0, 4ull * L->n, st)); L->maskTotal = 0; SDF_TRY(L->masks.reserve(1)); continue; }
        MergeArgs ma{L->n, L->inner.p, L->childBase.p, L->list.p, L->listOff.p, L->listLen.p, L->ulist.p, L->uLen.p,
                     C->flag.p, C->list.p, C->listOff.p, C->listLen.p, C->ulist.p, C->uLen.p, nullptr, nullptr, stats.p + 1, (uint32_t)d == bitEnc ? 1 : 0};
        k_merge_union<<<gridFor(64ull * L->n, 256), 256, 0, st>>>(ma);
        k_mask_sizes<<<gridFor(L->n, 256), 256, 0, st>>>(L->n, L->inner.p, L->uLen.p, L->maskBytes.p);
        SDF_TRY(scan.exclusive(L->maskBytes.p, L->maskOff.p, L->n));
        SDF_TRY(lastPlus(st, L->maskOff.p, L->maskBytes.p, L->n, L->maskTotal));
        SDF_TRY(L->masks.reserve(L->maskTotal ? L->maskTotal : 1));
        ma.maskOff = L->maskOff.p; ma.masks = L->masks.p;
        k_merge_masks<<<gridFor(64ull * 8ull * L->n, 256), 256, 0, st>>>(ma);
        SDF_HIP_CHECK(hipGetLastError());
    }

    // ---- sizes bottom-up
    const uint32_t* cN = nullptr; const uint32_t* cS = nullptr; const uint32_t* cM = nullptr;
    for (int d = (int)maxDepth; d >= (int)startDepth; d--) {
        ExLevel* L = LV[d - sod].get();
        if (!L || L->n == 0) continue;
        SDF_TRY(L->aNode.reserve(L->n)); SDF_TRY(L->aSet.reserve(L->n)); SDF_TRY(L->aMask.reserve(L->n));
        SDF_TRY(L->pos.reserve(L->n)); SDF_TRY(L->blk.reserve(L->n)); SDF_TRY(L->setPos.reserve(L->n)); SDF_TRY(L->maskPos.reserve(L->n)); SDF_TRY(L->maskIdx.reserve(L->n));
        SizeArgs sa{L->n, (uint32_t)d, bitEnc, bits, L->flag.p, L->childBase.p, L->listLen.p, L->uLen.p, cN, cS, cM, L->aNode.p, L->aSet.p, L->aMask.p};
        k_ex_sizes<<<gridFor(L->n, 256), 256, 0, st>>>(sa);
        cN = L->aNode.p; cS = L->aSet.p; cM = L->aMask.p;
    }
    SDF_HIP_CHECK(hipGetLastError());
    // ---- cell offsets (relative to the shard's first body node / set word / mask byte) in the reference's single-thread DFS order
    ExLevel* S = LV[startDepth - sod].get();
    const uint32_t nCells = rankRange ? (uint32_t)E->shardCells.size() : G3;
    SDF_REQUIRE(S && S->n == nCells, "internal: start level missing");
    std::vector<uint32_t> hN(nCells), hS(nCells), hM(nCells);
    SDF_HIP_CHECK(hipMemcpyAsync(hN.data(), S->aNode.p, 4ull * nCells, hipMemcpyDeviceToHost, st));
    SDF_HIP_CHECK(hipMemcpyAsync(hS.data(), S->aSet.p, 4ull * nCells, hipMemcpyDeviceToHost, st));
    SDF_HIP_CHECK(hipMemcpyAsync(hM.data(), S->aMask.p, 4ull * nCells, hipMemcpyDeviceToHost, st));
    SDF_HIP_CHECK(hipStreamSynchronize(st));
    {
        std::vector<uint32_t> cellOfRank(G3), localOfCell(G3, 0xFFFFFFFFu);
        for (uint32_t z = 0; z < G; z++) for (uint32_t y = 0; y < G; y++) for (uint32_t x = 0; x < G; x++) cellOfRank[dfsRank(x, y, z, startDepth)] = z * G * G + y * G + x;
        if (rankRange) for (uint32_t i = 0; i < nCells; i++) localOfCell[E->shardCells[i]] = i;
        else for (uint32_t i = 0; i < G3; i++) localOfCell[i] = i;
        E->relB.assign(nCells, 0); E->relS.assign(nCells, 0); E->relM.assign(nCells, 0);
        uint64_t bn = 0, sw = 0, mb = 0;
        for (uint32_t r = rankRange ? rankRange[0] : 0u; r < (rankRange ? rankRange[1] : G3); r++) {
            const uint32_t li = localOfCell[cellOfRank[r]];
            E->relB[li] = (uint32_t)bn; E->relS[li] = (uint32_t)sw; E->relM[li] = (uint32_t)mb; bn += hN[li]; sw += hS[li]; mb += hM[li];
        }
        SDF_REQUIRE(G3 + bn < (1ull << 31) && sw < (1ull << 32) && mb < (1ull << 32), "structure too large");
        E->bodyNodes = bn;
        I.num_nodes = rankRange ? bn : G3 + bn; I.num_set_words = sw; I.num_mask_bytes = mb;
    }
    uint32_t hstats[2]; unsigned long long hcull = 0;
    SDF_HIP_CHECK(hipMemcpyAsync(hstats, stats.p, 8, hipMemcpyDeviceToHost, st));
    SDF_HIP_CHECK(hipMemcpyAsync(&hcull, cullTests.p, 8, hipMemcpyDeviceToHost, st));
    SDF_HIP_CHECK(hipStreamSynchronize(st));
    I.max_triangles_in_leafs = hstats[0]; I.max_triangles_encoded_in_leafs = hstats[1]; I.cull_tests = hcull;
    E->sod = sod;
    if (rankRange) {          // the shard keeps its levels until sdfhip_exact_emit_shard is told the absolute offsets
        E->isShard = true;
        I.seconds_total = nowSeconds() - tStart;
        *out = E.release();
        return SDFHIP_OK;
    }
    SDF_TRY(E->nodes.reserve(2 * I.num_nodes)); SDF_TRY(E->hasTri.reserve(I.num_nodes)); SDF_TRY(E->sets.reserve(I.num_set_words + 2)); SDF_TRY(E->masks.reserve(I.num_mask_bytes + 1));          // (+ the look-ahead padding k_exact_tiles reads, as the import paths add)
    SDF_HIP_CHECK(hipMemsetAsync(E->nodes.p, 0, 8ull * I.num_nodes, st)); SDF_HIP_CHECK(hipMemsetAsync(E->hasTri.p, 0, I.num_nodes, st));
    SDF_HIP_CHECK(hipMemsetAsync(E->sets.p, 0, 4ull * (I.num_set_words + 2), st));
    SDF_HIP_CHECK(hipMemsetAsync(E->masks.p, 0, I.num_mask_bytes + 1, st));
    SDF_TRY(exactEmit(E.get(), G3, 0u, 0u, E->nodes.p, E->hasTri.p, E->nodes.p + 2ull * G3, E->hasTri.p + G3, E->sets.p, E->masks.p));
    E->levels.clear();
    E->built = true;
    I.seconds_total = nowSeconds() - tStart;
    if (timing) {
        std::string per; char b[32];
        for (size_t i = 0; i < levelSeconds.size(); i++) { snprintf(b, sizeof b, "%s%.1f", i ? " " : "", 1e3 * levelSeconds[i]); per += b; }
        fprintf(stderr, "[sdfhip] exact build: %.1f ms (roots, then the levels from depth %u: %s ms); block cache / hipMalloc / hipFree inside: %ld calls, %.1f ms\n",
                1e3 * I.seconds_total, sod, per.c_str(), g_allocCalls() - allocCalls0, 1e3 * (g_allocSeconds() - alloc0));
    }
    *out = E.release();
    return SDFHIP_OK;
}

extern "C" {

int sdfhip_exact_build(sdfhip_ctx* ctx, sdfhip_mesh* mesh, const float box_min[3], const float box_max[3], uint32_t maxDepth, uint32_t startDepth,
                       uint32_t minTri, sdfhip_exact** out) {
    SDF_API_BEGIN
    return exactBuildImpl(ctx, mesh, box_min, box_max, maxDepth, startDepth, minTri, nullptr, out);
    SDF_API_END
}

int sdfhip_exact_build_shard(sdfhip_ctx* ctx, sdfhip_mesh* mesh, const float box_min[3], const float box_max[3], uint32_t maxDepth, uint32_t startDepth,
                             uint32_t minTri, uint32_t rank_begin, uint32_t rank_end, sdfhip_exact** out) {
    SDF_API_BEGIN
    SDF_REQUIRE(startDepth <= 10 && rank_begin < rank_end && rank_end <= (1u << (3 * startDepth)), "cell rank range must be a non-empty sub-range of [0, 8^start_depth)");
    const uint32_t range[2] = {rank_begin, rank_end};
    return exactBuildImpl(ctx, mesh, box_min, box_max, maxDepth, startDepth, minTri, range, out);
    SDF_API_END
}

int sdfhip_exact_shard_cells(sdfhip_exact* shard, uint32_t* out_cells) {
    SDF_API_BEGIN
    SDF_REQUIRE(shard && shard->isShard && out_cells, "not a shard");
    memcpy(out_cells, shard->shardCells.data(), 4ull * shard->shardCells.size());
    return SDFHIP_OK;
    SDF_API_END
}

int sdfhip_exact_emit_shard(sdfhip_exact* E, uint64_t node_offset, uint64_t set_offset, uint64_t mask_offset, uint32_t* dst_grid_nodes, uint8_t* dst_grid_has,
                            uint32_t* dst_body_nodes, uint8_t* dst_body_has, uint32_t* dst_sets, uint8_t* dst_masks, int where) {
    SDF_API_BEGIN
    SDF_REQUIRE(E && E->isShard && !E->levels.empty(), "not a shard, or already emitted");
    SDF_REQUIRE(dst_grid_nodes && dst_grid_has && dst_body_nodes && dst_body_has && dst_sets && dst_masks, "NULL destination");
    SDF_REQUIRE(node_offset + E->bodyNodes < (1ull << 31) && set_offset + E->info.num_set_words < (1ull << 32) && mask_offset + E->info.num_mask_bytes < (1ull << 32), "structure too large");
    SDF_HIP_CHECK(hipSetDevice(E->ctx->device));
    hipStream_t st = E->ctx->stream;
    AllocScope allocScope(st);       // device buffers of this call come from the stream-ordered pool
    const uint64_t nc = E->shardCells.size(), bn = E->bodyNodes, sw = E->info.num_set_words, mb = E->info.num_mask_bytes;
    DevBuf<uint32_t> tGrid, tBody, tSets; DevBuf<uint8_t> tGridHas, tBodyHas, tMasks;
    uint32_t* gN = dst_grid_nodes; uint8_t* gH = dst_grid_has; uint32_t* bN = dst_body_nodes; uint8_t* bH = dst_body_has; uint32_t* sS = dst_sets; uint8_t* mM = dst_masks;
    if (where == SDFHIP_HOST) {
        SDF_TRY(tGrid.reserve(2 * nc)); SDF_TRY(tGridHas.reserve(nc)); SDF_TRY(tBody.reserve(2 * bn + 1)); SDF_TRY(tBodyHas.reserve(bn + 1)); SDF_TRY(tSets.reserve(sw + 1)); SDF_TRY(tMasks.reserve(mb + 1));
        gN = tGrid.p; gH = tGridHas.p; bN = tBody.p; bH = tBodyHas.p; sS = tSets.p; mM = tMasks.p;
    }
    SDF_HIP_CHECK(hipMemsetAsync(gN, 0, 8 * nc, st)); SDF_HIP_CHECK(hipMemsetAsync(gH, 0, nc, st));
    if (bn) { SDF_HIP_CHECK(hipMemsetAsync(bN, 0, 8 * bn, st)); SDF_HIP_CHECK(hipMemsetAsync(bH, 0, bn, st)); }
    if (sw) SDF_HIP_CHECK(hipMemsetAsync(sS, 0, 4 * sw, st));
    if (mb) SDF_HIP_CHECK(hipMemsetAsync(mM, 0, mb, st));
    SDF_TRY(exactEmit(E, (uint32_t)node_offset, (uint32_t)set_offset, (uint32_t)mask_offset, gN, gH, bN, bH, sS, mM));
    if (where == SDFHIP_HOST) {
        SDF_HIP_CHECK(hipMemcpyAsync(dst_grid_nodes, gN, 8 * nc, hipMemcpyDeviceToHost, st)); SDF_HIP_CHECK(hipMemcpyAsync(dst_grid_has, gH, nc, hipMemcpyDeviceToHost, st));
        if (bn) { SDF_HIP_CHECK(hipMemcpyAsync(dst_body_nodes, bN, 8 * bn, hipMemcpyDeviceToHost, st)); SDF_HIP_CHECK(hipMemcpyAsync(dst_body_has, bH, bn, hipMemcpyDeviceToHost, st)); }
        if (sw) SDF_HIP_CHECK(hipMemcpyAsync(dst_sets, sS, 4 * sw, hipMemcpyDeviceToHost, st));
        if (mb) SDF_HIP_CHECK(hipMemcpyAsync(dst_masks, mM, mb, hipMemcpyDeviceToHost, st));
    }
    SDF_HIP_CHECK(hipStreamSynchronize(st));
    E->levels.clear();
    return SDFHIP_OK;
    SDF_API_END
}

int sdfhip_exact_destroy(sdfhip_exact* tree) { delete tree; return SDFHIP_OK; }

int sdfhip_exact_get_info(sdfhip_exact* tree, sdfhip_exact_info* out) {
    SDF_API_BEGIN
    SDF_REQUIRE(tree && out, "NULL argument");
    *out = tree->info;
    return SDFHIP_OK;
    SDF_API_END
}

int sdfhip_exact_download(sdfhip_exact* T, uint32_t* nodes, uint8_t* has, uint32_t* sets, uint8_t* masks) {
    SDF_API_BEGIN
    SDF_REQUIRE(T && nodes && has && sets && masks, "NULL argument");
    hipStream_t st = T->ctx->stream;
    SDF_HIP_CHECK(hipMemcpyAsync(nodes, T->nodes.p, 8ull * T->info.num_nodes, hipMemcpyDeviceToHost, st));
    SDF_HIP_CHECK(hipMemcpyAsync(has, T->hasTri.p, T->info.num_nodes, hipMemcpyDeviceToHost, st));
    if (T->info.num_set_words) SDF_HIP_CHECK(hipMemcpyAsync(sets, T->sets.p, 4ull * T->info.num_set_words, hipMemcpyDeviceToHost, st));
    if (T->info.num_mask_bytes) SDF_HIP_CHECK(hipMemcpyAsync(masks, T->masks.p, T->info.num_mask_bytes, hipMemcpyDeviceToHost, st));
    SDF_HIP_CHECK(hipStreamSynchronize(st));
    return SDFHIP_OK;
    SDF_API_END
}

}  // extern "C"

// (sdfhip_ctx_create: the runtime loads a translation unit's code object on the first use of one of its kernels — milliseconds that would
// otherwise land in the first build or the first query of a process)
namespace sdfhip { void loadKernelsExactBuild() { hipFuncAttributes a; (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_mul8)); (void)hipGetLastError(); } }
