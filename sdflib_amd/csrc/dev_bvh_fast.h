// Nearest-triangle search in two phases: an ORDER-FREE fp32 candidate search + an exact, order-aware resolution.  PRODUCT code.
//
// What has to come out is the id the reference's fp64 traversal returns (dev_bvh.h: nearer child first, strict '<', the farther child
// tested after the nearer subtree) — which, among triangles tied to within rounding (the nearest feature is an edge or a vertex for
// most points), depends on its visiting order.  dev_bvh.h reproduces that traversal step by step; its cost is the fp64 arithmetic of
// every visit (Eberly's point/triangle routine above all) executed by a wave in which few lanes need it at any moment.
// Here the work is split so that almost all of it is free of the order and of fp64:
//
//  1. k_near_quads — four lanes per query, any order, fp32 only: a branch-and-bound over the same tree (fp32 spheres, 48-byte triangle
//     records) with CONSERVATIVE tests: U2 is an upper bound of the true minimum squared distance (fp32 value + its error bound),
//     a child is skipped only if its sphere's lower bound (fp32 value - its error bound) exceeds U, a triangle is recorded as a
//     candidate iff its fp32 distance minus its error bound does not exceed U2.  Every triangle whose exact distance is within
//     ~1e-6 (relative) of the minimum ends up in the list (<= NEAR_K ids per query; overflow -> phase 3).
//  2. k_near_resolve — exact: the fp64 squared distance (the reference's formula, dev_bvh.h) of every candidate gives the minimum
//     d and the TIED set N = {d2 <= dmin2 (1 + 4e-12)}.  One member: it is the answer.  Otherwise the reference's decisions are
//     replayed on N alone: the triangles of N are visited in the reference's static order (at every inner node the child with the
//     smaller fp64 sphere distance first), a triangle is adopted iff d2 < best^2 (best = sqrt of the adopted d2), and after the
//     first adoption a subtree is entered only if its fp64 sphere distance is < best — the very comparisons of the reference.
//     Why that is the reference's answer: started with best = beta = dmin (1 + 1e-9) instead of +inf, the reference's traversal
//     visits a subset of its nodes in the same order; until it adopts its first triangle below beta the unbounded run has
//     best >= beta, so it visits and adopts that triangle too, and from then on the two runs are in the same state.  Triangles in
//     (dmin (1 + 4e-12), beta) can only change `best` by amounts far above the rounding that decides ties, i.e. no decision about
//     a member of N.  Before the first adoption every ancestor of a member of N passes its test (sphere bound <= d (1 + 4e-16) <
//     beta), so those tests need not be evaluated.  The tree is navigated without loads: inner nodes are numbered in pre-order and
//     a node over the sorted range [b, e) splits at (b + e) / 2, so ranges and child indices follow from arithmetic on the
//     triangles' ranks (triRank).
//  3. k_near_fallback — whatever 1 or 2 cannot decide with certainty (candidate list overflow, more than NEAR_MAX_TIES ties, a
//     zero distance) is answered by the exact traversal of dev_bvh.h.  The count is reported (sdfhip_octree_info.num_nearest_fallbacks).
#pragma once
#include "dev_bvh.h"
#include <string.h>

namespace sdfhip {
namespace {

#ifndef RESOLVE_WAVES
#define RESOLVE_WAVES 5
#endif
constexpr int NEAR_K = 16;
// (8 tie slots and a five-wave register budget: 16 KB of LDS per workgroup and 81 registers = five waves per SIMD — measured against
// 9 / 3: k_near_resolve 1.85 -> 1.42 ms per C2 build, tools/gpu_resolve_occ_ab.sh; a ninth tie goes to the exact traversal like a tenth)
#ifndef RESOLVE_MAX_TIES
#define RESOLVE_MAX_TIES 8
#endif
constexpr int NEAR_MAX_TIES = RESOLVE_MAX_TIES;
constexpr uint32_t NEAR_OVERFLOW = 0xFFu;
constexpr uint32_t NEAR_UNRESOLVED = 0xFFFFFFFFu;

// Eberly's routine in fp32 (same region logic as dev_bvh.h's fp64 one).  scale = |p - v0|^2 + |e0|^2 + |e1|^2 bounds the magnitude of
// every term of the result: |returned - exact| <= 4e-6 * scale for triangles that are not flagged degenerate (k_tri_verts).
SDF_DEV float pointTriangleSq32(F3 point, F3 v0, F3 v1, F3 v2, float& scale, float& c) {
    const F3 diff = v0 - point, e0 = v1 - v0, e1 = v2 - v0;
    const float a00 = dot(e0, e0), a01 = dot(e0, e1), a11 = dot(e1, e1);
    const float b0 = dot(diff, e0), b1 = dot(diff, e1);
    c = dot(diff, diff);
    scale = c + a00 + a11;
    const float det = fabsf(a00 * a11 - a01 * a01);
    const float s = a01 * b1 - a11 * b0;
    const float t = a01 * b0 - a00 * b1;
    enum { V0, V1, V2, E01, E02, R0, E12S, E12T };
    const float e12den = a00 - 2.0f * a01 + a11;
    const int alongE02 = (b1 >= 0) ? V0 : ((-b1 >= a11) ? V2 : E02);
    const int alongE01 = (b0 >= 0) ? V0 : ((-b0 >= a00) ? V1 : E01);
    const bool sNeg = s < 0, tNeg = t < 0;
    const int kIn = sNeg ? (tNeg ? ((b0 < 0) ? ((-b0 >= a00) ? V1 : E01) : alongE02) : alongE02) : (tNeg ? alongE01 : R0);
    const float r2a = a01 + b0, r2b = a11 + b1, n2 = r2b - r2a;
    const int k2 = (r2b > r2a) ? ((n2 >= e12den) ? V1 : E12S) : ((r2b <= 0) ? V2 : ((b1 >= 0) ? V0 : E02));
    const float r6a = a01 + b1, r6b = a00 + b0, n6 = r6b - r6a;
    const int k6 = (r6b > r6a) ? ((n6 >= e12den) ? V2 : E12T) : ((r6b <= 0) ? V1 : ((b0 >= 0) ? V0 : E01));
    const float n1 = a11 + b1 - a01 - b0;
    const int k1 = (n1 <= 0) ? V2 : ((n1 >= e12den) ? V1 : E12S);
    const bool inside = s + t <= det;
    const int kind = inside ? kIn : (sNeg ? k2 : (tNeg ? k6 : k1));
    float numer = sNeg ? n2 : (tNeg ? n6 : n1);
    numer = (kind == R0) ? 1.0f : ((kind == E01) ? -b0 : ((kind == E02) ? -b1 : ((kind >= E12S) ? numer : 0.0f)));
    const float denom = (kind == R0) ? det : ((kind == E01) ? a00 : ((kind == E02) ? a11 : ((kind >= E12S) ? e12den : 1.0f)));
    const float q = numer / denom;
    const float sq = (kind == R0) ? s * q : ((kind == E12S) ? q : 1.0f - q);
    const float tq = (kind == R0) ? t * q : ((kind == E12T) ? q : 1.0f - q);
    float d2 = sq * (a00 * sq + a01 * tq + 2.0f * b0) + tq * (a01 * sq + a11 * tq + 2.0f * b1) + c;
    d2 = (kind == V0) ? c : d2;
    d2 = (kind == V1) ? a00 + 2.0f * b0 + c : d2;
    d2 = (kind == V2) ? a11 + 2.0f * b1 + c : d2;
    d2 = (kind == E01) ? b0 * q + c : d2;
    d2 = (kind == E02) ? b1 * q + c : d2;
    return d2;
}

// fp32 distance of one triangle record with its error bound: (lower, upper) bounds of the exact squared distance
struct TriBounds { float lo, hi; };
SDF_DEV TriBounds triBounds32(float4 q0, float4 q1, float4 q2, F3 p);
SDF_DEV TriBounds triBounds32(const BvhDev& b, uint32_t t, F3 p) { return triBounds32(b.triV[3 * (size_t)t], b.triV[3 * (size_t)t + 1], b.triV[3 * (size_t)t + 2], p); }
SDF_DEV TriBounds triBounds32(float4 q0, float4 q1, float4 q2, F3 p) {
    float scale, c;
    const float d2 = pointTriangleSq32(p, F3{q0.x, q0.y, q0.z}, F3{q0.w, q1.x, q1.y}, F3{q1.z, q1.w, q2.x}, scale, c);
    const float slack = 4e-6f * scale + 1e-37f;
    if (q2.y != 0.f || !(scale < 1e37f) || !(d2 == d2)) return TriBounds{0.f, c * 1.000002f + 1e-37f};   // degenerate record, overflow or NaN: only |p - v0|^2 is trusted, as an upper bound
    return TriBounds{d2 - slack, d2 + slack};
}

// conservative lower bound of the distance to a sphere from its fp32 copy (sphereApprox32's bracket, doubled)
SDF_DEV float sphereLower32(float4 sp, F3 p, float coordScale) {
    const float dx = p.x - sp.x, dy = p.y - sp.y, dz = p.z - sp.z;
    const float a = __builtin_amdgcn_sqrtf(dx * dx + dy * dy + dz * dz);
    return (a - sp.w) - 2.0f * (1e-6f * (a + sp.w) + 5e-7f * coordScale);
}

SDF_DEV unsigned short halfRoundedDown(float f) {
    const _Float16 h = (_Float16)f;
    unsigned short bits = __builtin_bit_cast(unsigned short, h);
    if ((float)h > f) bits = (bits == 0u) ? (unsigned short)0x8001u : ((bits & 0x8000u) ? (unsigned short)(bits + 1u) : (unsigned short)(bits - 1u));
    return bits;
}
SDF_DEV unsigned short halfTowardsZero(float f) {
    typedef __fp16 half2_t __attribute__((ext_vector_type(2)));
    const half2_t h = __builtin_amdgcn_cvt_pkrtz(f, 0.f);
    return (unsigned short)(__builtin_bit_cast(uint32_t, h) & 0xFFFFu);
}
SDF_DEV float halfBitsToFloat(unsigned short bits) { return (float)__builtin_bit_cast(_Float16, bits); }

// Lower bounds of the distance from p to the four children of a wide node (layout: dev_bvh.h) and their references.
// A child is a sphere AND a slab |m . (x - c)| <= W: the distance to their intersection is at least
//   sqrt( max(h - W, 0)^2 / |m|^2 + max(rho - r, 0)^2 ),  h = |m . v|, rho^2 = |v|^2 - h^2 / |m|^2, v = p - c,
// and at least |v| - r.  For a point "above" a smooth patch of the surface the sphere bound is loose by the patch's radius, the
// slab's by its sagitta only; measured on the bumpy sphere it cuts the visits of a far-field query by three.
// Every term is rounded in the conservative direction (|m| = 1 +- 1e-4, fp32 products within 2e-6 of |v|).
// bound of ONE child record q of a wide node whose header is h
SDF_DEV float childBound(float4 h, float4 q, F3 p) {
    const uint32_t w0 = __float_as_uint(q.x), w1 = __float_as_uint(q.y), w2 = __float_as_uint(q.z), w3 = __float_as_uint(q.w);
    const float vx = p.x - fmaf((float)(w0 & 0xFFFFu), h.w, h.x), vy = p.y - fmaf((float)(w0 >> 16), h.w, h.y), vz = p.z - fmaf((float)(w1 & 0xFFFFu), h.w, h.z);
    const float rad = halfBitsToFloat((unsigned short)(w1 >> 16)), W = halfBitsToFloat((unsigned short)(w3 >> 16));
    const float mx = (float)(short)(w2 & 0xFFFFu) * (1.0f / 32767.0f), my = (float)(short)(w2 >> 16) * (1.0f / 32767.0f), mz = (float)(short)(w3 & 0xFFFFu) * (1.0f / 32767.0f);
    const float v2 = fmaf(vx, vx, fmaf(vy, vy, vz * vz));
    const float a = __builtin_amdgcn_sqrtf(v2);
    const float ls = (a - rad) - 2e-6f * (a + rad);
    const float hm = fabsf(fmaf(mx, vx, fmaf(my, vy, mz * vz)));
    const float up = fmaxf(hm * 0.9999f - 3e-6f * a - W, 0.f) * 0.9999f;                       // (h - W) / |m|, rounded down
    const float rho2 = fmaxf(v2 * 0.999996f - hm * hm * 1.0003f - 6e-6f * v2, 0.f);           // |v|^2 - h^2 / |m|^2, rounded down
    const float lat = fmaxf(__builtin_amdgcn_sqrtf(rho2) * 0.999999f - rad, 0.f);
    const float ld = __builtin_amdgcn_sqrtf(fmaf(up, up, lat * lat)) * 0.999998f;
    return rad < 0.f ? 3.4e38f : fmaxf(ls, (W < 6.0e4f) ? ld : -3.4e38f);
}
template <typename ND>
SDF_DEV void wideBounds(ND nd, F3 p, float l[4], uint32_t cr[4]) {
    const float4 h = nd[0], rf = nd[5];
    cr[0] = __float_as_uint(rf.x); cr[1] = __float_as_uint(rf.y); cr[2] = __float_as_uint(rf.z); cr[3] = __float_as_uint(rf.w);
#pragma unroll
    for (int c = 0; c < 4; c++) l[c] = childBound(h, nd[1 + c], p);
}

// ---- phase 1, QUAD form -------------------------------------------------------------------------------------------------
// Four lanes per query: lane c of a quad tests child c of the popped wide node (one 16-byte record each: a quad reads the node's
// 64 bytes of child records as one contiguous request), the survivors are ranked inside the quad (three DPP rotations) and pushed on
// the QUAD's stack, nearest on top; surviving triangles go to the quad's queue and are evaluated four at a time, one per lane.
// Against one lane per query: a quarter of the stack per lane (16 queries x 40 entries x 6 B = 3.8 KB per wave instead of 12 KB, so
// the LDS no longer caps the occupancy at three waves per SIMD), no compare-exchange network, 55 instead of 214 instructions between
// two dependent fetches, and the four lanes of a quad never diverge in the expansion.
constexpr int QUAD_STACK = 40, QUAD_TQ = 12;
template <int CTRL> SDF_DEV float quadPermF(float x) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, false)); }
template <int CTRL> SDF_DEV uint32_t quadPermU(uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xF, 0xF, false); }
SDF_DEV uint32_t quadBallot(bool pred, uint32_t lane) { return (uint32_t)(__ballot(pred) >> (lane & ~3u)) & 0xFu; }

// Launch constants of the candidate search, each the winner of a measured sweep (profiles/r04g_near_seeds_roundtrip_ab.txt):
constexpr uint32_t NEAR_TWO_PASS_MIN = 524288;  // batches of this many queries and more run as leaders + followers (32768 until late round 5: below ~0.5 M the leaders' pass is a
                                                // second latency-bound sweep that costs more than its seeds save - a C2 build 11.85 -> 11.65 ms, an eighth of it as a shard 5.6 -> 5.1 ms)
constexpr uint32_t NEAR_LEAD = 8;               // every 8th query of the Morton order is a leader (4: 9.2 ms, 16: 9.0 ms against 8.8 per C2 build); a power of two <= 128
constexpr uint32_t NEAR_QBLOCKS_PER_CU = 6;     // resident workgroups of 256 per CU (7: 9.06 ms, 8: 9.44 ms: more waves only add L2 misses)
constexpr uint32_t NEAR_QCHUNK = 16;            // queries a wave (16 quads) takes per atomic (32, 64: slower)
constexpr uint32_t NEAR_MULTISEED = 2;          // leaders a follower is seeded from: the one before it and the one after it
constexpr uint32_t NEAR_REFILL_MIN = 1;         // idle quads a refill waits for (2 - 4: 8.71 / 8.82 / 8.93 ms, within noise of 1)
constexpr int NEAR_DRAIN_QUAD_LANES = 40;       // lanes with a triangle to test that make a drain round worth its instructions (32: 8.85, 48: 9.0 ms)
constexpr uint32_t NEAR_MAX_STEPS = 1536;       // pops after which a query is handed to k_near_long (one wave per query; 768 / 384: the same build times, 256: +10 %, 160: x2.4)
// ... of a LARGE batch.  A pop is one dependent memory round trip (~0.5 us), so the longest query a launch may hold takes 1536 x 0.5 us = 0.75 ms
// however few queries there are.  Round 6 tried handing queries of batches below NEAR_TWO_PASS_MIN over after fewer pops, because every launch of
// a 1/8 shard (47 k / 172 k / 331 k queries) costs ~0.45 ms more than its work: no gain (profiles/r06_near_small_batches.txt) - the switch stays.
constexpr uint32_t NEAR_MAX_STEPS_SMALL = 1536;     // (measured: no gain from 768 / 384, a loss at 192 - the chains that set a small launch's time are a few hundred pops long)

// PROBE: the per-query counters and seeds of the dev probes (sdfhip_mesh_nearest_stats); the builds' instantiation has neither.  The launch
// constants above are compile-time values in the kernel (kernel arguments until late round 5: seven scalar registers and a division by `lead`).
template <int BLOCK, bool PROBE, uint32_t MAXSTEPS = NEAR_MAX_STEPS>
__global__ void __launch_bounds__(BLOCK) k_near_quads(BvhDev b, const float* __restrict__ pos, uint32_t numReps, uint32_t* __restrict__ cand, float* __restrict__ candLo,
                                                      uint8_t* __restrict__ candCount, float* __restrict__ candU2, uint32_t rank, uint32_t world, uint32_t* __restrict__ counters,
                                                      uint32_t* __restrict__ longList, uint32_t* __restrict__ longCount,
                                                      uint32_t* __restrict__ perQueryArg, const uint32_t* __restrict__ seedTriArg, int pass, uint32_t* __restrict__ best) {
    constexpr uint32_t maxSteps = MAXSTEPS, chunk = NEAR_QCHUNK, lead = NEAR_LEAD, multiSeed = NEAR_MULTISEED, refillMin = NEAR_REFILL_MIN;
    constexpr int drainQuads = NEAR_DRAIN_QUAD_LANES; constexpr bool seedFromNeighbour = true;
    uint32_t* const perQuery = PROBE ? perQueryArg : nullptr; const uint32_t* const seedTri = PROBE ? seedTriArg : nullptr;
    __shared__ uint32_t s_ref[BLOCK / 64][QUAD_STACK][16];
    __shared__ unsigned short s_lb[BLOCK / 64][QUAD_STACK][16];
    __shared__ uint32_t s_tq[BLOCK / 64][QUAD_TQ][16];
    const uint32_t lane = __lane_id(), c = lane & 3u, quad = lane >> 2, wv = threadIdx.x >> 6;
    uint32_t (*stk)[16] = s_ref[wv]; unsigned short (*lbs)[16] = s_lb[wv]; uint32_t (*tq)[16] = s_tq[wv];
    const uint32_t allBlocks = (numReps + 127u) / 128u;
    const uint32_t mine = allBlocks > rank ? (allBlocks - rank + world - 1u) / world : 0u;
    const uint32_t all = mine * 128u;
    int phase = pass;                            // wave-uniform: 0 = one sweep; 1 = the leaders (every lead-th query), then 2 = the followers
    uint32_t total = phase == 0 ? all : all / lead;
    uint32_t per = (((total + 127u) / 128u + 7u) / 8u) * 128u;
    uint32_t xcd = blockIdx.x & 7u, tried = 0, chunkNext = 0, chunkEnd = 0;
    // per query; identical in the four lanes of its quad
    uint32_t r = 0; F3 p = F3{0.f, 0.f, 0.f};
    float U = 3.0e38f, U2 = 3.0e38f;
    uint32_t nc = 0, steps = 0, lastTri = 0xFFFFFFFFu, stExpand = 0, stIter = 0, stTri = 0;
    uint32_t accExpand = 0, accTri = 0;          // this quad's totals over the launch (one atomic per wave at the end)
    int sp = 0, nq = 0, mode = 0, seedRef = 0, qpass = 0;
    bool have = false, done = false, overflow = false;
    for (;;) {
        // ---- refill: quads without a query draw from the wave's chunk (one atomic per chunk)
        // (the refill is a stretch of ~150 instructions and a memory round trip that the WHOLE wave executes for the quads that draw:
        // it waits until refillMin quads are idle — or nobody has work left)
        uint64_t idle = __ballot(c == 0u && !have && !done);
        if (idle != 0ull && (uint32_t)__popcll(idle) < refillMin && __ballot(have) != 0ull) idle = 0ull;
        while (idle != 0ull) {
            if (chunkNext >= chunkEnd) {
                if (tried >= 8u && phase == 1) {
                    phase = 2; tried = 0; xcd = blockIdx.x & 7u;
                    total = all - all / lead; per = (((total + 127u) / 128u + 7u) / 8u) * 128u;
                    continue;
                }
                if (tried >= 8u) { if (!have) done = true; break; }
                const uint32_t lo = xcd * per, hi = (lo + per < total) ? lo + per : total;
                uint32_t base = 0;
                if (lane == 0u) base = atomicAdd(counters + xcd + (phase == 2 ? 10u : 0u), chunk);
                base = __shfl(base, 0) + lo;
                if (base >= hi) { tried++; xcd = (xcd + 1u) & 7u; continue; }
                chunkNext = base; chunkEnd = (base + chunk < hi) ? base + chunk : hi;
            }
            const uint32_t avail = chunkEnd - chunkNext;
            const uint32_t slot = (uint32_t)__popcll(idle & ((1ull << (lane & ~3u)) - 1ull));
            if (!have && !done && slot < avail) {
                const uint32_t j = chunkNext + slot;
                const uint32_t q = phase == 0 ? j : (phase == 1 ? lead * j : j + j / (lead - 1u) + 1u);
                const uint32_t rr = ((q >> 7) * world + rank) * 128u + (q & 127u);
                if (rr < numReps) {
                    r = rr; have = true; qpass = phase;
                    p = F3{pos[3 * (size_t)r], pos[3 * (size_t)r + 1], pos[3 * (size_t)r + 2]};
                    U = 3.0e38f; U2 = 3.0e38f; nc = 0; overflow = false; nq = 0; sp = 0; steps = 0; stExpand = stIter = stTri = 0;
                    // Up to THREE seed triangles, one per lane (the quad evaluates them in one drain round, i.e. at the price of one): the
                    // triangles the leaders on either side of a follower ended on (the follower lies between them on the Morton curve) and
                    // the one this quad's previous query ended on.  The smallest of the three distances is the first bound.
                    uint32_t seed = 0xFFFFFFFFu;
                    if (seedTri) { if (c == 0u) seed = seedTri[r]; }
                    else if (c == 2u) { if (seedFromNeighbour) seed = lastTri; }
                    else if (qpass == 2 && c < multiSeed) {
                        uint32_t rl = r & ~(lead - 1u);
                        if (c == 1u) rl = ((rl & 127u) + lead < 128u) ? rl + lead : rl + lead + (world - 1u) * 128u;      // the next leader (the next of this rank's 128-query blocks)
                        if (rl < numReps) seed = __builtin_nontemporal_load(best + rl);
                    }
                    const uint32_t nibS = quadBallot(seed < b.numTriangles, lane);
                    if (b.numTriangles == 1u) { if (c == 0u) tq[0][quad] = 0u; nq = 1; mode = 0; }
                    else if (nibS != 0u) {                                                                       // one drain round gives the first bound
                        if (seed < b.numTriangles) tq[__popc(nibS & ((1u << c) - 1u))][quad] = seed;
                        nq = __popc(nibS); mode = 2;
                    }
                    else { mode = 1; seedRef = 0; }                                                              // a greedy descent does
                }
            }
            const uint32_t want = (uint32_t)__popcll(idle);
            chunkNext += want < avail ? want : avail;
            idle = __ballot(c == 0u && !have && !done);
        }
        if (__ballot(have) == 0ull) break;
        if (have) stIter++;
        __builtin_amdgcn_wave_barrier();
        // ---- one pop per walking quad (always an inner node: triangles never enter the stack)
        if (have && mode == 2 && nq == 0) { mode = 0; if (c == 0u) { stk[0][quad] = 0u; lbs[0][quad] = (unsigned short)0xFBFFu; } sp = 1; }      // seeded: the root, bound = -65504
        __builtin_amdgcn_wave_barrier();
        int ref = -1;
        if (have && mode != 2 && nq <= QUAD_TQ - 4) {
            if (mode == 1) ref = seedRef;
            else if (sp > 0) {
                sp--; steps++;
                const int e = (int)stk[sp][quad]; const float lbound = halfBitsToFloat(lbs[sp][quad]);
                if (!(lbound > U)) { if (sp + 4 > QUAD_STACK) steps = 0xFFFFFFF0u; else ref = e; }        // (a stack about to overflow: a job for k_near_long)
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (ref >= 0) {
            const float4* nd = b.wide + 8 * (size_t)ref;
            float4 h = nd[0], q = nd[1 + c];
            uint32_t cr = reinterpret_cast<const uint32_t*>(nd + 5)[c];
            // ONE memory round trip per expansion: left alone, the compiler fetches the radius word first, branches on childBound's
            // "no such child" early-out and only then asks for the header and the rest of the record (two dependent latencies per pop)
            asm volatile("" : "+v"(h.x), "+v"(h.y), "+v"(h.z), "+v"(h.w), "+v"(q.x), "+v"(q.y), "+v"(q.z), "+v"(q.w), "+v"(cr));
            const float lb = childBound(h, q, p);
            stExpand++;
            if (mode == 1) {
                float m = fminf(lb, quadPermF<0xB1>(lb)); m = fminf(m, quadPermF<0x4E>(m));
                const uint32_t first = (uint32_t)__builtin_ctz(quadBallot(lb == m, lane) | 16u);
                seedRef = (int)__shfl((int)cr, (int)((lane & ~3u) + (first & 3u)));
                if (seedRef < 0) { if (c == 0u) tq[nq][quad] = (uint32_t)~seedRef; nq++; mode = 2; }
            } else {
                const bool surv = !(lb > U);
                const bool isTri = surv && (int)cr < 0, isNode = surv && (int)cr >= 0;
                const uint32_t nibT = quadBallot(isTri, lane), nibN = quadBallot(isNode, lane);
                if (isTri) tq[nq + __popc(nibT & ((1u << c) - 1u))][quad] = ~cr;
                nq += __popc(nibT);
                // position from the bottom = the number of surviving nodes FARTHER than mine (ties: the higher lane counts as farther)
                const float key = isNode ? lb : -3.4e38f;
                const float k1 = quadPermF<0x39>(key), k2 = quadPermF<0x4E>(key), k3 = quadPermF<0x93>(key);
                const uint32_t c1 = (c + 1u) & 3u, c2 = (c + 2u) & 3u, c3 = (c + 3u) & 3u;
                const int farther = ((k1 > key || (k1 == key && c1 > c)) ? 1 : 0) + ((k2 > key || (k2 == key && c2 > c)) ? 1 : 0) + ((k3 > key || (k3 == key && c3 > c)) ? 1 : 0);
                // the bound as a half rounded TOWARDS ZERO (one instruction): down for a positive bound; a negative one (the point is inside the
                // child's volume) becomes a value <= 0, which like the exact one never exceeds U at the pop
                if (isNode) { stk[sp + farther][quad] = cr; lbs[sp + farther][quad] = halfTowardsZero(lb); }
                sp += __popc(nibN);
            }
        }
        __builtin_amdgcn_wave_barrier();
        // ---- one drain round (up to four triangles per quad, one per lane) when enough LANES would have one to test, a quad has no room
        // left for a node's four, or nobody walks any more
        const uint64_t pendQ = __ballot(have && (int)c < nq);         // the lanes a drain round would occupy (c < min(nq, 4))
        if (pendQ != 0ull && (__popcll(pendQ) >= drainQuads || __ballot(have && nq > QUAD_TQ - 4) != 0ull || __ballot(have && mode != 2 && (mode == 1 || sp > 0)) == 0ull)) {
            const int take = nq < 4 ? nq : 4;
            const bool mineT = have && (int)c < take;
            uint32_t t = 0u; TriBounds tb{3.4e38f, 3.4e38f};
            if (mineT) { t = tq[nq - 1 - (int)c][quad]; tb = triBounds32(b, t, p); }
            if (have) stTri += (uint32_t)take;
            if (have && take > 0) {
                nq -= take;
                const float hi = mineT ? tb.hi : 3.4e38f;
                float m = fminf(hi, quadPermF<0xB1>(hi)); m = fminf(m, quadPermF<0x4E>(m));
                if (m < U2) {
                    U2 = m; U = __builtin_amdgcn_sqrtf(U2) * 1.000001f + 1e-37f;
                    const uint32_t first = (uint32_t)__builtin_ctz(quadBallot(mineT && hi == m, lane) | 16u);
                    lastTri = (uint32_t)__shfl((int)t, (int)((lane & ~3u) + (first & 3u)));
                }
                if (mode != 2) {                        // (the seed triangle only lends its bound: the search meets it again)
                    const bool rec = mineT && tb.lo <= U2;
                    const uint32_t nibR = quadBallot(rec, lane);
                    const uint32_t k = (uint32_t)__popc(nibR);
                    if (nc + k > (uint32_t)NEAR_K) {    // rare: drop the entries the bound has overtaken since they were recorded (lane 0 of the quad)
                        // (lane 0 reads entries the quad's other lanes stored in earlier drain rounds, through __restrict__ pointers: the stores are
                        // made visible and the loads kept behind them explicitly)
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        uint32_t keep = nc;
                        if (c == 0u) {
                            uint32_t keepMask = 0;
                            for (uint32_t i = 0; i < nc; i++) keepMask |= (candLo[(size_t)i * numReps + r] <= U2) ? (1u << i) : 0u;
                            keep = 0;
                            for (uint32_t i = 0; i < nc; i++)
                                if ((keepMask >> i) & 1u) {
                                    if (i != keep) { cand[(size_t)keep * numReps + r] = cand[(size_t)i * numReps + r]; candLo[(size_t)keep * numReps + r] = candLo[(size_t)i * numReps + r]; }
                                    keep++;
                                }
                        }
                        nc = quadPermU<0x00>(keep);
                    }
                    if (nc + k <= (uint32_t)NEAR_K) {
                        if (rec) { const uint32_t at = nc + (uint32_t)__popc(nibR & ((1u << c) - 1u)); cand[(size_t)at * numReps + r] = t; candLo[(size_t)at * numReps + r] = tb.lo; }
                        nc += k;
                    } else overflow = true;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        // ---- finished queries; long ones go to k_near_long
        if (have && sp == 0 && nq == 0 && mode == 0) {
            if (c == 0u) {
                candCount[r] = (uint8_t)(overflow ? NEAR_OVERFLOW : nc); candU2[r] = U2;
                if (overflow) longList[atomicAdd(longCount, 1u)] = r;          // more candidates than the list holds: k_near_long collects up to 1024 and reduces them to the exact ties
                if (qpass == 1) best[r] = lastTri;
                if (perQuery) { perQuery[4 * (size_t)r + 1] = stExpand; perQuery[4 * (size_t)r + 2] = stIter; perQuery[4 * (size_t)r + 3] = stTri; }
            }
            accExpand += stExpand; accTri += stTri;
            have = false;
        }
        if (have && steps > maxSteps) {
            if (c == 0u) { longList[atomicAdd(longCount, 1u)] = r; candCount[r] = (uint8_t)NEAR_OVERFLOW; if (qpass == 1) best[r] = lastTri; }
            accExpand += stExpand; accTri += stTri;
            have = false;
        }
    }
    // the launch's work counters (sdfhip_octree_info.near_expansions / near_triangle_tests): a sum over the wave's quads, two atomics per wave
    unsigned long long e = c == 0u ? accExpand : 0u, t = c == 0u ? accTri : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { e += __shfl_xor(e, o); t += __shfl_xor(t, o); }
    if (lane == 0u) { atomicAdd(reinterpret_cast<unsigned long long*>(counters + 18), e); atomicAdd(reinterpret_cast<unsigned long long*>(counters + 20), t); }
}

// ---- phase 1b: one wave per long query ------------------------------------------------------------------------------------
// The stack is shared by the wave: every step pops up to 64 entries (one per lane), expands them, evaluates the popped triangles,
// lowers the common bound to the best of the 64 and pushes the survivors (farthest children first).  Candidates are collected in
// LDS with their lower bounds and filtered against the FINAL bound before they are written out.
constexpr int NEAR_LONG_STACK = 6144, NEAR_LONG_CAND = 1024;
__global__ void __launch_bounds__(64) k_near_long(BvhDev b, const float* __restrict__ pos, uint32_t numReps, const uint32_t* __restrict__ longList,
                                                  const uint32_t* __restrict__ longCount, uint32_t* __restrict__ cand, float* __restrict__ candLo, uint8_t* __restrict__ candCount, float* __restrict__ candU2,
                                                  uint32_t* __restrict__ why) {
    __shared__ uint2 s_stack[NEAR_LONG_STACK];
    __shared__ uint2 s_cand[NEAR_LONG_CAND];
    const uint32_t lane = threadIdx.x;
    const uint64_t below = (1ull << lane) - 1ull;
    const uint32_t count = *longCount;
    for (uint32_t qi = blockIdx.x; qi < count; qi += gridDim.x) {
        const uint32_t r = longList[qi];
        const F3 p = F3{pos[3 * (size_t)r], pos[3 * (size_t)r + 1], pos[3 * (size_t)r + 2]};
        float U = 3.0e38f, U2 = 3.0e38f;
        uint32_t size = 1, nc = 0; bool overflow = false;
        __syncthreads();
        if (lane == 0) s_stack[0] = make_uint2(0u, 0xFF7FFFFFu);      // the root, bound = -FLT_MAX
        __syncthreads();
        while (size > 0 && !overflow) {
            const uint32_t take = size < 64u ? size : 64u;
            const bool active = lane < take;
            const uint2 e = active ? s_stack[size - 1u - lane] : make_uint2(0u, 0u);
            size -= take;
            __syncthreads();                                           // everybody has read its entry before anything is pushed
            const int ref = (int)e.x;
            const bool live = active && !(__uint_as_float(e.y) > U);
            // triangles first: they lower the bound the children are tested against
            float lo = 3.4e38f, hi = 3.4e38f;
            if (live && ref < 0) { const TriBounds tb = triBounds32(b, (uint32_t)~ref, p); lo = tb.lo; hi = tb.hi; }
            float m = hi;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) m = fminf(m, __shfl_xor(m, o));
            if (m < U2) { U2 = m; U = __builtin_amdgcn_sqrtf(U2) * 1.000001f + 1e-37f; }
            const bool isCand = live && ref < 0 && lo <= U2;
            {
                const uint64_t cm = __ballot(isCand);
                if (nc + (uint32_t)__popcll(cm) > (uint32_t)NEAR_LONG_CAND) {      // compact the collected candidates against the current bound
                    __syncthreads();
                    uint32_t keep = 0;
                    for (uint32_t base = 0; base < nc; base += 64u) {
                        const bool in = base + lane < nc;
                        const uint2 c = in ? s_cand[base + lane] : make_uint2(0u, 0u);
                        const bool k = in && __uint_as_float(c.y) <= U2;
                        const uint64_t km = __ballot(k);
                        __syncthreads();
                        if (k) s_cand[keep + (uint32_t)__popcll(km & below)] = c;
                        keep += (uint32_t)__popcll(km);
                        __syncthreads();
                    }
                    nc = keep;
                    if (nc + (uint32_t)__popcll(cm) > (uint32_t)NEAR_LONG_CAND) { overflow = true; if (lane == 0) atomicAdd(why + 1, 1u); }
                }
                if (!overflow) {
                    if (isCand) s_cand[nc + (uint32_t)__popcll(cm & below)] = make_uint2((uint32_t)~ref, __float_as_uint(lo));
                    nc += (uint32_t)__popcll(cm);
                }
            }
            // inner nodes: bounds of the four children
            float l[4] = {3.4e38f, 3.4e38f, 3.4e38f, 3.4e38f}; uint32_t cr[4] = {0u, 0u, 0u, 0u};
            if (live && ref >= 0) {
                wideBounds(b.wide + 8 * (size_t)ref, p, l, cr);
#define SDF_CE(i, j) { const bool sw_ = l[j] < l[i]; const float tl = sw_ ? l[j] : l[i], th = sw_ ? l[i] : l[j]; const uint32_t rl = sw_ ? cr[j] : cr[i], rh = sw_ ? cr[i] : cr[j]; l[i] = tl; l[j] = th; cr[i] = rl; cr[j] = rh; }
                SDF_CE(0, 1) SDF_CE(2, 3) SDF_CE(0, 2) SDF_CE(1, 3) SDF_CE(1, 2)
#undef SDF_CE
            }
            // push: the lanes' farthest children first, their nearest last (popped next)
#pragma unroll
            for (int c = 3; c >= 0; c--) {
                const bool keep = !(l[c] > U);
                const uint64_t km = __ballot(keep);
                const uint32_t n = (uint32_t)__popcll(km);
                if (size + n > (uint32_t)NEAR_LONG_STACK) { overflow = true; if (lane == 0) atomicAdd(why + 2, 1u); break; }
                if (keep) s_stack[size + (uint32_t)__popcll(km & below)] = make_uint2(cr[c], __float_as_uint(l[c]));
                size += n;
            }
            __syncthreads();
        }
        // the candidates that survive the final bound
        uint32_t out = 0;
        if (!overflow) {
            for (uint32_t base = 0; base < nc; base += 64u) {
                const bool in = base + lane < nc;
                const uint2 c = in ? s_cand[base + lane] : make_uint2(0u, 0u);
                const bool k = in && __uint_as_float(c.y) <= U2;
                const uint64_t km = __ballot(k);
                const uint32_t at = out + (uint32_t)__popcll(km & below);
                if (k && at < (uint32_t)NEAR_K) { cand[(size_t)at * numReps + r] = c.x; candLo[(size_t)at * numReps + r] = __uint_as_float(c.y); }
                out += (uint32_t)__popcll(km);
            }
            if (out > (uint32_t)NEAR_K) {
                // More live candidates than the list holds (a point on a tube's axis: hundreds of triangles within the fp32 slack of the
                // minimum).  Left as an overflow, the query went to the exact traversal on ONE lane (1.6 ms each; twenty of them were 4 of a
                // torus-knot build's 20 ms).  Here the wave evaluates the fp64 distance of every live candidate (the traversal stack's LDS
                // is free by now), takes the minimum and hands k_near_resolve the candidates within its tie threshold only — the set it
                // would have reduced the full list to.
                double* s_d2 = reinterpret_cast<double*>(s_stack);
                const D3 pd = D3{(double)p.x, (double)p.y, (double)p.z};
                double dmin2 = BVH_NO_BOUND;
                __syncthreads();
                for (uint32_t base = 0; base < nc; base += 64u) {
                    const bool in = base + lane < nc;
                    const uint2 c = in ? s_cand[base + lane] : make_uint2(0u, 0u);
                    double d2 = BVH_NO_BOUND;
                    if (in && __uint_as_float(c.y) <= U2) d2 = triangleSq(b, c.x, pd);
                    if (in) s_d2[base + lane] = d2;
                    dmin2 = d2 < dmin2 ? d2 : dmin2;
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) { const double other = __shfl_xor(dmin2, o); dmin2 = other < dmin2 ? other : dmin2; }
                __syncthreads();
                out = 0;
                if (dmin2 >= 1e-200 && dmin2 < BVH_NO_BOUND) {
                    const double thr = dmin2 * (1.0 + 4e-12);
                    for (uint32_t base = 0; base < nc; base += 64u) {
                        const bool in = base + lane < nc;
                        const uint2 c = in ? s_cand[base + lane] : make_uint2(0u, 0u);
                        const bool k = in && s_d2[base + lane] <= thr;
                        const uint64_t km = __ballot(k);
                        const uint32_t at = out + (uint32_t)__popcll(km & below);
                        if (k && at < (uint32_t)NEAR_K) { cand[(size_t)at * numReps + r] = c.x; candLo[(size_t)at * numReps + r] = __uint_as_float(c.y); }
                        out += (uint32_t)__popcll(km);
                    }
                    if (out > (uint32_t)NEAR_K || out == 0u) { overflow = true; if (lane == 0) atomicAdd(why + 3, 1u); }
                } else { overflow = true; if (lane == 0) atomicAdd(why + 4, 1u); }          // (a zero distance is the exact traversal's in any case)
            }
        }
        if (lane == 0) { candCount[r] = (uint8_t)(overflow ? NEAR_OVERFLOW : out); candU2[r] = U2; atomicAdd(why, 1u); }
    }
}

// ---- phase 2 ---------------------------------------------------------------------------------------------------------
// node over the sorted range [b, e): left child = inner node + 1 over [b, mid), right child = inner node + (mid - b) over [mid, e)
struct SimFrame { uint32_t node, b, e, info; };      // info = lo | hi << 8 | side << 16 (the child of `node` to enter: tested at pop time)

template <int BLOCK, int FS>
SDF_DEV uint32_t resolveTies(const BvhDev& bvh, D3 p, double dmin2, int n2, uint32_t* __restrict__ ids, uint32_t* __restrict__ rk, uint32_t* __restrict__ frames, double* __restrict__ d2s) {
    // sort the tied candidates by rank (position in the tree's leaf order): subsets of a subtree are then contiguous.  d2s (optional):
    // their fp64 squared distances as the caller evaluated them, carried along so that a leaf costs no third evaluation
    for (int i = 0; i < n2; i++) rk[i * BLOCK] = bvh.triRank[ids[i * BLOCK]];
    for (int i = 1; i < n2; i++) {
        const uint32_t kr = rk[i * BLOCK], ki = ids[i * BLOCK];
        const double kd = d2s ? d2s[i * BLOCK] : 0.0;
        int j = i - 1;
        while (j >= 0 && rk[j * BLOCK] > kr) { rk[(j + 1) * BLOCK] = rk[j * BLOCK]; ids[(j + 1) * BLOCK] = ids[j * BLOCK]; if (d2s) d2s[(j + 1) * BLOCK] = d2s[j * BLOCK]; j--; }
        rk[(j + 1) * BLOCK] = kr; ids[(j + 1) * BLOCK] = ki; if (d2s) d2s[(j + 1) * BLOCK] = kd;
    }
    double best = sqrt(dmin2) * (1.0 + 1e-9);
    int bestTri = -1;
    bool adopted = false;
    int fsp = 0;
    uint32_t node = 0, b = 0, e = bvh.numTriangles; int lo = 0, hi = n2;
    uint32_t rmin = rk[0], rmax = rk[(n2 - 1) * BLOCK];      // the ranks of candidates lo and hi - 1: most levels of the way down need no more (every candidate on one side)
    bool pending = true;           // a subproblem (node, [b,e), [lo,hi)) is loaded
    for (;;) {
        if (!pending) {
            if (fsp == 0) break;
            fsp--;
            const uint32_t pn = frames[(4 * fsp) * FS], pb = frames[(4 * fsp + 1) * FS], pe = frames[(4 * fsp + 2) * FS], info = frames[(4 * fsp + 3) * FS];
            const int side = (int)((info >> 16) & 1u);
            // the deferred (farther) child: its test sees the distance found in the nearer subtree, as in the reference
            if (adopted && !sphereCloser(sphereTerms(bvh.sph + 4 * (size_t)pn, side, p), best)) continue;
            const uint32_t mid = (pb + pe) >> 1;
            node = side ? pn + (mid - pb) : pn + 1u; b = side ? mid : pb; e = side ? pe : mid;
            lo = (int)(info & 0xFFu); hi = (int)((info >> 8) & 0xFFu);
            rmin = rk[lo * BLOCK]; rmax = rk[(hi - 1) * BLOCK];
            pending = true;
        }
        if (e - b == 1u) {                                           // a leaf: exactly one candidate left, this triangle
            const uint32_t t = ids[lo * BLOCK];
            const double d2 = d2s ? d2s[lo * BLOCK] : triangleSq(bvh, t, p);
            if (d2 < best * best) { best = sqrt(d2); bestTri = (int)t; adopted = true; }
            pending = false;
            continue;
        }
        const uint32_t mid = (b + e) >> 1;
        int s = lo;
        if (rmax < mid) s = hi;
        else if (rmin < mid) while (s < hi && rk[s * BLOCK] < mid) s++;
        if (s == lo || s == hi) {                                    // all candidates on one side: the other subtree cannot touch a tie
            const int side = (s == lo) ? 1 : 0;
            if (adopted && !sphereCloser(sphereTerms(bvh.sph + 4 * (size_t)node, side, p), best)) { pending = false; continue; }
            const uint32_t nn = side ? node + (mid - b) : node + 1u;
            if (side) b = mid; else e = mid;
            node = nn;
            continue;
        }
        // candidates on both sides: the reference's order decides who is visited first
        const double2* nd = bvh.sph + 4 * (size_t)node;
        const double dL = sphereDistExact(sphereTerms(nd, 0, p)), dR = sphereDistExact(sphereTerms(nd, 1, p));
        const bool leftFirst = dL < dR;
        const int second = leftFirst ? 1 : 0;
        frames[(4 * fsp) * FS] = node; frames[(4 * fsp + 1) * FS] = b; frames[(4 * fsp + 2) * FS] = e;
        frames[(4 * fsp + 3) * FS] = (second ? ((uint32_t)s | ((uint32_t)hi << 8)) : ((uint32_t)lo | ((uint32_t)s << 8))) | ((uint32_t)second << 16);
        fsp++;
        if (adopted && !((leftFirst ? dL : dR) < best)) { pending = false; continue; }
        const uint32_t nn = leftFirst ? node + 1u : node + (mid - b);
        if (leftFirst) { e = mid; hi = s; rmax = rk[(s - 1) * BLOCK]; } else { b = mid; lo = s; rmin = rk[s * BLOCK]; }
        node = nn;
    }
    return bestTri < 0 ? NEAR_UNRESOLVED : (uint32_t)bestTri;
}

// one query through phase 2; s_ids / s_rk / s_d2: TIES x BLOCK entries of LDS each
template <int BLOCK, int TIES>
SDF_DEV uint32_t resolveOne(const BvhDev& b, const float* __restrict__ pos, uint32_t numReps, const uint32_t* __restrict__ cand, const float* __restrict__ candLo,
                            const uint8_t* __restrict__ candCount, const float* __restrict__ candU2, uint32_t r, uint32_t* s_ids, uint32_t* s_rk, double* s_d2) {
    uint32_t frames[4 * (TIES - 1)];          // (the replay's frames - a few pushes and pops per query - live in private memory: as 16 KB of LDS per
                                              // workgroup they were what capped the kernel at two waves per SIMD, and its time is dependent fetches)
    const F3 pf = F3{pos[3 * (size_t)r], pos[3 * (size_t)r + 1], pos[3 * (size_t)r + 2]};
    const D3 p = D3{(double)pf.x, (double)pf.y, (double)pf.z};
    const uint32_t nc = candCount[r];
    uint32_t res = NEAR_UNRESOLVED;
    if (nc != NEAR_OVERFLOW && nc > 0u) {
        // Only LIVE candidates — lower bound not above the search's final upper bound u2 — can be the minimum or tied with it (the
        // minimum's own bounds enclose its distance, and a tie lies within 4e-12 of it): the others were recorded while the bound was
        // still loose (6 candidates per query on average, 2-3 of them live) and cost no fp64 evaluation.  One live candidate with a
        // positive lower bound IS the answer, unevaluated.
        const float u2 = candU2[r];
        double dmin2 = BVH_NO_BOUND; uint32_t argmin = NEAR_UNRESOLVED, live = 0, liveId = 0; float liveLo = 0.f;
        for (uint32_t i = 0; i < nc; i++)
            if (candLo[(size_t)i * numReps + r] <= u2) { live++; liveId = cand[(size_t)i * numReps + r]; liveLo = candLo[(size_t)i * numReps + r]; }
        if (live == 1u && liveLo > 0.f) res = liveId;
        else if (live >= 1u) {
            // every live candidate is evaluated ONCE: the values stay in LDS (TIES slots: more live candidates than that are
            // rare and re-evaluated below) for the tie test and for the leaves of the replay
            uint32_t* ids = s_ids + threadIdx.x; double* d2s = s_d2 + threadIdx.x;
            const bool cached = live <= (uint32_t)TIES;
            uint32_t k = 0;
            for (uint32_t i = 0; i < nc; i++) {
                if (!(candLo[(size_t)i * numReps + r] <= u2)) continue;
                const uint32_t id = cand[(size_t)i * numReps + r];
                const double d2 = triangleSq(b, id, p);
                if (cached) { ids[k * BLOCK] = id; d2s[k * BLOCK] = d2; k++; }
                if (d2 < dmin2) { dmin2 = d2; argmin = id; }
            }
            if (dmin2 >= 1e-200) {
                const double thr = dmin2 * (1.0 + 4e-12);
                int n2 = 0;
                if (cached) {
                    // (a tie's lower bound cannot exceed u2, the upper bound of the minimum: every tie is among the live candidates)
                    for (uint32_t j = 0; j < k; j++) {
                        const double d2 = d2s[j * BLOCK];
                        if (d2 <= thr) { ids[n2 * BLOCK] = ids[j * BLOCK]; d2s[n2 * BLOCK] = d2; n2++; }
                    }
                } else {
                    for (uint32_t i = 0; i < nc; i++) {
                        const uint32_t id = cand[(size_t)i * numReps + r];
                        if ((double)candLo[(size_t)i * numReps + r] <= thr && triangleSq(b, id, p) <= thr) { if (n2 < TIES) ids[n2 * BLOCK] = id; n2++; }
                    }
                }
                if (n2 == 1) res = cached ? ids[0] : argmin;
                else if (n2 >= 2 && n2 <= TIES) res = resolveTies<BLOCK, 1>(b, p, dmin2, n2, ids, s_rk + threadIdx.x, frames, cached ? d2s : nullptr);
            }
        }
    }
    return res;
}

// (the body of resolveOne<BLOCK, NEAR_MAX_TIES>, spelled out: as a call it compiled to 96 registers with the frames in registers — selects per
// access — instead of 81 with the frames in scratch, and took 1.86 instead of 1.42 ms per C2 build)
template <int BLOCK>
__global__ void __launch_bounds__(BLOCK, RESOLVE_WAVES) k_near_resolve(BvhDev b, const float* __restrict__ pos, uint32_t numReps, const uint32_t* __restrict__ cand, const float* __restrict__ candLo,
                                                        const uint8_t* __restrict__ candCount, const float* __restrict__ candU2, uint32_t* __restrict__ out, uint32_t* __restrict__ fbList,
                                                        uint32_t* __restrict__ fbCount, uint32_t rank, uint32_t world) {
    // (the replay's frames — a few pushes and pops per query — live in private memory: as 16 KB of LDS per workgroup they were what
    // capped the kernel at two waves per SIMD, and its time is dependent fetches, 1.7 resident waves per SIMD measured)
    __shared__ uint32_t s_ids[NEAR_MAX_TIES * BLOCK], s_rk[NEAR_MAX_TIES * BLOCK];
    uint32_t frames[4 * (NEAR_MAX_TIES - 1)];
    __shared__ double s_d2[NEAR_MAX_TIES * BLOCK];
    const uint64_t r64 = ((uint64_t)blockIdx.x * world + rank) * BLOCK + threadIdx.x;
    if (r64 >= numReps) return;
    const uint32_t r = (uint32_t)r64;
    const F3 pf = F3{pos[3 * (size_t)r], pos[3 * (size_t)r + 1], pos[3 * (size_t)r + 2]};
    const D3 p = D3{(double)pf.x, (double)pf.y, (double)pf.z};
    const uint32_t nc = candCount[r];
    uint32_t res = NEAR_UNRESOLVED;
    if (nc != NEAR_OVERFLOW && nc > 0u) {
        // Only LIVE candidates — lower bound not above the search's final upper bound u2 — can be the minimum or tied with it (the
        // minimum's own bounds enclose its distance, and a tie lies within 4e-12 of it): the others were recorded while the bound was
        // still loose (6 candidates per query on average, 2-3 of them live) and cost no fp64 evaluation.  One live candidate with a
        // positive lower bound IS the answer, unevaluated.
        const float u2 = candU2[r];
        double dmin2 = BVH_NO_BOUND; uint32_t argmin = NEAR_UNRESOLVED, live = 0, liveId = 0; float liveLo = 0.f;
        for (uint32_t i = 0; i < nc; i++)
            if (candLo[(size_t)i * numReps + r] <= u2) { live++; liveId = cand[(size_t)i * numReps + r]; liveLo = candLo[(size_t)i * numReps + r]; }
        if (live == 1u && liveLo > 0.f) res = liveId;
        else if (live >= 1u) {
            // every live candidate is evaluated ONCE: the values stay in LDS (NEAR_MAX_TIES slots: more live candidates than that are
            // rare and re-evaluated below) for the tie test and for the leaves of the replay
            uint32_t* ids = s_ids + threadIdx.x; double* d2s = s_d2 + threadIdx.x;
            const bool cached = live <= (uint32_t)NEAR_MAX_TIES;
            uint32_t k = 0;
            for (uint32_t i = 0; i < nc; i++) {
                if (!(candLo[(size_t)i * numReps + r] <= u2)) continue;
                const uint32_t id = cand[(size_t)i * numReps + r];
                const double d2 = triangleSq(b, id, p);
                if (cached) { ids[k * BLOCK] = id; d2s[k * BLOCK] = d2; k++; }
                if (d2 < dmin2) { dmin2 = d2; argmin = id; }
            }
            if (dmin2 >= 1e-200) {
                const double thr = dmin2 * (1.0 + 4e-12);
                int n2 = 0;
                if (cached) {
                    // (a tie's lower bound cannot exceed u2, the upper bound of the minimum: every tie is among the live candidates)
                    for (uint32_t j = 0; j < k; j++) {
                        const double d2 = d2s[j * BLOCK];
                        if (d2 <= thr) { ids[n2 * BLOCK] = ids[j * BLOCK]; d2s[n2 * BLOCK] = d2; n2++; }
                    }
                } else {
                    for (uint32_t i = 0; i < nc; i++) {
                        const uint32_t id = cand[(size_t)i * numReps + r];
                        if ((double)candLo[(size_t)i * numReps + r] <= thr && triangleSq(b, id, p) <= thr) { if (n2 < NEAR_MAX_TIES) ids[n2 * BLOCK] = id; n2++; }
                    }
                }
                if (n2 == 1) res = cached ? ids[0] : argmin;
                else if (n2 >= 2 && n2 <= NEAR_MAX_TIES) res = resolveTies<BLOCK, 1>(b, p, dmin2, n2, ids, s_rk + threadIdx.x, frames, cached ? d2s : nullptr);
            }
        }
    }
    out[r] = res;
    if (res == NEAR_UNRESOLVED) fbList[atomicAdd(fbCount, 1u)] = r;
}

// The queries phase 2 left over, once more with room for every candidate the list can hold (NEAR_K ties: a vertex of valence 9 - 16
// under a sample point, the rings of a tube around its axis): few queries, so the 32 KB of LDS per workgroup this takes do not matter.
// An entry it decides is struck from the list (0xFFFFFFFF: k_near_fallback skips it) and counted in *late.
template <int BLOCK>
__global__ void __launch_bounds__(BLOCK) k_near_resolve_many(BvhDev b, const float* __restrict__ pos, uint32_t numReps, const uint32_t* __restrict__ cand, const float* __restrict__ candLo,
                                                             const uint8_t* __restrict__ candCount, const float* __restrict__ candU2, uint32_t* __restrict__ out, uint32_t* __restrict__ fbList,
                                                             const uint32_t* __restrict__ fbCount, uint32_t* __restrict__ late) {
    __shared__ uint32_t s_ids[NEAR_K * BLOCK], s_rk[NEAR_K * BLOCK];
    __shared__ double s_d2[NEAR_K * BLOCK];
    const uint32_t count = *fbCount;
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < count; i += gridDim.x * BLOCK) {
        const uint32_t r = fbList[i];
        const uint32_t res = resolveOne<BLOCK, NEAR_K>(b, pos, numReps, cand, candLo, candCount, candU2, r, s_ids, s_rk, s_d2);
        if (res != NEAR_UNRESOLVED) { out[r] = res; fbList[i] = 0xFFFFFFFFu; atomicAdd(late, 1u); }
    }
}

// ---- phase 3 ---------------------------------------------------------------------------------------------------------
template <int BLOCK>
__global__ void __launch_bounds__(BLOCK) k_near_fallback(BvhDev b, const float* __restrict__ pos, const uint32_t* __restrict__ fbList, uint32_t* __restrict__ fbCount,
                                                         uint32_t first, uint32_t* __restrict__ out, const float* __restrict__ candU2) {
    extern __shared__ uint32_t s_fb_stack[];
    const uint32_t count = *fbCount;
    if (blockIdx.x == 0 && threadIdx.x == 0 && first == 0u) atomicAdd(fbCount + 1, count);       // running total of a build (fbCount[0] restarts with every batch)
    // consecutive entries go to DIFFERENT workgroups (entry i = thread i / gridDim of block i % gridDim): the handful of points that get
    // here (a tube's axis: 20 per torus-knot build) each walk thousands of nodes in fp64, and as neighbouring lanes of one wave their
    // divergent walks ran one after the other (2.9 ms per batch); alone in a wave each takes its own time only (1.6 ms)
    for (uint32_t i = first + threadIdx.x * gridDim.x + blockIdx.x; i < count; i += gridDim.x * BLOCK) {
        const uint32_t r = fbList[i];
        if (r == 0xFFFFFFFFu) continue;          // decided by k_near_resolve_many after all
        // The traversal starts from the candidate search's bound (fp32 upper bound of the squared distance incl. its error margin, i.e.
        // strictly above the minimum) instead of from infinity: the same answer (dev_bvh.h) without the visits that precede the first
        // good triangle — the points that end up here sit among thousands of almost equidistant triangles (a tube's axis), and unbounded
        // a handful of them took 3 ms per batch of the torus knot's build.
        const F3 pf = F3{pos[3 * (size_t)r], pos[3 * (size_t)r + 1], pos[3 * (size_t)r + 2]};
        const float u2 = candU2 ? candU2[r] : 3.0e38f;
        uint32_t t = 0xFFFFFFFFu;
        if (u2 < 1.0e37f) t = bvhNearest<BLOCK>(b, pf, s_fb_stack + threadIdx.x, nullptr, sqrt((double)u2) * (1.0 + 1e-6));
        if (t == 0xFFFFFFFFu) t = bvhNearest<BLOCK>(b, pf, s_fb_stack + threadIdx.x);
        out[r] = t;
    }
}

// ---- host driver -------------------------------------------------------------------------------------------------------
typedef sdfhip_near_scratch NearScratch;
// the scratch buffers are ordinary allocations, whatever allocation scope the caller is in
struct NearPlainAlloc { AllocState saved; NearPlainAlloc() { saved = tlsAlloc(); tlsAlloc().active = false; } ~NearPlainAlloc() { tlsAlloc() = saved; } };
static inline bool nearestExactOnly() { static const bool v = getenv("SDFHIP_NEAREST") && !strcmp(getenv("SDFHIP_NEAREST"), "exact"); return v; }

// Nearest triangle of pos[0..n) into out (this rank's blocks only when world > 1).  seedTri (dev probe, sdfhip_mesh_nearest_stats): the
// search of query r starts from the bound of triangle seedTri[r].
static int nearestTwoPhase(hipStream_t st, const BvhDev& bvh, const float* pos, uint32_t n, uint32_t* out, NearScratch& S, int stackDepth, uint32_t rank, uint32_t world,
                           uint32_t* perQuery = nullptr, const uint32_t* seedTri = nullptr) {
    if (n == 0) return SDFHIP_OK;
    const uint32_t blocks = gridFor(n, 128);
    const uint32_t mine = blocks > rank ? (blocks - rank + world - 1) / world : 0;
    if (!mine) return SDFHIP_OK;
    NearPlainAlloc plain;
    SDF_TRY(S.cand.reserve((size_t)NEAR_K * n)); SDF_TRY(S.candLo.reserve((size_t)NEAR_K * n)); SDF_TRY(S.candCount.reserve(n)); SDF_TRY(S.candU2.reserve(n)); SDF_TRY(S.fbList.reserve(n)); SDF_TRY(S.longList.reserve(n));
    if (!S.counterReady) { SDF_TRY(S.fbCount.reserve(32)); SDF_HIP_CHECK(hipMemsetAsync(S.fbCount.p, 0, 128, st)); S.counterReady = true; S.evUsed = 0; }
    SDF_HIP_CHECK(hipMemsetAsync(S.fbCount.p, 0, 4, st));          // the fallback list is per batch
    SDF_HIP_CHECK(hipMemsetAsync(S.fbCount.p + 2, 0, 72, st));     // and so are the work counters of the persistent waves ([2..9], second pass [12..19]) and the long list
    // device time of this batch's launches, summed up by nearTotals() when the build has finished
    const bool timed = S.evUsed < sdfhip_near_scratch::kMaxTimed;
    hipEvent_t* ev = S.ev + 3 * S.evUsed;
    if (timed) {
        while (S.evMade < 3 * (S.evUsed + 1)) { SDF_HIP_CHECK(hipEventCreate(&S.ev[S.evMade])); S.evMade++; }
        S.evUsed++;
    }
    // Leaders first (batches of NEAR_TWO_PASS_MIN queries and more): every eighth query of the Morton order, then the rest, each seeded
    // with the triangles the leaders on either side of it ended on.  A search seeded with its own answer needs 67 expansions + 11 triangle
    // tests on the C2 mesh's level-7 samples (tools/gpu_near_hist.py), seeded by the quad's previous query alone 77 + 21.
    const bool twoPass = !seedTri && n >= NEAR_TWO_PASS_MIN;
    if (twoPass) { SDF_TRY(S.best.reserve(n)); SDF_HIP_CHECK(hipMemsetAsync(S.best.p, 0xFF, sizeof(uint32_t) * (size_t)n, st)); }
    if (timed) SDF_HIP_CHECK(hipEventRecord(ev[0], st));
    uint32_t qgrid = 256u * NEAR_QBLOCKS_PER_CU;          // all resident (72 VGPRs: 7 waves per SIMD)
    const uint32_t needBlocks = (mine * 128u + 63u) / 64u;           // 64 queries per block of 256 lanes
    if (qgrid > needBlocks) qgrid = needBlocks;
    if (perQuery || seedTri)
        k_near_quads<256, true><<<xcdGrid(qgrid), 256, 0, st>>>(bvh, pos, n, S.cand.p, S.candLo.p, S.candCount.p, S.candU2.p, rank, world, S.fbCount.p + 2, S.longList.p, S.fbCount.p + 10, perQuery, seedTri, twoPass ? 1 : 0, S.best.p);
    else {
        // (the small-batch hand-over of round 6 was measured through an environment switch, profiles/r06_near_small_batches.txt; no gain, the switch is gone)
        const uint32_t steps = n >= NEAR_TWO_PASS_MIN ? NEAR_MAX_STEPS : NEAR_MAX_STEPS_SMALL;
#define SDF_NEAR_LAUNCH(M) k_near_quads<256, false, M><<<xcdGrid(qgrid), 256, 0, st>>>(bvh, pos, n, S.cand.p, S.candLo.p, S.candCount.p, S.candU2.p, rank, world, S.fbCount.p + 2, S.longList.p, S.fbCount.p + 10, nullptr, nullptr, twoPass ? 1 : 0, S.best.p)
        if (steps >= 1536u) SDF_NEAR_LAUNCH(1536u); else SDF_NEAR_LAUNCH(NEAR_MAX_STEPS_SMALL);
#undef SDF_NEAR_LAUNCH
    }
    if (timed) SDF_HIP_CHECK(hipEventRecord(ev[1], st));
    k_near_long<<<2048, 64, 0, st>>>(bvh, pos, n, S.longList.p, S.fbCount.p + 10, S.cand.p, S.candLo.p, S.candCount.p, S.candU2.p, S.fbCount.p + 24);
    k_near_resolve<128><<<mine, 128, 0, st>>>(bvh, pos, n, S.cand.p, S.candLo.p, S.candCount.p, S.candU2.p, out, S.fbList.p, S.fbCount.p, rank, world);
    k_near_resolve_many<128><<<64, 128, 0, st>>>(bvh, pos, n, S.cand.p, S.candLo.p, S.candCount.p, S.candU2.p, out, S.fbList.p, S.fbCount.p, S.fbCount.p + 30);
    k_near_fallback<128><<<256, 128, (size_t)stackDepth * 128 * sizeof(uint32_t), st>>>(bvh, pos, S.fbList.p, S.fbCount.p, 0u, out, S.candU2.p);
    SDF_HIP_CHECK(hipGetLastError());
    if (timed) SDF_HIP_CHECK(hipEventRecord(ev[2], st));
    return SDFHIP_OK;
}

// What the searches since the counters were reset (a build's start) amounted to; synchronises the stream.
struct NearTotals { uint64_t fallbacks = 0, expansions = 0, triangleTests = 0; double candidateSeconds = 0, searchSeconds = 0; };
static int nearTotals(hipStream_t st, NearScratch& S, NearTotals& out) {
    out = NearTotals();
    if (!S.counterReady) return SDFHIP_OK;
    uint32_t h[32];
    SDF_HIP_CHECK(hipMemcpyAsync(h, S.fbCount.p, sizeof(h), hipMemcpyDeviceToHost, st));
    SDF_HIP_CHECK(hipStreamSynchronize(st));
    out.fallbacks = h[1] - (h[1] >= h[30] ? h[30] : 0u);          // ([30]: struck from the lists by k_near_resolve_many)
    if (getenv("SDFHIP_TIMING")) {
        uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyAsync(w, S.fbCount.p + 24, 32, hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess)
            fprintf(stderr, "[sdfhip] nearest search: %u queries through k_near_long (of them left to the exact traversal: %u with more than %d candidates, %u stack overflows, %u with more than %d exact ties, %u at zero distance); %u left over by k_near_resolve, %u of them decided by k_near_resolve_many, the others by the exact traversal\n",
                    w[0], w[1], NEAR_LONG_CAND, w[2], w[3], NEAR_K, w[4], h[1], h[30]);
    }
    out.expansions = (uint64_t)h[20] | ((uint64_t)h[21] << 32); out.triangleTests = (uint64_t)h[22] | ((uint64_t)h[23] << 32);
    for (int i = 0; i < S.evUsed; i++) {
        float a = 0.f, b = 0.f;
        if (hipEventElapsedTime(&a, S.ev[3 * i], S.ev[3 * i + 1]) == hipSuccess && hipEventElapsedTime(&b, S.ev[3 * i], S.ev[3 * i + 2]) == hipSuccess) { out.candidateSeconds += 1e-3 * a; out.searchSeconds += 1e-3 * b; }
        else (void)hipGetLastError();
    }
    return SDFHIP_OK;
}

}  // namespace
}  // namespace sdfhip
