// OctreeSdf construction, CONTINUITY algorithm (SdfExporter's and the Unity plugin's default).  PRODUCT code —
// independent of oracle/.
//
// Reference behaviour reproduced: OctreeSdf::initOctreeWithContinuityNoDelay<VHQueries<TriCubic>> (src/sdf/
// OctreeSdfBreadthFirstNoDelay.h:84-1224; neighbour helpers src/sdf/OctreeSdfBreadthFirst.h:35-89), lattice cache
// disabled ("canonical" mode).  The reference runs, per depth: Iter 1 (parallel over nodes), Iter 2 (serial over nodes),
// and a serial post-pass that re-subdivides coarse leaves next to finer neighbours.
//
// MI355X form — everything runs in HIP kernels; the host only sizes buffers from a handful of counts it reads back:
//   * exact samples (two-phase nearest search), 8-slot 64x64 fits, termination rule, the "can the coarse neighbour's polynomial
//     stand in for this sample" tests, Hermite interpolation of the replaced samples, the hand-down of the 27-point stencil to children;
//   * Iter 2 is NOT serial here: within one level the leaf flags it reads are final after Iter 1, so neighbour masks are
//     order independent and the only serial thing, the allocation order, is an exclusive scan in node order;
//   * the post-pass's control flow depends only on INTEGER state (leaf / mark bits, child indices, neighbour words), never on
//     float values: kpp_* replay it on the device (roots -> generations of decide / expand -> allocation scan -> words), emitting
//     float-free "ops" (subdivide with sample mask / finalise leaf at slot) grouped by local BFS generation; kc_pp_* then execute
//     every generation of all scheduled leaves at once.  (Rounds 1-3 replayed it on the host over a mirror of the node words.)
// The reference's quirks are kept where they shape the output: child paths are truncated to 8 bits (its node constructor
// takes a uint8_t), re-created leaves are registered under parentChildrenIndex + childId even at the start depth, mark
// bits are cleared at the end.  mValueRange (never initialised in the reference) starts at 0.
// Compile with -ffp-contract=off.
#include "octree_internal.h"
#include "dev_bvh.h"
#include "dev_tricubic.h"
#include "octree_sampler.h"
#include "dev_prims.h"
#include <cmath>
#include <cstring>
#include <algorithm>
#include <cstdlib>

namespace sdfhip {

constexpr uint32_t NONE32 = 0xFFFFFFFFu;
constexpr uint32_t B31 = 1u << 31, B30 = 1u << 30;

using CMesh = MeshDev;

// mask of mid-points (bit 18-i) on the face / edge in direction dir (axis bits) with side code sign
SDF_HD uint32_t neighbourMask(uint32_t dir, uint32_t sign) {
    const unsigned char grid[19] = {1, 3, 4, 5, 7, 9, 10, 11, 12, 13, 14, 15, 16, 17, 19, 21, 22, 23, 25};
    int axes[3], na = 0;
    for (int a = 0; a < 3; a++) if (dir & (1u << a)) axes[na++] = a;
    if (sign >= (1u << na)) return 0;
    uint32_t m = 0;
    for (int i = 0; i < 19; i++) {
        const int g = grid[i];
        const int rel[3] = {g % 3 - 1, (g / 3) % 3 - 1, g / 9 - 1};
        bool on = true;
        for (int k = 0; k < na; k++) on = on && rel[axes[k]] == (((sign >> k) & 1u) ? 1 : -1);
        if (na == 2) for (int a = 0; a < 3; a++) if (!(dir & (1u << a))) on = on && rel[a] == 0;
        if (on) m |= 1u << (18 - i);
    }
    return m;
}
struct MaskTable { uint32_t m[24]; };
static MaskTable makeMaskTable() { MaskTable t; for (uint32_t d = 1; d <= 6; d++) for (uint32_t s = 0; s < 4; s++) t.m[4 * (d - 1) + s] = neighbourMask(d, s); return t; }

// the 18 face/edge neighbours of a node with child id c: f(blockSelector, dir, sign); blockSelector 0..5 = outward
// neighbour word nIdx[k], 6 = the node's own parent block (siblings)
template <typename F> SDF_HD void forEach18(uint32_t c, F f) {
    const uint32_t nc = ~c;
    f(0, 1u, c & 1u); f(0, 3u, 2u ^ (c & 3u)); f(0, 5u, ((nc >> 1) & 2u) + (c & 1u));
    f(1, 2u, (c >> 1) & 1u); f(1, 3u, 1u ^ (c & 3u)); f(1, 6u, 2u ^ ((c >> 1) & 3u));
    f(2, 3u, c & 3u);
    f(3, 4u, (c >> 2) & 1u); f(3, 5u, ((c >> 1) & 2u) + (nc & 1u)); f(3, 6u, 1u ^ ((c >> 1) & 3u));
    f(4, 5u, ((c >> 1) & 2u) + (c & 1u));
    f(5, 6u, (c >> 1) & 3u);
    f(6, 1u, nc & 1u); f(6, 2u, (nc >> 1) & 1u); f(6, 4u, (nc >> 2) & 1u);
    f(6, 3u, nc & 3u); f(6, 5u, ((nc >> 1) & 2u) + (nc & 1u)); f(6, 6u, (nc >> 1) & 3u);
}
template <typename F> SDF_HD void forEach18Grid(F f) {
    f(-1, 0, 0, 1u, 0u); f(1, 0, 0, 1u, 1u); f(0, -1, 0, 2u, 0u); f(0, 1, 0, 2u, 1u);
    f(-1, -1, 0, 3u, 0u); f(1, -1, 0, 3u, 1u); f(-1, 1, 0, 3u, 2u); f(1, 1, 0, 3u, 3u);
    f(0, 0, -1, 4u, 0u); f(0, 0, 1, 4u, 1u);
    f(-1, 0, -1, 5u, 0u); f(1, 0, -1, 5u, 1u); f(-1, 0, 1, 5u, 2u); f(1, 0, 1, 5u, 3u);
    f(0, -1, -1, 6u, 0u); f(0, 1, -1, 6u, 1u); f(0, -1, 1, 6u, 2u); f(0, 1, 1, 6u, 3u);
}
SDF_HD void neighboursVector(uint32_t o, uint32_t childId, uint32_t pci, uint32_t depth, const uint32_t* pN, const uint8_t* pD, uint32_t* oN, uint8_t* oD) {
    for (uint32_t n = 1; n <= 6; n++) {
        const uint32_t k = (~(o ^ childId)) & n;
        oN[n - 1] = k != 0 ? pN[k - 1] + (n ^ childId) * (1u - (pN[k - 1] >> 31)) : pci + (n ^ childId);
        oD[n - 1] = k != 0 ? pD[k - 1] : (uint8_t)depth;
    }
}
SDF_HD void neighboursInGrid(uint32_t o, int gx, int gy, int gz, int G, uint32_t* oN) {
    for (uint32_t n = 1; n <= 6; n++) {
        const int x = gx + ((n & 1) ? ((o & 1) ? 1 : -1) : 0), y = gy + ((n & 2) ? ((o & 2) ? 1 : -1) : 0), z = gz + ((n & 4) ? ((o & 4) ? 1 : -1) : 0);
        oN[n - 1] = (x >= 0 && x < G && y >= 0 && y < G && z >= 0 && z < G) ? (uint32_t)(z * G * G + y * G + x) : B30;
    }
}

// ---- device-side float helpers ---------------------------------------------------------------------------------------
// interpolateVertexValues (InterpolationMethods.h:457-497)
template <typename CF>
SDF_DEV void vertexValuesExact(CF c, F3 f, float nodeSize, float* __restrict__ out8) {
    out8[0] = tricubicValueLiteral(c, f);       // (interpolateVertexValues: literal in both flavours)
    out8[1] = tricubicDerivExact<1, 0, 0>(c, f) / nodeSize;
    out8[2] = tricubicDerivExact<0, 1, 0>(c, f) / nodeSize;
    out8[3] = tricubicDerivExact<0, 0, 1>(c, f) / nodeSize;
    const float sq = nodeSize * nodeSize;
    out8[4] = tricubicDerivExact<1, 1, 0>(c, f) / sq;
    out8[5] = tricubicDerivExact<1, 0, 1>(c, f) / sq;
    out8[6] = tricubicDerivExact<0, 1, 1>(c, f) / sq;
    out8[7] = tricubicDerivExact<1, 1, 1>(c, f) / (sq * nodeSize);
}
SDF_DEV F3 midFrac(int i) { const F3 r = midRel(i); return F3{0.5f * r.x + 0.5f, 0.5f * r.y + 0.5f, 0.5f * r.z + 0.5f}; }

SDF_DEV void exactSample(const CMesh& m, F3 p, float* __restrict__ out8, uint32_t* __restrict__ stk) {
    const uint32_t t = bvhNearest<128>(m.bvh, p, stk);
    const uint32_t a = m.idx[3 * t], b = m.idx[3 * t + 1], c = m.idx[3 * t + 2];
    F3 g;
    const float d = signedDistPointTriangleGrad(p, m.td + (size_t)TD_FLOATS * t, F3{m.verts[3 * a], m.verts[3 * a + 1], m.verts[3 * a + 2]},
                                                F3{m.verts[3 * b], m.verts[3 * b + 1], m.verts[3 * b + 2]}, F3{m.verts[3 * c], m.verts[3 * c + 1], m.verts[3 * c + 2]}, g);
    out8[0] = d; out8[1] = g.x; out8[2] = g.y; out8[3] = g.z; out8[4] = 0.f; out8[5] = 0.f; out8[6] = 0.f; out8[7] = 0.f;
}

// ---- level state (structure of arrays) --------------------------------------------------------------------------------
struct CLevelDev {
    uint32_t n; float half;
    float* center; uint32_t* coord; uint8_t* path; uint32_t* pci; uint32_t* nIdx; uint8_t* nDepth; uint32_t* word;
    float* vv; float* coeff; float* mid; uint8_t* terminal;
    uint32_t* cand; uint32_t* allocSize; uint32_t* allocOff; uint32_t* inner; uint32_t* childSlot;
};

// Iter 1a: refresh the six outward neighbour words (OctreeSdfBreadthFirstNoDelay.h:295-330)
__global__ void kc_refresh(CLevelDev L, uint32_t cd, const uint32_t* __restrict__ oc) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= L.n) return;
    const uint32_t path = L.path[i];
    for (uint32_t nb = 1; nb <= 6; nb++) {
        uint32_t ix = L.nIdx[6 * (size_t)i + nb - 1];
        uint32_t nd = L.nDepth[6 * (size_t)i + nb - 1];
        if (((ix >> 30) & 1u) == 0) {
            if (oc[ix & ~B31] & LEAF_BIT) ix = B31 | ix;
            else {
                ix = oc[ix & ~B31] & INDEX_MASK; nd++;
                while (nd < cd) {
                    const uint32_t dd = cd - nd;
                    const uint32_t cid = (3 * dd < 32) ? ((path >> (3 * dd)) & 7u) : 0u;
                    ix += (nb ^ cid);
                    if (oc[ix & ~B31] & LEAF_BIT) { ix = B31 | ix; break; }
                    ix = oc[ix & ~B31] & INDEX_MASK; nd++;
                }
            }
        }
        L.nIdx[6 * (size_t)i + nb - 1] = ix; L.nDepth[6 * (size_t)i + nb - 1] = (uint8_t)nd;
    }
}

// Iter 1b: the 19 exact mid-point samples per node come from sampleMidPoints (octree_sampler.h)

// Iter 1c: fit (8 slots), termination rule, provisional node word
__global__ void __launch_bounds__(128) kc_fit_rule(CLevelDev L, uint32_t cd, uint32_t startDepth, uint32_t maxDepth, int rule, float sqThr, float param1, uint32_t* __restrict__ oc) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= L.n) return;
    bool terminal = false;
    if (cd >= startDepth) {
        float s[64], c[64];
#pragma unroll
        for (int k = 0; k < 64; k++) s[k] = L.vv[64 * (size_t)i + k];
        tricubicFit(s, 2.0f * L.half, c);
#pragma unroll
        for (int k = 0; k < 64; k++) L.coeff[64 * (size_t)i + k] = c[k];
        if (cd < maxDepth && rule != 0) {
            const float* md = L.mid + 152 * (size_t)i;
            terminal = ruleValue(rule, [&](int n) { return c[n]; }, [&](int mm) { return md[8 * mm]; }, param1) < sqThr;
        }
    }
    if (cd >= maxDepth) return;            // the deepest level has no Iter 1 (its nodes become leaves in Iter 2)
    L.terminal[i] = terminal ? 1 : 0;
    const uint32_t w = L.word[i];
    if (w != NONE32) oc[w] = INDEX_MASK | (terminal ? LEAF_BIT : 0u);
}

// Iter 2a: which shared samples can be taken from this node's own polynomial, which coarse neighbours must be refined.
// THIRTY-TWO lanes per node since round 5 (a lane per node before: 0.10 lanes per vector instruction — few nodes border a coarser leaf, and
// those then evaluated up to nineteen 64-term polynomials alone): lane k < 18 looks at neighbour k, the sample masks are OR-ed over the
// lanes, the node's 64 coefficients go through LDS, lane m < 19 evaluates mid-point m, lane k < 24 writes candidate slot k.
__global__ void __launch_bounds__(256) kc_iter2_masks(CLevelDev L, uint32_t cd, uint32_t startDepth, uint32_t maxDepth, int G, float sqThr, MaskTable NM,
                                                      const uint32_t* __restrict__ oc) {
    __shared__ float s_c[8][64];
    __shared__ uint32_t s_nb[8][24];
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t i = gid >> 5, sub = gid & 31u, slot = threadIdx.x >> 5;
    if (i >= L.n) return;                                            // (uniform per group of 32 lanes; nothing below synchronises more than a wave)
    uint32_t* cand = L.cand + 24 * (size_t)i;
    if (sub < 24u) { cand[sub] = NONE32; s_nb[slot][sub] = NONE32; }
    const bool inner = !(cd >= maxDepth) && !L.terminal[i];
    if (sub == 0u) { L.inner[i] = inner ? 1u : 0u; L.allocSize[i] = inner ? (cd >= startDepth ? 8u : 0u) : 64u; }
    if (!inner || cd < startDepth) return;
    auto orOverGroup = [&](uint32_t v) {
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) v |= (uint32_t)__shfl_xor((int)v, o, 32);
        return v;
    };
    // neighbour `sub` of the eighteen
    uint32_t myMask = 0;
    if (sub < 18u) {
        int idx = 0;
        if (cd > startDepth) {
            const uint32_t c = L.path[i] & 7u, pci = L.pci[i];
            const uint32_t* N = L.nIdx + 6 * (size_t)i;
            int mySel = 0; uint32_t myDir = 1u, mySign = 0u;
            forEach18(c, [&](int sel, uint32_t dir, uint32_t sign) { if (idx++ == (int)sub) { mySel = sel; myDir = dir; mySign = sign; } });
            const uint32_t nodeId = (mySel < 6) ? N[mySel] : pci;
            if ((nodeId >> 31) || (!(nodeId >> 30) && (oc[nodeId + (myDir ^ c)] & LEAF_BIT))) {
                const uint32_t k = 4u * (myDir - 1u) + mySign;
                s_nb[slot][k] = (nodeId >> 31) ? (nodeId & ~B31) : nodeId + (myDir ^ c);
                myMask = NM.m[k];
            }
        } else {
            const uint32_t co = L.coord[i];
            const int gx = (int)(co & 1023u), gy = (int)((co >> 10) & 1023u), gz = (int)(co >> 20);
            int mx = 0, my = 0, mz = 0; uint32_t myDir = 1u, mySign = 0u;
            forEach18Grid([&](int dx, int dy, int dz, uint32_t dir, uint32_t sign) { if (idx++ == (int)sub) { mx = dx; my = dy; mz = dz; myDir = dir; mySign = sign; } });
            const int x = gx + mx, y = gy + my, z = gz + mz;
            if (x >= 0 && x < G && y >= 0 && y < G && z >= 0 && z < G) {
                const uint32_t at = (uint32_t)(z * G * G + y * G + x);
                if (oc[at] & LEAF_BIT) { const uint32_t k = 4u * (myDir - 1u) + mySign; s_nb[slot][k] = at; myMask = NM.m[k]; }
            }
        }
    }
    const uint32_t samplesMask = orOverGroup(myMask);
    if (samplesMask == 0) return;
    s_c[slot][sub] = L.coeff[64 * (size_t)i + sub]; s_c[slot][sub + 32u] = L.coeff[64 * (size_t)i + sub + 32u];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    uint32_t mySub = 0;
    if (sub < 19u && (samplesMask & (1u << (18u - sub)))) {
        const float* cs = s_c[slot];
        auto cf = [&](int n) { return cs[n]; };
        float* md = L.mid + 152 * (size_t)i + 8 * sub;
        const F3 f = midFrac((int)sub);
        const float iv = tricubicValueExact(cf, f);
        const float e = md[0] - iv;
        if (e * e > sqThr) mySub = samplesMask & (1u << (18u - sub));
        else vertexValuesExact(cf, f, 2.0f * L.half, md);
    }
    const uint32_t subdivisionMask = orOverGroup(mySub);
    if (sub < 24u) { const uint32_t nbk = s_nb[slot][sub]; if ((subdivisionMask & NM.m[sub]) && !(nbk >> 30)) cand[sub] = nbk; }
}

// Iter 2b: node words, children blocks, leaf payloads
__global__ void __launch_bounds__(256) kc_iter2_write(CLevelDev L, uint32_t cd, uint32_t startDepth, uint32_t base, uint32_t* __restrict__ oc, uint32_t* __restrict__ valueRangeBits) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t i = gid >> 6, lane = gid & 63u;
    if (i >= L.n) return;
    const uint32_t w = L.word[i];
    const uint32_t at = base + L.allocOff[i];
    if (L.inner[i]) {
        if (cd < startDepth) return;
        if (lane == 0) oc[w] = at & INDEX_MASK;
        if (lane < 8) oc[at + lane] = ~(7u << 29);
    } else {
        if (lane == 0) oc[w] = (at & INDEX_MASK) | LEAF_BIT;
        oc[at + lane] = __float_as_uint(L.coeff[64 * (size_t)i + lane]);
        // max |corner value| of the leaves: folded over the eight corner lanes first and compared with the value already there (the
        // maximum only grows): 5.5 M atomics on ONE address were 3.8 ms of a C2 build's 31
        uint32_t vb = (lane < 8) ? __float_as_uint(fabsf(L.vv[64 * (size_t)i + 8 * lane])) : 0u;
        vb = max(vb, (uint32_t)__shfl_xor((int)vb, 1)); vb = max(vb, (uint32_t)__shfl_xor((int)vb, 2)); vb = max(vb, (uint32_t)__shfl_xor((int)vb, 4));
        if (lane == 0 && vb > *reinterpret_cast<volatile uint32_t*>(valueRangeBits)) atomicMax(valueRangeBits, vb);
    }
}

// children of the inner nodes of a level -> next level
__global__ void __launch_bounds__(256) kc_children(CLevelDev L, CLevelDev N, uint32_t cd, uint32_t startDepth, uint32_t base, int G) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t i = gid >> 6;
    if (i >= L.n || !L.inner[i]) return;
    const uint32_t c = (gid >> 3) & 7u, j = gid & 7u;
    const uint32_t child = L.childSlot[i] + c;
    const int src = kStencilDev.src[c][j];
    const float* s = (src >= 0) ? (L.mid + 152 * (size_t)i + 8 * src) : (L.vv + 64 * (size_t)i + 8 * (-src - 1));
    float* d = N.vv + 64 * (size_t)child + 8 * j;
#pragma unroll
    for (int k = 0; k < 8; k++) d[k] = s[k];
    if (j != 0) return;
    const float ns = 0.5f * L.half;
    N.center[3 * (size_t)child] = L.center[3 * (size_t)i] + ((c & 1u) ? ns : -ns);
    N.center[3 * (size_t)child + 1] = L.center[3 * (size_t)i + 1] + ((c & 2u) ? ns : -ns);
    N.center[3 * (size_t)child + 2] = L.center[3 * (size_t)i + 2] + ((c & 4u) ? ns : -ns);
    const uint32_t co = L.coord[i];
    const uint32_t x = 2u * (co & 1023u) + (c & 1u), y = 2u * ((co >> 10) & 1023u) + ((c >> 1) & 1u), z = 2u * (co >> 20) + (c >> 2);
    N.coord[child] = x | (y << 10) | (z << 20);
    N.path[child] = (uint8_t)((L.path[i] << 3) | c);
    const uint32_t childIndex = (cd >= startDepth) ? base + L.allocOff[i] : NONE32;
    N.pci[child] = childIndex;
    N.terminal[child] = 0;
    uint32_t oN[6]; uint8_t oD[6];
    if (cd == startDepth) {
        neighboursInGrid(c, (int)(co & 1023u), (int)((co >> 10) & 1023u), (int)(co >> 20), G, oN);
        for (int k = 0; k < 6; k++) oD[k] = (uint8_t)cd;
    } else {
        uint32_t pN[6]; uint8_t pD[6];
        for (int k = 0; k < 6; k++) { pN[k] = L.nIdx[6 * (size_t)i + k]; pD[k] = L.nDepth[6 * (size_t)i + k]; }
        neighboursVector(c, L.path[i] & 7u, L.pci[i], cd, pN, pD, oN, oD);
    }
    for (int k = 0; k < 6; k++) { N.nIdx[6 * (size_t)child + k] = oN[k]; N.nDepth[6 * (size_t)child + k] = oD[k]; }
    // word of the child: inside its parent's block, or its start-grid slot when the NEXT level is the start depth
    if (cd >= startDepth) N.word[child] = childIndex + c;
    else if (cd + 1 == startDepth) N.word[child] = z * (uint32_t)G * (uint32_t)G + y * (uint32_t)G + x;
    else N.word[child] = NONE32;
}

// the totals of three exclusive scans (last offset + last value each), side by side for one read-back
__global__ void kc_totals3(const uint32_t* __restrict__ s0, const uint32_t* __restrict__ v0, const uint32_t* __restrict__ s1, const uint32_t* __restrict__ v1,
                           const uint32_t* __restrict__ s2, const uint32_t* __restrict__ v2, uint32_t* __restrict__ out3) {
    if (threadIdx.x == 0) { out3[0] = *s0 + *v0; out3[1] = *s1 + *v1; out3[2] = *s2 + *v2; }
}
__global__ void kc_flag_cand(const uint32_t* __restrict__ cand, uint32_t n, uint32_t* __restrict__ flag) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[i] = (cand[i] != NONE32) ? 1u : 0u;
}
__global__ void kc_compact_cand(const uint32_t* __restrict__ cand, const uint32_t* __restrict__ flag, const uint32_t* __restrict__ scan, uint32_t n, uint32_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && flag[i]) out[scan[i]] = cand[i];
}
__global__ void kc_mul8(uint32_t n, uint32_t* __restrict__ v) { const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) v[i] *= 8u; }

// ---- post-pass ops (float part) ------------------------------------------------------------------------------------------
// A node of the post-pass lives either in a level (src = depth << 26 | slot is NOT used; see refs below) or in the pool.
struct PoolDev { float* center; float* half; float* vv; };
struct OpDev {
    uint32_t kind;          // 0 = subdivide, 1 = leaf
    uint32_t srcLevel;      // depth of the level holding the source node, or NONE32 for a pool node
    uint32_t srcSlot;
    uint32_t samplesMask;   // subdivide: bit (18-i) set -> mid-point i is interpolated, clear -> exact sample
    uint32_t recycle;       // subdivide: reuse the level node's Iter-1 coefficients and samples
    uint32_t childPool;     // subdivide: first of 8 pool slots for the children
    uint32_t coeffIndex;    // leaf: where the 64 coefficients go in the node array
    uint32_t scratch;       // index into the per-generation scratch (coeff 64 + mid 152 floats)
};
struct LevelPtrs { const float* center; const float* vv; const float* coeff; const float* mid; float half; };
struct LevelTable { LevelPtrs lv[12]; };

SDF_DEV void opSource(const OpDev& op, const LevelTable& LT, const PoolDev& P, const float*& vv, F3& center, float& half) {
    if (op.srcLevel != NONE32) {
        const LevelPtrs& l = LT.lv[op.srcLevel];
        vv = l.vv + 64 * (size_t)op.srcSlot; half = l.half;
        center = F3{l.center[3 * (size_t)op.srcSlot], l.center[3 * (size_t)op.srcSlot + 1], l.center[3 * (size_t)op.srcSlot + 2]};
    } else {
        vv = P.vv + 64 * (size_t)op.srcSlot; half = P.half[op.srcSlot];
        center = F3{P.center[3 * (size_t)op.srcSlot], P.center[3 * (size_t)op.srcSlot + 1], P.center[3 * (size_t)op.srcSlot + 2]};
    }
}
// fit for every op that needs one (all leaves; subdivisions that do not recycle)
__global__ void __launch_bounds__(128) kc_pp_fit(const OpDev* __restrict__ ops, uint32_t nOps, LevelTable LT, PoolDev P, float* __restrict__ scratch, uint32_t* __restrict__ oc) {
    const uint32_t o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= nOps) return;
    const OpDev op = ops[o];
    float* sc = scratch + 216 * (size_t)op.scratch;
    if (op.kind == 0 && op.recycle) {
        const float* c = LT.lv[op.srcLevel].coeff + 64 * (size_t)op.srcSlot;
        for (int k = 0; k < 64; k++) sc[k] = c[k];
        return;
    }
    const float* vv; F3 ce; float half;
    opSource(op, LT, P, vv, ce, half);
    float s[64], c[64];
#pragma unroll
    for (int k = 0; k < 64; k++) s[k] = vv[k];
    tricubicFit(s, 2.0f * half, c);
    if (op.kind == 1) {
#pragma unroll
        for (int k = 0; k < 64; k++) oc[op.coeffIndex + k] = __float_as_uint(c[k]);
    } else {
#pragma unroll
        for (int k = 0; k < 64; k++) sc[k] = c[k];
    }
}
// the 19 mid-points of every subdividing op
__global__ void __launch_bounds__(128) kc_pp_mid(CMesh m, const OpDev* __restrict__ ops, uint32_t nOps, LevelTable LT, PoolDev P, float* __restrict__ scratch, float thr, float sqThr) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= 19u * nOps) return;
    const uint32_t o = gid / 19u, mi = gid - 19u * o;
    const OpDev op = ops[o];
    if (op.kind != 0) return;
    float* sc = scratch + 216 * (size_t)op.scratch;
    float* md = sc + 64 + 8 * mi;
    const float* vv; F3 ce; float half;
    opSource(op, LT, P, vv, ce, half);
    auto cf = [&](int n) { return sc[n]; };
    const F3 f = midFrac((int)mi);
    const bool masked = (op.samplesMask & (1u << (18 - mi))) != 0;
    if (op.recycle) {
        const float* src = LT.lv[op.srcLevel].mid + 152 * (size_t)op.srcSlot + 8 * mi;
        for (int k = 0; k < 8; k++) md[k] = src[k];
    } else if (masked) vertexValuesExact(cf, f, 2.0f * half, md);
    // else: md already holds the exact sample (kc_pp_sample_all)
    if (!masked) {
        const float iv = tricubicValueExact(cf, f);
        const float e = md[0] - iv;
        if (e * e < sqThr) vertexValuesExact(cf, f, 2.0f * half, md);
    } else if (op.recycle) vertexValuesExact(cf, f, 2.0f * half, md);
}
// geometry of the children of every subdividing op (centres only: lets the exact samples of ALL generations of a post-pass be
// taken in one launch before the value passes run generation by generation)
__global__ void kc_pp_geom(const OpDev* __restrict__ ops, uint32_t nOps, LevelTable LT, PoolDev P, float* __restrict__ pcenter, float* __restrict__ phalf) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t o = gid >> 3, c = gid & 7u;
    if (o >= nOps) return;
    const OpDev op = ops[o];
    if (op.kind != 0) return;
    const float* vv; F3 ce; float half;
    opSource(op, LT, P, vv, ce, half);
    const uint32_t child = op.childPool + c;
    const float ns = 0.5f * half;
    pcenter[3 * (size_t)child] = ce.x + ((c & 1u) ? ns : -ns);
    pcenter[3 * (size_t)child + 1] = ce.y + ((c & 2u) ? ns : -ns);
    pcenter[3 * (size_t)child + 2] = ce.z + ((c & 4u) ? ns : -ns);
    phalf[child] = ns;
}
// the exact mid-point samples of every subdividing op of a post-pass (all generations): few, long traversals -> one launch
__global__ void __launch_bounds__(128) kc_pp_sample_all(CMesh m, const OpDev* __restrict__ ops, uint32_t nOps, LevelTable LT, PoolDev P, float* __restrict__ scratch) {
    extern __shared__ uint32_t s_stack[];
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= 19u * nOps) return;
    const uint32_t o = gid / 19u, mi = gid - 19u * o;
    const OpDev op = ops[o];
    if (op.kind != 0 || op.recycle || (op.samplesMask & (1u << (18 - mi)))) return;
    const float* vv; F3 ce; float half;
    opSource(op, LT, P, vv, ce, half);
    exactSample(m, ce + midRel((int)mi) * half, scratch + 216 * (size_t)op.scratch + 64 + 8 * mi, s_stack + threadIdx.x);
}
// The same samples through the two-phase search (dev_bvh_fast.h): list the positions, search, then evaluate.
__global__ void __launch_bounds__(128) kc_pp_list(const OpDev* __restrict__ ops, uint32_t nOps, LevelTable LT, PoolDev P, float* __restrict__ pos, uint32_t* __restrict__ slot,
                                                  uint32_t* __restrict__ count) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    bool active = gid < 19u * nOps;
    uint32_t o = 0, mi = 0; OpDev op{};
    if (active) { o = gid / 19u; mi = gid - 19u * o; op = ops[o]; active = !(op.kind != 0 || op.recycle || (op.samplesMask & (1u << (18 - mi)))); }
    const uint64_t mask = __ballot(active);
    if (mask == 0ull) return;
    const uint32_t lane = __lane_id();
    const int leader = __ffsll((unsigned long long)mask) - 1;
    uint32_t base = 0;
    if ((int)lane == leader) base = atomicAdd(count, (uint32_t)__popcll(mask));
    base = __shfl(base, leader);
    if (!active) return;
    const uint32_t at = base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
    const float* vv; F3 ce; float half;
    opSource(op, LT, P, vv, ce, half);
    const F3 p = ce + midRel((int)mi) * half;
    pos[3 * (size_t)at] = p.x; pos[3 * (size_t)at + 1] = p.y; pos[3 * (size_t)at + 2] = p.z;
    slot[at] = gid;
}
__global__ void __launch_bounds__(128) kc_pp_values(CMesh m, const OpDev* __restrict__ ops, const float* __restrict__ pos, const uint32_t* __restrict__ slot,
                                                    const uint32_t* __restrict__ triOf, uint32_t n, float* __restrict__ scratch) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t gid = slot[i], o = gid / 19u, mi = gid - 19u * o, t = triOf[i];
    const F3 p = F3{pos[3 * (size_t)i], pos[3 * (size_t)i + 1], pos[3 * (size_t)i + 2]};
    const uint32_t a = m.idx[3 * t], b = m.idx[3 * t + 1], c = m.idx[3 * t + 2];
    F3 g;
    const float d = signedDistPointTriangleGrad(p, m.td + (size_t)TD_FLOATS * t, F3{m.verts[3 * a], m.verts[3 * a + 1], m.verts[3 * a + 2]},
                                                F3{m.verts[3 * b], m.verts[3 * b + 1], m.verts[3 * b + 2]}, F3{m.verts[3 * c], m.verts[3 * c + 1], m.verts[3 * c + 2]}, g);
    float* out8 = scratch + 216 * (size_t)ops[o].scratch + 64 + 8 * mi;
    out8[0] = d; out8[1] = g.x; out8[2] = g.y; out8[3] = g.z; out8[4] = 0.f; out8[5] = 0.f; out8[6] = 0.f; out8[7] = 0.f;
}
// children of every subdividing op -> pool
__global__ void __launch_bounds__(256) kc_pp_children(const OpDev* __restrict__ ops, uint32_t nOps, LevelTable LT, PoolDev P, const float* __restrict__ scratch,
                                                      float* __restrict__ pcenter, float* __restrict__ phalf, float* __restrict__ pvv) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t o = gid >> 6;
    if (o >= nOps) return;
    const OpDev op = ops[o];
    if (op.kind != 0) return;
    const uint32_t c = (gid >> 3) & 7u, j = gid & 7u;
    const float* vv; F3 ce; float half;
    opSource(op, LT, P, vv, ce, half);
    const float* sc = scratch + 216 * (size_t)op.scratch;
    const int src = kStencilDev.src[c][j];
    const float* s = (src >= 0) ? (sc + 64 + 8 * src) : (vv + 8 * (-src - 1));
    const uint32_t child = op.childPool + c;
    float* d = pvv + 64 * (size_t)child + 8 * j;
#pragma unroll
    for (int k = 0; k < 8; k++) d[k] = s[k];
    if (j == 0) {
        const float ns = 0.5f * half;
        pcenter[3 * (size_t)child] = ce.x + ((c & 1u) ? ns : -ns);
        pcenter[3 * (size_t)child + 1] = ce.y + ((c & 2u) ? ns : -ns);
        pcenter[3 * (size_t)child + 2] = ce.z + ((c & 4u) ? ns : -ns);
        phalf[child] = ns;
    }
}

// ---- post-pass, integer part ON THE DEVICE (round 4) ------------------------------------------------------------------------
// The reference's post-pass (OctreeSdfBreadthFirstNoDelay.h:739-1181) is a serial loop: every scheduled leaf is subdivided by a local
// breadth-first search that reads and appends to mOctreeData.  Two facts make it a decide -> scan -> write scheme:
//  (1) WHAT a visited node does — split or stay a leaf — does not depend on the order in which the scheduled leaves are processed.  A
//      node splits iff one of its neighbour look-ups lands on an inner node WITHOUT the mark bit, i.e. on a node subdivided by the
//      regular iterations.  The post-pass itself only ever turns a leaf into a MARKED inner node and creates leaves / marked nodes
//      below it: a word that is "leaf or marked" stays so for the rest of the build, a regular inner word never changes, and a
//      neighbour walk that enters leaf-or-marked territory can only stay inside it and ends without a contribution whichever state
//      it finds there.  So the walks are evaluated against the array as it was before the pass, every root by itself; a pointer that
//      would lead into blocks the pass is about to create (the siblings of a node created by the pass) carries bit 30 — the
//      reference's own "nothing there" marker, skipped by every look-up — instead of an index that does not exist yet;
//  (2) WHERE things are allocated follows from the order alone: roots in list order (first occurrence of every still-leaf word),
//      inside a root its nodes generation by generation in queue order; a split takes 8 words, a leaf 64 — except the first leaf
//      of a root, which recycles the root's old coefficient block.  That is an exclusive scan over (root, generation, position).
// Leaves are found through their coefficient block (a leaf word holds its index; blocks are allocated in units of 8 words after the
// start grid, so (index - G^3) / 8 addresses a direct table): recOf[] = the level node or pool node owning the block.
struct PPNode {
    uint32_t path, pci, coord, depth;
    uint32_t nIdx[6];
    uint8_t nDepth[6]; uint8_t split, gen;
    uint32_t srcLevel, srcSlot;       // level + slot of a level node, or NONE32 + pool slot
    uint32_t root, parent;            // number of the root in this pass; index of the parent in the pass's node array (NONE32: a root)
    uint32_t word;                    // roots: their node word
    uint32_t samplesMask;
    uint32_t oldCoeff;                // the coefficient block the root owned
    uint32_t ignore;                  // a node finalised by an earlier post-pass (no Iter-1 data to recycle)
    uint32_t childPool;               // splits: first of the 8 pool slots of the children
};
struct PoolRec { uint32_t path, pci, coord, depth; uint32_t nIdx[6]; uint8_t nDepth[6]; uint8_t pad[2]; };
struct LevelInt { const uint8_t* path; const uint32_t* pci; const uint32_t* coord; const uint32_t* nIdx; const uint8_t* nDepth; };
struct LevelIntTable { LevelInt lv[12]; };
constexpr uint32_t PP_MAXGEN = 12;
constexpr uint32_t REC_POOL = 1u << 31;

SDF_DEV uint32_t coeffSlot(uint32_t leafWord, uint32_t G3) { return ((leafWord & INDEX_MASK) - G3) >> 3; }

// level leaves own the blocks Iter 2 gave them
__global__ void kc_register_leaves(CLevelDev L, uint32_t cd, uint32_t base, uint32_t G3, uint32_t* __restrict__ recOf) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= L.n || L.inner[i]) return;
    recOf[(base + L.allocOff[i] - G3) >> 3] = (cd << 27) | i;
}
__global__ void kpp_first(const uint32_t* __restrict__ clist, uint32_t n, const uint32_t* __restrict__ oc, uint32_t G3, uint32_t* __restrict__ firstOcc) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t w = oc[clist[i]];
    if (w & LEAF_BIT) atomicMin(&firstOcc[coeffSlot(w, G3)], i);
}
__global__ void kpp_rootflag(const uint32_t* __restrict__ clist, uint32_t n, const uint32_t* __restrict__ oc, uint32_t G3, const uint32_t* __restrict__ firstOcc, uint32_t* __restrict__ flag) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t w = oc[clist[i]];
    flag[i] = ((w & LEAF_BIT) && firstOcc[coeffSlot(w, G3)] == i) ? 1u : 0u;
}
__global__ void kpp_roots(const uint32_t* __restrict__ clist, uint32_t n, const uint32_t* __restrict__ oc, uint32_t G3, uint32_t* __restrict__ firstOcc, const uint32_t* __restrict__ flag,
                          const uint32_t* __restrict__ scan, const uint32_t* __restrict__ recOf, LevelIntTable LI, const PoolRec* __restrict__ pool, PPNode* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !flag[i]) return;
    const uint32_t W = clist[i], w = oc[W], slot = coeffSlot(w, G3);
    firstOcc[slot] = NONE32;                    // the table is all ones again when the pass ends
    const uint32_t ref = recOf[slot];
    PPNode nd{};
    if (ref & REC_POOL) {
        const PoolRec r = pool[ref & ~REC_POOL];
        nd.path = r.path; nd.pci = r.pci; nd.coord = r.coord; nd.depth = r.depth;
        for (int k = 0; k < 6; k++) { nd.nIdx[k] = r.nIdx[k]; nd.nDepth[k] = r.nDepth[k]; }
        nd.srcLevel = NONE32; nd.srcSlot = ref & ~REC_POOL; nd.ignore = 1u;
    } else {
        const uint32_t lvl = ref >> 27, s = ref & ((1u << 27) - 1u);
        const LevelInt& l = LI.lv[lvl];
        nd.path = l.path[s]; nd.pci = l.pci[s]; nd.coord = l.coord[s]; nd.depth = lvl;
        for (int k = 0; k < 6; k++) { nd.nIdx[k] = l.nIdx[6 * (size_t)s + k]; nd.nDepth[k] = l.nDepth[6 * (size_t)s + k]; }
        nd.srcLevel = lvl; nd.srcSlot = s; nd.ignore = 0u;
    }
    nd.root = scan[i]; nd.parent = NONE32; nd.word = W; nd.oldCoeff = w & INDEX_MASK; nd.gen = 0;
    out[scan[i]] = nd;
}
// one node of the local searches: refresh its six outward pointers, collect the mid-points next to regularly subdivided
// neighbours, decide (OctreeSdfBreadthFirstNoDelay.h:770-912; the planner's loop body above, reading the device array)
__global__ void __launch_bounds__(128) kpp_decide(PPNode* __restrict__ nodes, uint32_t begin, uint32_t n, uint32_t cd, uint32_t startDepth, int G, MaskTable NM,
                                                  const uint32_t* __restrict__ oc, uint32_t* __restrict__ splitFlag) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    PPNode& nd = nodes[begin + i];
    const bool first = nd.parent == NONE32;
    const uint32_t depthN = nd.depth, c = nd.path & 7u, path = nd.path;
    uint32_t nI[6]; uint32_t nD[6];
    for (int k = 0; k < 6; k++) { nI[k] = nd.nIdx[k]; nD[k] = nd.nDepth[k]; }
    auto leaf = [&](uint32_t at) { return (oc[at] & LEAF_BIT) != 0; };
    auto leafish = [&](uint32_t at) { return (oc[at] & (LEAF_BIT | MARK_BIT)) != 0; };
    uint32_t subdividedMask = 0;
    if (depthN > startDepth) {
        for (uint32_t nb = 1; nb <= 6; nb++) {
            const uint32_t sign = ((((nb & c) >> 2) & 1u) << ((nb & 1u) | ((nb & 2u) >> 1))) + ((((nb & c) >> 1) & 1u) << (nb & 1u)) + (nb & c & 1u);
            uint32_t ix = nI[nb - 1], dp = nD[nb - 1];
            if (((ix >> 30) & 1u) != 0) continue;
            if ((!first || (ix >> 31)) && leaf(ix & ~B31)) ix = B31 | ix;
            else {
                if (!first || (ix >> 31)) { ix = oc[ix & ~B31] & INDEX_MASK; dp++; }
                while (dp < depthN && dp < cd) {
                    const uint32_t dd = depthN - dp;
                    const uint32_t cid = (3 * dd < 32) ? ((path >> (3 * dd)) & 7u) : 0u;
                    ix += (nb ^ cid);
                    if (leaf(ix & ~B31)) { ix = B31 | ix; break; }
                    ix = oc[ix & ~B31] & INDEX_MASK; dp++;
                }
                if (cd >= depthN && !(ix >> 31)) subdividedMask |= leafish((ix & ~B31) + (nb ^ c)) ? 0u : NM.m[4 * (nb - 1) + sign];
            }
            nI[nb - 1] = ix; nD[nb - 1] = dp;
        }
    }
    if (cd >= depthN) {
        if (depthN > startDepth) {
            const uint32_t nc = ~c;
            auto upd = [&](uint32_t nid, uint32_t dir, uint32_t sign) {
                const bool lf = (nid >> 31) || (nid >> 30) || leafish(nid + (dir ^ c));
                subdividedMask |= lf ? 0u : NM.m[4 * (dir - 1) + sign];
            };
            // the six siblings: blocks created by the post-pass hold nothing but leaves and marked nodes
            const uint32_t pci = first ? nd.pci : B30;
            upd(pci, 1u, nc & 1u); upd(pci, 2u, (nc >> 1) & 1u); upd(pci, 4u, (nc >> 2) & 1u);
            upd(pci, 3u, nc & 3u); upd(pci, 5u, ((nc >> 1) & 2u) + (nc & 1u)); upd(pci, 6u, (nc >> 1) & 3u);
            upd(nI[0], 3u, 2u ^ (c & 3u)); upd(nI[0], 5u, ((nc >> 1) & 2u) + (c & 1u));
            upd(nI[1], 3u, 1u ^ (c & 3u)); upd(nI[1], 6u, 2u ^ ((c >> 1) & 3u));
            upd(nI[3], 5u, ((c >> 1) & 2u) + (nc & 1u)); upd(nI[3], 6u, 1u ^ ((c >> 1) & 3u));
        } else if (depthN == startDepth) {
            const int gx = (int)(nd.coord & 1023u), gy = (int)((nd.coord >> 10) & 1023u), gz = (int)(nd.coord >> 20);
            forEach18Grid([&](int dx, int dy, int dz, uint32_t dir, uint32_t sign) {
                const int x = gx + dx, y = gy + dy, z = gz + dz;
                if (x >= 0 && x < G && y >= 0 && y < G && z >= 0 && z < G) subdividedMask |= leafish((uint32_t)(z * G * G + y * G + x)) ? 0u : NM.m[4 * (dir - 1) + sign];
            });
        }
    }
    const bool split = cd >= depthN && subdividedMask != 0u;
    for (int k = 0; k < 6; k++) { nd.nIdx[k] = nI[k]; nd.nDepth[k] = (uint8_t)nD[k]; }
    nd.samplesMask = ~subdividedMask; nd.split = split ? 1 : 0;
    splitFlag[i] = split ? 1u : 0u;
}
// the eight children of every splitting node of a generation -> the next generation (queue order: parents in order, children 0..7)
__global__ void kpp_expand(PPNode* __restrict__ nodes, uint32_t begin, uint32_t n, const uint32_t* __restrict__ splitFlag, const uint32_t* __restrict__ splitScan,
                           uint32_t nextBegin, uint32_t poolBase, uint32_t startDepth, int G) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t i = gid >> 3, ch = gid & 7u;
    if (i >= n || !splitFlag[i]) return;
    const PPNode& p = nodes[begin + i];
    const uint32_t rank = splitScan[i], depthN = p.depth, c = p.path & 7u;
    const uint32_t gx = p.coord & 1023u, gy = (p.coord >> 10) & 1023u, gz = p.coord >> 20;
    PPNode k{};
    k.path = (uint8_t)((p.path << 3) | ch); k.pci = NONE32; k.depth = depthN + 1;
    k.coord = (2u * gx + (ch & 1u)) | ((2u * gy + ((ch >> 1) & 1u)) << 10) | ((2u * gz + (ch >> 2)) << 20);
    uint32_t oN[6]; uint8_t oD[6];
    if (depthN == startDepth) { neighboursInGrid(ch, (int)gx, (int)gy, (int)gz, G, oN); for (int q = 0; q < 6; q++) oD[q] = (uint8_t)depthN; }
    else {
        uint32_t pN[6]; uint8_t pD[6];
        for (int q = 0; q < 6; q++) { pN[q] = p.nIdx[q]; pD[q] = p.nDepth[q]; }
        // the parent's own block is real for a root; for a node the pass created it does not exist yet: "nothing there"
        neighboursVector(ch, c, (p.parent == NONE32) ? p.pci : B30, depthN, pN, pD, oN, oD);
    }
    for (int q = 0; q < 6; q++) { k.nIdx[q] = oN[q]; k.nDepth[q] = oD[q]; }
    k.srcLevel = NONE32; k.srcSlot = poolBase + 8u * rank + ch; k.root = p.root; k.parent = begin + i; k.oldCoeff = p.oldCoeff; k.gen = (uint8_t)(p.gen + 1);
    nodes[nextBegin + 8u * rank + ch] = k;
    if (ch == 0) nodes[begin + i].childPool = poolBase + 8u * rank;
}
__global__ void kpp_firstleaf(const PPNode* __restrict__ nodes, uint32_t N, uint32_t* __restrict__ firstLeaf) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < N && !nodes[j].split) atomicMin(&firstLeaf[nodes[j].root], j);
}
__global__ void kpp_sizes(const PPNode* __restrict__ nodes, uint32_t N, const uint32_t* __restrict__ firstLeaf, uint32_t* __restrict__ size) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < N) size[j] = nodes[j].split ? 8u : (firstLeaf[nodes[j].root] == j ? 0u : 64u);
}
// segments (root, generation) are contiguous in the generation-major node array
__global__ void kpp_bounds(const PPNode* __restrict__ nodes, uint32_t N, const uint32_t* __restrict__ size, const uint32_t* __restrict__ ex, uint32_t* __restrict__ segStart, uint32_t* __restrict__ segEnd) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= N) return;
    const uint32_t r = nodes[j].root, g = nodes[j].gen;
    if (j == 0 || nodes[j - 1].root != r || nodes[j - 1].gen != g) segStart[PP_MAXGEN * (size_t)r + g] = ex[j];
    if (j + 1 == N || nodes[j + 1].root != r || nodes[j + 1].gen != g) segEnd[PP_MAXGEN * (size_t)r + g] = ex[j] + size[j];
}
__global__ void kpp_roottot(uint32_t numRoots, const uint32_t* __restrict__ segStart, const uint32_t* __restrict__ segEnd, uint32_t* __restrict__ rootTot) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= numRoots) return;
    uint32_t t = 0;
    for (uint32_t g = 0; g < PP_MAXGEN; g++) t += segEnd[PP_MAXGEN * (size_t)r + g] - segStart[PP_MAXGEN * (size_t)r + g];
    rootTot[r] = t;
}
__global__ void kpp_offsets(const PPNode* __restrict__ nodes, uint32_t N, const uint32_t* __restrict__ ex, const uint32_t* __restrict__ segStart, const uint32_t* __restrict__ segEnd,
                            const uint32_t* __restrict__ rootBase, uint32_t* __restrict__ off) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= N) return;
    const uint32_t r = nodes[j].root, g = nodes[j].gen;
    uint32_t o = rootBase[r];
    for (uint32_t q = 0; q < g; q++) o += segEnd[PP_MAXGEN * (size_t)r + q] - segStart[PP_MAXGEN * (size_t)r + q];
    off[j] = o + ex[j] - segStart[PP_MAXGEN * (size_t)r + g];
}
// node words, ownership of the coefficient blocks, the pool records of the new leaves, and the float part's op list
__global__ void kpp_finalise(const PPNode* __restrict__ nodes, uint32_t N, const uint32_t* __restrict__ off, const uint32_t* __restrict__ firstLeaf, uint32_t base, uint32_t G3,
                             uint32_t* __restrict__ oc, uint32_t* __restrict__ recOf, PoolRec* __restrict__ pool, OpDev* __restrict__ ops) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= N) return;
    const PPNode nd = nodes[j];
    const bool isRoot = nd.parent == NONE32;
    const uint32_t at = base + off[j];
    const uint32_t pci = isRoot ? nd.pci : base + off[nd.parent];
    const uint32_t W = isRoot ? nd.word : pci + (nd.path & 7u);
    OpDev op{}; op.srcLevel = nd.srcLevel; op.srcSlot = nd.srcSlot; op.scratch = j;
    if (nd.split) {
        oc[W] = (at & INDEX_MASK) | MARK_BIT;
        op.kind = 0; op.samplesMask = nd.samplesMask; op.recycle = (isRoot && !nd.ignore) ? 1u : 0u; op.childPool = nd.childPool;
    } else {
        const uint32_t ci = (firstLeaf[nd.root] == j) ? nd.oldCoeff : at;
        oc[W] = (ci & INDEX_MASK) | LEAF_BIT;
        op.kind = 1; op.coeffIndex = ci;
        if (nd.srcLevel == NONE32) {           // a pool node: it stays reachable for later post-passes with its refreshed pointers
            PoolRec r{}; r.path = nd.path; r.pci = pci; r.coord = nd.coord; r.depth = nd.depth;
            for (int k = 0; k < 6; k++) { r.nIdx[k] = nd.nIdx[k]; r.nDepth[k] = nd.nDepth[k]; }
            pool[nd.srcSlot] = r;
            recOf[(ci - G3) >> 3] = REC_POOL | nd.srcSlot;
        }
    }
    ops[j] = op;
}

// final walk: leaf histogram + min border value (computeMinBorderValue, OctreeSdf.cpp:155-230).  Level-synchronous: one lane
// per node of the current frontier; inner nodes append their 8 children to the next frontier.  Mark bits left by the post-pass are
// cleared on the way (OctreeSdfBreadthFirstNoDelay.h:1191-1217: the reference's last step is such a walk).
__global__ void kc_walk_init(int G, uint32_t* __restrict__ at, uint32_t* __restrict__ co) {
    const uint32_t cell = blockIdx.x * blockDim.x + threadIdx.x;
    if (cell >= (uint32_t)(G * G * G)) return;
    at[cell] = cell; co[cell] = (cell % G) | (((cell / G) % G) << 10) | ((cell / (G * G)) << 20);
}
__global__ void kc_walk_level(uint32_t* __restrict__ oc, const uint32_t* __restrict__ at, const uint32_t* __restrict__ coIn, uint32_t n, uint32_t d,
                              uint32_t* __restrict__ nextAt, uint32_t* __restrict__ nextCo, uint32_t* __restrict__ nextCount,
                              unsigned long long* __restrict__ leavesPerDepth, uint32_t* __restrict__ minBorderKey) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = i < n;
    const uint32_t w = valid ? oc[at[i]] : 0u;
    if (w & MARK_BIT) oc[at[i]] = w & ~MARK_BIT;
    const uint32_t co = valid ? coIn[i] : 0u;
    const bool leaf = valid && (w & LEAF_BIT);
    const unsigned long long leafMask = __ballot(leaf);
    if (leafMask != 0ull && (threadIdx.x & 63u) == (uint32_t)(__ffsll((long long)leafMask) - 1)) atomicAdd(&leavesPerDepth[d], (unsigned long long)__popcll(leafMask));
    if (!valid) return;
    const uint32_t x = co & 1023u, y = (co >> 10) & 1023u, z = co >> 20;
    if (!leaf) {
        const uint32_t base = atomicAdd(nextCount, 8u);
        for (uint32_t c = 0; c < 8; c++) {
            nextAt[base + c] = (w & INDEX_MASK) + c;
            nextCo[base + c] = (2 * x + (c & 1u)) | ((2 * y + ((c >> 1) & 1u)) << 10) | ((2 * z + (c >> 2)) << 20);
        }
        return;
    }
    const uint32_t last = (1u << d) - 1u;
    if (!(x == 0 || y == 0 || z == 0 || x == last || y == last || z == last)) return;
    const uint32_t* cw = oc + (w & INDEX_MASK);
    auto cf = [&](int k) { return __uint_as_float(cw[k]); };
    float mn = INFINITY;
    for (uint32_t k = 0; k < 8; k++) {
        const uint32_t bx = k & 1u, by = (k >> 1) & 1u, bz = k >> 2;
        const bool onBorder = (x + bx == 0) || (y + by == 0) || (z + bz == 0) || (x + bx == last + 1) || (y + by == last + 1) || (z + bz == last + 1);
        if (!onBorder) continue;
        const float v = tricubicValueExact(cf, F3{(float)bx, (float)by, (float)bz});
        mn = gmin(mn, v);
    }
    if (mn < INFINITY) { const uint32_t b = __float_as_uint(mn); atomicMin(minBorderKey, (b & 0x80000000u) ? ~b : (b | 0x80000000u)); }
}

// ---- host side ---------------------------------------------------------------------------------------------------------
struct CLevelHost {
    bool sampled = false;      // the 19 exact mid-point samples were already enqueued (overlapped with the previous level's post-pass)
    uint32_t depth = 0, n = 0; float half = 0.f;
    DevBuf<float> center, vv, coeff, mid; DevBuf<uint32_t> coord, pci, nIdx, word, cand, allocSize, allocOff, inner, childSlot; DevBuf<uint8_t> path, nDepth, terminal;
    int alloc(uint32_t count) {
        n = count;
        SDF_TRY(center.reserve(3ull * n)); SDF_TRY(vv.reserve(64ull * n)); SDF_TRY(coeff.reserve(64ull * n)); SDF_TRY(mid.reserve(152ull * n));
        SDF_TRY(coord.reserve(n)); SDF_TRY(pci.reserve(n)); SDF_TRY(nIdx.reserve(6ull * n)); SDF_TRY(word.reserve(n)); SDF_TRY(cand.reserve(24ull * n));
        SDF_TRY(allocSize.reserve(n)); SDF_TRY(allocOff.reserve(n)); SDF_TRY(inner.reserve(n)); SDF_TRY(childSlot.reserve(n));
        SDF_TRY(path.reserve(n)); SDF_TRY(nDepth.reserve(6ull * n)); SDF_TRY(terminal.reserve(n));
        return SDFHIP_OK;
    }
    CLevelDev dev() { return CLevelDev{n, half, center.p, coord.p, path.p, pci.p, nIdx.p, nDepth.p, word.p, vv.p, coeff.p, mid.p, terminal.p, cand.p, allocSize.p, allocOff.p, inner.p, childSlot.p}; }
};

static uint32_t outwardSign(uint32_t n, uint32_t c) {
    return ((((n & c) >> 2) & 1u) << ((n & 1u) | ((n & 2u) >> 1))) + ((((n & c) >> 1) & 1u) << (n & 1u)) + (n & c & 1u);
}

static int scanExclusive(hipStream_t st, DevBuf<unsigned char>& tmp, size_t& tmpBytes, const uint32_t* in, uint32_t* out, uint32_t n) {
    if (n == 0) return SDFHIP_OK;
    size_t need = 0;
    SDF_HIP_CHECK(devExclusiveSum(nullptr, need, in, out, (size_t)n, st));
    if (need > tmpBytes) { SDF_TRY(tmp.reserve(need)); tmpBytes = need; }
    SDF_HIP_CHECK(devExclusiveSum(tmp.p, need, in, out, (size_t)n, st));
    return SDFHIP_OK;
}
static int lastPlus(hipStream_t st, const uint32_t* scan, const uint32_t* val, uint32_t n, uint32_t& total) {
    total = 0;
    if (n == 0) return SDFHIP_OK;
    return readBackWords(st, scan + (n - 1), val + (n - 1), 1, &total);
}

int continuityBuildImpl(sdfhip_ctx* ctx, sdfhip_mesh* mesh, const sdfhip_octree_params* P, sdfhip_octree** out) {
    SDF_REQUIRE(P->depth >= 1, "depth must be at least 1");
    if (P->depth > 10) { setError("depth %u is above this build's limit of 10: node coordinates are packed 10 bits per axis (the reference's own limit is its 30-bit word index, OctreeSdf.h:53-55, which a depth-11 tree of a real surface exceeds anyway)", (unsigned)P->depth); return SDFHIP_E_UNSUPPORTED; }
    SDF_REQUIRE(P->start_depth <= P->depth, "start_depth > depth");
    SDF_REQUIRE(P->rule >= SDFHIP_RULE_NONE && P->rule <= SDFHIP_RULE_BY_DISTANCE, "unknown termination rule");
    SDF_REQUIRE(P->fit_mode == SDFHIP_FIT_EXACT, "the CONTINUITY builder feeds coefficients back into the tree: only FIT_EXACT is provided");
    SDF_REQUIRE(P->cell_begin == 0 && (P->cell_end == 0 || P->cell_end == (1u << (3 * P->start_depth))), "the CONTINUITY builder is not sharded");
    SDF_TRY(sdfhip_mesh_ensure_bvh(mesh));
    SDF_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    AllocScope allocScope(st);       // device buffers of this call come from the stream-ordered pool
    const double tStart = nowSeconds();
    const uint32_t maxDepth = P->depth, startDepth = P->start_depth;
    const int G = 1 << startDepth; const uint32_t G3 = (uint32_t)(G * G * G);
    std::unique_ptr<sdfhip_octree> T(new sdfhip_octree());
    T->ctx = ctx; T->params = *P;
    const float sx = P->box_max[0] - P->box_min[0], sy = P->box_max[1] - P->box_min[1], sz = P->box_max[2] - P->box_min[2];
    SDF_REQUIRE(sx > 0 && sy > 0 && sz > 0 && std::isfinite(sx) && std::isfinite(sy) && std::isfinite(sz), "empty or non-finite box");
    const float maxSize = gmax(gmax(sx, sy), sz);
    const float cx = P->box_min[0] + 0.5f * sx, cy = P->box_min[1] + 0.5f * sy, cz = P->box_min[2] + 0.5f * sz;
    float bmin[3] = {cx - 0.5f * maxSize, cy - 0.5f * maxSize, cz - 0.5f * maxSize};
    float bmax[3] = {cx + 0.5f * maxSize, cy + 0.5f * maxSize, cz + 0.5f * maxSize};
    memcpy(T->info.box_min, bmin, 12); memcpy(T->info.box_max, bmax, 12);
    T->info.start_grid_size = G; T->info.max_depth = maxDepth; T->cellSize = maxSize / (float)G; T->info.start_grid_cell_size = T->cellSize;
    const uint32_t sod = startDepth < 1u ? startDepth : 1u;
    const float thr = P->rule_params[0], sqThr = thr * thr, param1 = P->rule_params[1];
    CMesh md{meshBvh(mesh), mesh->dVerts.p, mesh->dIdx.p, mesh->dTri.p};
    size_t stackBytes; { int depth = 1; while ((1ull << (depth - 1)) < mesh->numTriangles) depth++; stackBytes = (size_t)(depth + 2) * 128 * sizeof(uint32_t); }
    DevBuf<unsigned char> scanTmp; size_t scanTmpBytes = 0;
    DevBuf<uint32_t> stats; SDF_TRY(stats.reserve(2));
    { const uint32_t init[2] = {0u, 0xFFFFFFFFu}; SDF_HIP_CHECK(hipMemcpyAsync(stats.p, init, 8, hipMemcpyHostToDevice, st)); }

    // the node array grows on the device; the post-pass's integer part runs there too (kpp_*)
    DevBuf<uint32_t> oc; size_t ocCap = 0; uint32_t ocSize = G3;
    DevBuf<uint32_t> recOf, firstOcc; size_t recCap = 0;      // per coefficient block: its owner / the first candidate naming it (device post-pass)
    auto ensureOc = [&](size_t need) -> int {
        if (need <= ocCap) return SDFHIP_OK;
        size_t cap = ocCap ? ocCap : std::max<size_t>((size_t)1 << 22, std::min<size_t>(ctx->contCaps[0], (size_t)1 << 27));      // at most 512 MB up front, however large the last tree was      // (every growth is an allocation, copies and two waits: a repeated build starts where the last one ended)
        while (cap < need) cap *= 2;
        DevBuf<uint32_t> bigger; SDF_TRY(bigger.reserve(cap));
        if (oc.p) SDF_HIP_CHECK(hipMemcpyAsync(bigger.p, oc.p, 4ull * ocSize, hipMemcpyDeviceToDevice, st));
        {
            const size_t rc = (cap - G3) / 8 + 1;
            DevBuf<uint32_t> r2, f2; SDF_TRY(r2.reserve(rc)); SDF_TRY(f2.reserve(rc));
            if (recCap) SDF_HIP_CHECK(hipMemcpyAsync(r2.p, recOf.p, 4ull * recCap, hipMemcpyDeviceToDevice, st));
            SDF_HIP_CHECK(hipMemsetAsync(f2.p, 0xFF, 4ull * rc, st));
            SDF_HIP_CHECK(hipStreamSynchronize(st));
            recOf = std::move(r2); firstOcc = std::move(f2); recCap = rc;
        }
        SDF_HIP_CHECK(hipStreamSynchronize(st));
        oc = std::move(bigger); ocCap = cap;
        return SDFHIP_OK;
    };
    SDF_TRY(ensureOc(G3));
    SDF_HIP_CHECK(hipMemsetAsync(oc.p, 0, 4ull * G3, st));
    const MaskTable NM = makeMaskTable();
    // pool of post-pass nodes on the device
    DevBuf<float> pCenter, pHalf, pVv; DevBuf<PoolRec> pRec; size_t poolCap = 0; uint32_t poolCount = 0;
    auto ensurePool = [&](size_t need) -> int {
        if (need <= poolCap) return SDFHIP_OK;
        size_t cap = poolCap ? poolCap : std::max<size_t>(4096, std::min<size_t>(ctx->contCaps[1], (size_t)1 << 21));
        while (cap < need) cap *= 2;
        DevBuf<float> c2, h2, v2; DevBuf<PoolRec> r2; SDF_TRY(c2.reserve(3 * cap)); SDF_TRY(h2.reserve(cap)); SDF_TRY(v2.reserve(64 * cap));
        SDF_TRY(r2.reserve(cap));
        if (poolCap) {
            SDF_HIP_CHECK(hipMemcpyAsync(c2.p, pCenter.p, 12 * poolCap, hipMemcpyDeviceToDevice, st));
            SDF_HIP_CHECK(hipMemcpyAsync(h2.p, pHalf.p, 4 * poolCap, hipMemcpyDeviceToDevice, st));
            SDF_HIP_CHECK(hipMemcpyAsync(v2.p, pVv.p, 256 * poolCap, hipMemcpyDeviceToDevice, st));
            SDF_HIP_CHECK(hipMemcpyAsync(r2.p, pRec.p, sizeof(PoolRec) * poolCap, hipMemcpyDeviceToDevice, st));
            SDF_HIP_CHECK(hipStreamSynchronize(st));
        }
        pCenter = std::move(c2); pHalf = std::move(h2); pVv = std::move(v2); pRec = std::move(r2); poolCap = cap;
        return SDFHIP_OK;
    };
    DevBuf<PPNode> ppNodesBuf; size_t ppNodesCap = 0;
    auto ensureNodes = [&](size_t have, size_t need) -> int {
        if (need <= ppNodesCap) return SDFHIP_OK;
        size_t cap = ppNodesCap ? ppNodesCap : std::max<size_t>(4096, std::min<size_t>(ctx->contCaps[2], (size_t)1 << 22));
        while (cap < need) cap *= 2;
        DevBuf<PPNode> b2; SDF_TRY(b2.reserve(cap));
        if (have) { SDF_HIP_CHECK(hipMemcpyAsync(b2.p, ppNodesBuf.p, sizeof(PPNode) * have, hipMemcpyDeviceToDevice, st)); SDF_HIP_CHECK(hipStreamSynchronize(st)); }
        ppNodesBuf = std::move(b2); ppNodesCap = cap;
        return SDFHIP_OK;
    };
    DevBuf<uint32_t> ppFlag, ppScan, ppSplitFlag, ppSplitScan, ppFirstLeaf, ppSize, ppEx, ppOff, ppSegStart, ppSegEnd, ppRootTot, ppRootBase;

    std::vector<std::unique_ptr<CLevelHost>> LV(maxDepth + 1);
    {   // root level
        std::unique_ptr<CLevelHost> L(new CLevelHost());
        L->depth = sod; SDF_TRY(L->alloc(1u << (3 * sod)));
        const float newSize = (float)(0.5f * (bmax[0] - bmin[0]) * std::pow(0.5f, sod));
        L->half = newSize;
        const uint32_t n = L->n, vpa = 1u << sod;
        std::vector<float> hc(3 * n); std::vector<uint32_t> hco(n), hpci(n, NONE32), hword(n, NONE32), hn(6 * n, 0u); std::vector<uint8_t> hp(n, 0), hnd(6 * n, 0), ht(n, 0);
        const float scx = bmin[0] + newSize, scy = bmin[1] + newSize, scz = bmin[2] + newSize;
        for (uint32_t k = 0; k < vpa; k++) for (uint32_t j = 0; j < vpa; j++) for (uint32_t i = 0; i < vpa; i++) {
            const uint32_t r = i + vpa * j + vpa * vpa * k;
            hc[3 * r] = scx + ((float)i * 2.0f) * newSize; hc[3 * r + 1] = scy + ((float)j * 2.0f) * newSize; hc[3 * r + 2] = scz + ((float)k * 2.0f) * newSize;
            hco[r] = i | (j << 10) | (k << 20);
            if (sod == startDepth) hword[r] = k * G * G + j * G + i;
        }
        SDF_HIP_CHECK(hipMemcpyAsync(L->center.p, hc.data(), 4 * hc.size(), hipMemcpyHostToDevice, st));
        SDF_HIP_CHECK(hipMemcpyAsync(L->coord.p, hco.data(), 4 * n, hipMemcpyHostToDevice, st));
        SDF_HIP_CHECK(hipMemcpyAsync(L->pci.p, hpci.data(), 4 * n, hipMemcpyHostToDevice, st));
        SDF_HIP_CHECK(hipMemcpyAsync(L->word.p, hword.data(), 4 * n, hipMemcpyHostToDevice, st));
        SDF_HIP_CHECK(hipMemcpyAsync(L->nIdx.p, hn.data(), 24 * n, hipMemcpyHostToDevice, st));
        SDF_HIP_CHECK(hipMemcpyAsync(L->path.p, hp.data(), n, hipMemcpyHostToDevice, st));
        SDF_HIP_CHECK(hipMemcpyAsync(L->nDepth.p, hnd.data(), 6 * n, hipMemcpyHostToDevice, st));
        SDF_HIP_CHECK(hipMemcpyAsync(L->terminal.p, ht.data(), n, hipMemcpyHostToDevice, st));
        SDF_HIP_CHECK(hipStreamSynchronize(st));
        T->info.num_samples += 8ull * n;
        LV[sod] = std::move(L);
    }

    uint64_t numRescheduled = 0, ppRoots = 0, ppNodes = 0, ppSplits = 0;
    SampleScratch SS;
    SS.near = &ctx->nearScratch; ctx->nearScratch.counterReady = false;      // (builds on one context are serialised by its buildLock)
    if (ctx->exchange.world >= 1 && ctx->exchange.acquire) SS.exchange = &ctx->exchange;      // multi-GPU: every rank traverses its share of each sample batch
    DevBuf<float> ppPos; DevBuf<uint32_t> ppSlot, ppTri, ppCount;      // post-pass samples through the two-phase search
    DevBuf<OpDev> dops; DevBuf<float> scratch; DevBuf<uint32_t> cflag, cscan, clist, totals3;      // post-pass device buffers, grow-only
    {   // Levels down to the start depth exist a priori (every node above it subdivides): create their geometry now — kc_children will
        // write the same centres / coordinates again together with everything else — and take the root corners and all their
        // mid-point samples in ONE deduplicated batch instead of one latency-bound launch per level.
        SampleBatch B;
        CLevelHost* R0 = LV[sod].get();
        B.add(R0->center.p, R0->coord.p, R0->half, R0->n, 8, R0->vv.p, 8);
        for (uint32_t d = sod; d <= startDepth && d < maxDepth; d++) {
            CLevelHost* L = LV[d].get();
            B.add(L->center.p, L->coord.p, L->half, L->n, 19, L->mid.p, 8);
            L->sampled = true;
            if (d < startDepth && d + 1 <= maxDepth) {
                std::unique_ptr<CLevelHost> N(new CLevelHost());
                N->depth = d + 1; N->half = 0.5f * L->half; SDF_TRY(N->alloc(8u * L->n));
                k_expand_geometry<<<gridFor(8ull * L->n, 256), 256, 0, st>>>(L->center.p, L->coord.p, L->half, L->n, N->center.p, N->coord.p);
                LV[d + 1] = std::move(N);
            }
        }
        SDF_TRY(sampleBatch(st, md, B, SS, stackBytes, T->info.num_traversals));
    }
    double tIter = 0, tPlan = 0, tOps = 0; double tMark = nowSeconds();
    auto lap = [&](double& acc) { const double now = nowSeconds(); acc += now - tMark; tMark = now; };
    for (uint32_t cd = sod; cd <= maxDepth; cd++) {
        SDF_TRY(sampleBatchEnd(st, md, SS));       // the samples begun during the previous level's planning
        CLevelHost* L = LV[cd].get();
        if (!L || L->n == 0) continue;
        const uint32_t n = L->n;
        CLevelDev Ld = L->dev();
        // ---------------- Iter 1
        if (cd < maxDepth) {
            if (cd > startDepth) kc_refresh<<<gridFor(n, 256), 256, 0, st>>>(Ld, cd, oc.p);
            if (!L->sampled) SDF_TRY(sampleMidPoints(st, md, L->coord.p, L->center.p, L->half, n, L->mid.p, 8, SS, stackBytes, T->info.num_traversals));
            T->info.num_samples += 19ull * n;
        }
        kc_fit_rule<<<gridFor(n, 128), 128, 0, st>>>(Ld, cd, startDepth, maxDepth, P->rule, sqThr, param1, oc.p);
        // ---------------- Iter 2
        kc_iter2_masks<<<gridFor(32ull * n, 256), 256, 0, st>>>(Ld, cd, startDepth, maxDepth, G, sqThr, NM, oc.p);
        // the three counts the host needs before it can go on - words to allocate, inner nodes, post-pass candidates - all follow from
        // kc_iter2_masks: their scans run back to back and ONE read-back brings the totals (three read-backs per level until round 5)
        SDF_TRY(cflag.reserve(24ull * n)); SDF_TRY(cscan.reserve(24ull * n)); SDF_TRY(totals3.reserve(3));
        SDF_TRY(scanExclusive(st, scanTmp, scanTmpBytes, L->allocSize.p, L->allocOff.p, n));
        SDF_TRY(scanExclusive(st, scanTmp, scanTmpBytes, L->inner.p, L->childSlot.p, n));
        kc_flag_cand<<<gridFor(24ull * n, 256), 256, 0, st>>>(L->cand.p, 24 * n, cflag.p);
        SDF_TRY(scanExclusive(st, scanTmp, scanTmpBytes, cflag.p, cscan.p, 24 * n));
        kc_totals3<<<1, 64, 0, st>>>(L->allocOff.p + (n - 1), L->allocSize.p + (n - 1), L->childSlot.p + (n - 1), L->inner.p + (n - 1), cscan.p + (24ull * n - 1), cflag.p + (24ull * n - 1), totals3.p);
        uint32_t h3[3] = {0, 0, 0};
        SDF_TRY(readBackWords(st, totals3.p, nullptr, 3, h3));
        const uint32_t allocTotal = h3[0], numInner = h3[1], numCand = h3[2];
        SDF_REQUIRE((uint64_t)ocSize + allocTotal < (uint64_t)INDEX_MASK, "octree exceeds the 30-bit node index of the reference layout");
        const uint32_t base = ocSize;
        SDF_TRY(ensureOc((size_t)ocSize + allocTotal));
        kc_iter2_write<<<gridFor(64ull * n, 256), 256, 0, st>>>(Ld, cd, startDepth, base, oc.p, stats.p);
        if (cd >= startDepth) kc_register_leaves<<<gridFor(n, 256), 256, 0, st>>>(Ld, cd, base, G3, recOf.p);
        kc_mul8<<<gridFor(n, 256), 256, 0, st>>>(n, L->childSlot.p);
        if (numInner > 0) {
            if (!LV[cd + 1]) {
                std::unique_ptr<CLevelHost> N(new CLevelHost());
                N->depth = cd + 1; N->half = 0.5f * L->half; SDF_TRY(N->alloc(8u * numInner));
                LV[cd + 1] = std::move(N);
            }
            SDF_REQUIRE(LV[cd + 1]->n == 8u * numInner, "internal: pre-created level has the wrong size");
            kc_children<<<gridFor(64ull * n, 256), 256, 0, st>>>(Ld, LV[cd + 1]->dev(), cd, startDepth, base, G);
        }
        // candidates of the post-pass, in node order then slot order
        if (numCand) {
            SDF_TRY(clist.reserve(numCand));
            kc_compact_cand<<<gridFor(24ull * n, 256), 256, 0, st>>>(L->cand.p, cflag.p, cscan.p, 24 * n, clist.p);
        }
        SDF_HIP_CHECK(hipGetLastError());
        ocSize += allocTotal;
        static const bool timingSync = getenv("SDFHIP_TIMING") != nullptr;
        if (timingSync) SDF_HIP_CHECK(hipStreamSynchronize(st));          // (only the phase times want it)
        lap(tIter);
        // The exact samples of the NEXT level depend only on its node centres, not on the post-pass below: enqueue them now so
        // that the GPU traverses the BVH while the host plans (the post-pass device ops queue up behind them on the stream).
        if (cd + 1 < maxDepth && LV[cd + 1] && LV[cd + 1]->n > 0 && !LV[cd + 1]->sampled) {
            CLevelHost* N = LV[cd + 1].get();
            SampleBatch B;
            B.add(N->center.p, N->coord.p, N->half, N->n, 19, N->mid.p, 8);
            SDF_TRY(sampleBatchBegin(st, md, B, SS, stackBytes, T->info.num_traversals));      // ended at the top of the next iteration
            N->sampled = true;
        }
        numRescheduled += numCand;
        if (numCand == 0) continue;

        std::vector<size_t> gBegin; uint32_t na = 0;      // the pass's op list on the device (dops): generation g = [gBegin[g], gBegin[g + 1])
        {
            // ---------------- post-pass, integer part on the device: roots -> generations (decide, expand) -> allocation scan -> words + ops
            SDF_TRY(ppFlag.reserve(numCand)); SDF_TRY(ppScan.reserve(numCand));
            kpp_first<<<gridFor(numCand, 256), 256, 0, st>>>(clist.p, numCand, oc.p, G3, firstOcc.p);
            kpp_rootflag<<<gridFor(numCand, 256), 256, 0, st>>>(clist.p, numCand, oc.p, G3, firstOcc.p, ppFlag.p);
            SDF_TRY(scanExclusive(st, scanTmp, scanTmpBytes, ppFlag.p, ppScan.p, numCand));
            uint32_t numRoots = 0; SDF_TRY(lastPlus(st, ppScan.p, ppFlag.p, numCand, numRoots));
            if (numRoots == 0) { lap(tPlan); continue; }
            ppRoots += numRoots;
            LevelIntTable LI{};
            for (uint32_t d = 0; d <= maxDepth && d < 12; d++) if (LV[d]) LI.lv[d] = LevelInt{LV[d]->path.p, LV[d]->pci.p, LV[d]->coord.p, LV[d]->nIdx.p, LV[d]->nDepth.p};
            SDF_TRY(ensureNodes(0, std::max<size_t>(4096, 16ull * numRoots)));
            kpp_roots<<<gridFor(numCand, 256), 256, 0, st>>>(clist.p, numCand, oc.p, G3, firstOcc.p, ppFlag.p, ppScan.p, recOf.p, LI, pRec.p, ppNodesBuf.p);
            uint32_t count = numRoots, N = numRoots, passSplits = 0;
            gBegin.push_back(0);
            for (uint32_t g = 0;; g++) {
                SDF_REQUIRE(g < PP_MAXGEN, "internal: post-pass search deeper than the octree");
                SDF_TRY(ppSplitFlag.reserve(count)); SDF_TRY(ppSplitScan.reserve(count));
                kpp_decide<<<gridFor(count, 128), 128, 0, st>>>(ppNodesBuf.p, (uint32_t)gBegin[g], count, cd, startDepth, G, NM, oc.p, ppSplitFlag.p);
                SDF_TRY(scanExclusive(st, scanTmp, scanTmpBytes, ppSplitFlag.p, ppSplitScan.p, count));
                uint32_t ns = 0; SDF_TRY(lastPlus(st, ppSplitScan.p, ppSplitFlag.p, count, ns));
                gBegin.push_back(N);
                if (ns == 0) break;
                SDF_REQUIRE((uint64_t)N + 8ull * ns < (1ull << 31), "post-pass too large");
                SDF_TRY(ensureNodes(N, (size_t)N + 8ull * ns));
                kpp_expand<<<gridFor(8ull * count, 256), 256, 0, st>>>(ppNodesBuf.p, (uint32_t)gBegin[g], count, ppSplitFlag.p, ppSplitScan.p, N, poolCount + 8u * passSplits, startDepth, G);
                passSplits += ns; count = 8u * ns; N += count;
            }
            ppNodes += N; ppSplits += passSplits;
            SDF_TRY(ppFirstLeaf.reserve(numRoots)); SDF_TRY(ppSegStart.reserve((size_t)PP_MAXGEN * numRoots)); SDF_TRY(ppSegEnd.reserve((size_t)PP_MAXGEN * numRoots));
            SDF_TRY(ppRootTot.reserve(numRoots)); SDF_TRY(ppRootBase.reserve(numRoots)); SDF_TRY(ppSize.reserve(N)); SDF_TRY(ppEx.reserve(N)); SDF_TRY(ppOff.reserve(N));
            SDF_HIP_CHECK(hipMemsetAsync(ppFirstLeaf.p, 0xFF, 4ull * numRoots, st));
            SDF_HIP_CHECK(hipMemsetAsync(ppSegStart.p, 0, 4ull * PP_MAXGEN * numRoots, st));
            SDF_HIP_CHECK(hipMemsetAsync(ppSegEnd.p, 0, 4ull * PP_MAXGEN * numRoots, st));
            kpp_firstleaf<<<gridFor(N, 256), 256, 0, st>>>(ppNodesBuf.p, N, ppFirstLeaf.p);
            kpp_sizes<<<gridFor(N, 256), 256, 0, st>>>(ppNodesBuf.p, N, ppFirstLeaf.p, ppSize.p);
            SDF_TRY(scanExclusive(st, scanTmp, scanTmpBytes, ppSize.p, ppEx.p, N));
            kpp_bounds<<<gridFor(N, 256), 256, 0, st>>>(ppNodesBuf.p, N, ppSize.p, ppEx.p, ppSegStart.p, ppSegEnd.p);
            kpp_roottot<<<gridFor(numRoots, 256), 256, 0, st>>>(numRoots, ppSegStart.p, ppSegEnd.p, ppRootTot.p);
            SDF_TRY(scanExclusive(st, scanTmp, scanTmpBytes, ppRootTot.p, ppRootBase.p, numRoots));
            uint32_t passAlloc = 0; SDF_TRY(lastPlus(st, ppRootBase.p, ppRootTot.p, numRoots, passAlloc));
            kpp_offsets<<<gridFor(N, 256), 256, 0, st>>>(ppNodesBuf.p, N, ppEx.p, ppSegStart.p, ppSegEnd.p, ppRootBase.p, ppOff.p);
            SDF_REQUIRE((uint64_t)ocSize + passAlloc < (uint64_t)INDEX_MASK, "octree exceeds the 30-bit node index of the reference layout");
            SDF_TRY(ensureOc((size_t)ocSize + passAlloc));
            SDF_TRY(ensurePool((size_t)poolCount + 8ull * passSplits));
            SDF_TRY(dops.reserve(N)); SDF_TRY(scratch.reserve(216ull * N));
            kpp_finalise<<<gridFor(N, 256), 256, 0, st>>>(ppNodesBuf.p, N, ppOff.p, ppFirstLeaf.p, ocSize, G3, oc.p, recOf.p, pRec.p, dops.p);
            SDF_HIP_CHECK(hipGetLastError());
            ocSize += passAlloc; poolCount += 8u * passSplits; na = N;
            lap(tPlan);
        }
        LevelTable LT{};
        for (uint32_t d = 0; d <= maxDepth && d < 12; d++) if (LV[d]) LT.lv[d] = LevelPtrs{LV[d]->center.p, LV[d]->vv.p, LV[d]->coeff.p, LV[d]->mid.p, LV[d]->half};
        PoolDev PD{pCenter.p, pHalf.p, pVv.p};
        {
            // geometry first, then ONE launch for every exact sample of the post-pass, then fit / mid-point / children values
            // generation by generation — no host round trip between
            if (na) {
                for (size_t g = 0; g + 1 < gBegin.size(); g++) {
                    const uint32_t no = (uint32_t)(gBegin[g + 1] - gBegin[g]);
                    if (no) kc_pp_geom<<<gridFor(8ull * no, 256), 256, 0, st>>>(dops.p + gBegin[g], no, LT, PD, pCenter.p, pHalf.p);
                }
                if (nearestExactOnly()) kc_pp_sample_all<<<gridFor(19ull * na, 128), 128, stackBytes, st>>>(md, dops.p, na, LT, PD, scratch.p);
                else {
                    SDF_TRY(ppPos.reserve(57ull * na)); SDF_TRY(ppSlot.reserve(19ull * na)); SDF_TRY(ppTri.reserve(19ull * na)); SDF_TRY(ppCount.reserve(1));
                    SDF_HIP_CHECK(hipMemsetAsync(ppCount.p, 0, 4, st));
                    kc_pp_list<<<gridFor(19ull * na, 128), 128, 0, st>>>(dops.p, na, LT, PD, ppPos.p, ppSlot.p, ppCount.p);
                    uint32_t ns = 0;
                    SDF_TRY(readBackWords(st, ppCount.p, nullptr, 1, &ns));
                    if (ns) {
                        int depth = 1; while ((1ull << (depth - 1)) < mesh->numTriangles) depth++;
                        SDF_TRY(nearestTwoPhase(st, md.bvh, ppPos.p, ns, ppTri.p, ctx->nearScratch, depth + 2, 0u, 1u));
                        kc_pp_values<<<gridFor(ns, 128), 128, 0, st>>>(md, dops.p, ppPos.p, ppSlot.p, ppTri.p, ns, scratch.p);
                    }
                }
                for (size_t g = 0; g + 1 < gBegin.size(); g++) {
                    const uint32_t no = (uint32_t)(gBegin[g + 1] - gBegin[g]);
                    if (!no) continue;
                    const OpDev* ops = dops.p + gBegin[g];
                    kc_pp_fit<<<gridFor(no, 128), 128, 0, st>>>(ops, no, LT, PD, scratch.p, oc.p);
                    kc_pp_mid<<<gridFor(19ull * no, 128), 128, 0, st>>>(md, ops, no, LT, PD, scratch.p, thr, sqThr);
                    kc_pp_children<<<gridFor(64ull * no, 256), 256, 0, st>>>(ops, no, LT, PD, scratch.p, pCenter.p, pHalf.p, pVv.p);
                    T->info.num_samples += 19ull * no;
                }
                SDF_HIP_CHECK(hipGetLastError());
            }
        }
        lap(tOps);
    }
    SDF_TRY(sampleBatchEnd(st, md, SS));
    ctx->contCaps[0] = ocCap; ctx->contCaps[1] = poolCap; ctx->contCaps[2] = ppNodesCap;
    if (getenv("SDFHIP_TIMING")) fprintf(stderr, "[sdfhip] continuity: level kernels %.3f s, post-pass planning (device) %.3f s, post-pass ops %.3f s (host clock; phases overlap unless timing is on); post-pass: %llu scheduled, %llu still leaves, %llu nodes visited, %llu splits\n", tIter, tPlan, tOps,
                                            (unsigned long long)numRescheduled, (unsigned long long)ppRoots, (unsigned long long)ppNodes, (unsigned long long)ppSplits);
    // final statistics on the device
    DevBuf<unsigned long long> lpd; SDF_TRY(lpd.reserve(16));
    SDF_HIP_CHECK(hipMemsetAsync(lpd.p, 0, 128, st));
    {
        DevBuf<uint32_t> fa[2], fc[2], cnt; SDF_TRY(cnt.reserve(1));
        uint32_t n = G3, cur = 0;
        SDF_TRY(fa[0].reserve(n)); SDF_TRY(fc[0].reserve(n));
        kc_walk_init<<<gridFor(G3, 256), 256, 0, st>>>(G, fa[0].p, fc[0].p);
        for (uint32_t d = startDepth; d <= maxDepth && n > 0; d++) {
            const uint64_t cap = 8ull * n;
            SDF_TRY(fa[cur ^ 1].reserve(cap)); SDF_TRY(fc[cur ^ 1].reserve(cap));
            SDF_HIP_CHECK(hipMemsetAsync(cnt.p, 0, 4, st));
            kc_walk_level<<<gridFor(n, 256), 256, 0, st>>>(oc.p, fa[cur].p, fc[cur].p, n, d, fa[cur ^ 1].p, fc[cur ^ 1].p, cnt.p, lpd.p, stats.p + 1);
            SDF_TRY(readBackWords(st, cnt.p, nullptr, 1, &n));
            cur ^= 1;
        }
    }
    unsigned long long hl[16]; uint32_t hs[2];
    SDF_HIP_CHECK(hipMemcpyAsync(hl, lpd.p, 128, hipMemcpyDeviceToHost, st));
    SDF_HIP_CHECK(hipMemcpyAsync(hs, stats.p, 8, hipMemcpyDeviceToHost, st));
    SDF_HIP_CHECK(hipStreamSynchronize(st));
    for (int d = 0; d < 16; d++) { T->info.leaves_per_depth[d] = hl[d]; T->info.num_leaves += hl[d]; }
    memcpy(&T->info.value_range, &hs[0], 4);
    { const uint32_t k = hs[1]; const uint32_t b = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k; float f; memcpy(&f, &b, 4); T->info.min_border_value = (k == 0xFFFFFFFFu) ? INFINITY : f; }
    T->info.num_words = ocSize; T->info.cell_end = G3; T->info.body_words = ocSize - G3; T->info.body_offset = G3;
    T->info.post_pass_scheduled = numRescheduled;
    // The tree keeps its QUERY layout, made here from the working array (one sweep per level, the blocks come from this build's
    // allocation scope), and no resident copy of the array: download / device_words / .bin rebuild it from the layout bit for bit.
    SDF_TRY(octreeLayoutFromArray(T.get(), oc.p));
    if (T->qNodes + 64ull * T->qLeaves != (uint64_t)ocSize) {          // words that belong to no node or coefficient block: such an array cannot be rebuilt from the layout and stays
        SDF_TRY(T->data.reserve(ocSize));
        SDF_HIP_CHECK(hipMemcpyAsync(T->data.p, oc.p, 4ull * ocSize, hipMemcpyDeviceToDevice, st));
        SDF_HIP_CHECK(hipStreamSynchronize(st));
    }
    T->hasData = true; T->built = true;
    SDF_TRY(sampleFallbacks(st, SS, T->info));
    T->info.seconds_total = nowSeconds() - tStart;
    *out = T.release();
    return SDFHIP_OK;
}

}  // namespace sdfhip

// (sdfhip_ctx_create: the runtime loads a translation unit's code object on the first use of one of its kernels — milliseconds that would
// otherwise land in the first build or the first query of a process)
namespace sdfhip { void loadKernelsOctreeContinuity() { hipFuncAttributes a; (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&kc_mul8)); (void)hipGetLastError(); } }
