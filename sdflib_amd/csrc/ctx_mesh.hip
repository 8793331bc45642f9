// Context + mesh preparation (TriangleData on the device).  PRODUCT code — independent of oracle/.
//
// Reproduces TriangleUtils::calculateMeshTriangleData (reference src/utils/TriangleUtils.cpp:7-86, 422-427) as
// sort/scan passes instead of the reference's serial std::map walk:
//   k_triangle_frames : one lane per triangle, TriangleData ctor (TriangleUtils.h:23-42)
//   edge pseudonormals: 64-bit key (vmin,vmax) per half-edge, stable radix sort, consecutive entries of one key
//                       are paired (1st,2nd),(3rd,4th)... exactly like the map's insert/erase sequence (:63-83)
//   vertex pseudonormals: (vertex, 3t+k) pairs stable-sorted by vertex, then ONE lane sums a vertex's
//                       contributions sequentially in ascending 3t+k — the reference's float addition order (:85-86)
// Not reproduced: degenerate-triangle branches (dead in the reference: `if(false && ...)`, :45).  Non-manifold
// seam welding (:292-420) runs when sdfhip_mesh_create_ex is given the mesh bounding box (host planner weldSeams below);
// otherwise single-owner edges keep the default (0,0,1).  They are counted in mesh->unmatchedEdges either way.
#include "sdfhip_internal.h"
#include <memory>
#include <atomic>
#include <thread>
#include <cmath>
#include "dev_math.h"
#include "dev_prims.h"
#include <string.h>
#include <map>
#include <algorithm>

namespace sdfhip {

static thread_local std::string g_lastError;

void setError(const char* fmt, ...) {
    char buf[1024];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
    g_lastError = buf;
}

__global__ void k_triangle_frames(const float* __restrict__ verts, const uint32_t* __restrict__ idx, uint32_t numTriangles, float* __restrict__ td) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= numTriangles) return;
    const uint32_t a = idx[3 * t], b = idx[3 * t + 1], c = idx[3 * t + 2];
    const F3 p1 = F3{verts[3 * a], verts[3 * a + 1], verts[3 * a + 2]};
    const F3 p2 = F3{verts[3 * b], verts[3 * b + 1], verts[3 * b + 2]};
    const F3 p3 = F3{verts[3 * c], verts[3 * c + 1], verts[3 * c + 2]};
    float out[TD_FLOATS];
    makeTriangleData(p1, p2, p3, out);
    float* dst = td + (size_t)TD_FLOATS * t;
#pragma unroll
    for (int i = 0; i < TD_FLOATS; i++) dst[i] = out[i];
}

// The largest triangle index and the largest |coordinate| as a bit pattern (sign cleared: the patterns of non-negative floats order like
// their values, an infinity is 0x7F800000 and every NaN lies above it).
__global__ void __launch_bounds__(256) k_mesh_validate(const float* __restrict__ xyz, uint64_t nCoords, const uint32_t* __restrict__ idx, uint64_t nIdx, uint32_t* __restrict__ out2) {
    uint32_t mi = 0, mc = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nIdx; i += stride) { const uint32_t v = idx[i]; mi = v > mi ? v : mi; }
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nCoords; i += stride) { const uint32_t v = __float_as_uint(xyz[i]) & 0x7FFFFFFFu; mc = v > mc ? v : mc; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const uint32_t a = (uint32_t)__shfl_xor((int)mi, o), b = (uint32_t)__shfl_xor((int)mc, o); mi = a > mi ? a : mi; mc = b > mc ? b : mc; }
    if ((threadIdx.x & 63) == 0) { atomicMax(&out2[0], mi); atomicMax(&out2[1], mc); }
}

__global__ void k_halfedge_keys(const uint32_t* __restrict__ idx, uint32_t numHalfEdges, uint64_t* __restrict__ edgeKey,
                                uint32_t* __restrict__ vertKey, uint32_t* __restrict__ value) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= numHalfEdges) return;
    const uint32_t t = i / 3, k = i - 3 * t;
    const uint32_t a = idx[i], b = idx[3 * t + (k + 1) % 3];
    const uint32_t lo = a < b ? a : b, hi = a < b ? b : a;
    edgeKey[i] = ((uint64_t)lo << 32) | hi;
    vertKey[i] = a;
    value[i] = i;
}

__device__ inline void writeEdgeNormalPair(float* __restrict__ td, uint32_t later, uint32_t earlier) {
    const uint32_t t = later / 3, t2 = earlier / 3;
    const float* A = td + (size_t)TD_FLOATS * t;
    const float* B = td + (size_t)TD_FLOATS * t2;
    const F3 en = triNormal(A + 3) + triNormal(B + 3);
    const F3 ea = mulM(A + 3, en), eb = mulM(B + 3, en);
    float* da = td + (size_t)TD_FLOATS * t + 19 + 3 * (later % 3);
    float* db = td + (size_t)TD_FLOATS * t2 + 19 + 3 * (earlier % 3);
    da[0] = ea.x; da[1] = ea.y; da[2] = ea.z;
    db[0] = eb.x; db[1] = eb.y; db[2] = eb.z;
}

// seam welding (TriangleUtils.cpp:392-411): half-edge pairs found by the host planner get their summed normal
__global__ void k_weld_edges(const uint32_t* __restrict__ pairs, uint32_t n, float* __restrict__ td) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) writeEdgeNormalPair(td, pairs[2 * i], pairs[2 * i + 1]);
}

__global__ void k_edge_pair(const uint64_t* __restrict__ key, const uint32_t* __restrict__ val, uint32_t n, float* __restrict__ td,
                            uint32_t* __restrict__ unmatched, uint64_t* __restrict__ openKey, uint32_t* __restrict__ openHe) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t k = key[i];
    uint32_t r = 0;                              // position inside the run of equal keys
    while (r < i && key[i - 1 - r] == k) r++;
    if (r & 1u) {
        writeEdgeNormalPair(td, val[i], val[i - 1]);
    } else {
        const bool last = (i + 1 == n) || key[i + 1] != k;
        if (last) {                              // odd run: this half-edge has no partner
            const uint32_t slot = atomicAdd(unmatched, 1u);
            if (openKey) { openKey[slot] = k; openHe[slot] = val[i]; }
        }
    }
}

// corner angle of half-edge he = 3 t + k (corner k of triangle t): the arc cosine of the clamped cosine (TriangleUtils.cpp:85-86).
// The reference takes it with glm::acos, i.e. the platform's acosf, which is not correctly rounded — and one ulp in a vertex
// pseudonormal flips the sign of a sample that lies in the plane spanned by it.  Rounds 1-3 therefore sent the cosines to the HOST's
// libm and back (24 B per triangle over PCIe and a few ms of host threads); since round 4 the device runs glibc's algorithm itself
// (dev_math.h::acosfGlibc, equal to the running libm on every float of [-1, 1]: sdfhip_test_acosf_mismatches).  angle = false
// (a host whose libm is NOT that function: hostAcosNeeded below) writes the cosine and leaves the arc cosine to the host as before.
__global__ void k_corner_cos(const float* __restrict__ verts, const uint32_t* __restrict__ idx, uint32_t numHalfEdges, float* __restrict__ cs, bool angle) {
    const uint32_t he = blockIdx.x * blockDim.x + threadIdx.x;
    if (he >= numHalfEdges) return;
    const uint32_t t = he / 3, k = he - 3 * t;
    const uint32_t a = idx[he], b = idx[3 * t + (k + 1) % 3], c = idx[3 * t + (k + 2) % 3];
    const F3 pa = F3{verts[3 * a], verts[3 * a + 1], verts[3 * a + 2]};
    const F3 pb = F3{verts[3 * b], verts[3 * b + 1], verts[3 * b + 2]};
    const F3 pc = F3{verts[3 * c], verts[3 * c + 1], verts[3 * c + 2]};
    const float cosine = gclamp(dot(normalize(pb - pa), normalize(pc - pa)), -1.0f, 1.0f);
    cs[he] = angle ? acosfGlibc(cosine) : cosine;
}
__global__ void k_test_acosf(uint32_t firstBits, uint32_t stride, uint32_t count, float* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] = acosfGlibc(__uint_as_float(firstBits + i * stride));
}

__global__ void k_vertex_normal_sum(const uint32_t* __restrict__ vkey, const uint32_t* __restrict__ val, uint32_t n,
                                    const float* __restrict__ angle, const float* __restrict__ td,
                                    float* __restrict__ vnormal) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t v = vkey[i];
    if (i > 0 && vkey[i - 1] == v) return;       // only the head of a run works
    F3 acc = F3{0.f, 0.f, 0.f};
    for (uint32_t j = i; j < n && vkey[j] == v; j++) {
        const uint32_t he = val[j], t = he / 3;
        acc = acc + angle[he] * triNormal(td + (size_t)TD_FLOATS * t + 3);
    }
    vnormal[3 * v] = acc.x; vnormal[3 * v + 1] = acc.y; vnormal[3 * v + 2] = acc.z;
}

__global__ void k_vertex_normal_apply(const uint32_t* __restrict__ idx, uint32_t numHalfEdges, const float* __restrict__ vnormal, float* __restrict__ td) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= numHalfEdges) return;
    const uint32_t t = i / 3, k = i - 3 * t, v = idx[i];
    float* T = td + (size_t)TD_FLOATS * t;
    const F3 r = mulM(T + 3, F3{vnormal[3 * v], vnormal[3 * v + 1], vnormal[3 * v + 2]});
    T[28 + 3 * k] = r.x; T[29 + 3 * k] = r.y; T[30 + 3 * k] = r.z;
}

__global__ void k_pack_frames(const float* __restrict__ td, uint32_t n, float* __restrict__ frames) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t t = gid / 20u, k = gid - 20u * t;
    if (t >= n) return;
    frames[gid] = (k < 19u) ? td[(size_t)TD_FLOATS * t + k] : 0.f;
}
int packFrames(hipStream_t st, const float* td, uint32_t numTriangles, float* frames) {
    k_pack_frames<<<gridFor(20ull * numTriangles, 256), 256, 0, st>>>(td, numTriangles, frames);
    SDF_HIP_CHECK(hipGetLastError());
    return SDFHIP_OK;
}

// Host planner of the non-manifold seam welding (reference: src/utils/TriangleUtils.cpp:292-420).  Integer/set logic and
// the per-vertex normal merge run here (IEEE float adds, no contraction => same bits as the reference's loop); the
// per-edge normal writes go back to the device (k_weld_edges).  `openKey` = (min vertex << 32 | max vertex) of every
// single-owner edge, `openHe` its half-edge; `pairs` receives (later, earlier) half-edges whose edges coincide after welding.
static void weldSeams(const float* verts, const float* bbox6, const std::vector<uint64_t>& openKey, const std::vector<uint32_t>& openHe,
                      std::vector<uint32_t>& pairs, float* vnormal) {
    std::vector<uint32_t> order(openKey.size());
    for (uint32_t i = 0; i < order.size(); i++) order[i] = i;
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return openKey[a] < openKey[b]; });   // the reference's std::map order
    std::map<uint32_t, uint32_t> parent;
    auto parentOf = [&](uint32_t v) { auto it = parent.find(v); while (it != parent.end() && it->second != v) { v = it->second; it = parent.find(v); } return v; };
    std::vector<uint32_t> seam;
    for (uint64_t k : openKey) { seam.push_back((uint32_t)(k >> 32)); seam.push_back((uint32_t)k); }
    std::sort(seam.begin(), seam.end()); seam.erase(std::unique(seam.begin(), seam.end()), seam.end());
    const float sx = bbox6[3] - bbox6[0], sy = bbox6[4] - bbox6[1], sz = bbox6[5] - bbox6[2];
    const uint32_t axisRes = 2048;
    const float big = fmaxf(sx, fmaxf(sy, sz));
    const float gridScale = (float)axisRes / big;
    const float threshold = (float)(1e-5 / big);
    const float sqThr = threshold * threshold;
    auto cellOf = [&](uint32_t v, float offset) -> uint64_t {
        const int x = (int)((verts[3 * v] - bbox6[0]) * gridScale + offset), y = (int)((verts[3 * v + 1] - bbox6[1]) * gridScale + offset),
                  z = (int)((verts[3 * v + 2] - bbox6[2]) * gridScale + offset);
        return (uint32_t)((uint32_t)x + (uint32_t)y * axisRes + (uint32_t)z * axisRes * axisRes);
    };
    std::map<uint64_t, std::vector<uint32_t>> grids[2];
    // (p - start) * scale without the +0 keeps the reference's un-offset expression for the first grid
    auto cell0 = [&](uint32_t v) -> uint64_t {
        const int x = (int)((verts[3 * v] - bbox6[0]) * gridScale), y = (int)((verts[3 * v + 1] - bbox6[1]) * gridScale), z = (int)((verts[3 * v + 2] - bbox6[2]) * gridScale);
        return (uint32_t)((uint32_t)x + (uint32_t)y * axisRes + (uint32_t)z * axisRes * axisRes);
    };
    for (uint32_t v : seam) { grids[0][cell0(v)].push_back(v); grids[1][cellOf(v, 0.5f)].push_back(v); }
    for (uint32_t v : seam) {
        float offset = 0.0f;
        for (int g = 0; g < 2; g++) {
            auto it = grids[g].find(cellOf(v, offset));
            if (it != grids[g].end()) {
                for (uint32_t other : it->second) {
                    const float dx = verts[3 * v] - verts[3 * other], dy = verts[3 * v + 1] - verts[3 * other + 1], dz = verts[3 * v + 2] - verts[3 * other + 2];
                    if (dx * dx + dy * dy + dz * dz < sqThr) {
                        const uint32_t p1 = parentOf(v), p2 = parentOf(other);
                        if (v == p1) parent[p1] = p1;
                        parent[p2] = p1;
                        break;
                    }
                }
            }
            offset += 0.5f;
        }
    }
    std::map<uint64_t, uint32_t> repaired;
    for (uint32_t i : order) {
        const uint32_t a = parentOf((uint32_t)(openKey[i] >> 32)), b = parentOf((uint32_t)openKey[i]);
        const uint64_t k = a < b ? ((uint64_t)a << 32 | b) : ((uint64_t)b << 32 | a);
        auto ins = repaired.insert(std::make_pair(k, openHe[i]));
        if (!ins.second) { pairs.push_back(openHe[i]); pairs.push_back(ins.first->second); repaired.erase(ins.first); }
    }
    for (uint32_t v : seam) {
        const uint32_t p = parentOf(v);
        if (p != v) { vnormal[3 * p] = vnormal[3 * p] + vnormal[3 * v]; vnormal[3 * p + 1] = vnormal[3 * p + 1] + vnormal[3 * v + 1]; vnormal[3 * p + 2] = vnormal[3 * p + 2] + vnormal[3 * v + 2]; }
    }
    for (uint32_t v : seam) { const uint32_t p = parentOf(v); vnormal[3 * v] = vnormal[3 * p]; vnormal[3 * v + 1] = vnormal[3 * p + 1]; vnormal[3 * v + 2] = vnormal[3 * p + 2]; }
}


// ---- seam welding on the device (round 6) --------------------------------------------------------------------------------------------
// The same pass as weldSeams above, for meshes whose open edges number in the millions (an unwelded export: every edge is open).  What is
// independent of the union-find's state runs on the device: the seam-vertex list (ascending), the two 2048^3 hash grids (a stable sort by
// cell keeps each bucket in ascending vertex order = the reference's push_back order), per seam vertex and grid the FIRST bucket member
// within the threshold (TriangleUtils.cpp:361-369: the loop breaks there), the re-pairing of the open edges under the merged ids and the
// per-root normal sums (ascending member order, plain fp32 adds).  The union-find itself (:370-376) depends on the order of the unions
// (`verticesMap[p2] = p1`, no ranks): it stays sequential, on the host, over dense arrays - two words per seam vertex go down, one comes back.
__global__ void k_weld_flag_vertices(const uint64_t* __restrict__ openKey, uint32_t no, uint32_t* __restrict__ flag) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= no) return;
    const uint64_t k = openKey[i];
    flag[(uint32_t)(k >> 32)] = 1u; flag[(uint32_t)k] = 1u;
}
__global__ void k_weld_compact(const uint32_t* __restrict__ flag, const uint32_t* __restrict__ rank, uint32_t nv, uint32_t* __restrict__ seam) {
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v < nv && flag[v]) seam[rank[v]] = v;
}
struct WeldGrid { float sx, sy, sz, gridScale, sqThr; };
SDF_DEV uint32_t weldCell(const float* __restrict__ verts, uint32_t v, const WeldGrid& G, float offset, bool plain) {
    const uint32_t axisRes = 2048;
    // (p - start) * scale, then `+ offset` only where the reference's expression has it (the first grid is filled without the term)
    const float fx = (verts[3 * v] - G.sx) * G.gridScale, fy = (verts[3 * v + 1] - G.sy) * G.gridScale, fz = (verts[3 * v + 2] - G.sz) * G.gridScale;
    const int x = plain ? (int)fx : (int)(fx + offset), y = plain ? (int)fy : (int)(fy + offset), z = plain ? (int)fz : (int)(fz + offset);
    return (uint32_t)x + (uint32_t)y * axisRes + (uint32_t)z * axisRes * axisRes;
}
__global__ void k_weld_cells(const float* __restrict__ verts, const uint32_t* __restrict__ seam, uint32_t ns, WeldGrid G, uint32_t* __restrict__ key0, uint32_t* __restrict__ key1,
                             uint32_t* __restrict__ ident) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ns) return;
    key0[i] = weldCell(verts, seam[i], G, 0.0f, true); key1[i] = weldCell(verts, seam[i], G, 0.5f, false); ident[i] = i;
}
SDF_DEV uint32_t weldLowerBound(const uint32_t* __restrict__ keys, uint32_t n, uint32_t k) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (keys[mid] < k) lo = mid + 1; else hi = mid; }
    return lo;
}
// first[g][i] = seam index of the first member (ascending vertex id) of seam vertex i's bucket in grid g that lies within the threshold, or NONE
__global__ void k_weld_first(const float* __restrict__ verts, const uint32_t* __restrict__ seam, uint32_t ns, WeldGrid G,
                             const uint32_t* __restrict__ sKey0, const uint32_t* __restrict__ sIdx0, const uint32_t* __restrict__ sKey1, const uint32_t* __restrict__ sIdx1,
                             uint32_t* __restrict__ first0, uint32_t* __restrict__ first1) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ns) return;
    const uint32_t v = seam[i];
    const float px = verts[3 * v], py = verts[3 * v + 1], pz = verts[3 * v + 2];
    for (int g = 0; g < 2; g++) {
        const uint32_t* sKey = g ? sKey1 : sKey0; const uint32_t* sIdx = g ? sIdx1 : sIdx0;
        // the LOOK-UP key carries `+ offset` in both grids (offset = 0.0f for the first: x + 0.0f truncates like x)
        const uint32_t k = weldCell(verts, v, G, g ? 0.5f : 0.0f, false);
        uint32_t found = 0xFFFFFFFFu;
        for (uint32_t j = weldLowerBound(sKey, ns, k); j < ns && sKey[j] == k; j++) {
            const uint32_t o = seam[sIdx[j]];
            const float dx = px - verts[3 * o], dy = py - verts[3 * o + 1], dz = pz - verts[3 * o + 2];
            const float xx = dx * dx, yy = dy * dy, zz = dz * dz;
            if ((xx + yy) + zz < G.sqThr) { found = sIdx[j]; break; }
        }
        (g ? first1 : first0)[i] = found;
    }
}
// open edge j of the key-sorted sequence (= the reference's std::map order) under the merged vertex ids
__global__ void k_weld_rekey(const uint64_t* __restrict__ sKey, uint32_t no, const uint32_t* __restrict__ rank, const uint32_t* __restrict__ root, const uint32_t* __restrict__ seam,
                             uint64_t* __restrict__ newKey, uint32_t* __restrict__ ident) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= no) return;
    const uint64_t k = sKey[j];
    const uint32_t a = seam[root[rank[(uint32_t)(k >> 32)]]], b = seam[root[rank[(uint32_t)k]]];
    newKey[j] = a < b ? ((uint64_t)a << 32 | b) : ((uint64_t)b << 32 | a);
    ident[j] = j;
}
// runs of equal merged keys, in insertion order: members 0-1, 2-3, ... pair up (insert, meet-and-erase, insert ...: TriangleUtils.cpp:384-400)
__global__ void k_weld_pair(const uint64_t* __restrict__ key, const uint32_t* __restrict__ seq, uint32_t no, const uint32_t* __restrict__ sHe, float* __restrict__ td, uint32_t* __restrict__ paired) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= no) return;
    const uint64_t k = key[i];
    uint32_t r = 0;
    while (r < i && key[i - 1 - r] == k) r++;
    if (r & 1u) { writeEdgeNormalPair(td, sHe[seq[i]], sHe[seq[i - 1]]); atomicAdd(paired, 2u); }
}
__global__ void k_weld_rootkeys(const uint32_t* __restrict__ root, uint32_t ns, uint32_t* __restrict__ key, uint32_t* __restrict__ ident) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < ns) { key[i] = root[i]; ident[i] = i; }
}
// one thread per root (run head of the root-sorted member list): root's normal + its members' in ascending vertex order (:404-408)
__global__ void k_weld_sum(const uint32_t* __restrict__ sRoot, const uint32_t* __restrict__ sMember, uint32_t ns, const uint32_t* __restrict__ seam, const float* __restrict__ vn, float* __restrict__ sums) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ns) return;
    const uint32_t r = sRoot[i];
    if (i > 0 && sRoot[i - 1] == r) return;
    const uint32_t vr = seam[r];
    float ax = vn[3 * vr], ay = vn[3 * vr + 1], az = vn[3 * vr + 2];
    for (uint32_t j = i; j < ns && sRoot[j] == r; j++) {
        const uint32_t m = sMember[j];
        if (m == r) continue;
        const uint32_t vm = seam[m];
        ax = ax + vn[3 * vm]; ay = ay + vn[3 * vm + 1]; az = az + vn[3 * vm + 2];
    }
    sums[3 * r] = ax; sums[3 * r + 1] = ay; sums[3 * r + 2] = az;
}
__global__ void k_weld_spread(const uint32_t* __restrict__ root, uint32_t ns, const uint32_t* __restrict__ seam, const float* __restrict__ sums, float* __restrict__ vn) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ns) return;
    const uint32_t v = seam[i], r = root[i];
    vn[3 * v] = sums[3 * r]; vn[3 * v + 1] = sums[3 * r + 1]; vn[3 * v + 2] = sums[3 * r + 2];
}

template <typename K, typename V>
static int weldSort(hipStream_t st, DevBuf<unsigned char>& tmp, const K* kIn, K* kOut, const V* vIn, V* vOut, size_t n, unsigned bits) {
    size_t need = 0;
    SDF_HIP_CHECK(devSortPairs(nullptr, need, kIn, kOut, vIn, vOut, n, 0, bits, st));
    if (need > tmp.n) SDF_TRY(tmp.reserve(need));
    SDF_HIP_CHECK(devSortPairs(tmp.p, need, kIn, kOut, vIn, vOut, n, 0, bits, st));
    return SDFHIP_OK;
}

// dVerts / td / vnormal on the device; openKey / openHe = the `no` single-owner edges in any order.  Returns the number of welded half-edges.
static int weldSeamsDevice(hipStream_t st, const float* dVerts, uint32_t nv, const float* bbox6, const uint64_t* openKey, const uint32_t* openHe, uint32_t no,
                           float* td, float* vnormal, uint32_t* weldedHalfEdges, double* hostSeconds) {
    *weldedHalfEdges = 0;
    unsigned vbits = 1; while (vbits < 32u && (1ull << vbits) < (unsigned long long)nv) vbits++;
    DevBuf<unsigned char> tmp;
    // the reference's map order of the open edges
    DevBuf<uint64_t> sKey; DevBuf<uint32_t> sHe;
    SDF_TRY(sKey.reserve(no)); SDF_TRY(sHe.reserve(no));
    SDF_TRY(weldSort(st, tmp, openKey, sKey.p, openHe, sHe.p, no, 32u + vbits));
    // seam vertices, ascending
    DevBuf<uint32_t> flag, rank, counts; SDF_TRY(flag.reserve(nv)); SDF_TRY(rank.reserve(nv)); SDF_TRY(counts.reserve(2));
    SDF_HIP_CHECK(hipMemsetAsync(flag.p, 0, 4ull * nv, st));
    k_weld_flag_vertices<<<gridFor(no, 256), 256, 0, st>>>(openKey, no, flag.p);
    { size_t need = 0; SDF_HIP_CHECK(devExclusiveSum(nullptr, need, flag.p, rank.p, (size_t)nv, st)); if (need > tmp.n) SDF_TRY(tmp.reserve(need));
      SDF_HIP_CHECK(devExclusiveSum(tmp.p, need, flag.p, rank.p, (size_t)nv, st)); }
    uint32_t ns = 0;
    SDF_TRY(readBackWords(st, rank.p + (nv - 1), flag.p + (nv - 1), 1, &ns));
    if (ns == 0) return SDFHIP_OK;
    DevBuf<uint32_t> seam; SDF_TRY(seam.reserve(ns));
    k_weld_compact<<<gridFor(nv, 256), 256, 0, st>>>(flag.p, rank.p, nv, seam.p);
    // the grids
    WeldGrid G;
    {
        const float sx = bbox6[3] - bbox6[0], sy = bbox6[4] - bbox6[1], sz = bbox6[5] - bbox6[2];
        const float big = fmaxf(sx, fmaxf(sy, sz));
        const float threshold = (float)(1e-5 / big);
        G = WeldGrid{bbox6[0], bbox6[1], bbox6[2], (float)2048u / big, threshold * threshold};
    }
    DevBuf<uint32_t> key0, key1, ident, sKey0, sIdx0, sKey1, sIdx1, first0, first1;
    SDF_TRY(key0.reserve(ns)); SDF_TRY(key1.reserve(ns)); SDF_TRY(ident.reserve(no > ns ? no : ns)); SDF_TRY(sKey0.reserve(ns)); SDF_TRY(sIdx0.reserve(ns));
    SDF_TRY(sKey1.reserve(ns)); SDF_TRY(sIdx1.reserve(ns)); SDF_TRY(first0.reserve(ns)); SDF_TRY(first1.reserve(ns));
    k_weld_cells<<<gridFor(ns, 256), 256, 0, st>>>(dVerts, seam.p, ns, G, key0.p, key1.p, ident.p);
    SDF_TRY(weldSort(st, tmp, key0.p, sKey0.p, ident.p, sIdx0.p, ns, 32u));
    SDF_TRY(weldSort(st, tmp, key1.p, sKey1.p, ident.p, sIdx1.p, ns, 32u));
    k_weld_first<<<gridFor(ns, 128), 128, 0, st>>>(dVerts, seam.p, ns, G, sKey0.p, sIdx0.p, sKey1.p, sIdx1.p, first0.p, first1.p);
    SDF_HIP_CHECK(hipGetLastError());
    // the union-find, sequential over the seam vertices in ascending order (host; dense arrays over seam indices)
    std::vector<uint32_t> f0(ns), f1(ns), parent(ns, 0xFFFFFFFFu), root(ns);
    SDF_HIP_CHECK(hipMemcpyAsync(f0.data(), first0.p, 4ull * ns, hipMemcpyDeviceToHost, st));
    SDF_HIP_CHECK(hipMemcpyAsync(f1.data(), first1.p, 4ull * ns, hipMemcpyDeviceToHost, st));
    SDF_HIP_CHECK(hipStreamSynchronize(st));
    const double t0 = nowSeconds();
    {
        uint32_t* P = parent.data();
        auto parentOf = [P](uint32_t x) { while (P[x] != 0xFFFFFFFFu && P[x] != x) x = P[x]; return x; };
        for (uint32_t i = 0; i < ns; i++) {
            for (int g = 0; g < 2; g++) {
                const uint32_t o = g ? f1[i] : f0[i];
                if (o == 0xFFFFFFFFu) continue;
                const uint32_t p1 = parentOf(i), p2 = parentOf(o);
                if (i == p1) P[p1] = p1;
                P[p2] = p1;
            }
        }
        for (uint32_t i = 0; i < ns; i++) root[i] = parentOf(i);
    }
    if (hostSeconds) *hostSeconds = nowSeconds() - t0;
    DevBuf<uint32_t> dRoot; SDF_TRY(dRoot.reserve(ns));
    SDF_HIP_CHECK(hipMemcpyAsync(dRoot.p, root.data(), 4ull * ns, hipMemcpyHostToDevice, st));
    // re-pair the open edges under the merged ids
    DevBuf<uint64_t> newKey, newKeyS; DevBuf<uint32_t> seq, paired; SDF_TRY(newKey.reserve(no)); SDF_TRY(newKeyS.reserve(no)); SDF_TRY(seq.reserve(no)); SDF_TRY(paired.reserve(1));
    SDF_HIP_CHECK(hipMemsetAsync(paired.p, 0, 4, st));
    k_weld_rekey<<<gridFor(no, 256), 256, 0, st>>>(sKey.p, no, rank.p, dRoot.p, seam.p, newKey.p, ident.p);
    SDF_TRY(weldSort(st, tmp, newKey.p, newKeyS.p, ident.p, seq.p, no, 32u + vbits));
    k_weld_pair<<<gridFor(no, 256), 256, 0, st>>>(newKeyS.p, seq.p, no, sHe.p, td, paired.p);
    // parents' normals: root + members in ascending order, then handed to every member
    DevBuf<uint32_t> rKey, rKeyS, rMem; DevBuf<float> sums; SDF_TRY(rKey.reserve(ns)); SDF_TRY(rKeyS.reserve(ns)); SDF_TRY(rMem.reserve(ns)); SDF_TRY(sums.reserve(3ull * ns));
    k_weld_rootkeys<<<gridFor(ns, 256), 256, 0, st>>>(dRoot.p, ns, rKey.p, ident.p);
    unsigned sbits = 1; while (sbits < 32u && (1ull << sbits) < (unsigned long long)ns) sbits++;
    SDF_TRY(weldSort(st, tmp, rKey.p, rKeyS.p, ident.p, rMem.p, ns, sbits));
    k_weld_sum<<<gridFor(ns, 256), 256, 0, st>>>(rKeyS.p, rMem.p, ns, seam.p, vnormal, sums.p);
    k_weld_spread<<<gridFor(ns, 256), 256, 0, st>>>(dRoot.p, ns, seam.p, sums.p, vnormal);
    SDF_HIP_CHECK(hipGetLastError());
    SDF_TRY(readBackWords(st, paired.p, nullptr, 1, weldedHalfEdges));       // (also: the host vectors above outlive every copy that reads them)
    return SDFHIP_OK;
}

}  // namespace sdfhip

namespace sdfhip {
__global__ void k_mailbox(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, int count, volatile uint32_t* mb, uint32_t seq) {
    for (int i = 0; i < count; i++) mb[2 + i] = a[i] + (b ? b[i] : 0u);
    __threadfence_system();
    mb[0] = seq;
}
namespace {
// Pinned, mapped mailboxes: a POOL per device that calls borrow from (as many boxes come to exist as calls have ever overlapped: the
// build threads of multi.hip and the callers' query threads come and go, and a box per thread was lost with its thread).
struct Mailbox { uint32_t* host = nullptr; uint32_t* dev = nullptr; uint32_t seq = 0; };
struct MailboxPool {
    std::mutex lock; std::vector<Mailbox> idle[16];
    bool take(int device, Mailbox& M) {
        {
            std::lock_guard<std::mutex> g(lock);
            if (!idle[device].empty()) { M = idle[device].back(); idle[device].pop_back(); return true; }
        }
        void* h = nullptr;
        if (hipHostMalloc(&h, 256, hipHostMallocMapped | hipHostMallocPortable | hipHostMallocCoherent) == hipSuccess && hipHostGetDevicePointer(reinterpret_cast<void**>(&M.dev), h, 0) == hipSuccess) {
            M.host = static_cast<uint32_t*>(h); M.host[0] = 0; M.seq = 0; return true;
        }
        (void)hipGetLastError(); if (h) (void)hipHostFree(h);
        return false;
    }
    void give(int device, const Mailbox& M) { std::lock_guard<std::mutex> g(lock); idle[device].push_back(M); }
};
MailboxPool& mailboxes() { static MailboxPool* p = new MailboxPool; return *p; }       // (never destroyed: no HIP call at process exit)
}
int readBackWords(hipStream_t st, const uint32_t* a, const uint32_t* b, int count, uint32_t* out) {
    SDF_REQUIRE(count >= 1 && count <= 8, "internal: readBackWords takes 1 to 8 words");
    int device = 0;
    SDF_HIP_CHECK(hipGetDevice(&device));
    Mailbox M;
    if (device >= 0 && device < 16 && mailboxes().take(device, M)) {
        // the mailbox goes back to the pool when the wait ends normally; on an error return k_mailbox may still be queued to write into it,
        // so it is dropped instead (a few bytes of pinned memory leak on a path that ends the build anyway)
        struct Return { int device; Mailbox& M; bool ok; ~Return() { if (ok) mailboxes().give(device, M); } } giveBack{device, M, false};
        const uint32_t seq = ++M.seq ? M.seq : ++M.seq;          // never 0
        k_mailbox<<<1, 1, 0, st>>>(a, b, count, M.dev, seq);
        SDF_HIP_CHECK(hipGetLastError());
        volatile uint32_t* mb = M.host;
        const double t0 = nowSeconds();
        uint32_t spins = 0;
        while (__atomic_load_n(&M.host[0], __ATOMIC_ACQUIRE) != seq) {
            __builtin_ia32_pause();
            if ((++spins & 0x3FFFu) == 0u && nowSeconds() - t0 > 0.05) {          // something long, or something wrong: let the runtime say which
                const hipError_t q = hipStreamQuery(st);
                if (q != hipSuccess && q != hipErrorNotReady) { setError("HIP error while waiting for a read-back: %s", hipGetErrorString(q)); return SDFHIP_E_HIP; }
                if (q == hipSuccess && __atomic_load_n(&M.host[0], __ATOMIC_ACQUIRE) != seq) { SDF_HIP_CHECK(hipStreamSynchronize(st)); }
            }
        }
        for (int i = 0; i < count; i++) out[i] = mb[2 + i];
        giveBack.ok = true;
        return SDFHIP_OK;
    }
    // (no pinned memory to be had: two pageable copies)
    uint32_t ha[8], hb[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    SDF_HIP_CHECK(hipMemcpyAsync(ha, a, 4 * (size_t)count, hipMemcpyDeviceToHost, st));
    if (b) SDF_HIP_CHECK(hipMemcpyAsync(hb, b, 4 * (size_t)count, hipMemcpyDeviceToHost, st));
    SDF_HIP_CHECK(hipStreamSynchronize(st));
    for (int i = 0; i < count; i++) out[i] = ha[i] + hb[i];
    return SDFHIP_OK;
}
}  // namespace sdfhip

namespace sdfhip {
// The corner angles' arc cosines run on the device as glibc 2.35's algorithm restated (dev_math.h::acosfGlibc).  The reference calls the
// RUNNING libm, and one ulp in a pseudonormal can flip the sign of a sample: on a host whose acosf is another function (a newer glibc with
// the correctly rounded CORE-MATH routine, musl ...) the restatement would silently differ from what the reference computes there.  So
// the first mesh of a process compares the two on 80 000 bit patterns — a stride through [-1, 1] plus the neighbourhoods of the
// algorithm's branch points (0, 2^-26, 0.5, 1) — and on any mismatch the arc cosines are taken on the host by the running libm, as in
// rounds 1-3 (logged once).  -1: not decided yet; the test hook sdfhip_test_set_host_acos forces either path.
static std::atomic<int> g_hostAcos{-1};
static bool hostAcosNeeded() {
    int v = g_hostAcos.load(std::memory_order_acquire);
    if (v >= 0) return v != 0;
    uint64_t bad = 0;
    auto probe = [&](uint32_t bits) { float x; memcpy(&x, &bits, 4); if (!(std::fabs(x) <= 1.0f)) return; const float a = ::acosf(x), m = acosfGlibc(x); if (memcmp(&a, &m, 4)) bad++; };
    for (uint32_t i = 0; i < 32768u; i++) { probe(i * 32537u); probe(0x80000000u | (i * 32537u)); }       // 0 .. 0x3F80xxxx: every exponent of [0, 1], both signs
    const uint32_t centres[4] = {0x00000000u, 0x32800000u /* 2^-26 */, 0x3F000000u /* 0.5 */, 0x3F800000u /* 1 */};
    for (uint32_t c : centres)
        for (uint32_t k = 0; k < 2048u; k++) { probe(c + k); probe((c - k) & 0x7FFFFFFFu); probe(0x80000000u | (c + k)); probe(0x80000000u | ((c - k) & 0x7FFFFFFFu)); }
    v = bad ? 1 : 0;
    int expected = -1;
    if (g_hostAcos.compare_exchange_strong(expected, v) && bad)
        fprintf(stderr, "[sdfhip] the running libm's acosf differs from the device's restatement of glibc's (%llu of 81 920 probes): arc cosines are taken on the host\n", (unsigned long long)bad);
    return g_hostAcos.load(std::memory_order_acquire) != 0;
}
}  // namespace sdfhip

namespace sdfhip { void loadKernelsCtxMesh(); void loadKernelsBvh(); void loadKernelsOctreeBuild(); void loadKernelsOctreeContinuity(); void loadKernelsOctreeQuery(); void loadKernelsOctreeLattice();
                   void loadKernelsBlocks(); void loadKernelsExactBuild(); void loadKernelsExactQuery(); void loadKernelsMulti(); }
using namespace sdfhip;

extern "C" {

const char* sdfhip_last_error(void) { return g_lastError.c_str(); }

// test hook: 1 forces the host's acosf for the corner angles, 0 the device's, -1 lets the next mesh decide by the self-check again
void sdfhip_test_set_host_acos(int mode) { g_hostAcos.store(mode < 0 ? -1 : (mode ? 1 : 0), std::memory_order_release); }
// test hooks: acosfGlibc against the running libm.  Host compilation on the bit patterns first, first + stride, ... (count of them; the
// values outside [-1, 1] are skipped), on `threads` host threads; and the DEVICE compilation on the same patterns.
uint64_t sdfhip_test_acosf_mismatches(uint32_t first_bits, uint32_t stride, uint64_t count, int threads) {
    if (threads < 1) threads = 1;
    std::vector<uint64_t> bad((size_t)threads, 0);
    auto work = [&](int q) {
        for (uint64_t i = (uint64_t)q; i < count; i += (uint64_t)threads) {
            const uint32_t b = first_bits + (uint32_t)(i * stride);
            float x; memcpy(&x, &b, 4);
            if (!(std::fabs(x) <= 1.0f)) continue;
            const float a = ::acosf(x), m = acosfGlibc(x);
            if (memcmp(&a, &m, 4)) bad[(size_t)q]++;
        }
    };
    std::vector<std::thread> th;
    for (int q = 1; q < threads; q++) th.emplace_back(work, q);
    work(0);
    for (std::thread& t : th) t.join();
    uint64_t total = 0; for (uint64_t v : bad) total += v;
    return total;
}
int sdfhip_test_acosf_device(sdfhip_ctx* ctx, uint32_t first_bits, uint32_t stride, uint32_t count, uint64_t* out_mismatches) {
    SDF_REQUIRE(ctx && out_mismatches, "null argument");
    SDF_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    DevBuf<float> d; SDF_TRY(d.reserve(count ? count : 1));
    k_test_acosf<<<gridFor(count, 256), 256, 0, st>>>(first_bits, stride, count, d.p);
    std::vector<float> h(count);
    SDF_HIP_CHECK(hipMemcpyAsync(h.data(), d.p, sizeof(float) * (size_t)count, hipMemcpyDeviceToHost, st));
    SDF_HIP_CHECK(hipStreamSynchronize(st));
    uint64_t bad = 0;
    for (uint32_t i = 0; i < count; i++) {
        const uint32_t b = first_bits + i * stride;
        float x; memcpy(&x, &b, 4);
        if (!(std::fabs(x) <= 1.0f)) continue;
        const float a = ::acosf(x);
        if (memcmp(&a, &h[i], 4)) bad++;
    }
    *out_mismatches = bad;
    return SDFHIP_OK;
}
int sdfhip_interpolation_flavour(void) {
#ifdef SDFHIP_ENOKI_ORDER
    return 1;
#else
    return 0;
#endif
}
const char* sdfhip_version(void) { return "sdfhip 0.1 (gfx950)"; }
void sdfhip_abi_sizes(uint64_t out[3]) { out[0] = sizeof(sdfhip_octree_info); out[1] = sizeof(sdfhip_octree_params); out[2] = sizeof(sdfhip_exact_info); }

int sdfhip_ctx_create(int device_id, void* stream, int stream_mode, sdfhip_ctx** out) {
    SDF_API_BEGIN
    SDF_REQUIRE(out != nullptr, "out is NULL");
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        setError("no HIP device available (%s); libsdfhip has no CPU fallback", e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
        return SDFHIP_E_NO_DEVICE;
    }
    SDF_REQUIRE(device_id >= 0 && device_id < count, "device_id out of range");
    SDF_HIP_CHECK(hipSetDevice(device_id));
    sdfhip_ctx* c = new sdfhip_ctx();
    c->device = device_id;
    SDF_HIP_CHECK(hipGetDeviceProperties(&c->prop, device_id));
    if (stream_mode == SDFHIP_STREAM_BORROWED) { c->stream = (hipStream_t)stream; c->ownsStream = false; }
    else { SDF_HIP_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)); c->ownsStream = true; }
    BigBlockCache::get().addRef(c->device, c->stream);
    {   // every translation unit's code object onto this device now, once per device and process
        static std::mutex m; static bool loaded[64] = {};
        std::lock_guard<std::mutex> g(m);
        if (device_id < 64 && !loaded[device_id]) {
            loaded[device_id] = true;
            loadKernelsCtxMesh(); loadKernelsBvh(); loadKernelsOctreeBuild(); loadKernelsOctreeContinuity(); loadKernelsOctreeQuery(); loadKernelsOctreeLattice();
            loadKernelsBlocks(); loadKernelsExactBuild(); loadKernelsExactQuery(); loadKernelsMulti();
        }
    }
    *out = c;
    return SDFHIP_OK;
    SDF_API_END
}

int sdfhip_ctx_destroy(sdfhip_ctx* ctx) {
    SDF_API_BEGIN
    if (!ctx) return SDFHIP_OK;
    for (hipStream_t& s : ctx->bvhSide) if (s) { (void)hipStreamDestroy(s); s = nullptr; }
    for (hipStream_t& s : ctx->upSide) if (s) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); s = nullptr; }
    if (ctx->downStream) { (void)hipStreamSynchronize(ctx->downStream); (void)hipStreamDestroy(ctx->downStream); ctx->downStream = nullptr; }
    (void)hipSetDevice(ctx->device); (void)hipStreamSynchronize(ctx->stream);
    // the cached blocks of this context's stream, once its last context goes (a borrowed stream can serve several contexts)
    if (BigBlockCache::get().dropRef(ctx->device, ctx->stream) <= 0) BigBlockCache::get().trimStream(ctx->device, ctx->stream);
    if (ctx->ownsStream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return SDFHIP_OK;
    SDF_API_END
}

int sdfhip_ctx_trim(sdfhip_ctx* ctx, uint64_t keep_bytes) {
    SDF_API_BEGIN
    SDF_REQUIRE(ctx != nullptr, "ctx is NULL");
    SDF_HIP_CHECK(hipSetDevice(ctx->device));
    SDF_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    BigBlockCache::get().trimTo(ctx->device, ctx->stream, (size_t)keep_bytes);
    {   // the context's own grow-only scratch: the nearest search's candidate lists and the host-pointer staging buffers
        std::lock_guard<std::recursive_mutex> building(ctx->buildLock);
        if (ctx->nearScratch.bytes() > keep_bytes) ctx->nearScratch.release();
        ctx->contCaps[0] = ctx->contCaps[1] = ctx->contCaps[2] = 0;         // the next CONTINUITY build sizes itself from scratch
        std::lock_guard<std::mutex> staging(ctx->stage.lock);
        if (4 * (ctx->stage.pts.n + ctx->stage.dist.n + ctx->stage.grad.n + ctx->stage.ids.n) > keep_bytes) { ctx->stage.pts.release(); ctx->stage.dist.release(); ctx->stage.grad.release(); ctx->stage.ids.release(); }
    }
    return SDFHIP_OK;
    SDF_API_END
}

int sdfhip_ctx_cached_bytes(sdfhip_ctx* ctx, uint64_t* out_bytes) {
    SDF_API_BEGIN
    SDF_REQUIRE(ctx != nullptr && out_bytes != nullptr, "NULL argument");
    std::lock_guard<std::recursive_mutex> building(ctx->buildLock);
    *out_bytes = BigBlockCache::get().cachedBytes(ctx->device, ctx->stream) + ctx->nearScratch.bytes()
               + 4 * (ctx->stage.pts.n + ctx->stage.dist.n + ctx->stage.grad.n + ctx->stage.ids.n);
    return SDFHIP_OK;
    SDF_API_END
}

int sdfhip_ctx_synchronize(sdfhip_ctx* ctx) {
    SDF_API_BEGIN
    SDF_REQUIRE(ctx != nullptr, "ctx is NULL");
    SDF_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return SDFHIP_OK;
    SDF_API_END
}

void* sdfhip_ctx_stream(sdfhip_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int sdfhip_ctx_set_exchange(sdfhip_ctx* ctx, const sdfhip_exchange* x) {
    SDF_API_BEGIN
    SDF_REQUIRE(ctx, "null context");
    if (!x || x->world < 1) { ctx->exchange = sdfhip_exchange{}; return SDFHIP_OK; }
    SDF_REQUIRE(x->acquire && x->all_reduce_sum, "exchange without callbacks");
    SDF_REQUIRE(x->rank >= 0 && x->rank < x->world, "exchange rank outside [0, world)");
    ctx->exchange = *x;
    return SDFHIP_OK;
    SDF_API_END
}

int sdfhip_mesh_create(sdfhip_ctx* ctx, const float* xyz, uint32_t nv, const uint32_t* indices, uint32_t nt, sdfhip_mesh** out) {
    SDF_API_BEGIN
    return sdfhip_mesh_create_ex(ctx, xyz, nv, indices, nt, nullptr, out);
    SDF_API_END
}

int sdfhip_mesh_create_ex(sdfhip_ctx* ctx, const float* xyz, uint32_t nv, const uint32_t* indices, uint32_t nt, const float* bbox6, sdfhip_mesh** out) {
    SDF_API_BEGIN
    return sdfhip_mesh_create_opt(ctx, xyz, nv, indices, nt, bbox6, 0u, out);
    SDF_API_END
}

int sdfhip_mesh_create_opt(sdfhip_ctx* ctx, const float* xyz, uint32_t nv, const uint32_t* indices, uint32_t nt, const float* bbox6, uint32_t flags, sdfhip_mesh** out) {
    SDF_API_BEGIN
    SDF_REQUIRE(ctx && xyz && indices && out, "NULL argument");
    std::lock_guard<std::recursive_mutex> building(ctx->buildLock);
    SDF_REQUIRE(nv >= 3 && nt >= 1, "empty mesh");
    SDF_REQUIRE((uint64_t)nt * 3 < (1ull << 32), "too many triangles");
    // (the arrays are checked on the DEVICE, below: as host loops in front of the upload — the largest index, the largest |coordinate|, a NaN
    // test — they were 2 - 6 ms of pure latency per 1.31 M-triangle mesh)
    SDF_HIP_CHECK(hipSetDevice(ctx->device));
    std::unique_ptr<sdfhip_mesh> owner(new sdfhip_mesh());      // released on success only: every early return below frees it
    sdfhip_mesh* m = owner.get();
    m->ctx = ctx; m->numVertices = nv; m->numTriangles = nt;
    // host copies of the arrays are what the HOST planner reads: with the tree built on the device (the default) they are fetched back from
    // the device only if that build hands over (bvh.hip, hostArrays) — 24 MB of copies and page faults less per 1.31 M-triangle mesh
    if (!sdfhip::bvhBuildOnDevice()) { m->hVerts.assign(xyz, xyz + 3ull * nv); m->hIdx.assign(indices, indices + 3ull * nt); }
    hipStream_t st = ctx->stream;
    const uint32_t nhe = 3 * nt;
    AllocScope allocScope(st);       // device buffers of this call come from the stream-ordered pool
    int rc = SDFHIP_OK;
    auto fail = [&](int code) { return code; };
    if ((rc = m->dVerts.reserve(3ull * nv)) || (rc = m->dIdx.reserve(nhe)) || (rc = m->dTri.reserve((size_t)TD_FLOATS * nt)) || (rc = m->dFrames.reserve((size_t)FRAME_FLOATS * nt))) return fail(rc);
    SDF_HIP_CHECK(hipMemcpyAsync(m->dVerts.p, xyz, sizeof(float) * 3ull * nv, hipMemcpyHostToDevice, st));
    SDF_HIP_CHECK(hipMemcpyAsync(m->dIdx.p, indices, sizeof(uint32_t) * nhe, hipMemcpyHostToDevice, st));
    {   // nothing below may index the vertices before the indices are known to be in range; NaN / infinite coordinates make the
        // reference's std::sort comparator inconsistent (undefined behaviour): rejected
        DevBuf<uint32_t> chk;
        if ((rc = chk.reserve(2))) return fail(rc);
        SDF_HIP_CHECK(hipMemsetAsync(chk.p, 0, 8, st));
        k_mesh_validate<<<1024, 256, 0, st>>>(m->dVerts.p, 3ull * nv, m->dIdx.p, (uint64_t)nhe, chk.p);
        uint32_t h[2] = {0, 0};
        if ((rc = readBackWords(st, chk.p, nullptr, 2, h))) return fail(rc);
        SDF_REQUIRE(h[0] < nv, "triangle index out of range");
        SDF_REQUIRE(h[1] <= 0x7F7FFFFFu, "non-finite vertex coordinate");
        memcpy(&m->bvhCoordScale, &h[1], 4);          // max |coordinate|: bounds the fp32 rounding of the BVH's sphere centres
    }
    if (flags & SDFHIP_MESH_PLAN_BVH_EARLY) sdfhip::startEarlyBvhPlan(m);      // the planner (host threads) runs under everything below
    k_triangle_frames<<<gridFor(nt, 256), 256, 0, st>>>(m->dVerts.p, m->dIdx.p, nt, m->dTri.p);
    DevBuf<float> cornerAngle;
    if ((rc = cornerAngle.reserve(nhe))) return fail(rc);
    const bool hostAcos = hostAcosNeeded();
    k_corner_cos<<<gridFor(nhe, 256), 256, 0, st>>>(m->dVerts.p, m->dIdx.p, nhe, cornerAngle.p, !hostAcos);
    std::vector<float> hAngle;
    if (hostAcos) {
        hAngle.resize(nhe);
        SDF_HIP_CHECK(hipMemcpyAsync(hAngle.data(), cornerAngle.p, sizeof(float) * nhe, hipMemcpyDeviceToHost, st));
        SDF_HIP_CHECK(hipStreamSynchronize(st));
    }
    if ((rc = packFrames(st, m->dTri.p, nt, m->dFrames.p))) return fail(rc);

    DevBuf<uint64_t> eKey, eKeyS; DevBuf<uint32_t> vKey, vKeyS, val, valS, valS2, counter; DevBuf<float> vnormal; DevBuf<unsigned char> tmp;
    if ((rc = eKey.reserve(nhe)) || (rc = eKeyS.reserve(nhe)) || (rc = vKey.reserve(nhe)) || (rc = vKeyS.reserve(nhe)) ||
        (rc = val.reserve(nhe)) || (rc = valS.reserve(nhe)) || (rc = valS2.reserve(nhe)) || (rc = counter.reserve(1)) ||
        (rc = vnormal.reserve(3ull * nv))) return fail(rc);
    k_halfedge_keys<<<gridFor(nhe, 256), 256, 0, st>>>(m->dIdx.p, nhe, eKey.p, vKey.p, val.p);
    size_t tb1 = 0, tb2 = 0;
    unsigned vbits = 1; while (vbits < 32u && (1ull << vbits) < (unsigned long long)nv) vbits++;      // bits of a vertex index: the sorts skip the passes over bits that are zero in every key
    SDF_HIP_CHECK(devSortPairs(nullptr, tb1, eKey.p, eKeyS.p, val.p, valS.p, (size_t)nhe, 0, 32u + vbits, st));
    SDF_HIP_CHECK(devSortPairs(nullptr, tb2, vKey.p, vKeyS.p, val.p, valS2.p, (size_t)nhe, 0, vbits, st));
    if ((rc = tmp.reserve(tb1 > tb2 ? tb1 : tb2))) return fail(rc);
    SDF_HIP_CHECK(devSortPairs(tmp.p, tb1, eKey.p, eKeyS.p, val.p, valS.p, (size_t)nhe, 0, 32u + vbits, st));
    SDF_HIP_CHECK(hipMemsetAsync(counter.p, 0, sizeof(uint32_t), st));
    DevBuf<uint64_t> openKey; DevBuf<uint32_t> openHe;
    if (bbox6 && ((rc = openKey.reserve(nhe)) || (rc = openHe.reserve(nhe)))) return fail(rc);
    k_edge_pair<<<gridFor(nhe, 256), 256, 0, st>>>(eKeyS.p, valS.p, nhe, m->dTri.p, counter.p, bbox6 ? openKey.p : nullptr, bbox6 ? openHe.p : nullptr);
    SDF_HIP_CHECK(devSortPairs(tmp.p, tb2, vKey.p, vKeyS.p, val.p, valS2.p, (size_t)nhe, 0, vbits, st));
    SDF_HIP_CHECK(hipMemsetAsync(vnormal.p, 0, sizeof(float) * 3ull * nv, st));
    if (hostAcos) {   // the arc cosines on host threads while the device sorts (see k_corner_cos)
        unsigned parts = (unsigned)(nhe / 65536u); const unsigned hc = std::thread::hardware_concurrency();
        if (parts > (hc ? hc : 1u)) parts = hc ? hc : 1u;
        if (parts > 64u) parts = 64u;
        if (parts < 1u) parts = 1u;
        float* a = hAngle.data();
        auto work = [a](uint64_t i0, uint64_t i1) { for (uint64_t i = i0; i < i1; i++) a[i] = std::acos(a[i]); };
        std::vector<std::thread> th;
        for (unsigned q = 1; q < parts; q++) th.emplace_back(work, (uint64_t)nhe * q / parts, (uint64_t)nhe * (q + 1) / parts);
        work(0, (uint64_t)nhe / parts);
        for (std::thread& t : th) t.join();
        SDF_HIP_CHECK(hipMemcpyAsync(cornerAngle.p, hAngle.data(), sizeof(float) * nhe, hipMemcpyHostToDevice, st));
    }
    k_vertex_normal_sum<<<gridFor(nhe, 256), 256, 0, st>>>(vKeyS.p, valS2.p, nhe, cornerAngle.p, m->dTri.p, vnormal.p);
    SDF_HIP_CHECK(hipGetLastError());
    SDF_HIP_CHECK(hipMemcpyAsync(&m->unmatchedEdges, counter.p, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    SDF_HIP_CHECK(hipStreamSynchronize(st));
    if (bbox6 && m->unmatchedEdges > 0) {
        const uint32_t no = m->unmatchedEdges;
        static const bool hostWeld = getenv("SDFHIP_WELD") && !strcmp(getenv("SDFHIP_WELD"), "host");       // the round-1..5 planner (std::map based), kept as a cross-check
        if (!hostWeld) {
            double hs = 0.0;
            if ((rc = weldSeamsDevice(st, m->dVerts.p, nv, bbox6, openKey.p, openHe.p, no, m->dTri.p, vnormal.p, &m->weldedEdges, &hs))) return fail(rc);
            if (getenv("SDFHIP_TIMING")) fprintf(stderr, "[sdfhip] seam welding: %u open edges, %u half-edges welded, union-find on the host %.2f ms\n", no, m->weldedEdges, 1e3 * hs);
        } else {
        std::vector<uint64_t> hKey(no); std::vector<uint32_t> hHe(no); std::vector<float> hVn(3ull * nv);
        SDF_HIP_CHECK(hipMemcpyAsync(hKey.data(), openKey.p, sizeof(uint64_t) * no, hipMemcpyDeviceToHost, st));
        SDF_HIP_CHECK(hipMemcpyAsync(hHe.data(), openHe.p, sizeof(uint32_t) * no, hipMemcpyDeviceToHost, st));
        SDF_HIP_CHECK(hipMemcpyAsync(hVn.data(), vnormal.p, sizeof(float) * 3ull * nv, hipMemcpyDeviceToHost, st));
        SDF_HIP_CHECK(hipStreamSynchronize(st));
        std::vector<uint32_t> pairs;
        weldSeams(xyz, bbox6, hKey, hHe, pairs, hVn.data());
        m->weldedEdges = (uint32_t)pairs.size();       // half-edges that found a partner = 2 per welded edge
        if (!pairs.empty()) {
            DevBuf<uint32_t> dPairs;
            if ((rc = dPairs.reserve(pairs.size()))) return fail(rc);
            SDF_HIP_CHECK(hipMemcpyAsync(dPairs.p, pairs.data(), sizeof(uint32_t) * pairs.size(), hipMemcpyHostToDevice, st));
            k_weld_edges<<<gridFor(pairs.size() / 2, 256), 256, 0, st>>>(dPairs.p, (uint32_t)(pairs.size() / 2), m->dTri.p);
            SDF_HIP_CHECK(hipStreamSynchronize(st));
        }
        SDF_HIP_CHECK(hipMemcpyAsync(vnormal.p, hVn.data(), sizeof(float) * 3ull * nv, hipMemcpyHostToDevice, st));
        SDF_HIP_CHECK(hipStreamSynchronize(st));
        }
    }
    k_vertex_normal_apply<<<gridFor(nhe, 256), 256, 0, st>>>(m->dIdx.p, nhe, vnormal.p, m->dTri.p);
    SDF_HIP_CHECK(hipGetLastError());
    SDF_HIP_CHECK(hipStreamSynchronize(st));
    *out = owner.release();
    return SDFHIP_OK;
    SDF_API_END
}

int sdfhip_mesh_destroy(sdfhip_mesh* mesh) { delete mesh; return SDFHIP_OK; }

int sdfhip_mesh_edge_stats(sdfhip_mesh* mesh, uint32_t* unmatched_edges, uint32_t* welded_half_edges) {
    SDF_API_BEGIN
    SDF_REQUIRE(mesh != nullptr, "mesh is NULL");
    if (unmatched_edges) *unmatched_edges = mesh->unmatchedEdges;
    if (welded_half_edges) *welded_half_edges = mesh->weldedEdges;
    return SDFHIP_OK;
    SDF_API_END
}

int sdfhip_mesh_triangle_data(sdfhip_mesh* mesh, float* out_host) {
    SDF_API_BEGIN
    SDF_REQUIRE(mesh && out_host, "NULL argument");
    SDF_HIP_CHECK(hipMemcpyAsync(out_host, mesh->dTri.p, sizeof(float) * TD_FLOATS * (size_t)mesh->numTriangles, hipMemcpyDeviceToHost, mesh->ctx->stream));
    SDF_HIP_CHECK(hipStreamSynchronize(mesh->ctx->stream));
    return SDFHIP_OK;
    SDF_API_END
}

}  // extern "C"

// (sdfhip_ctx_create: the runtime loads a translation unit's code object on the first use of one of its kernels — milliseconds that would
// otherwise land in the first build or the first query of a process)
namespace sdfhip { void loadKernelsCtxMesh() { hipFuncAttributes a; (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_triangle_frames)); (void)hipGetLastError(); } }
