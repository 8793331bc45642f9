// Leaf-driven lattice evaluation: the two column kernels of octree_query.hip's lattice plan (see there for the plan, the tables and the
// launch shapes).  PRODUCT code.  A translation unit of its own because of ONE compiler flag: everything else is built with
// -fno-slp-vectorize (packed fp32 multiplies / adds cost two issue slots and register-pair moves on gfx950), but these kernels are faster
// with the SLP vectoriser — k_lattice_columns is separable Horner on FMAs, which pack at full rate (v_pk_fma_f32), and the reference-order
// kernel's 222 registers schedule better with it (measured: 0.111 / 0.317 ms with, 0.122 / 0.345 ms without, 256^3 value + gradient).
// Compile with -ffp-contract=off.
#include "octree_internal.h"
#include "dev_tricubic.h"

namespace sdfhip {

template <bool GRAD>
__global__ void __launch_bounds__(256) k_lattice_columns(const float* __restrict__ coef, const float* __restrict__ F, const uint16_t* __restrict__ ranges,
                                                         const uint4* __restrict__ desc, const uint32_t* __restrict__ sortedLeaf, uint32_t waves, LatCols C,
                                                         uint32_t nx, uint32_t ny, float* __restrict__ dist, float* __restrict__ grad) {
    const uint32_t w = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + (threadIdx.x >> 6));
    if (w >= waves) return;
    const uint4 D = desc[w];
    const uint32_t level = __builtin_amdgcn_readfirstlane(D.z);
    const uint32_t mx = C.mx[level], my = C.my[level];
    const float scale = __uint_as_float((127u + level) << 23);               // 2^level
    const uint32_t wInGroup = __builtin_amdgcn_readfirstlane(D.w), lane = threadIdx.x & 63u, cols = mx * my;
    uint32_t rel, lx, ly;
    if (cols <= 32u) {
        // k whole leaves per wave, lanes x-fastest ACROSS the leaves: where the leaves adjoin in x (the list is sorted that way)
        // consecutive lanes write consecutive addresses
        const uint32_t k = 64u / cols, rowLen = k * mx;
        ly = lane / rowLen;
        const uint32_t rem = lane - ly * rowLen, li = rem / mx;
        lx = rem - li * mx;
        rel = wInGroup * k + li;
        if (ly >= my) return;
    } else {
        const uint32_t tid = wInGroup * 64u + lane;
        rel = tid / cols;
        const uint32_t col = tid - rel * cols;
        ly = col / mx; lx = col - ly * mx;
    }
    if (rel >= D.y) return;
    const uint32_t leaf = sortedLeaf[D.x + rel];
    const uint32_t* r32 = reinterpret_cast<const uint32_t*>(ranges + 6 * (size_t)leaf);
    const uint32_t rx = r32[0], ry = r32[1], rz = r32[2];
    const uint32_t x = (rx & 0xFFFFu) + lx, y = (ry & 0xFFFFu) + ly, z0 = rz & 0xFFFFu, z1 = rz >> 16;
    if (x >= (rx >> 16) || y >= (ry >> 16) || z0 >= z1) return;
    const float* Fz = F + nx + ny;
    float tzNext = Fz[z0];
    const float tx = F[x] * scale, ty = F[nx + y] * scale;
    const float fx = tx - floorf(tx), fy = ty - floorf(ty);
    const float4* src = reinterpret_cast<const float4*>(coef + 64ull * leaf);
    float yv[4], ygx[4], ygy[4];
#pragma unroll
    for (int k = 3; k >= 0; k--) {
        float v = 0.f, gx = 0.f, gy = 0.f;
#pragma unroll
        for (int j = 3; j >= 0; j--) {
            const float4 q = src[j + 4 * k];
            const float px = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(q.w, fx, q.z), fx, q.y), fx, q.x);
            if (GRAD) {
                const float dx = __builtin_fmaf(__builtin_fmaf(3.f * q.w, fx, 2.f * q.z), fx, q.y);
                gy = __builtin_fmaf(gy, fy, v);
                gx = __builtin_fmaf(gx, fy, dx);
            }
            v = __builtin_fmaf(v, fy, px);
        }
        yv[k] = v; ygx[k] = gx; ygy[k] = gy;
    }
    size_t at = ((size_t)z0 * ny + y) * nx + x;
    const size_t plane = (size_t)nx * ny;
    for (uint32_t z = z0; z < z1; z++, at += plane) {
        const float tz = tzNext * scale, fz = tz - floorf(tz);
        tzNext = Fz[z + 1];                         // (the table has one spare entry) in flight while this point is evaluated and stored
        float v = 0.f, gx = 0.f, gy = 0.f, gz = 0.f;
#pragma unroll
        for (int k = 3; k >= 0; k--) {
            if (GRAD) { gz = __builtin_fmaf(gz, fz, v); gx = __builtin_fmaf(gx, fz, ygx[k]); gy = __builtin_fmaf(gy, fz, ygy[k]); }
            v = __builtin_fmaf(v, fz, yv[k]);
        }
        dist[at] = v;
        if (GRAD) { const F3 g = normalize(F3{gx, gy, gz}); grad[3 * at] = g.x; grad[3 * at + 1] = g.y; grad[3 * at + 2] = g.z; }
    }
}

// EVAL_EXACT by columns.  The reference's literal order (dev_tricubic.h: every term is c * x..x * y..y * z..z from left to right, the terms
// added one after the other) cannot be contracted — but the part of every term that precedes its z factors, c * x^i * y^j (times the
// derivative's integer factor), is the same for all points of a column.  Those prefixes are computed once per column; a point then
// costs the z multiplications and the additions only: about 500 flop for value + gradient instead of 1 100, with results that are
// the point kernel's bit for bit (same operations in the same order on the same operands).
template <int EX, int EY, int EZ>
SDF_DEV void columnPrefixes(const float4* __restrict__ src, float fx, float fy, float (&P)[64]) {
#pragma unroll
    for (int n = 0; n < 64; n++) {
        const int i = n & 3, j = (n >> 2) & 3, k = n >> 4;
        const int fac = (EX ? i : 1) * (EY ? j : 1) * (EZ ? k : 1);
        if (fac == 0) { P[n] = 0.f; continue; }
        const float4 q = src[n >> 2];
        const float c = (n & 3) == 0 ? q.x : ((n & 3) == 1 ? q.y : ((n & 3) == 2 ? q.z : q.w));
        float t = (EX || EY || EZ) ? (float)fac * c : c;
#pragma unroll
        for (int a = 0; a < i - EX; a++) t = t * fx;
#pragma unroll
        for (int a = 0; a < j - EY; a++) t = t * fy;
        P[n] = t;
    }
}
template <int EX, int EY, int EZ>
SDF_DEV float columnPoint(const float (&P)[64], float fz) {
    float acc = 0.0f;
    bool first = true;
#pragma unroll
    for (int n = 0; n < 64; n++) {
        const int i = n & 3, j = (n >> 2) & 3, k = n >> 4;
        const int fac = (EX ? i : 1) * (EY ? j : 1) * (EZ ? k : 1);
        if (fac == 0) continue;
        float t = P[n];
#pragma unroll
        for (int a = 0; a < k - EZ; a++) t = t * fz;
        if ((EX || EY || EZ) && first) { acc = t; first = false; } else acc = acc + t;       // tricubicValueExact starts from 0.0f + t, the derivatives from t
    }
    return acc;
}
template <bool GRAD>
__global__ void __launch_bounds__(256) k_lattice_columns_exact(const float* __restrict__ coef, const float* __restrict__ F, const uint16_t* __restrict__ ranges,
                                                               const uint4* __restrict__ desc, const uint32_t* __restrict__ sortedLeaf, uint32_t waves, LatCols C,
                                                               uint32_t nx, uint32_t ny, float* __restrict__ dist, float* __restrict__ grad) {
    const uint32_t w = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + (threadIdx.x >> 6));
    if (w >= waves) return;
    const uint4 D = desc[w];
    const uint32_t level = __builtin_amdgcn_readfirstlane(D.z);
    const uint32_t mx = C.mx[level], my = C.my[level];
    const float scale = __uint_as_float((127u + level) << 23);
    const uint32_t wInGroup = __builtin_amdgcn_readfirstlane(D.w), lane = threadIdx.x & 63u, cols = mx * my;
    uint32_t rel, lx, ly;
    if (cols <= 32u) {
        const uint32_t k = 64u / cols, rowLen = k * mx;
        ly = lane / rowLen;
        const uint32_t rem = lane - ly * rowLen, li = rem / mx;
        lx = rem - li * mx;
        rel = wInGroup * k + li;
        if (ly >= my) return;
    } else {
        const uint32_t tid = wInGroup * 64u + lane;
        rel = tid / cols;
        const uint32_t col = tid - rel * cols;
        ly = col / mx; lx = col - ly * mx;
    }
    if (rel >= D.y) return;
    const uint32_t leaf = sortedLeaf[D.x + rel];
    const uint32_t* r32 = reinterpret_cast<const uint32_t*>(ranges + 6 * (size_t)leaf);
    const uint32_t rx = r32[0], ry = r32[1], rz = r32[2];
    const uint32_t x = (rx & 0xFFFFu) + lx, y = (ry & 0xFFFFu) + ly, z0 = rz & 0xFFFFu, z1 = rz >> 16;
    if (x >= (rx >> 16) || y >= (ry >> 16) || z0 >= z1) return;
    const float* Fz = F + nx + ny;
    const float tx = F[x] * scale, ty = F[nx + y] * scale;
    const float fx = tx - floorf(tx), fy = ty - floorf(ty);
    const float4* src = reinterpret_cast<const float4*>(coef + 64ull * leaf);
    const size_t plane = (size_t)nx * ny, at0 = ((size_t)z0 * ny + y) * nx + x;
#ifdef SDFHIP_ENOKI_ORDER
    {   // the Enoki flavour's value multiplies its power vectors by z BEFORE the coefficients: nothing of a term but the vectors is shared along z
        float cf[64];
#pragma unroll
        for (int n = 0; n < 16; n++) { const float4 q = src[n]; cf[4 * n] = q.x; cf[4 * n + 1] = q.y; cf[4 * n + 2] = q.z; cf[4 * n + 3] = q.w; }
        size_t at = at0;
        for (uint32_t z = z0; z < z1; z++, at += plane) {
            const float tz = Fz[z] * scale, fz = tz - floorf(tz);
            dist[at] = tricubicValueEnoki([&](int n) { return cf[n]; }, F3{fx, fy, fz});
        }
    }
#else
    {
        float P[64];
        columnPrefixes<0, 0, 0>(src, fx, fy, P);
        size_t at = at0;
        for (uint32_t z = z0; z < z1; z++, at += plane) {
            const float tz = Fz[z] * scale, fz = tz - floorf(tz);
            dist[at] = columnPoint<0, 0, 0>(P, fz);
        }
    }
#endif
    if (GRAD) {
        float PX[64], PY[64], PZ[64];
        columnPrefixes<1, 0, 0>(src, fx, fy, PX); columnPrefixes<0, 1, 0>(src, fx, fy, PY); columnPrefixes<0, 0, 1>(src, fx, fy, PZ);
        size_t at = at0;
        for (uint32_t z = z0; z < z1; z++, at += plane) {
            const float tz = Fz[z] * scale, fz = tz - floorf(tz);
            const F3 g = normalize(F3{columnPoint<1, 0, 0>(PX, fz), columnPoint<0, 1, 0>(PY, fz), columnPoint<0, 0, 1>(PZ, fz)});
            grad[3 * at] = g.x; grad[3 * at + 1] = g.y; grad[3 * at + 2] = g.z;
        }
    }
}


int latticeColumnsLaunch(hipStream_t st, bool exact, const float* coef, const float* F, const uint16_t* ranges, const uint4* desc, const uint32_t* sortedLeaf, uint32_t waves,
                         const LatCols& C, uint32_t nx, uint32_t ny, float* d, float* g) {
    const unsigned grid = (unsigned)gridFor(waves, 4);
    if (exact) {
        if (g) k_lattice_columns_exact<true><<<grid, 256, 0, st>>>(coef, F, ranges, desc, sortedLeaf, waves, C, nx, ny, d, g);
        else k_lattice_columns_exact<false><<<grid, 256, 0, st>>>(coef, F, ranges, desc, sortedLeaf, waves, C, nx, ny, d, nullptr);
    } else {
        if (g) k_lattice_columns<true><<<grid, 256, 0, st>>>(coef, F, ranges, desc, sortedLeaf, waves, C, nx, ny, d, g);
        else k_lattice_columns<false><<<grid, 256, 0, st>>>(coef, F, ranges, desc, sortedLeaf, waves, C, nx, ny, d, nullptr);
    }
    return hipGetLastError() == hipSuccess ? SDFHIP_OK : SDFHIP_E_HIP;
}

}  // namespace sdfhip

// (sdfhip_ctx_create: the runtime loads a translation unit's code object on the first use of one of its kernels — milliseconds that would
// otherwise land in the first build or the first query of a process)
namespace sdfhip { void loadKernelsOctreeLattice() { hipFuncAttributes a; (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_lattice_columns<true>)); (void)hipGetLastError(); } }
