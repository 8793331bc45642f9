// 64x64 tricubic fit on the matrix cores (v_mfma_f32_32x32x2_f32).  PRODUCT code — independent of oracle/.
//
// coefficients[node][0..63] = M (64 x 64, integer) x scaled Hermite vector[node][0..63], i.e. the GEMM
// C[nodes x 64] = S[nodes x K] . Mt[K x 64] with K = 8 vertices x SLOTS values (SLOTS = 4 when the mixed derivatives are
// identically zero, as in the NO_CONTINUITY builder; 8 in general).  One wave owns a 32-node tile: the tile of S is
// staged through LDS with coalesced dwordx4 loads (row pitch K+1 floats -> conflict-free column reads), Mt lives in LDS
// ([k][j], lanes read consecutive j), and 2 x (K/2) MFMA steps accumulate the two 32-coefficient halves.
// Fragment layout of mfma_f32_32x32x2f32 (cdna_hip_programming.md section 3): A[i = lane&31][k = lane>>5],
// B[k = lane>>5][j = lane&31], C/D reg r -> row (r&3) + 8*(r>>2) + 4*(lane>>5), col lane&31.
//
// Numerics: an fp32 MFMA is a k-ordered fmaf chain (one rounding per step) while the reference rounds every product and
// every sum separately: neither is "the" exact product and the 64-term sums (entries up to 27, heavy cancellation) carry
// ~1e-5 of fp32 noise either way.  To keep the MFMA path from ADDING noise, every scaled input is split per node into
// hi + lo with hi = s rounded to a grid of 2^-11 * 2^ceil(log2 max|s|): hi/grid is an integer <= 2^11, the matrix
// entries are integers <= 27, so every product and every partial sum of the hi pass is EXACT in fp32 (< 2^24 grid
// units) whatever the accumulation order; the lo pass (|lo| <= 2^-12 max|s|) contributes rounding errors 2^-12 times
// smaller.  hi-pass + lo-pass is therefore the real-arithmetic product rounded once (~0.5 ulp): the remaining
// difference to the reference is the reference's own rounding noise.  Decisions that depend on the coefficients are
// still re-checked with the reference-ordered scalar fit when they are close to the threshold (octree_build.hip).
#pragma once
#include "dev_tricubic.h"

namespace sdfhip {

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SLOTS>
struct FitMfmaSmem {
    static constexpr int K = 8 * SLOTS;
    float mt[K][64];                 // Mt[k][j] = M[j][col(k)]
    float s[4][32][K + 1];           // per wave: 32 nodes x K scaled values (+1 pad)
    float grid[4][32];               // per node: power-of-two quantum of the hi part
};

template <int SLOTS>
SDF_DEV int fitColumn(int k) { return 8 * (k / SLOTS) + (k % SLOTS); }     // active column k -> column of the 64-wide system

// Block = 256 threads (4 waves), 128 nodes per block.  in: [n][8][SLOTS] floats; out: [n][64].
template <int SLOTS>
__global__ void __launch_bounds__(256) k_fit_mfma(const float* __restrict__ in, const float* __restrict__ nodeSizes, float nodeSizeAll, uint32_t n,
                                                  float* __restrict__ out) {
    constexpr int K = 8 * SLOTS;
    __shared__ FitMfmaSmem<SLOTS> sm;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int e = tid; e < K * 64; e += 256) {
        const int k = e >> 6, j = e & 63;
        sm.mt[k][j] = (float)fitCoefRuntime(j, fitColumn<SLOTS>(k));
    }
    const uint32_t base = blockIdx.x * 128u + (uint32_t)w * 32u;
    // stage the wave's tile: 32 nodes x K floats = 8*K float4s... (K/4 float4 per node)
    constexpr int V4 = K / 4;
    for (int e = lane; e < 32 * V4; e += 64) {
        const int r = e / V4, c4 = e - r * V4;
        const uint32_t node = base + (uint32_t)r;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        float ns = 1.f;
        if (node < n) { v = reinterpret_cast<const float4*>(in)[(size_t)node * V4 + c4]; ns = nodeSizes ? nodeSizes[node] : nodeSizeAll; }
        const float sq = ns * ns, cu = sq * ns;
        float vals[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int k = 4 * c4 + q, slot = k % SLOTS;
            const float sc = (slot == 0) ? 1.f : (slot < 4 ? ns : (slot < 7 ? sq : cu));
            sm.s[w][r][k] = vals[q] * sc;
        }
    }
    __syncthreads();
    if (lane < 32) {                 // per-node quantum: 2^(ceil(log2 max|s|) - 11)
        float mx = 0.f;
        for (int k = 0; k < K; k++) mx = fmaxf(mx, fabsf(sm.s[w][lane][k]));
        int e = 0; (void)frexpf(mx, &e);                 // mx = m * 2^e, m in [0.5, 1)
        sm.grid[w][lane] = (mx > 0.f && mx < INFINITY) ? ldexpf(1.0f, e - 11) : 1.0f;
    }
    __syncthreads();
    const int i = lane & 31, kh = lane >> 5;
    const float q = sm.grid[w][i], rq = 1.0f / q;
#pragma unroll
    for (int half = 0; half < 2; half++) {
        f32x16 accHi = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        f32x16 accLo = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll 4
        for (int k0 = 0; k0 < K; k0 += 2) {
            const float a = sm.s[w][i][k0 + kh];
            const float hi = rintf(a * rq) * q;          // exact: q is a power of two
            const float lo = a - hi;                     // exact
            const float b = sm.mt[k0 + kh][32 * half + i];
            accHi = __builtin_amdgcn_mfma_f32_32x32x2f32(hi, b, accHi, 0, 0, 0);
            accLo = __builtin_amdgcn_mfma_f32_32x32x2f32(lo, b, accLo, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * kh;
            const uint32_t node = base + (uint32_t)row;
            if (node < n) out[(size_t)node * 64 + 32 * half + i] = accHi[r] + accLo[r];
        }
    }
}

}  // namespace sdfhip
