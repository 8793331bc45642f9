// Building blocks exposed through the C ABI for parity tests.  PRODUCT code — independent of oracle/.
#include "sdfhip_internal.h"
#include "dev_tricubic.h"
#include "dev_gjk.h"
#include "dev_fit_mfma.h"

namespace sdfhip {

__global__ void __launch_bounds__(128) k_fit_exact(const float* __restrict__ in, const float* __restrict__ nodeSize, uint64_t n, float* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s[64], c[64];
#pragma unroll
    for (int k = 0; k < 64; k++) s[k] = in[64 * i + k];
    tricubicFit(s, nodeSize[i], c);
#pragma unroll
    for (int k = 0; k < 64; k++) out[64 * i + k] = c[k];
}

__global__ void k_is_near(const float* __restrict__ half, const float* __restrict__ radius8, const float* __restrict__ tri9, const float* __restrict__ thr, uint64_t n, uint8_t* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float r[8];
    for (int k = 0; k < 8; k++) r[k] = radius8[8 * i + k];
    const float* t = tri9 + 9 * i;
    out[i] = isNearMinimize(half[i], r, F3{t[0], t[1], t[2]}, F3{t[3], t[4], t[5]}, F3{t[6], t[7], t[8]}, thr[i]) ? 1 : 0;
}

// Counter calibration for the query kernel's access pattern (MI355X_MICROARCH.md: FETCH_SIZE is calibrated for coalesced 16-B/lane
// streaming reads only): one 256-byte block per lane, fetched like k_octree_query_coop fetches its coefficients; the blocks are chosen by
// the caller (a permutation -> every block exactly once -> the bytes that must cross the fabric are known exactly).
// The blocks are fetched COOPERATIVELY: sixteen lanes read one lane's 256-byte block as one contiguous segment (an
// instruction touches 8 cache lines instead of 64), the rows go through LDS to their owners, 16 source lanes at a time.
constexpr int COOP_ROW = 68;                 // floats per LDS row: 64 + 4 of padding (b128 accesses of consecutive rows fall on different banks)
__global__ void __launch_bounds__(256) k_gather_blocks_coop(const uint32_t* __restrict__ data, const uint32_t* __restrict__ block, uint64_t n, float* __restrict__ out) {
    __shared__ float s_rows[4][16 * COOP_ROW];
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, grp = lane >> 4, sub = lane & 15;
    const bool live = i < n;
    const uint32_t mine = live ? block[i] : 0u;
    float acc = 0.f;
    for (int c = 0; c < 4; c++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = 4 * r + grp, srcLane = 16 * c + row;
            const uint32_t blk = __shfl(mine, srcLane);
            const float4 v = reinterpret_cast<const float4*>(data + 64ull * blk)[sub];
            *reinterpret_cast<float4*>(&s_rows[w][row * COOP_ROW + 4 * sub]) = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (grp == c) {
            const float4* rowp = reinterpret_cast<const float4*>(&s_rows[w][sub * COOP_ROW]);
#pragma unroll
            for (int q = 0; q < 16; q++) { const float4 v = rowp[q]; acc += (v.x + v.y) + (v.z + v.w); }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    if (live) out[i] = acc;
}

// VALU issue-ceiling calibration: 8 independent chains of plain (unpacked) v_fma_f32 per lane - the instruction class the reference-ordered
// kernels are made of (-ffp-contract=off gives separate v_mul / v_add, same issue rate); inline assembly, because the compiler would pair
// the chains into v_pk_fma_f32 (two lanes' worth per instruction), which those kernels cannot use.
__global__ void __launch_bounds__(256) k_valu_peak(uint32_t iters, float* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    float a0 = 1.0f + 1e-7f * (float)(i & 7), a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    const float m = 0.9999999f, c = 1e-9f;
    for (uint32_t k = 0; k < iters; k++) {
        asm volatile("v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\t"
                     "v_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
    }
    out[i] = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
}

}  // namespace sdfhip

using namespace sdfhip;

extern "C" {

int sdfhip_test_gather_blocks(sdfhip_ctx* ctx, const uint32_t* dev_data, const uint32_t* dev_block_ids, uint64_t n, float* dev_out) {
    SDF_API_BEGIN
    SDF_REQUIRE(ctx && dev_data && dev_block_ids && dev_out, "NULL argument");
    if (n == 0) return SDFHIP_OK;
    SDF_HIP_CHECK(hipSetDevice(ctx->device));
    k_gather_blocks_coop<<<gridFor(n, 256), 256, 0, ctx->stream>>>(dev_data, dev_block_ids, n, dev_out);          // the load pattern of k_octree_query_coop
    SDF_HIP_CHECK(hipGetLastError());
    return SDFHIP_OK;
    SDF_API_END
}

int sdfhip_test_valu_peak(sdfhip_ctx* ctx, uint32_t blocks, uint32_t iters, float* dev_out) {
    SDF_API_BEGIN
    SDF_REQUIRE(ctx && dev_out && blocks > 0, "NULL argument");
    SDF_HIP_CHECK(hipSetDevice(ctx->device));
    k_valu_peak<<<blocks, 256, 0, ctx->stream>>>(iters, dev_out);
    SDF_HIP_CHECK(hipGetLastError());
    return SDFHIP_OK;
    SDF_API_END
}

int sdfhip_tricubic_fit(sdfhip_ctx* ctx, const float* values_8x8, const float* node_sizes, uint64_t n, float* out64, int fit_mode) {
    SDF_API_BEGIN
    SDF_REQUIRE(ctx && values_8x8 && node_sizes && out64, "NULL argument");
    SDF_REQUIRE(fit_mode == SDFHIP_FIT_EXACT || fit_mode == SDFHIP_FIT_MFMA, "unknown fit_mode");
    if (n == 0) return SDFHIP_OK;
    SDF_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    DevBuf<float> din, dns, dout;
    SDF_TRY(din.reserve(64 * n)); SDF_TRY(dns.reserve(n)); SDF_TRY(dout.reserve(64 * n));
    SDF_HIP_CHECK(hipMemcpyAsync(din.p, values_8x8, 256 * n, hipMemcpyHostToDevice, st));
    SDF_HIP_CHECK(hipMemcpyAsync(dns.p, node_sizes, 4 * n, hipMemcpyHostToDevice, st));
    if (fit_mode == SDFHIP_FIT_EXACT) k_fit_exact<<<gridFor(n, 128), 128, 0, st>>>(din.p, dns.p, n, dout.p);
    else k_fit_mfma<8><<<gridFor(n, 128), 256, 0, st>>>(din.p, dns.p, 0.f, (uint32_t)n, dout.p);
    SDF_HIP_CHECK(hipGetLastError());
    SDF_HIP_CHECK(hipMemcpyAsync(out64, dout.p, 256 * n, hipMemcpyDeviceToHost, st));
    SDF_HIP_CHECK(hipStreamSynchronize(st));
    return SDFHIP_OK;
    SDF_API_END
}

int sdfhip_is_near_minimize(sdfhip_ctx* ctx, const float* half, const float* radius8, const float* tri9, const float* thr, uint64_t n, uint8_t* out) {
    SDF_API_BEGIN
    SDF_REQUIRE(ctx && half && radius8 && tri9 && thr && out, "NULL argument");
    if (n == 0) return SDFHIP_OK;
    SDF_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    DevBuf<float> dh, dr, dt, dth; DevBuf<uint8_t> dout;
    SDF_TRY(dh.reserve(n)); SDF_TRY(dr.reserve(8 * n)); SDF_TRY(dt.reserve(9 * n)); SDF_TRY(dth.reserve(n)); SDF_TRY(dout.reserve(n));
    SDF_HIP_CHECK(hipMemcpyAsync(dh.p, half, 4 * n, hipMemcpyHostToDevice, st));
    SDF_HIP_CHECK(hipMemcpyAsync(dr.p, radius8, 32 * n, hipMemcpyHostToDevice, st));
    SDF_HIP_CHECK(hipMemcpyAsync(dt.p, tri9, 36 * n, hipMemcpyHostToDevice, st));
    SDF_HIP_CHECK(hipMemcpyAsync(dth.p, thr, 4 * n, hipMemcpyHostToDevice, st));
    k_is_near<<<gridFor(n, 256), 256, 0, st>>>(dh.p, dr.p, dt.p, dth.p, n, dout.p);
    SDF_HIP_CHECK(hipGetLastError());
    SDF_HIP_CHECK(hipMemcpyAsync(out, dout.p, n, hipMemcpyDeviceToHost, st));
    SDF_HIP_CHECK(hipStreamSynchronize(st));
    return SDFHIP_OK;
    SDF_API_END
}

}  // extern "C"

// (sdfhip_ctx_create: the runtime loads a translation unit's code object on the first use of one of its kernels — milliseconds that would
// otherwise land in the first build or the first query of a process)
namespace sdfhip { void loadKernelsBlocks() { hipFuncAttributes a; (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_fit_exact)); (void)hipGetLastError(); } }
