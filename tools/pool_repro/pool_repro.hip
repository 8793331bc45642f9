// Stand-alone probe (no libsdfhip): does the HIP stream-ordered pool hand out, unmap or recycle memory that queued work still uses when a
// LARGE hipMallocAsync arrives while earlier blocks of the same stream have been hipFreeAsync'ed but their kernels have not run yet?
// Pattern of a build: many medium blocks written by long kernels, freed in stream order, then one request larger than anything the
// pool holds; results of the earlier kernels are checked through a reduction that was enqueued BEFORE the free.
// Build: hipcc --offload-arch=gfx950 -O2 -o pool_repro pool_repro.hip
// Run:   ./pool_repro [iterations] [big MB] [threshold: 0 = default, 1 = keep all] [big block given back: 0 = hipFreeAsync, 1 = sync + hipFree, 2 = mixed]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)

__global__ void k_fill(uint32_t* p, size_t n, uint32_t tag, int spin) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t v = tag ^ (uint32_t)i;
    for (int s = 0; s < spin; s++) v = v * 1664525u + 1013904223u;          // keeps the kernel on the device for a while
    p[i] = v;
}
__global__ void k_check(const uint32_t* p, size_t n, uint32_t tag, int spin, unsigned long long* bad, unsigned long long* smp) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t v = tag ^ (uint32_t)i;
    for (int s = 0; s < spin; s++) v = v * 1664525u + 1013904223u;
    const uint32_t got = p[i];
    if (got != v) {
        atomicAdd(bad, 1ull);
        if (got == 0u) atomicAdd(smp, 1ull);
        const unsigned long long k = atomicAdd(smp + 1, 1ull);
        if (k < 2) { smp[2 + 3 * k] = i; smp[3 + 3 * k] = v; smp[4 + 3 * k] = got; }
    }
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 200;
    const size_t bigMB = argc > 2 ? (size_t)atoll(argv[2]) : 170;
    const int keep = argc > 3 ? atoi(argv[3]) : 1;
    const int freeMode = argc > 4 ? atoi(argv[4]) : 2;
    const int maxMB = argc > 6 ? atoi(argv[6]) : 30;          // medium blocks are 1 .. maxMB MB (0: 64 KB .. 1 MB)
    const int syncBeforeFree = argc > 7 ? atoi(argv[7]) : 0;  // 1: the stream is idle whenever hipFreeAsync is called
    const int plain = argc > 5 ? atoi(argv[5]) : 0;          // control: 1 = hipMalloc / hipFree (after a synchronisation) instead of the pool
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    if (keep) { hipMemPool_t pool; uint64_t all = UINT64_MAX; CK(hipDeviceGetDefaultMemPool(&pool, 0)); CK(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &all)); }
    unsigned long long* bad; CK(hipMalloc(&bad, 256)); CK(hipMemset(bad, 0, 256));      // [0] before any free, [1] blocks still owned after the big request, [2] the big block
    srand(1);
    unsigned long long total = 0; long overlaps = 0;
    for (int it = 0; it < iters; it++) {
        const int nb = 8 + rand() % 24;
        std::vector<uint32_t*> blk(nb); std::vector<size_t> cnt(nb);
        for (int b = 0; b < nb; b++) {
            cnt[b] = maxMB > 0 ? ((size_t)(1 + rand() % maxMB) << 20) / 4 : ((size_t)(1 + rand() % 16) << 16) / 4;
            if (plain) CK(hipMalloc((void**)&blk[b], 4 * cnt[b])); else CK(hipMallocAsync((void**)&blk[b], 4 * cnt[b], st));
            k_fill<<<(unsigned)((cnt[b] + 255) / 256), 256, 0, st>>>(blk[b], cnt[b], (uint32_t)(it * 131 + b), 200);
        }
        for (int a = 0; a < nb; a++) for (int b = a + 1; b < nb; b++) {            // host-side sanity: live blocks must not overlap
            const uintptr_t a0 = (uintptr_t)blk[a], a1 = a0 + 4 * cnt[a], b0 = (uintptr_t)blk[b], b1 = b0 + 4 * cnt[b];
            if (a0 < b1 && b0 < a1) { overlaps++; if (overlaps <= 5) printf("iteration %d: LIVE BLOCKS OVERLAP: [%p, +%zu) and [%p, +%zu)\n", it, (void*)blk[a], 4 * cnt[a], (void*)blk[b], 4 * cnt[b]); }
        }
        // consumers are ENQUEUED, then the blocks are freed in stream order, then the large request comes while all of that is still queued
        for (int b = 0; b < nb; b++) k_check<<<(unsigned)((cnt[b] + 255) / 256), 256, 0, st>>>(blk[b], cnt[b], (uint32_t)(it * 131 + b), 200, bad, bad + 8);
        if (plain || syncBeforeFree) CK(hipStreamSynchronize(st));
        for (int b = 0; b < nb; b += 2) { if (plain) CK(hipFree(blk[b])); else CK(hipFreeAsync(blk[b], st)); }
        uint32_t* big; const size_t bigCnt = ((bigMB + (size_t)(rand() % 64)) << 20) / 4;
        if (plain) CK(hipMalloc((void**)&big, 4 * bigCnt)); else CK(hipMallocAsync((void**)&big, 4 * bigCnt, st));
        k_fill<<<(unsigned)((bigCnt + 255) / 256), 256, 0, st>>>(big, bigCnt, 0xB16B00u + it, 4);
        for (int b = 1; b < nb; b += 2) k_check<<<(unsigned)((cnt[b] + 255) / 256), 256, 0, st>>>(blk[b], cnt[b], (uint32_t)(it * 131 + b), 200, bad + 1, bad + 8);     // the blocks still owned must be intact
        k_check<<<(unsigned)((bigCnt + 255) / 256), 256, 0, st>>>(big, bigCnt, 0xB16B00u + it, 4, bad + 2, bad + 8);
        if (plain || syncBeforeFree) CK(hipStreamSynchronize(st));
        for (int b = 1; b < nb; b += 2) { if (plain) CK(hipFree(blk[b])); else CK(hipFreeAsync(blk[b], st)); }
        if (plain) { CK(hipFree(big)); continue; }
        if (freeMode == 0 || (freeMode == 2 && it % 3 == 0)) CK(hipFreeAsync(big, st)); else { CK(hipStreamSynchronize(st)); CK(hipFree(big)); }
        if (it % 7 == 0) CK(hipStreamSynchronize(st));
        (void)total;
    }
    CK(hipStreamSynchronize(st));
    unsigned long long h[4] = {0, 0, 0, 0}; CK(hipMemcpy(h, bad, 32, hipMemcpyDeviceToHost));
    printf("%d iterations, big block %zu MB, release threshold %s, big block freed by %s: corrupted words: %llu in blocks checked before any free, %llu in blocks still owned after the big request, %llu in the big block\n",
           iters, bigMB, keep ? "max" : "default", freeMode == 0 ? "hipFreeAsync" : freeMode == 1 ? "sync + hipFree" : "both", h[0], h[1], h[2]);
    printf("medium blocks up to %d MB, stream %s at hipFreeAsync\n", maxMB, syncBeforeFree ? "idle" : "busy");
    unsigned long long smp[8]; CK(hipMemcpy(smp, bad + 8, 64, hipMemcpyDeviceToHost));
    printf("first mismatches (word index, expected, found): (%llu, %08llx, %08llx) (%llu, %08llx, %08llx); zero words among the mismatches: %llu\n", smp[2], smp[3], smp[4], smp[5], smp[6], smp[7], smp[0]);
    printf("pairs of live blocks with overlapping address ranges: %ld\n", overlaps);
    return (h[0] | h[1] | h[2]) ? 1 : 0;
}
