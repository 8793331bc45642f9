// Dev probe: what a 4-byte device -> host read-back costs behind a small kernel, (a) as the builders do it (hipMemcpyAsync into a pageable
// word + hipStreamSynchronize), (b) hipMemcpyAsync into pinned memory + hipStreamSynchronize, (c) a one-thread kernel that stores the
// value and a sequence number into host-mapped pinned memory while the host spins on the sequence number.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/sync_probe/sync_probe tools/sync_probe/sync_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
__global__ void k_work(uint32_t* p, int n) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 3u + 1u; }
__global__ void k_post(const uint32_t* src, volatile uint32_t* mb, uint32_t seq) { mb[1] = src[0]; __threadfence_system(); mb[0] = seq; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipStream_t st; hipStreamCreate(&st);
    uint32_t* d; hipMalloc(&d, 1 << 22); hipMemset(d, 0, 1 << 22);
    uint32_t* pinned; hipHostMalloc(&pinned, 4096, hipHostMallocMapped); volatile uint32_t* mb = pinned; uint32_t* mbDev; hipHostGetDevicePointer((void**)&mbDev, pinned, 0);
    mb[0] = 0;
    const int reps = 2000;
    for (int work : {1 << 10, 1 << 20}) {
        uint32_t h = 0; double t0;
        for (int i = 0; i < 50; i++) { k_work<<<work / 256, 256, 0, st>>>(d, work); hipMemcpyAsync(&h, d, 4, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st); }
        t0 = now(); for (int i = 0; i < reps; i++) { k_work<<<work / 256, 256, 0, st>>>(d, work); hipMemcpyAsync(&h, d, 4, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st); }
        const double a = (now() - t0) / reps;
        t0 = now(); for (int i = 0; i < reps; i++) { k_work<<<work / 256, 256, 0, st>>>(d, work); hipMemcpyAsync(pinned + 8, d, 4, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st); }
        const double b = (now() - t0) / reps;
        uint32_t seq = 0;
        t0 = now(); for (int i = 0; i < reps; i++) { k_work<<<work / 256, 256, 0, st>>>(d, work); k_post<<<1, 1, 0, st>>>(d, mbDev, ++seq); while (mb[0] != seq) { __builtin_ia32_pause(); } }
        const double c = (now() - t0) / reps;
        t0 = now(); for (int i = 0; i < reps; i++) { k_work<<<work / 256, 256, 0, st>>>(d, work); } hipStreamSynchronize(st);
        const double base = (now() - t0) / reps;
        printf("kernel over %d words: back-to-back %.1f us per launch; + read-back: pageable copy + sync %.1f us, pinned copy + sync %.1f us, mailbox kernel + spin %.1f us\n", work, base * 1e6, a * 1e6, b * 1e6, c * 1e6);
    }
    return 0;
}
