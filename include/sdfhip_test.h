/* sdfhip_test.h - test and calibration hooks of libsdfhip.so.
 *
 * NOT part of the drop-in boundary (include/sdfhip.h): nothing a caller of the reference's API needs.  These entry points exist so that
 * tests/ and bench.py can check pieces of the product in isolation (the restated std::sort / glibc acosf, the BVH planner) and calibrate
 * profile counters (the 256-byte gather, the VALU issue ceiling).  They are exported by the same library and bound by sdflib_amd/_lib.py. */
#ifndef SDFHIP_TEST_H
#define SDFHIP_TEST_H
#include "sdfhip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* test hook, host only: mismatches between the BVH planner's threaded restatement of std::sort and std::sort itself on n keys */
int sdfhip_test_sort_matches_std(const double* keys, uint64_t n, int threads);
int sdfhip_test_heap_sort_matches_std(const double* keys, uint64_t n);     /* the restated libstdc++ heap sort vs std::make_heap + std::sort_heap */
/* test hooks: the restated glibc acosf of the mesh preparation (dev_math.h::acosfGlibc) against the running libm on the bit patterns
 * first_bits + i * stride, i < count (values outside [-1, 1] skipped) - its host compilation (no device needed; 0, 1, 2^32 = every float),
 * and its device compilation.  Both return / store the number of differing results. */
uint64_t sdfhip_test_acosf_mismatches(uint32_t first_bits, uint32_t stride, uint64_t count, int threads);
int sdfhip_test_acosf_device(sdfhip_ctx* ctx, uint32_t first_bits, uint32_t stride, uint32_t count, uint64_t* out_mismatches);
/* Which acosf the corner angles use is decided by the first mesh of a process (a self-check of the restatement against the running libm:
 * on a host whose libm is another function the arc cosines are taken there, as the reference does).  Test hook: 1 forces the host's,
 * 0 the device's, -1 makes the next mesh decide again. */
void sdfhip_test_set_host_acos(int mode);
/* the BVH planner alone, host memory in and out (no device needed): 8 doubles + 2 ints per inner node, max(num_triangles - 1, 1) nodes.
 * Replaces the tree half of tmd::TriangleMeshDistance::construct (TriangleMeshDistance.h:421-490); CPU tests compare it with the oracle's. */
int sdfhip_test_plan_bvh(const float* xyz, uint32_t num_vertices, const uint32_t* indices, uint32_t num_triangles, double* out_spheres, int32_t* out_children, double* seconds);

/* Profiling aid (device pointers, enqueued on the context's stream): lane i reads the 256-byte block dev_block_ids[i] of dev_data with the
 * query kernel's load pattern (16 x dwordx4) and writes one float.  With a permutation of block ids the bytes that must cross the fabric
 * are known exactly, which calibrates rocprofv3's FETCH_SIZE for this access pattern (bench.py, tools/profile_bench.sh). */
int sdfhip_test_gather_blocks(sdfhip_ctx* ctx, const uint32_t* dev_data, const uint32_t* dev_block_ids, uint64_t n, float* dev_out);

/* Calibration of the VALU issue ceiling (device pointer, enqueued on the context's stream): `blocks` workgroups of 256 lanes, every wave
 * issues 8 independent chains x `iters` v_fma_f32 (8 * iters wave-instructions + a few of loop control); lane results are summed into
 * dev_out[blocks * 256] so that nothing is eliminated.  Timed live it gives the chip's fp32 wave-instruction rate; profiled, its
 * SQ_ACTIVE_INST_VALU per second is the counter's own ceiling, against which bench.py prices the VALU-bound kernels. */
int sdfhip_test_valu_peak(sdfhip_ctx* ctx, uint32_t blocks, uint32_t iters, float* dev_out);

#ifdef __cplusplus
}
#endif
#endif /* SDFHIP_TEST_H */
