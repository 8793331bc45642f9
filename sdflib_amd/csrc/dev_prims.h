// Device-wide primitives of the product path: exclusive prefix sums and key / value radix sorts on rocPRIM (ROCm's own library,
// called directly — no CUB-shaped front end).  Both follow rocPRIM's two-call protocol: tmp == nullptr returns the scratch size.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_radix_sort.hpp>

namespace sdfhip {

template <typename T>
static inline hipError_t devExclusiveSum(void* tmp, size_t& bytes, const T* in, T* out, size_t n, hipStream_t st) {
    return rocprim::exclusive_scan(tmp, bytes, in, out, (T)0, n, rocprim::plus<T>(), st);
}
// ascending, stable, on the key bits [beginBit, endBit)
template <typename K, typename V>
static inline hipError_t devSortPairs(void* tmp, size_t& bytes, const K* keysIn, K* keysOut, const V* valsIn, V* valsOut, size_t n, unsigned beginBit, unsigned endBit, hipStream_t st) {
    return rocprim::radix_sort_pairs(tmp, bytes, keysIn, keysOut, valsIn, valsOut, n, beginBit, endBit, st);
}

}  // namespace sdfhip
