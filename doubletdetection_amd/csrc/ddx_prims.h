// Device-wide primitives used by the stages: direct calls into rocPRIM (header-only, ships with ROCm), wave64-native.
// Every call follows rocPRIM's two-phase protocol: temp == nullptr returns the temporary-storage need in `bytes`.
// NOTE radix sorts pick their algorithm (single block / merge / onesweep) by the element count, and each algorithm has
// its own temporary-storage need: always query with the count that is then sorted.
#pragma once

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_reduce_by_key.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_segmented_radix_sort.hpp>
#include <rocprim/functional.hpp>

namespace ddx {
namespace prim {

template <typename K, typename V>
inline hipError_t sort_pairs(void* temp, size_t& bytes, const K* kin, K* kout, const V* vin, V* vout, size_t n, int begin_bit, int end_bit,
                             hipStream_t s) {
    return rocprim::radix_sort_pairs(temp, bytes, kin, kout, vin, vout, n, (unsigned)begin_bit, (unsigned)end_bit, s);
}

template <typename K, typename V>
inline hipError_t sort_pairs_desc(void* temp, size_t& bytes, const K* kin, K* kout, const V* vin, V* vout, size_t n, int begin_bit,
                                  int end_bit, hipStream_t s) {
    return rocprim::radix_sort_pairs_desc(temp, bytes, kin, kout, vin, vout, n, (unsigned)begin_bit, (unsigned)end_bit, s);
}

template <typename K>
inline hipError_t sort_keys(void* temp, size_t& bytes, const K* kin, K* kout, size_t n, int begin_bit, int end_bit, hipStream_t s) {
    return rocprim::radix_sort_keys(temp, bytes, kin, kout, n, (unsigned)begin_bit, (unsigned)end_bit, s);
}

// independent sorts of the segments [begin[i], end[i]) (here: the rows of a CSR matrix)
template <typename K, typename V, typename O>
inline hipError_t segmented_sort_pairs(void* temp, size_t& bytes, const K* kin, K* kout, const V* vin, V* vout, size_t n, size_t segments,
                                       const O* begin, const O* end, int begin_bit, int end_bit, hipStream_t s) {
    return rocprim::segmented_radix_sort_pairs(temp, bytes, kin, kout, vin, vout, (unsigned)n, (unsigned)segments, begin, end,
                                               (unsigned)begin_bit, (unsigned)end_bit, s);
}

// out[i] = in[0] + ... + in[i-1], accumulated in the output type
template <typename I, typename O>
inline hipError_t exclusive_sum(void* temp, size_t& bytes, const I* in, O* out, size_t n, hipStream_t s) {
    return rocprim::exclusive_scan(temp, bytes, in, out, O(0), n, rocprim::plus<O>(), s);
}

// runs of equal consecutive keys -> (key, sum of the run's values); *runs = number of runs
template <typename K, typename V, typename R>
inline hipError_t reduce_by_key_sum(void* temp, size_t& bytes, const K* keys, K* unique_out, const V* vals, V* sums, R* runs, size_t n,
                                    hipStream_t s) {
    return rocprim::reduce_by_key(temp, bytes, keys, vals, (unsigned)n, unique_out, sums, runs, rocprim::plus<V>(), rocprim::equal_to<K>(), s);
}

}  // namespace prim
}  // namespace ddx
