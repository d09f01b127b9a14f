// Stable key-value radix sort of a batch's statistic events by group id
// (rocPRIM through hipCUB: a library primitive, kept in its own translation
// unit so that the kernels of dist_hip.hip compile in seconds).  Used by the
// ordered replay of order-dependent float statistics (kernels.h,
// k_replay_sorted): stability is what keeps each group's events in row order.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <stdexcept>
#include <string>

namespace dist {

size_t sort_pairs_temp_bytes(size_t n, int bits) {
    size_t bytes = 0;
    const uint32_t * k = nullptr;
    uint32_t * ko = nullptr;
    hipError_t err = hipcub::DeviceRadixSort::SortPairs(
        nullptr, bytes, k, ko, k, ko, (int)n, 0, bits, (hipStream_t) nullptr);
    if (err != hipSuccess)
        throw std::runtime_error(std::string("radix sort sizing: ") +
                                 hipGetErrorString(err));
    return bytes;
}

void sort_pairs(void * temp, size_t temp_bytes, const uint32_t * keys_in,
                uint32_t * keys_out, const uint32_t * vals_in,
                uint32_t * vals_out, size_t n, int bits, hipStream_t stream) {
    hipError_t err = hipcub::DeviceRadixSort::SortPairs(
        temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, (int)n, 0,
        bits, stream);
    if (err != hipSuccess)
        throw std::runtime_error(std::string("radix sort: ") +
                                 hipGetErrorString(err));
}

}  // namespace dist
