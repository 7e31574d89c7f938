// lane_sort.hip — the one library primitive on the path: a device radix sort (rocPRIM through hipCUB) of
// (64-bit key, 32-bit lane position) pairs, kept in its own translation unit because the sort templates take longer to
// compile than everything else together.  Used by the lane re-ordering of g2048.hip (LaneSort): the keys are the lanes'
// big-tile patterns, the sorted values are the permutation k_td_play reads its lanes through.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>

// temp == nullptr: only report the scratch size in *temp_bytes.  Returns a hipError_t.
extern "C" __attribute__((visibility("hidden"))) int g2048_lane_sort_pairs(void* temp, size_t* temp_bytes, const uint64_t* keys_in, uint64_t* keys_out,
                                                                             const uint32_t* vals_in, uint32_t* vals_out, uint32_t n, int begin_bit,
                                                                             int end_bit, hipStream_t stream) {
    return (int)hipcub::DeviceRadixSort::SortPairs(temp, *temp_bytes, keys_in, keys_out, vals_in, vals_out, (int)n, begin_bit, end_bit, stream);
}
