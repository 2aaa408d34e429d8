// svr_sort.hip -- device radix sort and prefix sum for the work lists of the cell-owned scatter (svr_cell.inc), taken from
// the vendor's primitives library (hipCUB over rocPRIM: plumbing that runs once per slice geometry, not on the hot path).
// A translation unit of its own so that the library's templates are not re-instantiated with every build of svr_hip.hip.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>

// tmp == NULL: *tmp_bytes = the scratch the call needs.  Returns a hipError_t as int.
int svr_sort_keys_u64(void *tmp, size_t *tmp_bytes, const uint64_t *keys_in, uint64_t *keys_out, size_t n, int end_bit, hipStream_t stream) {
  return (int)hipcub::DeviceRadixSort::SortKeys(tmp, *tmp_bytes, keys_in, keys_out, (int)n, 0, end_bit, stream);
}
int svr_inclusive_sum_u32(void *tmp, size_t *tmp_bytes, const uint32_t *in, uint32_t *out, size_t n, hipStream_t stream) {
  return (int)hipcub::DeviceScan::InclusiveSum(tmp, *tmp_bytes, in, out, (int)n, stream);
}
