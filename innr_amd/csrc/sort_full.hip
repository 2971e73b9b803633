// sort_full.hip -- the "k larger than the candidate lists" path (k > INNR_MAX_K): the reference's own algorithm on the
// device -- all N scores of a query, a full sort by (score, index), truncate (batch.rs:754-763, 790-799;
// scalar.rs:383-392) -- with the sort done by rocPRIM's radix sort on the same 64-bit composites the candidate
// lists use ([preference(32) | ~index(32)], larger = better: one descending key sort IS the stable sort by score
// with ties in index order). A translation unit of its own: the rocPRIM headers would double api.hip's build time.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include "common.h"

namespace innr {

__global__ void make_sort_keys_kernel(const float* __restrict__ scores, uint32_t N, bool smaller_is_better,
                                      uint64_t* __restrict__ keys) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const uint32_t o = f32_ord(scores[i]);
    keys[i] = cand_make(smaller_is_better ? ~o : o, i);
}

// bytes of scratch the sort of n keys needs
hipError_t full_sort_scratch_bytes(size_t n, size_t* bytes) {
    *bytes = 0;
    return hipcub::DeviceRadixSort::SortKeysDescending(nullptr, *bytes, (const uint64_t*)nullptr, (uint64_t*)nullptr, (int)n);
}

// scores[0..n) -> sorted[0..n) composites, best first
hipError_t full_sort_scores(const float* scores, size_t n, bool smaller_is_better, uint64_t* keys, uint64_t* sorted,
                            void* scratch, size_t scratch_bytes, hipStream_t stream) {
    make_sort_keys_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(scores, (uint32_t)n, smaller_is_better, keys);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return hipcub::DeviceRadixSort::SortKeysDescending(scratch, scratch_bytes, keys, sorted, (int)n, 0, 64, stream);
}

}  // namespace innr
