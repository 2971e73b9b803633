// sort_full.hip -- the "k larger than the candidate lists" path (k > INNR_MAX_K): the reference's own algorithm on the
// device -- all N scores of a query, a full sort by (score, index), truncate (batch.rs:754-763, 790-799;
// scalar.rs:383-392) -- with the sort done by rocPRIM's radix sort on the same 64-bit composites the candidate
// lists use ([preference(32) | ~index(32)], larger = better: one descending key sort IS the stable sort by score
// with ties in index order). A translation unit of its own: the rocPRIM headers would double api.hip's build time.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include "common.h"

namespace innr {

// mask (nullable): predicate bytes of batch_knn_filtered (batch.rs:839); a vector that does not pass gets composite 0,
// below every real one (a real composite has ~idx != 0), so the passing vectors fill the front of the sorted array
__global__ void make_sort_keys_kernel(const float* __restrict__ scores, uint32_t N, bool smaller_is_better,
                                      const uint8_t* __restrict__ mask, uint64_t* __restrict__ keys) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const uint32_t o = f32_ord(scores[i]);
    keys[i] = (mask && !mask[i]) ? 0ull : cand_make(smaller_is_better ? ~o : o, i);
}

__global__ void segment_offsets_kernel(uint32_t nseg, uint32_t len, uint32_t* __restrict__ off) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= nseg) off[i] = i * len;
}

// bytes of scratch the sort of n keys needs
hipError_t full_sort_scratch_bytes(size_t n, size_t* bytes) {
    *bytes = 0;
    return hipcub::DeviceRadixSort::SortKeysDescending(nullptr, *bytes, (const uint64_t*)nullptr, (uint64_t*)nullptr, (int)n);
}

// scores[0..n) -> sorted[0..n) composites, best first
hipError_t full_sort_scores(const float* scores, size_t n, bool smaller_is_better, uint64_t* keys, uint64_t* sorted,
                            void* scratch, size_t scratch_bytes, hipStream_t stream, const uint8_t* mask) {
    make_sort_keys_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(scores, (uint32_t)n, smaller_is_better, mask, keys);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return hipcub::DeviceRadixSort::SortKeysDescending(scratch, scratch_bytes, keys, sorted, (int)n, 0, 64, stream);
}

// nseg segments of `len` composites each (innr_batch_rerank with more candidates per query than a list holds): every
// segment sorted best-first. off: device scratch of nseg + 1 uint32.
hipError_t segmented_sort_scratch_bytes(size_t nseg, size_t len, size_t* bytes) {
    *bytes = 0;
    return hipcub::DeviceSegmentedRadixSort::SortKeysDescending(nullptr, *bytes, (const uint64_t*)nullptr, (uint64_t*)nullptr,
                                                                (int)(nseg * len), (int)nseg, (const uint32_t*)nullptr,
                                                                (const uint32_t*)nullptr);
}
hipError_t segmented_sort_keys(const uint64_t* keys, uint64_t* sorted, size_t nseg, size_t len, uint32_t* off, void* scratch,
                               size_t scratch_bytes, hipStream_t stream) {
    segment_offsets_kernel<<<(unsigned)((nseg + 256) / 256), 256, 0, stream>>>((uint32_t)nseg, (uint32_t)len, off);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return hipcub::DeviceSegmentedRadixSort::SortKeysDescending(scratch, scratch_bytes, keys, sorted, (int)(nseg * len), (int)nseg,
                                                                off, off + 1, 0, 64, stream);
}

}  // namespace innr
