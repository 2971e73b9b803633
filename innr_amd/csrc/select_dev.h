// select_dev.h -- workgroup-level sorting / selection of 64-bit composites in LDS (device inline functions only: shared by
// kernels_topk.h and sort_full.hip, which are separate translation units).
#pragma once

#include "common.h"

namespace innr {

constexpr int kSelThreads = 256;
constexpr int kSelSlots = 4096;  // LDS sort window (32 KiB of u64)

// Bitonic sort, DESCENDING, of s[0..n) (n a power of two <= kSelSlots) by one 256-thread workgroup.
__device__ __forceinline__ void wg_bitonic_desc(uint64_t* s, int n) {
    for (int k = 2; k <= n; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < n; i += kSelThreads) {
                int ixj = i ^ j;
                if (ixj > i) {
                    uint64_t a = s[i], b = s[ixj];
                    bool desc = ((i & k) == 0);
                    if (desc ? (a < b) : (a > b)) {
                        s[i] = b;
                        s[ixj] = a;
                    }
                }
            }
            __syncthreads();
        }
    }
}

__device__ __forceinline__ int next_pow2_i(int x) {
    int p = 1;
    while (p < x) p <<= 1;
    return p;
}

// Best kout of ONE segment of n distinct 64-bit composites (larger = better), n up to 2^32, by one 256-thread workgroup: the
// reference's "sort all scores, truncate(k)" (batch.rs:754-763) restricted to what the truncation keeps.
//   n <= kSelSlots: one LDS bitonic sort.
//   larger: RADIX SELECT of the kout-th largest key -- eight passes over the segment, most significant byte first, a 256-bin
//   histogram in LDS per pass (only keys that match the prefix fixed so far are counted) -- then the keys >= it (exactly kout:
//   composites are distinct) are gathered into LDS and sorted. kout <= kSelSlots.
// keys: the segment (global memory); returns the sorted best kout in s[0..kout).
__device__ __forceinline__ void wg_segment_topk(const uint64_t* __restrict__ keys, uint32_t n, uint32_t kout, uint64_t* s /*[kSelSlots]*/,
                                                uint32_t* hist /*[258]*/) {
    if (n <= (uint32_t)kSelSlots) {
        const int np = next_pow2_i((int)n > 1 ? (int)n : 1);
        for (int e = threadIdx.x; e < np; e += kSelThreads) s[e] = (uint32_t)e < n ? keys[e] : 0ull;
        __syncthreads();
        wg_bitonic_desc(s, np);
        return;
    }
    uint64_t prefix = 0;      // the bytes of the kout-th largest key fixed so far
    uint32_t want = kout;     // its rank (1 = largest) among the keys that share `prefix`
    for (int byte = 7; byte >= 0; --byte) {
        for (int i = threadIdx.x; i < 256; i += kSelThreads) hist[i] = 0;
        __syncthreads();
        const int sh = 8 * byte;
        const uint64_t himask = byte == 7 ? 0ull : (~0ull << (sh + 8));
        for (uint32_t e = threadIdx.x; e < n; e += kSelThreads) {
            const uint64_t v = keys[e];
            if ((v & himask) == prefix) atomicAdd(&hist[(uint32_t)(v >> sh) & 0xffu], 1u);
        }
        __syncthreads();
        if (threadIdx.x == 0) {  // walk the bins from the largest byte down to the one that holds rank `want`
            uint32_t acc = 0, b = 255;
            for (;; --b) {
                if (acc + hist[b] >= want || b == 0) break;
                acc += hist[b];
            }
            hist[256] = b;
            hist[257] = want - acc;
        }
        __syncthreads();
        prefix |= (uint64_t)hist[256] << sh;
        want = hist[257];
        __syncthreads();  // (the next pass clears the histogram)
    }
    // prefix == the kout-th largest key; gather the keys above it, then copies of it until kout are there (composites are
    // distinct except for duplicate candidates of a re-rank)
    if (threadIdx.x == 0) hist[0] = 0;
    for (int e = threadIdx.x; e < kSelSlots; e += kSelThreads) s[e] = 0ull;
    __syncthreads();
    for (int phase = 0; phase < 2; ++phase) {
        for (uint32_t e = threadIdx.x; e < n; e += kSelThreads) {
            const uint64_t v = keys[e];
            if (phase == 0 ? v > prefix : v == prefix) {
                const uint32_t pos = atomicAdd(&hist[0], 1u);
                if (pos < kout) s[pos] = v;
            }
        }
        __syncthreads();
    }
    wg_bitonic_desc(s, next_pow2_i((int)kout > 1 ? (int)kout : 1));
}

}  // namespace innr
