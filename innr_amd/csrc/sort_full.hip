// sort_full.hip -- the "k larger than the candidate lists" path (k > INNR_MAX_K): the reference's own algorithm on the
// device -- all N scores of a query, a stable sort by (score, index), truncate(k) (batch.rs:754-763, 790-799;
// scalar.rs:383-392) -- restricted to what the truncation keeps. The keys are the 64-bit composites the candidate lists
// use ([preference(32) | ~index(32)], larger = better: descending key order IS the stable sort by score with ties in index
// order), all distinct. Hand-written, no library sort:
//   1. RADIX SELECT of the k-th largest key: eight passes over the N keys, most significant byte first -- a 256-bin
//      histogram per pass (per-block in LDS, merged by atomics), only keys that match the prefix fixed so far are counted;
//      one tiny kernel per pass picks the bin that holds rank k. 8 reads of the keys instead of the ~16 reads + writes of a
//      full radix sort, and no 2^31 limit on N.
//   2. the keys >= it (exactly k: composites are distinct) are gathered, unordered, into a buffer of next_pow2(k);
//   3. that buffer is sorted: one workgroup's LDS bitonic sort up to 4096 keys, beyond that a global bitonic network whose
//      steps below the LDS window run inside one launch per stage (k = N = 10M, "rank the whole corpus": 91 launches).
// A translation unit of its own (it shares nothing with the scan / GEMM kernels but the composite layout).
#include <hip/hip_runtime.h>

#include <algorithm>

#include "common.h"
#include "select_dev.h"

namespace innr {

// mask (nullable): predicate bytes of batch_knn_filtered (batch.rs:839); a vector that does not pass gets composite 0,
// below every real one (a real composite has ~idx != 0), so the passing vectors come first
__global__ void make_sort_keys_kernel(const float* __restrict__ scores, size_t N, bool smaller_is_better,
                                      const uint8_t* __restrict__ mask, uint64_t* __restrict__ keys) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const uint32_t o = f32_ord(scores[i]);
    keys[i] = (mask && !mask[i]) ? 0ull : cand_make(smaller_is_better ? ~o : o, (uint32_t)i);
}

// selection state in device memory: hist[256], then {prefix, want} as two uint64, then the gather counter
struct SelectState {
    uint32_t hist[256];
    uint64_t prefix, want;
    uint32_t gathered, pad;
};

__global__ void select_init_kernel(SelectState* st, uint64_t k) {
    if (threadIdx.x < 256) st->hist[threadIdx.x] = 0;
    if (threadIdx.x == 0) {
        st->prefix = 0;
        st->want = k;
        st->gathered = 0;
    }
}

// histogram of byte `byte` over the keys whose higher bytes equal the prefix fixed so far
__global__ __launch_bounds__(256) void select_hist_kernel(const uint64_t* __restrict__ keys, size_t n, SelectState* st, int byte) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int sh = 8 * byte;
    const uint64_t himask = byte == 7 ? 0ull : (~0ull << (sh + 8));
    const uint64_t prefix = st->prefix;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
        const uint64_t v = keys[e];
        if ((v & himask) == prefix) atomicAdd(&h[(uint32_t)(v >> sh) & 0xffu], 1u);
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&st->hist[threadIdx.x], h[threadIdx.x]);
}

// the bin that holds rank `want` (1 = largest) becomes the next byte of the prefix; the histogram is cleared for the next pass
__global__ void select_pick_kernel(SelectState* st, int byte) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = st->hist[threadIdx.x];
    __syncthreads();
    st->hist[threadIdx.x] = 0;
    if (threadIdx.x == 0) {
        uint64_t acc = 0, want = st->want;
        uint32_t b = 255;
        for (;; --b) {
            if (acc + h[b] >= want || b == 0) break;
            acc += h[b];
        }
        st->prefix |= (uint64_t)b << (8 * byte);
        st->want = want - acc;
    }
}

// out[0..k) = the keys above the selected one (phase 0), then copies of the selected one until k are there (phase 1: keys are
// distinct except for masked vectors -- composite 0 -- and duplicate candidates of a re-rank), in no particular order;
// out[k..npad) = 0
__global__ __launch_bounds__(256) void select_gather_kernel(const uint64_t* __restrict__ keys, size_t n, SelectState* st,
                                                             uint64_t* __restrict__ out, size_t k, size_t npad, int phase) {
    const uint64_t t = st->prefix;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
        const uint64_t v = keys[e];
        if (phase == 0 ? v > t : v == t) {
            const uint32_t pos = atomicAdd(&st->gathered, 1u);
            if (pos < k) out[pos] = v;
        }
    }
    if (phase == 0)
        for (size_t e = k + (size_t)blockIdx.x * 256 + threadIdx.x; e < npad; e += (size_t)gridDim.x * 256) out[e] = 0ull;
}

// ---- descending bitonic network over npad = 2^m keys, window = kSelSlots keys per workgroup ----------------------------
// all steps of the stages k = 2 .. kSelSlots on one window (stage kSelSlots takes its direction from the window's position)
__global__ __launch_bounds__(kSelThreads) void bitonic_window_sort_kernel(uint64_t* __restrict__ keys, size_t npad) {
    __shared__ uint64_t s[kSelSlots];
    const size_t base = (size_t)blockIdx.x * kSelSlots;
    const int w = npad < (size_t)kSelSlots ? (int)npad : kSelSlots;
    for (int e = threadIdx.x; e < w; e += kSelThreads) s[e] = keys[base + e];
    __syncthreads();
    for (int k = 2; k <= w; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < w; i += kSelThreads) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const uint64_t a = s[i], b = s[ixj];
                    const bool desc = (((base + (size_t)i) & (size_t)k) == 0);
                    if (desc ? (a < b) : (a > b)) {
                        s[i] = b;
                        s[ixj] = a;
                    }
                }
            }
            __syncthreads();
        }
    }
    for (int e = threadIdx.x; e < w; e += kSelThreads) keys[base + e] = s[e];
}
// one step (j >= kSelSlots) of stage k
__global__ __launch_bounds__(256) void bitonic_global_step_kernel(uint64_t* __restrict__ keys, size_t npad, size_t j, size_t k) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npad) return;
    const size_t ixj = i ^ j;
    if (ixj > i) {
        const uint64_t a = keys[i], b = keys[ixj];
        const bool desc = ((i & k) == 0);
        if (desc ? (a < b) : (a > b)) {
            keys[i] = b;
            keys[ixj] = a;
        }
    }
}
// the steps j = kSelSlots/2 .. 1 of stage k (> kSelSlots) on one window
__global__ __launch_bounds__(kSelThreads) void bitonic_window_merge_kernel(uint64_t* __restrict__ keys, size_t k) {
    __shared__ uint64_t s[kSelSlots];
    const size_t base = (size_t)blockIdx.x * kSelSlots;
    for (int e = threadIdx.x; e < kSelSlots; e += kSelThreads) s[e] = keys[base + e];
    __syncthreads();
    const bool desc = ((base & k) == 0);  // the whole window lies on one side of bit k
    for (int j = kSelSlots >> 1; j > 0; j >>= 1) {
        for (int i = threadIdx.x; i < kSelSlots; i += kSelThreads) {
            const int ixj = i ^ j;
            if (ixj > i) {
                const uint64_t a = s[i], b = s[ixj];
                if (desc ? (a < b) : (a > b)) {
                    s[i] = b;
                    s[ixj] = a;
                }
            }
        }
        __syncthreads();
    }
    for (int e = threadIdx.x; e < kSelSlots; e += kSelThreads) keys[base + e] = s[e];
}

static size_t next_pow2_sz(size_t x) {
    size_t p = 1;
    while (p < x) p <<= 1;
    return p;
}

// bytes of scratch: the selection state (256-byte aligned slot); the callers size their key buffers themselves
hipError_t full_sort_scratch_bytes(size_t /*n*/, size_t* bytes) {
    *bytes = 4096;
    return hipSuccess;
}
// capacity (in keys) the `sorted` buffer of full_topk_keys needs for the best k
size_t full_topk_out_capacity(size_t k) { return next_pow2_sz(k > 1 ? k : 1); }

// keys[0..n) (distinct composites; 0 = "not a candidate") -> sorted[0..k) best first. sorted: full_topk_out_capacity(k) keys.
hipError_t full_topk_keys(const uint64_t* keys, size_t n, size_t k, uint64_t* sorted, void* scratch, hipStream_t stream) {
    if (k > n) k = n;
    if (k == 0) return hipSuccess;
    SelectState* st = static_cast<SelectState*>(scratch);
    const unsigned nb = (unsigned)std::min<size_t>((n + 255) / 256, 4096);
    select_init_kernel<<<1, 256, 0, stream>>>(st, (uint64_t)k);
    for (int byte = 7; byte >= 0; --byte) {
        select_hist_kernel<<<nb, 256, 0, stream>>>(keys, n, st, byte);
        select_pick_kernel<<<1, 256, 0, stream>>>(st, byte);
    }
    const size_t npad = next_pow2_sz(k);
    select_gather_kernel<<<nb, 256, 0, stream>>>(keys, n, st, sorted, k, npad, 0);
    select_gather_kernel<<<nb, 256, 0, stream>>>(keys, n, st, sorted, k, npad, 1);
    const unsigned nwin = (unsigned)((npad + kSelSlots - 1) / kSelSlots);
    bitonic_window_sort_kernel<<<nwin, kSelThreads, 0, stream>>>(sorted, npad);
    for (size_t kk = (size_t)kSelSlots * 2; kk <= npad; kk <<= 1) {
        for (size_t j = kk >> 1; j >= (size_t)kSelSlots; j >>= 1)
            bitonic_global_step_kernel<<<(unsigned)((npad + 255) / 256), 256, 0, stream>>>(sorted, npad, j, kk);
        bitonic_window_merge_kernel<<<nwin, kSelThreads, 0, stream>>>(sorted, kk);
    }
    return hipGetLastError();
}

// scores[0..n) -> sorted[0..k) composites, best first (keys: scratch of n composites)
hipError_t full_sort_scores(const float* scores, size_t n, bool smaller_is_better, uint64_t* keys, uint64_t* sorted, void* scratch,
                            size_t /*scratch_bytes*/, hipStream_t stream, const uint8_t* mask, size_t k) {
    make_sort_keys_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(scores, n, smaller_is_better, mask, keys);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return full_topk_keys(keys, n, k, sorted, scratch, stream);
}

// nseg segments of `len` composites each (innr_batch_rerank with more candidates per query than a list holds): the best k of
// every segment, best first, at sorted[seg * len ..]. k <= kSelSlots: one workgroup per segment (radix select + LDS sort);
// beyond: the segments one after the other through full_topk_keys (tmp: full_topk_out_capacity(k) keys).
__global__ __launch_bounds__(kSelThreads) void segment_sort_kernel(const uint64_t* __restrict__ keys, uint64_t* __restrict__ sorted,
                                                                    uint32_t len, uint32_t k) {
    __shared__ uint64_t s[kSelSlots];
    __shared__ uint32_t hist[258];
    const size_t base = (size_t)blockIdx.x * len;
    wg_segment_topk(keys + base, len, k, s, hist);
    __syncthreads();
    for (uint32_t r = threadIdx.x; r < k; r += kSelThreads) sorted[base + r] = s[r];
}
hipError_t segmented_sort_scratch_bytes(size_t /*nseg*/, size_t /*len*/, size_t* bytes) {
    *bytes = 4096;
    return hipSuccess;
}
hipError_t segmented_topk_keys(const uint64_t* keys, uint64_t* sorted, size_t nseg, size_t len, size_t k, uint64_t* tmp, void* scratch,
                               hipStream_t stream) {
    if (k > len) k = len;
    if (k == 0 || nseg == 0) return hipSuccess;
    if (k <= (size_t)kSelSlots) {
        segment_sort_kernel<<<(unsigned)nseg, kSelThreads, 0, stream>>>(keys, sorted, (uint32_t)len, (uint32_t)k);
        return hipGetLastError();
    }
    for (size_t sgm = 0; sgm < nseg; ++sgm) {
        hipError_t e = full_topk_keys(keys + sgm * len, len, k, tmp, scratch, stream);
        if (e != hipSuccess) return e;
        e = hipMemcpyAsync(sorted + sgm * len, tmp, k * sizeof(uint64_t), hipMemcpyDeviceToDevice, stream);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace innr
