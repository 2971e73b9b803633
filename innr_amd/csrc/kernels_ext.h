// kernels_ext.h -- the remaining L2 helpers of src/batch.rs: per-dimension variance (:572-592) and the
// order-preserving survivor compaction behind batch_l2_squared_pruning (:320-365).
#pragma once

#include "common.h"

namespace innr {

// batch_dimension_variance (batch.rs:572-592): per dimension d, SEQUENTIALLY over i = 0..N-1 (the order is part
// of the result): mean = sum(x)/n, var = sum((x-mean)*(x-mean))/n, both sums folded from -0.0 like
// <f32 as Sum>::sum. One lane per dimension: a serial dependency chain is what the reference computes, so the
// only parallelism is across dimensions; float4 loads keep each lane on its own cache lines. One-time per batch.
__global__ __launch_bounds__(64) void dimension_variance_kernel(const float* __restrict__ V, size_t ldN, uint32_t N,
                                                                uint32_t D, float* __restrict__ var) {
    const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= D) return;
    const float* row = V + (size_t)d * ldN;
    if (N <= 1) {  // batch.rs:573-575
        var[d] = 0.0f;
        return;
    }
    const float nf = (float)N;
    float s = -0.0f;
    const uint32_t n4 = N / 4;
    const float4* r4 = reinterpret_cast<const float4*>(row);
#pragma unroll 4
    for (uint32_t i = 0; i < n4; ++i) {
        const float4 v = r4[i];
        s = ex::add(ex::add(ex::add(ex::add(s, v.x), v.y), v.z), v.w);
    }
    for (uint32_t i = n4 * 4; i < N; ++i) s = ex::add(s, row[i]);
    const float mean = ex::div(s, nf);
    float acc = -0.0f;
#pragma unroll 4
    for (uint32_t i = 0; i < n4; ++i) {
        const float4 v = r4[i];
        const float a = ex::sub(v.x, mean), b = ex::sub(v.y, mean), c = ex::sub(v.z, mean), e = ex::sub(v.w, mean);
        acc = ex::mad2(ex::mad2(ex::mad2(ex::mad2(acc, a, a), b, b), c, c), e, e);
    }
    for (uint32_t i = n4 * 4; i < N; ++i) {
        const float a = ex::sub(row[i], mean);
        acc = ex::mad2(acc, a, a);
    }
    var[d] = ex::div(acc, nf);
}

// batch_l2_squared_pruning survivors (batch.rs:339-364): a vector is dropped the first time its partial sum
// exceeds `threshold`; partial sums of squares are monotone (also in f32), so the survivors are exactly the
// vectors whose FULL distance is not > threshold (NaN survives: `dist > threshold` is false), reported in index
// order with their full distance. Pass 1 counts per 256-vector chunk, pass 2 (after an exclusive scan of the
// counts) scatters in order.
__global__ __launch_bounds__(256) void prune_count_kernel(const float* __restrict__ dist, uint32_t N, float threshold,
                                                          uint32_t* __restrict__ chunk_count) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const bool keep = (i < N) && !(dist[i] > threshold);
    const unsigned long long m = __ballot(keep);
    __shared__ uint32_t wc[4];
    if ((threadIdx.x & 63) == 0) wc[threadIdx.x >> 6] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) chunk_count[blockIdx.x] = wc[0] + wc[1] + wc[2] + wc[3];
}

// single-workgroup exclusive scan of `n` counts (n <= a few 100k): offsets[i] = sum(counts[0..i)), total -> *total
__global__ __launch_bounds__(1024) void exclusive_scan_kernel(const uint32_t* __restrict__ counts, uint32_t n,
                                                               uint32_t* __restrict__ offsets,
                                                               uint32_t* __restrict__ total) {
    __shared__ uint32_t part[1024];
    const uint32_t per = (n + 1023) / 1024;
    const uint32_t b = threadIdx.x * per, e = (b + per < n) ? b + per : n;
    uint32_t s = 0;
    for (uint32_t i = b; i < e; ++i) s += counts[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {  // Hillis-Steele inclusive scan of the partials
        uint32_t v = (threadIdx.x >= off) ? part[threadIdx.x - off] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = (threadIdx.x == 0) ? 0 : part[threadIdx.x - 1];
    for (uint32_t i = b; i < e; ++i) {
        offsets[i] = run;
        run += counts[i];
    }
    if (threadIdx.x == 1023) *total = part[1023];
}

__global__ __launch_bounds__(256) void prune_scatter_kernel(const float* __restrict__ dist, uint32_t N, float threshold,
                                                            const uint32_t* __restrict__ chunk_offset, uint64_t index_base,
                                                            uint64_t* __restrict__ out_idx, float* __restrict__ out_dist,
                                                            uint32_t cap) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const float dv = (i < N) ? dist[i] : 0.0f;
    const bool keep = (i < N) && !(dv > threshold);
    const unsigned long long m = __ballot(keep);
    __shared__ uint32_t wc[4];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) wc[w] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t base = chunk_offset[blockIdx.x];
    for (int j = 0; j < w; ++j) base += wc[j];
    if (keep) {
        const uint32_t pos = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        if (pos < cap) {
            out_idx[pos] = index_base + i;
            out_dist[pos] = dv;
        }
    }
}

}  // namespace innr
