// kernels_topk.h -- cross-producer selection, exact re-score, finalisation and shard merge.
#pragma once

#include "common.h"
#include "topk_dev.h"

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

// One workgroup per query: best KP composites over all producer lists of that query.
//   lists  : [nslots][qstride][cap] composites, counts : [nslots][qstride]; query q uses column `q`
//   out    : [Q][KP] best-first, zero padded; out_cnt[q] = min(KP, total)
//   Producers leave at most KP entries per list, so list `s` of a part owns the fixed window
//   [s*KP, (s+1)*KP): no prefix sums, one gather, one sort. Part p (blockIdx.y) reduces the
//   kSelSlots/KP lists [p*spp, (p+1)*spp) and writes out[(p*Q + q)*KP ..], out_cnt[p*Q + q]; the host
//   re-launches on that output (as lists with qstride = Q, cap = KP) until one part remains.
__global__ __launch_bounds__(kSelThreads) void select_topk_kernel(const uint64_t* __restrict__ lists,
                                                                   const uint32_t* __restrict__ counts,
                                                                   uint32_t nslots, uint32_t qstride, uint32_t cap,
                                                                   uint32_t KP, uint32_t Q, uint64_t* __restrict__ out,
                                                                   uint32_t* __restrict__ out_cnt) {
    __shared__ uint64_t s[kSelSlots];
    __shared__ uint32_t s_total;
    const uint32_t q = blockIdx.x, part = blockIdx.y;
    const uint32_t spp = kSelSlots / KP;  // lists per part
    const uint32_t slot0 = part * spp;
    const uint32_t nloc = (slot0 >= nslots) ? 0u : ((nslots - slot0 < spp) ? nslots - slot0 : spp);
    if (threadIdx.x == 0) s_total = 0;
    __syncthreads();
    const int n = next_pow2_i((int)(nloc * KP) > 1 ? (int)(nloc * KP) : 1);
    uint32_t mine = 0;
    for (int e = threadIdx.x; e < n; e += kSelThreads) {
        const uint32_t ls = (uint32_t)e / KP, i = (uint32_t)e % KP;
        uint64_t v = 0;
        if (ls < nloc) {
            uint32_t c = counts[(size_t)(slot0 + ls) * qstride + q];
            c = c < KP ? c : KP;
            if (i < c) {
                v = lists[((size_t)(slot0 + ls) * qstride + q) * cap + i];
                ++mine;
            }
        }
        s[e] = v;
    }
    if (mine) atomicAdd(&s_total, mine);
    __syncthreads();
    wg_bitonic_desc(s, n);
    const uint32_t total = s_total;
    const uint32_t keep = total < KP ? total : KP;
    for (uint32_t i = threadIdx.x; i < KP; i += kSelThreads)
        out[((size_t)part * Q + q) * KP + i] = (i < keep) ? s[i] : 0ull;
    if (threadIdx.x == 0) out_cnt[(size_t)part * Q + q] = keep;
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
    // prefix == the kout-th largest key; gather everything >= it
    if (threadIdx.x == 0) hist[0] = 0;
    for (int e = threadIdx.x; e < kSelSlots; e += kSelThreads) s[e] = 0ull;
    __syncthreads();
    for (uint32_t e = threadIdx.x; e < n; e += kSelThreads) {
        const uint64_t v = keys[e];
        if (v >= prefix) {
            const uint32_t pos = atomicAdd(&hist[0], 1u);
            if (pos < (uint32_t)kSelSlots) s[pos] = v;
        }
    }
    __syncthreads();
    wg_bitonic_desc(s, next_pow2_i((int)kout > 1 ? (int)kout : 1));
}

// Exact engines: composites already carry the reference's exact score bits. One thread per (q, r).
__global__ void emit_results_kernel(const uint64_t* __restrict__ sel, uint32_t KP, uint32_t Q, uint32_t kout,
                                    bool smaller_is_better, uint64_t index_base, uint64_t* __restrict__ out_idx,
                                    float* __restrict__ out_score) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= Q * kout) return;
    const uint32_t q = t / kout, r = t % kout;
    const uint64_t c = sel[(size_t)q * KP + r];
    out_idx[t] = index_base + cand_idx(c);
    out_score[t] = pref_score(cand_pref(c), smaller_is_better);
}

// Shard merge: one wave per query; G*kin candidates (global idx, exact score) -> best kout
// by (score order, idx ascending). 64-bit indices, so ranks are counted on (pref, idx) pairs directly.
__global__ __launch_bounds__(64) void merge_topk_kernel(const uint64_t* __restrict__ idx,
                                                        const float* __restrict__ score, uint32_t G, uint32_t Q,
                                                        uint32_t kin, uint32_t kout, bool smaller_is_better,
                                                        uint64_t* __restrict__ out_idx,
                                                        float* __restrict__ out_score) {
    const uint32_t q = blockIdx.x;
    const uint32_t total = G * kin;
    const int lane = threadIdx.x;
    // each lane owns candidates lane, lane+64, ... ; rank = number of candidates strictly better
    for (uint32_t c = lane; c < total; c += 64) {
        const uint32_t g = c / kin, r = c % kin;
        const size_t off = ((size_t)g * Q + q) * kin + r;
        const uint64_t my_i = idx[off];
        const float my_sf = score[off];
        // idx == UINT64_MAX marks an empty slot (a shard that holds fewer than kin vectors): ranks last
        const uint32_t my_p = (my_i == ~0ull) ? 0u : (smaller_is_better ? ~f32_ord(my_sf) : f32_ord(my_sf));
        uint32_t rank = 0;
        for (uint32_t o = 0; o < total; ++o) {
            const uint32_t g2 = o / kin, r2 = o % kin;
            const size_t off2 = ((size_t)g2 * Q + q) * kin + r2;
            const float sf = score[off2];
            const uint64_t i2 = idx[off2];
            const uint32_t p = (i2 == ~0ull) ? 0u : (smaller_is_better ? ~f32_ord(sf) : f32_ord(sf));
            rank += (p > my_p || (p == my_p && (i2 < my_i || (i2 == my_i && o < c)))) ? 1u : 0u;
        }
        if (rank < kout) {
            out_idx[(size_t)q * kout + rank] = my_i;
            out_score[(size_t)q * kout + rank] = my_sf;
        }
    }
}

}  // namespace innr
