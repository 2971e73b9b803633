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
__global__ __launch_bounds__(kSelThreads) void select_topk_kernel(const uint64_t* __restrict__ lists,
                                                                   const uint32_t* __restrict__ counts,
                                                                   uint32_t nslots, uint32_t qstride, uint32_t cap,
                                                                   uint32_t KP, uint64_t* __restrict__ out,
                                                                   uint32_t* __restrict__ out_cnt) {
    __shared__ uint64_t s[kSelSlots];
    const uint32_t q = blockIdx.x;
    int fill = 0;  // uniform across the workgroup
    for (uint32_t slot = 0; slot < nslots; ++slot) {
        const uint32_t c = counts[(size_t)slot * qstride + q];
        if (c == 0) continue;
        if (fill + (int)c > kSelSlots) {  // window full: reduce to the best KP first
            const int n = next_pow2_i(fill);
            for (int i = fill + threadIdx.x; i < n; i += kSelThreads) s[i] = 0;
            __syncthreads();
            wg_bitonic_desc(s, n);
            fill = fill < (int)KP ? fill : (int)KP;
        }
        const uint64_t* src = lists + ((size_t)slot * qstride + q) * cap;
        for (uint32_t i = threadIdx.x; i < c; i += kSelThreads) s[fill + i] = src[i];
        fill += (int)c;
        __syncthreads();
    }
    const int n = next_pow2_i(fill > 1 ? fill : 1);
    for (int i = fill + threadIdx.x; i < n; i += kSelThreads) s[i] = 0;
    __syncthreads();
    wg_bitonic_desc(s, n);
    const int keep = fill < (int)KP ? fill : (int)KP;
    for (int i = threadIdx.x; i < (int)KP; i += kSelThreads) out[(size_t)q * KP + i] = (i < keep) ? s[i] : 0ull;
    if (threadIdx.x == 0) out_cnt[q] = (uint32_t)keep;
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
