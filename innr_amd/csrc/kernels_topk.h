// kernels_topk.h -- cross-producer selection, exact re-score, finalisation and shard merge.
#pragma once

#include "common.h"
#include "select_dev.h"
#include "topk_dev.h"

namespace innr {

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
    // Gather DENSELY (eight threads per list, the list's place handed out by one LDS atomic; the order is the sort's business):
    // the sort then runs over the entries that exist, not over the part's whole window -- the small-batch int8 filter has 2048
    // producers per query, nearly all of them with a handful of entries (a 64-query call spent 1.1 ms in four launches of this
    // kernel sorting 4096-slot windows of zeros).
    const uint32_t g8 = threadIdx.x >> 3, l8 = threadIdx.x & 7;
    for (uint32_t ls = g8; ls < nloc; ls += kSelThreads / 8) {
        uint32_t c = counts[(size_t)(slot0 + ls) * qstride + q];
        c = c < KP ? c : KP;
        uint32_t base = 0;
        if (l8 == 0 && c) base = atomicAdd(&s_total, c);
        base = (uint32_t)__shfl((int)base, (int)(threadIdx.x & 63u & ~7u), 64);
        const uint64_t* src = lists + ((size_t)(slot0 + ls) * qstride + q) * cap;
        for (uint32_t i = l8; i < c; i += 8) s[base + i] = src[i];
    }
    __syncthreads();
    const int n = next_pow2_i((int)s_total > 1 ? (int)s_total : 1);
    for (int e = (int)s_total + (int)threadIdx.x; e < n; e += kSelThreads) s[e] = 0ull;
    for (int e = n + (int)threadIdx.x; e < (int)KP; e += kSelThreads) s[e] = 0ull;  // (read below when fewer than KP exist)
    __syncthreads();
    wg_bitonic_desc(s, n);
    const uint32_t total = s_total;
    const uint32_t keep = total < KP ? total : KP;
    for (uint32_t i = threadIdx.x; i < KP; i += kSelThreads)
        out[((size_t)part * Q + q) * KP + i] = (i < keep) ? s[i] : 0ull;
    if (threadIdx.x == 0) out_cnt[(size_t)part * Q + q] = keep;
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
