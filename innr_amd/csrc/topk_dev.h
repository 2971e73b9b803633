// topk_dev.h -- wave-level candidate lists: threshold filter + append + rank-compaction.
//
// Replaces, on the device, the reference's "materialise N scores, stable-sort all of them, truncate(k)"
// (src/batch.rs:756-758, 792-794) and its streaming TopK (src/topk.rs:96-121): a producer wave keeps,
// per query, a list of 64-bit composites [pref | ~idx] (common.h) in global scratch, admits a score only
// if it is not worse than the list's threshold (the KP-th best it has proven to exist), and compacts the
// list to its best KP entries when it runs short of room. Anything rejected or dropped is strictly worse
// than KP other entries of the same query in (score, index) order, so it can never be in the top KP --
// the final select kernel therefore reproduces the stable-sort result exactly.
//
// All functions are wave-collective: every lane of the wave must reach them (wave-uniform control flow).
#pragma once

#include "common.h"

namespace innr {

__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, int lane) {
    uint32_t lo = __builtin_amdgcn_readlane((uint32_t)v, lane);
    uint32_t hi = __builtin_amdgcn_readlane((uint32_t)(v >> 32), lane);
    return ((uint64_t)hi << 32) | lo;
}

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        uint32_t o = (uint32_t)__shfl_xor((int)v, off, 64);
        v = v > o ? v : o;
    }
    return v;
}

// L1-bypassing 8-byte load (agent scope): the list was written by this wave's own earlier stores.
__device__ __forceinline__ uint64_t load_entry(const uint64_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Append one candidate. `cnt` lives in LDS and is private to the owning wave; lanes of that wave may
// append concurrently (LDS atomic hands out distinct slots). The compaction invariant keeps pos < cap;
// the guard turns a broken invariant into a reported error instead of an out-of-bounds store.
__device__ __forceinline__ void cand_append(uint64_t* list, uint32_t* cnt, uint32_t cap, uint64_t comp,
                                            uint32_t* errflag) {
    uint32_t pos = atomicAdd(cnt, 1u);
    if (pos < cap) list[pos] = comp;
    else atomicOr(errflag, 1u);
}

// ---- chip-wide threshold ("global threshold slots") ---------------------------------------------------
// With S producers per query each list's own threshold only knows 1/S of the corpus seen so far, so almost
// every tile still yields candidates. The producers therefore share, per query, S = kSlotMul*KP monotone slots:
//   slots[idx % S] = max pref of any admitted candidate whose corpus index falls in that residue class.
// Distinct slots were raised by distinct corpus vectors, so t = the KP-th largest slot value certifies "at least KP
// vectors score >= t": anything below t cannot be in the top KP, whichever producer sees it (ties pass).
// Everything is relaxed agent-scope atomicMax / loads: a stale or lost update only makes the bound weaker, never
// wrong. slots and gthr are zeroed before every launch (0 = "no bound").
// Two halves, so that the appending loop never waits on memory: gthr_raise is a fire-and-forget atomic max per admitted
// candidate; gthr_publish_select, once per query that admitted anything in a tile, re-reads the slots and raises the
// published bound. (One combined call per candidate -- load the slot, atomic max with return, rescan, load and raise
// the bound -- put 3-4 dependent L2 round trips on every append: tools/gemm_probe.hip measured 87K cycles per visit of
// the append path, 5 % of the C2 kernel.)
__device__ __forceinline__ void gthr_raise(uint32_t* slots, uint32_t KP, uint32_t pref, uint32_t idx) {
    (void)__hip_atomic_fetch_max(slots + (idx & (KP - 1)), pref, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // KP = 2^n
}
// Slots per query = kSlotMul * KP. With exactly KP slots the bound is the MINIMUM of KP bucket maxima, which sits near
// global rank KP*ln(KP) (coupon collector: every bucket must have been hit); with more slots the bound is the KP-th
// LARGEST bucket maximum -- still a certificate of KP distinct vectors -- and sits much closer to rank KP. 2*KP slots:
// measured at C2 with tools/gemm_probe.hip: KP = 32: 109.3 / 108.8 / 108.9 ms for 1 / 2 / 4 * KP slots, KP = 128:
// 114.5 / 112.7 / 114.3 (more slots = tighter bound but more loads and compares per re-derivation).
#ifndef INNR_SLOT_MUL
#define INNR_SLOT_MUL 2
#endif
constexpr uint32_t kSlotMul = INNR_SLOT_MUL;
// A visit of the append path re-derives the chip-wide bound of a query only when it admitted a candidate whose corpus
// index is a multiple of kPubEvery (a power of two): one admission in kPubEvery, chosen by a property of the data, not of
// the schedule. Every admission still raises its slot (gthr_raise), so nothing is lost, the published bound just lags.
// A re-derivation loads the query's slots, and that wait sits out everything the wave has in flight (the corpus DMA six steps
// ahead): on the fast pipes it is the expensive part of a visit. C2 shape, builds with -DINNR_PUB_EVERY (profiles/r02_metrics_*):
// bf16 filter kernel 14.49 ms at 4, 13.58 at 16, 13.61 at 32; the f32 kernel does not notice (106.6 ms either way).
#ifndef INNR_PUB_EVERY
#define INNR_PUB_EVERY 16
#endif
constexpr uint32_t kPubEvery = INNR_PUB_EVERY;

// The chip-wide bound of one query, re-derived from its S = kSlotMul*KP slot values (wave-uniform arguments). Two rules, both
// certificates, the better one wins:
//   KP rule  t_KP = the KP-th largest slot value: at least KP vectors score >= t_KP, so nothing below it is in the top KP.
//   k rule   t_k = the kk-th largest slot value (kk = the k the caller asked for): at least k vectors have an APPROXIMATE score
//            >= t_k, hence an exact score >= t_k - E (E = the query's bound on |approx - exact|), hence the k-th best EXACT
//            score s_k >= t_k - E, and a vector whose approximate score is below t_k - 2E has an exact score below s_k: it
//            is not in the top k. The caller passes kmargin = 2E (+ rounding room; +inf switches the rule off) and the bound
//            is set one key below t_k - kmargin, so that the re-score's proof "k-th exact > bound + E" is strict.
//   With the KP rule alone the bound sits near global rank 1.4 KP whatever the filter's precision; the k rule puts it where
//   the proof needs it: at C2 (k = 10, int8 filter, 2E = 3.1 against a gap of 5.0 between the 10th and the 128th score) near
//   rank 50 instead of 180, i.e. a third of the survivors per tile -- and the lists' length KP becomes a capacity, not the
//   quantity that sets the threshold. The re-score (rescore_kernel, rescore_u8_kernel) proves against the FINAL bound of the
//   query (gthr only grows, so every threshold a site was rejected with is <= it).
// Bisection on the 32 key bits with ballot counts: select(n) = max{x : #(slots >= x) >= n}; 0 while fewer than n are filled.
template <int NR>  // NR = S / 64 slot values per lane
__device__ __forceinline__ void gthr_publish_select(const uint32_t* slots, uint32_t* gthr_q, uint32_t KP, int lane,
                                                    uint32_t kk = 0u, float kmargin = __builtin_inff()) {
    uint32_t v[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) v[r] = __hip_atomic_load(slots + r * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    auto count_ge = [&](uint32_t x) -> uint32_t {
        uint32_t cnt = 0;
#pragma unroll
        for (int r = 0; r < NR; ++r) cnt += (uint32_t)__popcll(__ballot(v[r] >= x));
        return cnt;
    };
    auto select = [&](uint32_t n) -> uint32_t {
        uint32_t t = 0;
#pragma unroll 1
        for (int bit = 31; bit >= 0; --bit) {
            const uint32_t cand = t | (1u << bit);
            if (count_ge(cand) >= n) t = cand;
        }
        return t;
    };
    uint32_t g = 0;
    if (kk) {
        const uint32_t tk = select(kk);
        if (tk) {
            const float x = ord_f32(tk) - kmargin;
            if (x - x == 0.0f) g = f32_ord(x) - 1u;  // finite x: its key is >= 0x00800000
        }
    }
    uint32_t t = g;
    if (!g || count_ge(g) >= KP) {  // the KP rule may be the stronger one (always, while the k rule has nothing to say)
        const uint32_t tkp = select(KP);
        t = tkp > t ? tkp : t;
    }
    if (lane == 0 && t) (void)__hip_atomic_fetch_max(gthr_q, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The publish half, through the SCALAR memory path (slots, gthr_q, KP wave-uniform; KP a multiple of 32): s_load ... glc
// reads L2, where the atomics execute, and is tracked by lgkmcnt. A vector load would have to be waited for with
// vmcnt(0) -- vmcnt retires in order -- i.e. behind every prefetch the wave has in flight (gemm_filter_kernel: two
// K-steps of corpus DMA). Same speed as the wave-wide vector version at C2 (the DMA is usually back by then); kept
// because the append path now holds no vmcnt wait at all.
// Not ordered against this wave's own gthr_raise atomics (different path to L2): a raise that has not landed yet
// yields a smaller minimum, i.e. a weaker but still valid bound, and the next publish of that query picks it up.
typedef uint32_t u32x16_t __attribute__((ext_vector_type(16)));
__device__ __forceinline__ void gthr_publish_scalar(const uint32_t* slots, uint32_t* gthr_q, uint32_t KP, int lane) {
    uint32_t mn = 0xffffffffu;
    for (uint32_t j = 0; j < KP; j += 32) {
        u32x16_t a, b;
        uint64_t base;  // copied on the scalar unit first: the compiler adds no hazard padding in front of inline asm
        asm volatile(
            "s_mov_b64 %2, %3\n\t"
            "s_load_dwordx16 %0, %2, 0x0 glc\n\t"
            "s_load_dwordx16 %1, %2, 0x40 glc\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=&s"(a), "=&s"(b), "=&s"(base)
            : "s"(slots + j)
            : "memory");
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            mn = mn < a[i] ? mn : a[i];
            mn = mn < b[i] ? mn : b[i];
        }
    }
    if (lane == 0 && mn) (void)__hip_atomic_fetch_max(gthr_q, mn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Compact a list of `cnt` (<= 64*R) distinct composites to its best min(cnt, KP), written back sorted
// best-first at list[0..keep). Returns keep; *thr_pref = pref of the KP-th best (0 = "no threshold yet"
// while fewer than KP entries exist). Rank counting: rank(e) = #entries greater than e; entries are
// distinct (one per corpus index), so ranks are a permutation and rank < keep selects AND orders.
template <int R>
__device__ __forceinline__ uint32_t wave_compact(uint64_t* list, uint32_t cnt, uint32_t KP, uint32_t* thr_pref) {
    const int lane = threadIdx.x & 63;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's appends have reached L2
    uint64_t e[R];
    uint32_t rank[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        uint32_t slot = r * 64 + lane;
        e[r] = (slot < cnt) ? load_entry(list + slot) : 0ull;
        rank[r] = 0;
    }
#pragma unroll
    for (int r2 = 0; r2 < R; ++r2) {
        if ((uint32_t)(r2 * 64) < cnt) {  // wave-uniform
            const int lim = (cnt - r2 * 64) < 64u ? (int)(cnt - r2 * 64) : 64;
            for (int l = 0; l < lim; ++l) {
                const uint64_t b = readlane_u64(e[r2], l);
#pragma unroll
                for (int r = 0; r < R; ++r) rank[r] += (b > e[r]) ? 1u : 0u;
            }
        }
    }
    const uint32_t keep = cnt < KP ? cnt : KP;
    uint32_t t = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t slot = r * 64 + lane;
        if (slot < cnt && rank[r] < keep) list[rank[r]] = e[r];
        if (slot < cnt && rank[r] == KP - 1) t = cand_pref(e[r]);
    }
    *thr_pref = wave_max_u32(t);  // exactly one lane/reg holds rank KP-1 when cnt >= KP, else all 0
    return keep;
}

// rank of every re-scored candidate among the first `done` of them (composite order = score desc, index asc); candidate c lives
// in slot c / 64 of lane c % 64 (rescore_kernel, rescore_u8_kernel)
template <int RK>
__device__ __forceinline__ void rescore_rank(const uint64_t (&e)[RK], uint32_t done, uint32_t (&rank)[RK]) {
#pragma unroll
    for (int r = 0; r < RK; ++r) rank[r] = 0;
#pragma unroll
    for (int r2 = 0; r2 < RK; ++r2) {
        if ((uint32_t)(r2 * 64) < done) {
            const int lim = (done - r2 * 64) < 64u ? (int)(done - r2 * 64) : 64;
            for (int l = 0; l < lim; ++l) {
                const uint64_t bcast = readlane_u64(e[r2], l);
#pragma unroll
                for (int r = 0; r < RK; ++r) rank[r] += (bcast > e[r]) ? 1u : 0u;
            }
        }
    }
}


}  // namespace innr
