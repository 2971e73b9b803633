// kernels_maxsim.h -- ColBERT-style late interaction over a document corpus: maxsim / maxsim_cosine
// (src/maxsim.rs:96-194) for ONE query against EVERY document, exact.
//
// Reference per document: sum_i max_j dot(q_i, d_j)   (portable path maxsim.rs:142-152 -> dense::dot_portable,
// dense.rs:103-125: four strided accumulators, ((s0+s1)+s2)+s3, then a sequential tail; max = f32::max folded
// from -inf (ignores NaN); sum folded from -0.0 in query-token order). The caller loops over documents and sorts
// (examples/maxsim_colbert.rs:171-187); here one launch scores the whole corpus and a second selects the top-k.
//
// Roofline: HBM, 4*docs*T*dim bytes per query (C4: 32.8 GB); arithmetic intensity 2*Tq/4 = 16 flop/B at Tq = 32,
// i.e. the reference-order VALU formulation (mul and add as separate roundings: 2 VALU ops per MAC) needs
// ~1.2x the time HBM does. So this kernel keeps the reference's arithmetic bit for bit -- no approximate pass,
// no re-score, no proof -- and still sits within ~30 % of the memory roof.
//
// Mapping: document tokens stay in their natural layout tok[(doc*T + t)*dim + d]. One lane owns one document
// token (a wave = 64/Tp documents, Tp = T rounded up to a power of two <= 64) and keeps the 4 x Tq accumulators
// of dot_portable in registers; the query tokens are wave-uniform scalar loads. Each lane streams its own 512-B
// token row in 128-B bursts (8 x dwordx4 in flight: a burst is one cache line, fetched once).
#pragma once

#include <utility>

#include "common.h"

namespace innr {

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kMsThreads = 256;
constexpr int kMsQ = 32;       // query tokens per pass (4 accumulators each = 128 VGPRs)
constexpr int kMsBurst = 8;    // float4 chunks per lane per burst (128 B)

__device__ __forceinline__ float wave_max_seg(float v, int seg) {  // fmaxf over aligned groups of `seg` lanes
    for (int off = seg >> 1; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// COS: maxsim_cosine (maxsim.rs:168-194 -> dense::cosine_portable dense.rs:288-346): per pair
//   ab/(sqrt(aa)*sqrt(bb)) if aa > eps^2 and bb > eps^2 else 0, with aa, bb accumulated in the same 4-way order.
// partial_in / partial_out: running totals when the query has more than kMsQ tokens (passes in token order).
// NQ (8/16/32) = query tokens per pass, a compile-time bound so the hot loop has no per-token branch. The query
// buffer is zero-padded to NQ tokens; padded tokens are computed and ignored (only qi < nq enter the sum).
//
// Query operands. Every lane needs every query value q[qi][d] as a wave-uniform multiplier. Measured on 100K docs x
// 64 x 128, Tq = 32 (tools/maxsim_probe.hip, tools/valu_rate.hip):
//   * LDS broadcast reads (ds_read_b128 per (qi, chunk)): 32 reads per chunk per wave keep the CU's one LDS pipe
//     as busy as the VALU -- 4.3 ms;
//   * compiler-scheduled scalar loads (wave-uniform global reads -> s_load_dwordx4): hipcc serialises
//     {s_load, wait, 8 VALU} and cannot keep loads in flight -- 3.2 ms;
//   * query in LDS, one float4 per lane per two chunks, v_readlane_b32 into an SGPR before each mul: v_readlane
//     issues at ~5-6 cycles, 4 of them per 4 packed mul/add -- 2.3 ms, VALU-issue bound;
//   * THIS: the query is re-packed in global memory as qpk[chunk][qi][4] so that ONE s_load_dwordx16 brings the
//     operands of 4 query tokens x 4 dims (= 16 packed VALU ops); loads are issued by hand one group ahead of
//     their use, double-buffered in 2 x 16 SGPRs. The scalar cache holds the whole packed query (16 KB).
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Asynchronous: dst is valid only after swait0(dst). The compiler sees dst as defined here and keeps its
// registers untouched until swait0's tied operand "uses" them (checked in the ISA: no copy in between).
template <int BYTE_OFF>
__device__ __forceinline__ void sload16(f32x16& dst, const float* p) {
    asm volatile("s_load_dwordx16 %0, %1, %2" : "=s"(dst) : "s"(p), "n"(BYTE_OFF));
}
__device__ __forceinline__ void swait0(f32x16& v) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(v)); }

// 4 query tokens (qi0..qi0+3) x one 4-dim chunk: fl(acc + fl(q*d)) per element, as packed pairs
// (v_pk_mul_f32 + v_pk_add_f32; the library is built -ffp-contract=off).
template <int NQ>
__device__ __forceinline__ void maxsim_group(f32x2 (&acc)[NQ][2], const f32x16& q, int qi0, const float4& d) {
    const f32x2 d01 = {d.x, d.y}, d23 = {d.z, d.w};
    // two tokens at a time: four products, then four sums -- a dependent packed op issued within 3 slots of its
    // producer costs s_nops on gfx950
#pragma unroll
    for (int j = 0; j < 4; j += 2) {
        const f32x2 qa01 = {q[4 * j + 0], q[4 * j + 1]}, qa23 = {q[4 * j + 2], q[4 * j + 3]};
        const f32x2 qb01 = {q[4 * j + 4], q[4 * j + 5]}, qb23 = {q[4 * j + 6], q[4 * j + 7]};
        f32x2 ta01 = qa01 * d01, ta23 = qa23 * d23, tb01 = qb01 * d01, tb23 = qb23 * d23;
        asm volatile("" : "+v"(ta01), "+v"(ta23), "+v"(tb01), "+v"(tb23));
        acc[qi0 + j][0] = acc[qi0 + j][0] + ta01;
        acc[qi0 + j][1] = acc[qi0 + j][1] + ta23;
        acc[qi0 + j + 1][0] = acc[qi0 + j + 1][0] + tb01;
        acc[qi0 + j + 1][1] = acc[qi0 + j + 1][1] + tb23;
        // pin the results here: pure arithmetic would otherwise drift away from the loads that feed it
        asm volatile("" : "+v"(acc[qi0 + j][0]), "+v"(acc[qi0 + j][1]), "+v"(acc[qi0 + j + 1][0]), "+v"(acc[qi0 + j + 1][1]));
    }
}

__device__ __forceinline__ void maxsim_bb(float (&bb)[4], const float4& d) {
    bb[0] = ex::mad2(bb[0], d.x, d.x);
    bb[1] = ex::mad2(bb[1], d.y, d.y);
    bb[2] = ex::mad2(bb[2], d.z, d.z);
    bb[3] = ex::mad2(bb[3], d.w, d.w);
}

// One full burst: 8 chunks x NQ/4 groups, software-pipelined over the flat group sequence g = 0..G-1 (a pack
// expansion, so that every load offset is an immediate). qb = packed query at chunk c0: group (b, gi) sits at
// qb + (b*NQ + 4*gi)*4 floats.
// The groups go in PAIRS (round 3): scalar loads return out of order, so a wave can only ever wait for ALL of its loads
// (lgkmcnt(0)) -- with one s_load_dwordx16 requested per group, one group ahead, a load had the 16 packed VALU instructions of
// one group (~85 cycles, twice that with the SIMD's other wave) to come back in, and the scan ran at 32 T results/s against
// the 60 T/s the bare mul + add chains reach (tools/valu_rate.hip). Two loads per wait, for the pair after the current one,
// double the cover at the price of 32 more SGPRs (4 x 16 hold query operands).
template <int NQ>
__device__ __forceinline__ constexpr int maxsim_group_off(int g) {  // byte offset of group g's operands from qb
    return ((g / (NQ / 4)) * NQ + 4 * (g % (NQ / 4))) * 16;
}
__device__ __forceinline__ void swait0_pair(f32x16& a, f32x16& b) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(b)); }

#ifdef INNR_MS_PROBE_NOMATH  // tools/maxsim_probe.hip: loads only (one group's arithmetic per chunk keeps them live)
template <int NQ, bool COS, int g>
__device__ __forceinline__ void maxsim_step(f32x2 (&acc)[NQ][2], float (&bb)[4], const float4 (&dv)[kMsBurst],
                                            const float* qb, f32x16 (&buf)[2]) {
    constexpr int GPC = NQ / 4, G = kMsBurst * GPC, b = g / GPC, gi = g % GPC;
    if (gi > 0) return;
    swait0(buf[b & 1]);
    if (g + GPC < G) sload16<((g + GPC) / GPC) * NQ * 16>(buf[(b + 1) & 1], qb);
    maxsim_group<NQ>(acc, buf[b & 1], 0, dv[b]);
}
template <int NQ, bool COS, int... g>
__device__ __forceinline__ void maxsim_burst_seq(f32x2 (&acc)[NQ][2], float (&bb)[4], const float4 (&dv)[kMsBurst],
                                                 const float* qb, std::integer_sequence<int, g...>) {
    f32x16 buf[2];
    sload16<0>(buf[0], qb);
    (maxsim_step<NQ, COS, g>(acc, bb, dv, qb, buf), ...);
}
template <int NQ, bool COS>
__device__ __forceinline__ void maxsim_burst(f32x2 (&acc)[NQ][2], float (&bb)[4], const float4 (&dv)[kMsBurst],
                                             const float* qb) {
    maxsim_burst_seq<NQ, COS>(acc, bb, dv, qb, std::make_integer_sequence<int, kMsBurst * (NQ / 4)>{});
}
#else
template <int NQ, bool COS, int p>
__device__ __forceinline__ void maxsim_step2(f32x2 (&acc)[NQ][2], float (&bb)[4], const float4 (&dv)[kMsBurst],
                                             const float* qb, f32x16 (&buf)[2][2]) {
    constexpr int GPC = NQ / 4, G = kMsBurst * GPC, g0 = 2 * p, g1 = 2 * p + 1;
    static_assert(G % 2 == 0, "groups are visited in pairs");
    swait0_pair(buf[p & 1][0], buf[p & 1][1]);
#ifdef INNR_MS_PROBE_NOSLOAD  // tools/maxsim_probe.hip: the arithmetic on stale operands -- what the scalar loads cost
    if (p == 0) {
#else
    if (g0 + 2 < G) {
#endif
        sload16<maxsim_group_off<NQ>(g0 + 2)>(buf[(p + 1) & 1][0], qb);
        sload16<maxsim_group_off<NQ>(g1 + 2)>(buf[(p + 1) & 1][1], qb);
    }
    if (COS && g0 % GPC == 0) maxsim_bb(bb, dv[g0 / GPC]);
    maxsim_group<NQ>(acc, buf[p & 1][0], 4 * (g0 % GPC), dv[g0 / GPC]);
    if (COS && g1 % GPC == 0) maxsim_bb(bb, dv[g1 / GPC]);
    maxsim_group<NQ>(acc, buf[p & 1][1], 4 * (g1 % GPC), dv[g1 / GPC]);
}
template <int NQ, bool COS, int... p>
__device__ __forceinline__ void maxsim_burst_seq(f32x2 (&acc)[NQ][2], float (&bb)[4], const float4 (&dv)[kMsBurst],
                                                 const float* qb, std::integer_sequence<int, p...>) {
    f32x16 buf[2][2];
    sload16<maxsim_group_off<NQ>(0)>(buf[0][0], qb);
    sload16<maxsim_group_off<NQ>(1)>(buf[0][1], qb);
    (maxsim_step2<NQ, COS, p>(acc, bb, dv, qb, buf), ...);
}
template <int NQ, bool COS>
__device__ __forceinline__ void maxsim_burst(f32x2 (&acc)[NQ][2], float (&bb)[4], const float4 (&dv)[kMsBurst],
                                             const float* qb) {
    maxsim_burst_seq<NQ, COS>(acc, bb, dv, qb, std::make_integer_sequence<int, kMsBurst * (NQ / 4) / 2>{});
}
#endif

__device__ __forceinline__ void maxsim_load_burst(float4 (&dv)[kMsBurst], const float* __restrict__ row, uint32_t c0, int lane) {
#pragma unroll
    for (int b = 0; b < kMsBurst; ++b) {
#ifdef INNR_MS_PROBE_NOLOAD  // tools/maxsim_probe.hip: arithmetic only
        dv[b] = make_float4((float)(c0 + b), (float)lane, 1.0f, 2.0f);
#else
        dv[b] = *reinterpret_cast<const float4*>(row + 4 * (c0 + b));
#endif
    }
}

// The max over a document's TP token lanes for NQ query tokens at once, as a TRANSPOSING butterfly: a plain butterfly per query
// token costs log2(TP) exchanges each (32 x 6 at C4); here the step with lane offset OFF also halves what a lane holds -- lanes
// with bit OFF clear keep the lower half of the values, the others the upper half, each sends the half it gives up -- until
// one value per lane is left (then, or at offset 1, plain steps): NQ/2 + NQ/4 + ... exchanges instead of NQ log2(TP). Query
// token qi's maximum ends up in slot ms_tmax_slot(qi) of the segment's lanes whose bits match ms_tmax_lane(qi).
// (max is order-free here: v_max_f32 ignores a NaN operand unless both are, and orders -0 < +0.)
template <int NQ, int N, int OFF>
__device__ __forceinline__ void ms_tmax(float (&v)[NQ], int lane) {
    if constexpr (OFF >= 1) {
        if constexpr (N > 1 && OFF >= 2) {
            constexpr int H = N / 2;
            const bool up = (lane & OFF) != 0;
#pragma unroll
            for (int i = 0; i < H; ++i) {
                const float send = up ? v[i] : v[i + H];
                const float keep = up ? v[i + H] : v[i];
                v[i] = fmaxf(keep, __shfl_xor(send, OFF, 64));
            }
            ms_tmax<NQ, H, OFF / 2>(v, lane);
        } else {
#pragma unroll
            for (int i = 0; i < N; ++i) v[i] = fmaxf(v[i], __shfl_xor(v[i], OFF, 64));
            ms_tmax<NQ, N, OFF / 2>(v, lane);
        }
    }
}
constexpr int ms_tmax_left(int NQ, int TP) {  // values per lane after the transposing steps
    int n = NQ;
    for (int off = TP / 2; off >= 2; off /= 2)
        if (n > 1) n /= 2;
    return n;
}
constexpr int ms_tmax_lane(int NQ, int TP, int qi) {  // lane bits (inside the segment) of the lanes that hold query token qi
    int n = NQ, idx = qi, bits = 0;
    for (int off = TP / 2; off >= 2; off /= 2)
        if (n > 1) {
            n /= 2;
            if (idx >= n) { bits |= off; idx -= n; }
        }
    return bits;
}
constexpr int ms_tmax_slot(int NQ, int TP, int qi) {
    int n = NQ, idx = qi;
    for (int off = TP / 2; off >= 2; off /= 2)
        if (n > 1) {
            n /= 2;
            if (idx >= n) idx -= n;
        }
    return idx;
}

// qpk[(c*NQ + qi)*4 + e] = qtok[qi*dim + 4c + e] (qi >= nq_valid rows are zero in qtok already)
__global__ void maxsim_pack_query_kernel(const float* __restrict__ qtok, uint32_t NQ, uint32_t dim, float* __restrict__ qpk) {
    const uint32_t chunks = dim / 4;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= chunks * NQ * 4) return;
    const uint32_t e = i & 3, qi = (i >> 2) % NQ, c = (i >> 2) / NQ;
    qpk[i] = qtok[(size_t)qi * dim + 4 * c + e];
}

// MULTI: documents longer than 64 tokens (several token groups per document, a running max per query token kept
// across them: 32 more live registers, so it is a separate instantiation).
template <bool COS, int NQ, bool MULTI>
__global__ __launch_bounds__(kMsThreads, 2) void maxsim_scan_kernel(
    const float* __restrict__ tok, const uint32_t* __restrict__ doc_len, uint32_t ndocs, uint32_t T, uint32_t Tp,
    uint32_t dim, const float* __restrict__ qtok /*[NQ][dim], zero-padded*/,
    const float* __restrict__ qpk /*[dim/4][NQ][4] packed copy*/, uint32_t nq,
    const float* __restrict__ q_aa /*[nq] COS*/, const float* __restrict__ partial_in, float* __restrict__ out,
    bool first_pass, const uint32_t* __restrict__ doc_ids /*null: slot s is document s; else document doc_ids[s]*/,
    const float* __restrict__ q_saa = nullptr /*[nq] COS: sqrt(q_aa), the same rounding, once per query instead of per lane*/) {
    const int lane = threadIdx.x & 63;
    const uint32_t docs_per_wave = 64 / Tp;
    const uint32_t wave = (blockIdx.x * kMsThreads + threadIdx.x) >> 6;
    const uint32_t nwaves = (gridDim.x * kMsThreads) >> 6;
    const uint32_t chunks = dim / 4;
    // the epilogue's form (wave-uniform, fixed for the launch): 1 = the transposing butterfly over a compile-time segment width
    const int tmode = (dim % 4 == 0 && (Tp == 64 || Tp == 32 || Tp == 16 || Tp == 8)) ? 1 : 0;
    for (uint32_t dbase = wave * docs_per_wave; dbase < ndocs; dbase += nwaves * docs_per_wave) {
        const uint32_t slot = dbase + lane / Tp;  // output position; ndocs = number of slots
        const uint32_t doc = (slot < ndocs && doc_ids) ? doc_ids[slot] : slot;
        const uint32_t t = lane % Tp;
        const uint32_t len = (slot < ndocs) ? (doc_len ? min(doc_len[doc], T) : T) : 0;
        // documents with T > 64 tokens: walk the tokens in groups of Tp = 64, keeping a running max per query token
        float best[NQ];
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi) best[qi] = -INFINITY;
        for (uint32_t t0 = 0; t0 < (MULTI ? T : 1u); t0 += Tp) {
            const bool live = (t0 + t) < len;
            // lanes without a token read token 0 of the corpus (always mapped) and are masked out of the max below:
            // no per-load predication in the hot loop
            const float* row = tok + (live ? ((size_t)doc * T + t0 + t) * dim : (size_t)0);
            f32x2 acc[NQ][2];
            float bb[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int qi = 0; qi < NQ; ++qi) acc[qi][0] = acc[qi][1] = f32x2{0.0f, 0.0f};
            // bursts of 8 chunks (32 dims = one 128-B line of this lane's token row), the next burst's line already
            // in flight while this one is multiplied; per accumulator the chunks are visited in ascending order,
            // which is all dot_portable's result depends on
            uint32_t c0 = 0;
            float4 dv[kMsBurst], dn[kMsBurst];
            if (kMsBurst <= chunks) maxsim_load_burst(dv, row, 0, lane);
#pragma unroll 1
            for (; c0 + kMsBurst <= chunks; c0 += kMsBurst) {
                const bool more = c0 + 2 * kMsBurst <= chunks;  // wave-uniform
                if (more) maxsim_load_burst(dn, row, c0 + kMsBurst, lane);
                maxsim_burst<NQ, COS>(acc, bb, dv, qpk + (size_t)c0 * NQ * 4);
                if (more) {
#pragma unroll
                    for (int b = 0; b < kMsBurst; ++b) dv[b] = dn[b];
                }
            }
            // remaining chunks (dim not a multiple of 32): one group at a time, not pipelined
            for (; c0 < chunks; ++c0) {
                const float4 d4 = *reinterpret_cast<const float4*>(row + 4 * c0);
                if (COS) maxsim_bb(bb, d4);
#pragma unroll
                for (int gi = 0; gi < NQ / 4; ++gi) {
                    f32x16 q;
                    sload16<0>(q, qpk + ((size_t)c0 * NQ + 4 * gi) * 4);
                    swait0(q);
                    maxsim_group<NQ>(acc, q, 4 * gi, d4);
                }
            }
            // ((s0+s1)+s2)+s3 then the sequential tail (dense.rs:119-124)
            float sbb = ex::add(ex::add(ex::add(bb[0], bb[1]), bb[2]), bb[3]);
            float tailv[3] = {0.f, 0.f, 0.f};
            const uint32_t ntail = dim - chunks * 4;
            for (uint32_t e = 0; e < ntail; ++e) {
                tailv[e] = row[chunks * 4 + e];
                if (COS) sbb = ex::mad2(sbb, tailv[e], tailv[e]);
            }
#ifdef INNR_MS_PROBE_NOEPI  // tools/maxsim_probe.hip: the hot loop alone (the accumulators folded into one value per lane)
            {
                float f = 0.0f;
#pragma unroll
                for (int qi = 0; qi < NQ; ++qi) f += (acc[qi][0].x + acc[qi][0].y) + (acc[qi][1].x + acc[qi][1].y);
                best[0] = fmaxf(best[0], f + sbb + (float)ntail);
            }
#else
            // Per query token: the lane's dot product, then the max over the document's Tp token lanes. With the segment width a
            // run-time value the 32 butterflies were 32 loops (and the tail's per-token pointers 146 spilled SGPRs): this epilogue
            // cost 31 % of the kernel at C4 (tools/maxsim_probe.hip, -DINNR_MS_PROBE_NOEPI: 3.20 -> 2.19 ms on 200 K documents).
            // Now one wave-uniform switch picks straight-line code for the segment width -- the transposing butterfly ms_tmax --;
            // a dimension that is not a multiple of four, or fewer than 8 tokens per document, keep the general form.
            const float sqrt_bb = COS ? ex::sqrt(sbb) : 0.0f;
            auto epilogue = [&](auto tp_tag, auto tail_tag) {
                constexpr int TPC = decltype(tp_tag)::value;  // 0: run-time width
                constexpr bool TAIL = decltype(tail_tag)::value;
                float val[NQ];
#pragma unroll
                for (int qi = 0; qi < NQ; ++qi) {
                    if ((uint32_t)qi < nq) {
                        float sdot = ex::add(ex::add(ex::add(acc[qi][0].x, acc[qi][0].y), acc[qi][1].x), acc[qi][1].y);
                        if (TAIL)
                            for (uint32_t e = 0; e < ntail; ++e) sdot = ex::mad2(sdot, qtok[(size_t)qi * dim + chunks * 4 + e], tailv[e]);
                        float sc = sdot;
                        if (COS) {  // dense.rs:341-345
                            const float aa = q_aa[qi];
                            constexpr float kEpsSq = INNR_NORM_EPSILON * INNR_NORM_EPSILON;  // lib.rs:184
                            sc = (aa > kEpsSq && sbb > kEpsSq) ? ex::div(sdot, ex::mul(q_saa ? q_saa[qi] : ex::sqrt(aa), sqrt_bb)) : 0.0f;
                        }
                        sc = live ? sc : -INFINITY;  // tokens beyond the document's length do not take part in the max
                        if (TPC == 0) best[qi] = fmaxf(best[qi], wave_max_seg(sc, (int)Tp));
                        else val[qi] = sc;
                    } else if (TPC != 0) {
                        val[qi] = -INFINITY;  // (a padded query token: computed, never summed)
                    }
                }
                if constexpr (TPC != 0) {  // transposed: best[s] = running max of slot s (ms_tmax)
                    ms_tmax<NQ, NQ, TPC / 2>(val, lane);
#pragma unroll
                    for (int s2 = 0; s2 < ms_tmax_left(NQ, TPC); ++s2) best[s2] = fmaxf(best[s2], val[s2]);
                }
            };
            using std::integral_constant;
            if (tmode == 0 && ntail != 0) epilogue(integral_constant<int, 0>(), integral_constant<bool, true>());
            else if (tmode == 0) epilogue(integral_constant<int, 0>(), integral_constant<bool, false>());
            else if (Tp == 64) epilogue(integral_constant<int, 64>(), integral_constant<bool, false>());
            else if (Tp == 32) epilogue(integral_constant<int, 32>(), integral_constant<bool, false>());
            else if (Tp == 16) epilogue(integral_constant<int, 16>(), integral_constant<bool, false>());
            else epilogue(integral_constant<int, 8>(), integral_constant<bool, false>());
#endif
        }
        // sum over query tokens in token order, folded from -0.0 (first pass) or from the previous passes' total
        float total = (t == 0 && slot < ndocs && !first_pass) ? partial_in[slot] : -0.0f;
        auto fold = [&](auto tp_tag) {
            constexpr int TPC = decltype(tp_tag)::value;
#pragma unroll
            for (int qi = 0; qi < NQ; ++qi)
                if ((uint32_t)qi < nq) {
                    float b;
                    if constexpr (TPC == 0) b = best[qi];
                    else b = __shfl(best[ms_tmax_slot(NQ, TPC, qi)], (lane & ~(TPC - 1)) | ms_tmax_lane(NQ, TPC, qi), 64);  // (every lane takes part)
                    total = ex::add(total, b);
                }
        };
        if (tmode == 0) fold(std::integral_constant<int, 0>());
        else if (Tp == 64) fold(std::integral_constant<int, 64>());
        else if (Tp == 32) fold(std::integral_constant<int, 32>());
        else if (Tp == 16) fold(std::integral_constant<int, 16>());
        else fold(std::integral_constant<int, 8>());
        if (t == 0 && slot < ndocs) out[slot] = (len == 0) ? 0.0f : total;  // empty document -> 0.0 (maxsim.rs:97-99)
    }
}

__global__ void sqrt_kernel(const float* __restrict__ x, uint32_t n, float* __restrict__ y) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = ex::sqrt(x[i]);
}

// aa[i] = sum of squares of query token i in cosine_portable's 4-way order (dense.rs:288-339)
__global__ void query_token_sq_kernel(const float* __restrict__ qtok, uint32_t nq, uint32_t dim, float* __restrict__ aa) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    const float* a = qtok + (size_t)i * dim;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    const uint32_t chunks = dim / 4;
    for (uint32_t c = 0; c < chunks; ++c)
        for (int e = 0; e < 4; ++e) s[e] = ex::mad2(s[e], a[4 * c + e], a[4 * c + e]);
    float r = ex::add(ex::add(ex::add(s[0], s[1]), s[2]), s[3]);
    for (uint32_t d = chunks * 4; d < dim; ++d) r = ex::mad2(r, a[d], a[d]);
    aa[i] = r;
}

// ---- MFMA engine: approximate scores of every document, then exact re-score of the best (api.hip) ---------------
// D[i][j] = sum_k A[i][k] B[k][j] on v_mfma_f32_32x32x2_f32 with A = 32 document tokens (rows i), B = 32 query tokens
// (columns j). Lane l feeds A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31] and receives column j = l&31, rows
// i = (g&3) + 8(g>>2) + 4(l>>5), g = 0..15: the max over document tokens is an in-lane max over the 16 accumulator
// registers plus one exchange between the wave halves; the sum over query tokens is a 32-lane reduction.
// K order: the sum over dimensions is order-free here, so dimension 8c + 4h + e is K-step (4c + e), half h: every
// lane loads ONE float4 (16 B) of its token row per 8 dimensions, and a token's 128-B line is consumed by 4
// consecutive loads of lanes i and i+32. Query operands wait in LDS in exactly that order (one ds_read_b128 per 4
// MFMAs). Requires dim % 8 == 0.
typedef float msf32x16 __attribute__((ext_vector_type(16)));

// qB[(c*64 + l)*4 + e] = q[j = l&31][8c + 4(l>>5) + e]; rows j >= nq are zero in qtok (padded buffer)
__global__ void maxsim_pack_mfma_kernel(const float* __restrict__ qtok, uint32_t dim, float* __restrict__ qB) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= dim * 32) return;
    const uint32_t e = i & 3, l = (i >> 2) & 63, c = i >> 8;
    qB[i] = qtok[(size_t)(l & 31) * dim + 8 * c + 4 * (l >> 5) + e];
}

// per token: bb = sum of squares in cosine_portable's order (dense.rs:288-339), inv[t] = bb > eps^2 ? 1/sqrt(bb) : 0
// (the reference's zero-norm guard, dense.rs:341-345) and the corpus-wide maximum of sqrt(bb) (as uint bits: norms
// are non-negative, NaN compares above everything and poisons the maximum on purpose)
__global__ void maxsim_token_norms_kernel(const float* __restrict__ tok, size_t ntok, uint32_t dim, float* __restrict__ inv,
                                          uint32_t* __restrict__ max_bits) {
    const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    float nrm = 0.0f;
    if (r < ntok) {
        const float* a = tok + r * dim;
        float s4[4] = {0.f, 0.f, 0.f, 0.f};
        const uint32_t chunks = dim / 4;
        for (uint32_t c = 0; c < chunks; ++c) {
            const float4 v = *reinterpret_cast<const float4*>(a + 4 * c);
            s4[0] = ex::mad2(s4[0], v.x, v.x);
            s4[1] = ex::mad2(s4[1], v.y, v.y);
            s4[2] = ex::mad2(s4[2], v.z, v.z);
            s4[3] = ex::mad2(s4[3], v.w, v.w);
        }
        float bb = ex::add(ex::add(ex::add(s4[0], s4[1]), s4[2]), s4[3]);
        for (uint32_t d = chunks * 4; d < dim; ++d) bb = ex::mad2(bb, a[d], a[d]);
        constexpr float kEpsSq = INNR_NORM_EPSILON * INNR_NORM_EPSILON;
        nrm = ex::sqrt(bb);
        inv[r] = (bb > kEpsSq) ? ex::div(1.0f, nrm) : 0.0f;
    }
    uint32_t m = __float_as_uint(nrm);
    for (int off = 32; off >= 1; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off, 64));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(max_bits, m);
}

// COS: q_scale[j] = aa_j > eps^2 ? 1/sqrt(aa_j) : 0 from the exact squared norms of the query tokens
__global__ void maxsim_query_scale_kernel(const float* __restrict__ aa, uint32_t n, float* __restrict__ scale) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    constexpr float kEpsSq = INNR_NORM_EPSILON * INNR_NORM_EPSILON;
    scale[i] = (aa[i] > kEpsSq) ? ex::div(1.0f, ex::sqrt(aa[i])) : 0.0f;
}

template <bool COS>
__global__ __launch_bounds__(kMsThreads) void maxsim_mfma_kernel(
    const float* __restrict__ tok, const uint32_t* __restrict__ doc_len, const float* __restrict__ tok_inv /*COS*/,
    uint32_t ndocs, uint32_t T, uint32_t dim, const float* __restrict__ qB /*[dim/8][64][4]*/, uint32_t nq,
    const float* __restrict__ q_scale /*[32] COS*/, const float* __restrict__ partial_in, float* __restrict__ out,
    bool first_pass) {
    extern __shared__ __attribute__((aligned(16))) float s_qB[];
    const int lane = threadIdx.x & 63;
    const uint32_t nch = dim / 8;
    for (uint32_t i = threadIdx.x; i < nch * 64; i += kMsThreads)
        reinterpret_cast<float4*>(s_qB)[i] = reinterpret_cast<const float4*>(qB)[i];
    __syncthreads();
    const float4* bl = reinterpret_cast<const float4*>(s_qB) + lane;
    const uint32_t wave = (blockIdx.x * kMsThreads + threadIdx.x) >> 6;
    const uint32_t nwaves = (gridDim.x * kMsThreads) >> 6;
    const uint32_t j = lane & 31, h = lane >> 5;
    const float qs = COS ? q_scale[j] : 1.0f;
    for (uint32_t doc = wave; doc < ndocs; doc += nwaves) {  // wave-uniform
        const uint32_t len = doc_len ? min(doc_len[doc], T) : T;
        float best = -INFINITY;  // max over this document's tokens of (token . query token j)
        for (uint32_t t0 = 0; t0 < len; t0 += 32) {
            // rows past the document's end re-read its last token (mapped memory) and are masked out of the max
            const uint32_t r = min(t0 + j, T - 1);
            const float* rowp = tok + ((size_t)doc * T + r) * dim + 4 * h;
            msf32x16 acc;
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[g] = 0.0f;
            float4 a[4], an[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if ((uint32_t)u < nch) a[u] = *reinterpret_cast<const float4*>(rowp + 8 * u);
            for (uint32_t cb = 0; cb < nch; cb += 4) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (cb + 4 + u < nch) an[u] = *reinterpret_cast<const float4*>(rowp + 8 * (cb + 4 + u));
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (cb + u < nch) {
                        const float4 b4 = bl[(size_t)(cb + u) * 64];
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].x, b4.x, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].y, b4.y, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].z, b4.z, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].w, b4.w, acc, 0, 0, 0);
                    }
#pragma unroll
                for (int u = 0; u < 4; ++u) a[u] = an[u];
            }
            float m = -INFINITY;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const uint32_t row = t0 + (g & 3) + 8 * (g >> 2) + 4 * h;
                float v = acc[g];
                if (COS) v *= tok_inv[(size_t)doc * T + min(row, T - 1)];
                m = (row < len) ? fmaxf(m, v) : m;
            }
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            best = fmaxf(best, m);
        }
        float v = (j < nq) ? best * qs : 0.0f;
#pragma unroll
        for (int off = 16; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0) {
            float total = (first_pass ? 0.0f : partial_in[doc]) + v;
            if (total != total) total = INFINITY;  // a NaN anywhere: force the document into the exact re-score
            out[doc] = (len == 0) ? 0.0f : total;
        }
    }
}

// ---- MFMA engine, tile-unrolled variant for dim = 32 * NB, NB <= 4 (ColBERT's 128) ------------------------------
// Same mathematics as maxsim_mfma_kernel; the difference is the memory pipeline. The generic kernel lets the compiler
// place the waits and ends every 32-dim burst on s_waitcnt vmcnt(0) + 16 register moves (its next-burst buffer is
// copied into the current one), i.e. one burst of lead. Here a whole 32-token tile lives in NB x 4 float4 registers per
// lane, every (tile, burst) has its own registers, and burst b of the NEXT tile is requested the moment burst b of the
// current one has been multiplied: NB-1 bursts of lead across tile and document boundaries. All loads are plain C
// (every VMEM op visible to the compiler, no register copies of in-flight loads), so hipcc's own s_waitcnt vmcnt(N)
// comes out as the exact count of younger requests; past its last tile a wave re-requests the current one (mapped
// addresses, registers nobody reads) so that the loop body is branch-free.
// NQ queries per corpus pass (1, 2 or 4; all with <= 32 tokens): the tile is multiplied against each query's operands
// before its registers are re-requested, so the corpus is streamed once for NQ queries and the pass turns MFMA-bound
// (NQ x 64 MFMAs per 16 KiB tile). Query qi's operands sit at qB + qi * (4*NB*64*4) floats, its token count in nq[qi],
// its scale row at q_scale + 32*qi, its scores in out + qi * ndocs. partial_in/first_pass (queries of more than 32
// tokens, accumulated over passes) are honoured for NQ == 1 only.
template <bool COS, int NB, int NQ>
__global__ __launch_bounds__(kMsThreads) void maxsim_mfma_tile_kernel(
    const float* __restrict__ tok, const uint32_t* __restrict__ doc_len, const float* __restrict__ tok_inv /*COS, padded*/,
    uint32_t ndocs, uint32_t T, const float* __restrict__ qB /*[NQ][4*NB][64][4]*/, const uint32_t* __restrict__ nq /*[NQ]*/,
    const float* __restrict__ q_scale /*[NQ][32] COS*/, const float* __restrict__ partial_in, float* __restrict__ out,
    bool first_pass) {
    constexpr uint32_t dim = 32 * NB, nch = 4 * NB;
    extern __shared__ __attribute__((aligned(16))) float s_qB[];
    const int lane = threadIdx.x & 63;
    for (uint32_t i = threadIdx.x; i < NQ * nch * 64; i += kMsThreads)
        reinterpret_cast<float4*>(s_qB)[i] = reinterpret_cast<const float4*>(qB)[i];
    __syncthreads();
    const float4* bl = reinterpret_cast<const float4*>(s_qB) + lane;
    // wave index as an SGPR value: document bookkeeping (lengths, loop control) then stays on the scalar unit --
    // as a VGPR value the doc_len read became a vector load whose use drained every token load in flight
    const uint32_t wave = __builtin_amdgcn_readfirstlane((blockIdx.x * kMsThreads + threadIdx.x) >> 6);
    const uint32_t nwaves = (gridDim.x * kMsThreads) >> 6;
    const uint32_t j = lane & 31, h = lane >> 5;
    float qs[NQ];
    uint32_t nqv[NQ];
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) {
        qs[qi] = COS ? q_scale[32 * qi + j] : 1.0f;
        nqv[qi] = nq[qi];
    }
    if (wave >= ndocs) return;  // wave-uniform, after the only barrier
    auto doclen = [&](uint32_t d) { return doc_len ? min(doc_len[d], T) : T; };
    // rows past the document's end re-read its last token (mapped memory) and are masked out of the max
    auto rowptr = [&](uint32_t d, uint32_t t0) { return tok + ((size_t)d * T + min(t0 + j, T - 1)) * dim + 4 * h; };
    auto invptr = [&](uint32_t d, uint32_t t0, int q) { return tok_inv + (size_t)d * T + t0 + 8 * q + 4 * h; };
    // 4 consecutive floats at a 4-byte-aligned address (doc*T need not be a multiple of 4)
    auto ldinv = [&](const float* p) { return make_float4(p[0], p[1], p[2], p[3]); };

    uint32_t cdoc = wave, ct0 = 0, clen = doclen(cdoc);
    const float* rp = rowptr(cdoc, 0);
    float4 a[NB][4], iv[4];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int u = 0; u < 4; ++u) a[b][u] = *reinterpret_cast<const float4*>(rp + 32 * b + 8 * u);
    if (COS) {
#pragma unroll
        for (int q = 0; q < 4; ++q) iv[q] = ldinv(invptr(cdoc, 0, q));
    }
    float best[NQ];
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) best[qi] = -INFINITY;
    while (true) {  // every exit condition is wave-uniform
        uint32_t ndoc = cdoc, nt0 = ct0 + 32, nlen = clen;
        if (nt0 >= clen) {
            ndoc = cdoc + nwaves;
            nt0 = 0;
            nlen = ndoc < ndocs ? doclen(ndoc) : 0;
        }
        const bool has_next = ndoc < ndocs;
        const float* nrp = has_next ? rowptr(ndoc, nt0) : rp;
        msf32x16 acc[NQ];
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi)
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[qi][g] = 0.0f;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int qi = 0; qi < NQ; ++qi) {
                    const float4 b4 = bl[(size_t)(qi * nch + 4 * b + u) * 64];
                    acc[qi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[b][u].x, b4.x, acc[qi], 0, 0, 0);
                    acc[qi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[b][u].y, b4.y, acc[qi], 0, 0, 0);
                    acc[qi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[b][u].z, b4.z, acc[qi], 0, 0, 0);
                    acc[qi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[b][u].w, b4.w, acc[qi], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 4; ++u) a[b][u] = *reinterpret_cast<const float4*>(nrp + 32 * b + 8 * u);
        }
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi) {
            float m = -INFINITY;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const uint32_t row = ct0 + (g & 3) + 8 * (g >> 2) + 4 * h;
                float v = acc[qi][g];
                if (COS) {
                    const float4 t = iv[g >> 2];
                    v *= (g & 3) == 0 ? t.x : ((g & 3) == 1 ? t.y : ((g & 3) == 2 ? t.z : t.w));
                }
                m = (row < clen) ? fmaxf(m, v) : m;
            }
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            best[qi] = fmaxf(best[qi], m);
        }
        if (COS) {
#pragma unroll
            for (int q = 0; q < 4; ++q) iv[q] = ldinv(has_next ? invptr(ndoc, nt0, q) : invptr(cdoc, ct0, q));
        }
        if (ndoc != cdoc) {  // last tile of this document: sum over query tokens, publish
#pragma unroll
            for (int qi = 0; qi < NQ; ++qi) {
                float v = (j < nqv[qi]) ? best[qi] * qs[qi] : 0.0f;
#pragma unroll
                for (int off = 16; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
                if (lane == 0) {
                    float total = ((NQ > 1 || first_pass) ? 0.0f : partial_in[cdoc]) + v;
                    if (total != total) total = INFINITY;  // a NaN anywhere: force the document into the exact re-score
                    out[(size_t)qi * ndocs + cdoc] = (clen == 0) ? 0.0f : total;
                }
                best[qi] = -INFINITY;
            }
        }
        if (!has_next) break;
        cdoc = ndoc;
        ct0 = nt0;
        clen = nlen;
        rp = nrp;
    }
}

// Re-score epilogue (one block of 256 threads, ncand <= 256): order the candidates by (exact score desc, document
// index asc), emit the best kout, and PROVE them: every document outside the candidate set has approximate score
// <= t_approx (the worst candidate's), hence exact score <= t_approx + E; if the kout-th exact score is strictly
// above that, no outsider belongs in the answer. flag[0] = 1 when proven (or when every document was a candidate).
__global__ __launch_bounds__(256) void maxsim_finish_kernel(const uint64_t* __restrict__ sel /*approx composites, best first*/,
                                                            const uint32_t* __restrict__ sel_cnt, const float* __restrict__ exact,
                                                            uint32_t kout, uint32_t ndocs, float E, uint64_t index_base,
                                                            uint64_t* __restrict__ out_idx, float* __restrict__ out_score,
                                                            uint32_t* __restrict__ flag) {
    __shared__ uint64_t s_c[256];
    __shared__ float s_kth;
    const uint32_t n = min(sel_cnt[0], 256u), t = threadIdx.x;
    uint64_t mine = 0;
    if (t < n) mine = cand_make(f32_ord(exact[t]), cand_idx(sel[t]));
    s_c[t] = mine;
    __syncthreads();
    if (t < n) {
        uint32_t rank = 0;
        for (uint32_t o = 0; o < n; ++o) rank += s_c[o] > mine;
        if (rank < kout) {
            out_idx[rank] = index_base + cand_idx(mine);
            out_score[rank] = exact[t];
            if (rank == kout - 1) s_kth = exact[t];
        }
    }
    __syncthreads();
    if (t == 0) {
        bool ok = n >= kout;
        if (ok && n < ndocs) {
            const float t_approx = pref_score(cand_pref(sel[n - 1]), false);
            const float bound = t_approx + E;
            ok = (s_kth - s_kth == 0.0f) && (bound - bound == 0.0f) && s_kth > bound;
        }
        flag[0] = ok ? 1u : 0u;
    }
}

// candidate document ids (uint32) from the selected composites
__global__ void maxsim_cand_ids_kernel(const uint64_t* __restrict__ sel, const uint32_t* __restrict__ sel_cnt, uint32_t KP,
                                       uint32_t* __restrict__ ids) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < KP) ids[t] = (t < sel_cnt[0]) ? cand_idx(sel[t]) : 0u;
}

// synthetic documents: token (doc, t) = generate_normalized-style row of the uniform stream:
// x = uniform row (row0 + doc*T + t); norm = sqrt(sum x*x) sequential from -0.0; x /= norm if norm > f32::EPSILON
// (examples/maxsim_colbert.rs:212-228 with the uniform generator). One thread per token.
__global__ void generate_tokens_kernel(float* __restrict__ tok, size_t ntok, uint32_t dim, uint64_t seed, uint64_t row0) {
    const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= ntok) return;
    float* row = tok + r * dim;
    float ss = -0.0f;
    for (uint32_t d = 0; d < dim; ++d) {
        const float x = uniform_embedding(seed, row0 + r, dim, d);
        ss = ex::mad2(ss, x, x);
    }
    const float norm = ex::sqrt(ss);
    for (uint32_t d = 0; d < dim; ++d) {
        const float x = uniform_embedding(seed, row0 + r, dim, d);
        row[d] = (norm > 1.1920929e-07f) ? ex::div(x, norm) : x;
    }
}

// Top-k of a dense score array (one "query"): threshold filter + per-wave lists, see topk_dev.h.
template <int R>
__global__ __launch_bounds__(256) void dense_filter_kernel(const float* __restrict__ scores, uint32_t N,
                                                           uint64_t* __restrict__ lists, uint32_t* __restrict__ counts,
                                                           uint32_t KP, uint32_t chunks_per_slot,
                                                           uint32_t* __restrict__ errflag) {
    constexpr uint32_t cap = 64 * R;
    __shared__ uint32_t s_cnt[4], s_thr[4];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t slot = (size_t)blockIdx.x * 4 + w;
    if (lane == 0) {
        s_cnt[w] = 0;
        s_thr[w] = 0;
    }
    __builtin_amdgcn_wave_barrier();
    const size_t nchunks = ((size_t)N + 255) / 256;
    size_t ch0 = slot * chunks_per_slot, ch1 = ch0 + chunks_per_slot;
    if (ch1 > nchunks) ch1 = nchunks;
    uint64_t* my = lists + slot * cap;
    for (size_t ch = ch0; ch < ch1; ++ch) {
        const uint32_t thr = __hip_atomic_load(&s_thr[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const size_t i = ch * 256 + (size_t)c * 64 + lane;
            if (i < N) {
                const uint32_t pref = f32_ord(scores[i]);
                if (pref >= thr) cand_append(my, &s_cnt[w], cap, cand_make(pref, (uint32_t)i), errflag);
            }
        }
        __builtin_amdgcn_wave_barrier();
        const uint32_t c = __builtin_amdgcn_readfirstlane(
            __hip_atomic_load(&s_cnt[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT));
        if (c > cap - kBurst) {
            uint32_t t;
            const uint32_t keep = wave_compact<R>(my, c, KP, &t);
            if (lane == 0) {
                s_cnt[w] = keep;
                s_thr[w] = t;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    uint32_t c = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&s_cnt[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT));
    if (c > KP) {
        uint32_t t;
        c = wave_compact<R>(my, c, KP, &t);
    }
    if (lane == 0) counts[slot] = c;
}

}  // namespace innr
