// kernels_gemm.h -- the batched-query engine: Q x N scores as an f32 MFMA GEMM over the PDX corpus with a
// fused per-query threshold-filter top-k, so the Q x N score matrix (41 GB at 1024 x 10M) never exists.
//
// Replaces the reference's per-query loop  batch_dot_into (src/batch.rs:284-297) + full stable sort
// (:756-758) / batch_cosine_into epilogue (:713-727)  for a whole batch of queries.
//
// Roofline: MFMA-bound. 2*Q*N*D flop; v_mfma_f32_32x32x2_f32 = 64 flop/clk/SIMD = 157.3 TFLOP/s chip peak.
// The corpus is streamed once per query tile (Q/512 or Q/256 times): far below the HBM roof.
//
// Mapping (CDNA4, wave64):
//   S^T[corpus i][query j] = sum_d V[d][i] * Qt[d][j]     A = corpus (MFMA rows), B = queries (MFMA cols)
//   Both operands are K-major in memory: the PDX layout V[d*ldN + i] IS the A operand's layout (lane l needs
//   A[i = l&31][k = l>>5], i.e. 32 consecutive floats of one dimension row), and the queries are transposed
//   once per call to Qt[d*Qpad + j]. No in-kernel transpose, every global and LDS access is contiguous.
//   Block = 4 or 8 waves (template parameter WAVES), tile 128 corpus x 64*WAVES queries x BK 16; wave w owns queries
//   [64w, 64w+64) x all 128 corpus rows = 4 x 2 MFMA tiles of 32x32 (128 accumulator VGPRs).
//   Corpus tile (shared by the block's waves): LDS-DMA (global_load_lds_dwordx4) into a 3-stage ring of 8 KiB, two
//   K-steps ahead; one ds_read_b128 per lane yields the A fragments of FOUR row tiles at once (tile rt holds
//   corpus rows 4r+rt), conflict-free. Query operands (private to a wave): plain 8-byte loads from the L2-resident
//   Qt straight into registers, one K-step ahead -- they never touch LDS. One barrier per K-step; 26-28 KiB LDS and
//   <= 256 VGPRs keep 8 waves resident per CU: two 4-wave blocks (one block's epilogue overlaps the other's MFMAs) or
//   one 8-wave block (corpus streamed half as often).
//   All global addresses are a wave-uniform 64-bit base (advanced on the scalar unit) + a constant 32-bit lane offset.
//   The wave that owns a query owns its candidate list: no cross-wave synchronisation in the epilogue.
//   Block -> (slice, query tile): one query tile per XCD group (its 768 KiB stay in that L2); placement only
//   affects speed.
#pragma once

#include "common.h"
#include "topk_dev.h"

namespace innr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// waves per block (template parameter WAVES): 4 = 256-query tile, two blocks per CU; 8 = 512-query tile, one block per
// CU, the corpus streamed once per 512 queries instead of once per 256 (half the L2-miss traffic). plan_gemm (api.hip)
// picks 8 for the dot kind with more than 256 queries (same speed) and 4 for cosine / L2 / u8 (0.6-1 ms faster at C2).
constexpr int kBC = 128;   // corpus rows per block tile
constexpr int kBQmax = 512;  // queries per block tile = 64 per wave
constexpr int kBK = 16;    // K-step
constexpr int kGemmBurst = kBC;  // most appends one tile can make to one query's list

constexpr int kStages = 3;  // LDS ring: the DMA for K-step s+2 is issued during step s (two steps of cover)

template <int WAVES>
struct alignas(16) GemmLds {
    alignas(16) float A[kStages][kBK][kBC];  // 3 x 8 KiB: the corpus tile, shared by the block's four waves
    uint32_t cnt[64 * WAVES];
    uint32_t thr[64 * WAVES];
};
constexpr uint32_t kStageBytesA = kBK * kBC * 4;

// One LDS-DMA instruction: 64 lanes x 16 B from per-lane global addresses to LDS [M0 .. M0 + 1 KiB), linear.
// Written as inline asm on purpose: for the builtin, hipcc (ROCm 7.2) treats every later ds_read as aliasing
// the pending DMA and drains vmcnt(0) before the first fragment read of the SAME K-step, which serialises
// prefetch and compute. The asm form is invisible to that bookkeeping; the K-loop waits for it explicitly
// (s_waitcnt vmcnt(0) before the barrier that publishes the stage). M0 is saved/restored around the DMA and
// the SALU->M0->DMA hazard is padded with s_nop 0 inside the statement.
// Addresses are wave-uniform 64-bit base (SGPR pair) + per-lane 32-bit byte offset: the K-loop advances the bases on
// the scalar unit and carries one VGPR per stream instead of a 64-bit per-lane pointer.
__device__ __forceinline__ void glds16(const char* base_uniform, uint32_t lane_off, uint32_t lds_byte_addr_uniform) {
    uint32_t keep;
    uint64_t base;  // see gload2 for the s_mov_b64
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %4\n\t"
        "s_mov_b64 %1, %3\n\t"
        "global_load_lds_dwordx4 %2, %1\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep), "=&s"(base)
        : "v"(lane_off), "s"(base_uniform), "s"(lds_byte_addr_uniform)
        : "memory");
}

__device__ __forceinline__ uint32_t lds_addr_uniform(const void* p) {
    const uint32_t a = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
    return __builtin_amdgcn_readfirstlane(a);
}

// Operand feed of one K-step (16 dimensions):
//   corpus tile (16 x 128 floats, shared by the four waves): 2 LDS-DMA instructions of 1 KiB per wave into the ring,
//     LDS destination = wave-uniform base + lane*16 (linear), source address per lane; issued two steps ahead;
//   queries: every wave multiplies its OWN 64 queries, so their fragments never need to be shared: each lane loads
//     its B operands (2 floats per k-pair: queries 64w + 2(l&31) + {0,1}, dimension 2kp + (l>>5)) straight from the
//     K-major query matrix (L2-resident) into the registers the previous step just freed, one step ahead.
//     (Staging them through LDS as well cost 16 KiB of DMA writes + 16 KiB of ds_reads per block per step:
//      tools/gemm_probe.hip attributed 12 % of the kernel time to exactly that traffic.)
//
// vmcnt counts every VMEM op of the wave in issue order (DMA, query loads, epilogue loads/stores alike). Per step a
// wave issues, in this order and in EVERY step (past the end of its slice into stages / registers nobody reads): the
// corpus DMA of step s+2 (2 ops; 1 for u8 and for 8-wave blocks), then 4 x 2 query loads for step s+1, each pair
// right after the MFMA group that consumed its registers. All waits are hand-counted inline asm (use_after<N> before
// a group's operands, wait_but_youngest<N> at the end of the step): the constant sequence is what makes the counts
// constants. tools/check_gemm_asm.py verifies in the ISA that the operand registers are only touched by those loads
// and the MFMAs.
__device__ __forceinline__ void wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#ifdef INNR_GEMM_PROBE_NOWAIT  // tools/gemm_probe.hip: issue the DMA but never wait for it
template <int N> __device__ __forceinline__ void wait_but_youngest() {}
#else
template <int N> __device__ __forceinline__ void wait_but_youngest() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
#endif

// Query-operand load (8 B per lane) and its use-side wait, both inline asm: left to the compiler, the waits for
// these loop-carried loads come out as vmcnt(1..3) -- it cannot see the DMA ops in between and resolves the loop
// back-edge conservatively -- which exposes an L2 round trip in every group. dst is valid only after use_after<N>.
__device__ __forceinline__ void gload2(float2& dst, const char* base_uniform, uint32_t lane_off) {
    // The base goes through an in-asm s_mov_b64: a VALU-written SGPR (v_readlane of a spilled SGPR, v_readfirstlane) read
    // by a VMEM instruction needs 5 wait states that hipcc's hazard recogniser does not insert in front of inline asm
    // (seen: the R = 20 instantiations reload spilled bases right before these loads); SALU reads are interlocked.
    uint64_t base;
    asm volatile("s_mov_b64 %1, %3\n\tglobal_load_dwordx2 %0, %2, %1"
                 : "=v"(dst), "=&s"(base)
                 : "v"(lane_off), "s"(base_uniform)
                 : "memory");
}
template <int N> __device__ __forceinline__ void use_after(float2& a, float2& b) {
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N));
}
// agent-coherent dword load (sc1: served by L2, where the atomics that update it execute), same contract as gload2
__device__ __forceinline__ void gload1_agent(uint32_t& dst, const uint32_t* base_uniform, uint32_t lane_off) {
    uint64_t base;
    asm volatile("s_mov_b64 %1, %3\n\tglobal_load_dword %0, %2, %1 sc1"
                 : "=v"(dst), "=&s"(base)
                 : "v"(lane_off), "s"(base_uniform)
                 : "memory");
}
template <int N> __device__ __forceinline__ void use_after(uint32_t& a, uint32_t& b) {
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N));
}

enum { kGemmDot = 0, kGemmCos = 1, kGemmU8 = 2, kGemmL2 = 3 };

// KIND kGemmDot: score = q.v            (batch_knn_dot)
//      kGemmCos: score = q.v * invn[i] * invq[j]   (approximate cosine; exact one in the re-score)
//      kGemmL2 : score = 2 q.v + invq[j] - invn[i] with invn[i] = |v_i|^2 and invq[j] = C_j - |q_j|^2, i.e.
//                C_j - |q_j - v_i|^2 up to rounding: larger = closer, and non-negative for C_j = (|q_j| + max|v|)^2 so
//                the raw-bit fast reject stays valid (batch_knn; exact direct-difference distance in the re-score)
//      kGemmU8 : the corpus is u8 codes C[d*ldN + i] (scalar.rs): a tile stage is 16 x 128 BYTES (2 DMA pieces per
//                block instead of 8), fragments are widened u8 -> f32 in registers (v_cvt_f32_ubyte0..3: one
//                ds_read_b32 feeds the four row tiles) and score = scale * (q.c) + invq[j]   (invq = offset*sum(q)):
//                "path B" of SURVEY.md -- the f32 MFMA pipe, a quarter of the corpus bytes.
// MODE 0: fused top-k filter (product path).  MODE 1: dump the dense score matrix (layout test only).
// MODE 2: COLLECT -- the completion pass of queries whose proof failed (knn_complete, api.hip): every query has a FIXED
//   threshold (gthr[q], set by the host: its k-th exact score so far, less E), and every site whose approximate score clears
//   it is appended to the query's GLOBAL list (`lists` = uint32 indices [Qpad][KP], `counts` = uint32 [Qpad] list lengths, KP =
//   the capacity; an overfull list keeps counting, the host sees the overflow). No slots, no compaction, no bound moves.
// WAVES: 8 = 512-query tile, one block per CU; 4 = 256 queries, two blocks per CU; 2 / 1 = 128 / 64 queries for small query
// batches (a 64-query batch on a 256-query tile spends three quarters of its MFMAs on padding: 27 ms instead of ~8 at C2),
// several blocks per CU, each streaming its own corpus slice (not for the u8 kind, whose small batches take the int8 engine).
template <int KIND, int R, int MODE, int WAVES>
__global__ __launch_bounds__(64 * WAVES, WAVES >= 8 ? 1 : 2) void gemm_filter_kernel(
    const void* __restrict__ Vraw, size_t ldN, uint32_t N, uint32_t Dpad, const float* __restrict__ Qt, size_t Qpad,
    uint32_t nqt, uint32_t qtg, uint32_t tiles_per_slice, const float* __restrict__ invn, const float* __restrict__ invq, float scale,
    uint64_t* __restrict__ lists, uint32_t* __restrict__ counts, uint32_t KP, uint32_t kk, uint32_t* __restrict__ errflag,
    uint32_t* gslots /*[Qpad][kSlotMul * KP]*/, uint32_t* gthr /*[Qpad] bounds, then [Qpad] k-rule margins (float)*/,
    float* __restrict__ dump, size_t ld_dump) {
    const float* const kmargin = reinterpret_cast<const float*>(gthr + Qpad);  // 2E per query: the k rule of topk_dev.h
    constexpr bool COS = KIND == kGemmCos;
    constexpr bool U8 = KIND == kGemmU8;
    constexpr bool L2K = KIND == kGemmL2;
    static_assert(WAVES == 1 || WAVES == 2 || WAVES == 4 || WAVES == 8, "block = 1, 2, 4 or 8 waves");
    static_assert(!(KIND == kGemmU8 && WAVES < 4), "the u8 kind stages two 1-KiB pieces per K-step: 4 or 8 waves");
    // corpus DMA pieces (1 KiB = two dimension rows of the 16 x 128 stage) this wave issues per K-step
    constexpr int NP = (KIND == kGemmU8) ? 1 : 8 / WAVES;
    constexpr int kEpiTgWait = 8 + NP;  // gemm_epilogue.inc: at most this many VMEM ops of the wave are in flight at a tile end
    const float* V = static_cast<const float*>(Vraw);
    const uint8_t* C8 = static_cast<const uint8_t*>(Vraw);
    constexpr int kBQ = 64 * WAVES;
    __shared__ GemmLds<WAVES> s;
    constexpr uint32_t cap = 64 * R;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // Blocks b, b+8, ... share an XCD and its 4 MB L2 (round-robin dispatch). An XCD works on a group of `qtg`
    // query tiles (qtg | nqt, (nqt/qtg) | 8): its consecutive blocks take the qtg tiles of ONE corpus slice, so a
    // streamed corpus tile is fetched from HBM once per group while only qtg x 768 KB of queries must stay in L2.
    // qtg = nqt: corpus read once, all queries resident per XCD; qtg = 1: corpus read nqt times, one query tile.
    const uint32_t b = blockIdx.x;
    const uint32_t xcd = b & 7, lb = b >> 3, groups = nqt / qtg;
    const uint32_t qt = (xcd % groups) * qtg + lb % qtg;
    const uint32_t slice = (lb / qtg) * (8 / groups) + xcd / groups;
    const size_t q0 = (size_t)qt * kBQ;
    const uint32_t ntiles = (uint32_t)(ldN / kBC);
    uint32_t t0 = slice * tiles_per_slice, t1 = t0 + tiles_per_slice;
    if (t1 > ntiles) t1 = ntiles;
    if (t0 > t1) t0 = t1;
    const uint32_t nk = Dpad / kBK;
    const uint32_t total = (t1 - t0) * nk;

    s.cnt[threadIdx.x] = 0;  // kBQ == kGemmThreads
    s.thr[threadIdx.x] = 0;
    uint64_t* my_lists = lists + ((size_t)slice * Qpad + q0) * cap;

    f32x16 acc[4][2];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[rt][ct][g] = 0.0f;

    // Source addresses of this wave's corpus DMA pieces and query operands: a wave-uniform base, advanced on the scalar
    // unit, plus a constant per-lane byte offset (< 4 GiB: the host keeps ldN below 2^29 for this engine).
    const int wu = __builtin_amdgcn_readfirstlane(w);
    const char* sa;     // base of this wave's first corpus piece; piece j adds j * piece_step (f32: two dimension rows)
    uint32_t la;        // its stage-0 LDS destination; piece j adds j KiB, stage k adds k * kStageBytesA
    uint32_t va;        // per-lane offset inside a piece
    // u8 corpus: the 16 x 128-byte stage is two 1-KiB pieces (8 rows each): piece (w & 1) from every wave
    if (U8) va = (uint32_t)(lane >> 3) * (uint32_t)ldN + (uint32_t)(lane & 7) * 16u;
    else va = ((uint32_t)(lane >> 5) * (uint32_t)ldN + (uint32_t)(lane & 31) * 4u) * 4u;
    if (U8) {
        sa = reinterpret_cast<const char*>(C8 + (size_t)(8 * (wu & 1)) * ldN + (size_t)t0 * kBC);
        la = lds_addr_uniform(reinterpret_cast<const uint8_t*>(&s.A[0][0][0]) + 1024 * (wu & 1));
    } else {  // 8 pieces of 2 rows: NP consecutive ones per wave
        sa = reinterpret_cast<const char*>(V + (size_t)(2 * NP * wu) * ldN + (size_t)t0 * kBC);
        la = lds_addr_uniform(&s.A[0][2 * NP * wu][0]);
    }
    const size_t piece_step = (size_t)2 * ldN * 4;
    // B operands: k-pair kp of the K-step the base refers to sits at sb + kp * b_kp
    const char* sb = reinterpret_cast<const char*>(Qt + q0 + 64 * wu);
    const uint32_t vb = ((uint32_t)(lane >> 5) * (uint32_t)Qpad + 2u * (uint32_t)(lane & 31)) * 4u;
    const size_t b_kp = 2 * Qpad * 4;
    // base strides in bytes
    const size_t a_step = (size_t)kBK * ldN * (U8 ? 1 : 4), q_step = (size_t)kBK * Qpad * 4;
    // subtract at a tile change: back to row 0, next tile
#ifdef INNR_GEMM_PROBE_L2HOT  // tools/gemm_probe.hip: every block re-reads corpus tile 0 (L2-resident operands)
    const size_t a_wrap = (size_t)(Dpad - kBK) * ldN * (U8 ? 1 : 4);
    sa -= (size_t)t0 * kBC * (U8 ? 1 : 4);
#else
    const size_t a_wrap = ((size_t)(Dpad - kBK) * ldN - kBC) * (U8 ? 1 : 4);
#endif
    const size_t q_wrap = (size_t)(Dpad - kBK) * Qpad * 4;
    uint32_t pks = 0, bks = 0;  // K-step index (within a tile) that sa / sb refer to
    auto advance = [&]() {
        if (++pks == nk) {
            pks = 0;
            sa -= a_wrap;
        } else {
            sa += a_step;
        }
    };
    auto advance_b = [&]() {
        if (++bks == nk) {
            bks = 0;
            sb -= q_wrap;
        } else {
            sb += q_step;
        }
    };
    auto issue_a = [&](uint32_t stage_off) {  // u8: waves 2-3 repeat the pieces of waves 0-1: every wave's op count is the same
#pragma unroll
        for (int j = 0; j < NP; ++j) glds16(sa + j * piece_step, va, la + 1024u * j + stage_off);
    };
    // 1- and 2-wave blocks issue 8 / 4 pieces per K-step: spread over the four MFMA groups of the step (NP / 4 behind each
    // group's query loads) instead of in front of them, where one wave per SIMD would leave the matrix pipe idle for the
    // ~1000 cycles the eight issues take. The waits do not move: a group's operands still have 6 + NP younger ops (NP = sum of
    // the pieces, wherever they sit in the step), the end-of-step wait still leaves the step's own 8 + NP ops in flight.
    constexpr bool kSpreadDma = NP >= 4;
    auto issue_a_part = [&](uint32_t stage_off, int grp) {
#pragma unroll
        for (int j = grp * (NP / 4); j < (grp + 1) * (NP / 4); ++j) glds16(sa + j * piece_step, va, la + 1024u * j + stage_off);
    };
    // prologue: corpus K-steps 0 and 1 into stages 0 and 1, query operands of K-step 0 into registers
    // (pa always refers to the last K-step issued: advance, then issue -- so a pointer never leaves the slice)
    if (total) issue_a(0);
    if (total > 1) {
        advance();
        issue_a(kStageBytesA);
    }
    float2 breg[kBK / 2];  // this wave's B operands of the current K-step, refilled pair by pair for the next one
#pragma unroll
    for (int kp = 0; kp < kBK / 2; ++kp) breg[kp] = make_float2(0.f, 0.f);
    if (total) {
#pragma unroll
        for (int kp = 0; kp < kBK / 2; ++kp) gload2(breg[kp], sb + kp * b_kp, vb);
        advance_b();  // sb wraps inside the query matrix at every tile change: always a mapped address
    }
    wait_all();
#pragma unroll
    for (int kp = 0; kp < kBK / 2; kp += 2) use_after<0>(breg[kp], breg[kp + 1]);
    __syncthreads();  // stages 0 and 1 visible to every wave

    // chip-wide thresholds of this lane's two queries: read here (they may be seeded, see seed_thresholds_kernel) and
    // refreshed at the end of every tile's epilogue
    uint32_t tg_next[2] = {0u, 0u};
    if (MODE != 1) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) gload1_agent(tg_next[ct], gthr + q0 + 64 * wu + ct, 8u * (uint32_t)(lane & 31));
    }
    float iq_lane[2] = {1.0f, 1.0f};  // per-query epilogue constant: COS 1/||q||; U8 offset * sum(q); L2 C_j - |q_j|^2
    if (COS || U8 || L2K) {
        iq_lane[0] = invq[q0 + 64 * w + 2 * (lane & 31) + 0];
        iq_lane[1] = invq[q0 + 64 * w + 2 * (lane & 31) + 1];
    }
    uint32_t tile = t0, ks = 0;
    uint32_t st = 0;  // stage holding the current K-step; the DMA of step+2 goes to stage (st + 2) % 3
    for (uint32_t step = 0; step < total; ++step) {
        const bool has_next = step + 2 < total;   // wave-uniform: is there a corpus K-step to prefetch?
        const uint32_t dst = (st == 0) ? 2u : st - 1;
        const uint32_t da = dst * kStageBytesA;
        // 4 groups of 2 k-pairs: A-fragment reads (software-pipelined one group ahead: two register sets, so only the
        // first group of a step, right after the barrier, exposes LDS latency), 16 MFMAs, then the two B registers the
        // group just consumed are reloaded for the next K-step.
        float4 avs[2][2];
        // (Re-deriving the per-lane fragment address inside the loop, to spare the one VGPR hipcc spills in the cosine / L2
        //  instantiations, was tried: it removes the scratch reload at the top of the K-step and is slower all the same --
        //  dot 114 -> 117 ms, cosine 116 -> 121 ms.)
        const int lane_k = lane;
        auto read_frags = [&](int grp, int set) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int kp = 2 * grp + h;
                if (U8) {  // 4 codes (rows 4r..4r+3 of dimension k) in one dword; widened to f32 right before the MFMAs
                    const uint8_t* a8 = reinterpret_cast<const uint8_t*>(&s.A[st][0][0]);
                    avs[set][h].x = __uint_as_float(
                        *reinterpret_cast<const uint32_t*>(a8 + (2 * kp + (lane_k >> 5)) * kBC + 4 * (lane_k & 31)));
                } else {
                    avs[set][h] = *reinterpret_cast<const float4*>(&s.A[st][2 * kp + (lane_k >> 5)][4 * (lane_k & 31)]);
                }
            }
        };
        read_frags(0, 0);
        // the corpus DMA of step s+2 goes out first: every later wait of this step then sees it among the younger ops
#ifndef INNR_GEMM_PROBE_NODMA_A  // tools/gemm_probe.hip
        if (has_next) advance();
        if (!kSpreadDma) issue_a(da);
#endif
#pragma unroll
        for (int grp = 0; grp < 4; ++grp) {
            const int cur = grp & 1;
            if (grp < 3) read_frags(grp + 1, cur ^ 1);
            float4 av[2] = {avs[cur][0], avs[cur][1]};
            // Every step issues the same VMEM sequence (corpus DMA, then 4 x 2 query loads; past the end of the slice
            // they re-fetch the last step / wrap, into a stage and registers nobody reads), so the ops younger than
            // this group's operands (loaded one step ago, after the same group) are always 6 query loads + the corpus
            // DMA (f32: 2 ops, u8: 1). One unconditional wait: a branch here made hipcc copy the registers BEFORE it.
            use_after<6 + NP>(breg[2 * grp], breg[2 * grp + 1]);
            const float2 bv[2] = {breg[2 * grp], breg[2 * grp + 1]};
            if (U8) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const uint32_t c4 = __float_as_uint(avs[cur][h].x);
                    av[h].x = (float)(c4 & 0xffu);
                    av[h].y = (float)((c4 >> 8) & 0xffu);
                    av[h].z = (float)((c4 >> 16) & 0xffu);
                    av[h].w = (float)(c4 >> 24);
                }
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float a[4] = {av[h].x, av[h].y, av[h].z, av[h].w};
                const float bb[2] = {bv[h].x, bv[h].y};
#pragma unroll
                for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct)
                        acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[rt], bb[ct], acc[rt][ct], 0, 0, 0);
            }
#ifndef INNR_GEMM_PROBE_NODMA_B  // tools/gemm_probe.hip
            gload2(breg[2 * grp], sb + (2 * grp) * b_kp, vb);
            gload2(breg[2 * grp + 1], sb + (2 * grp + 1) * b_kp, vb);
#endif
            if (kSpreadDma) issue_a_part(da, grp);
            __builtin_amdgcn_sched_barrier(0);
        }
        advance_b();

#ifdef INNR_GEMM_PROBE_NOEPI  // tools/gemm_probe.hip: K-loop only (accumulators keep running, results meaningless)
        if (false) {
#elif defined(INNR_GEMM_PROBE_SKIPEPI)  // same code as the product kernel, epilogue never taken (dump == nullptr at run time)
        if (ks + 1 == nk && dump != nullptr) {
#else
        if (ks + 1 == nk) {
#endif
#include "gemm_epilogue.inc"
            ks = 0;
            ++tile;
        } else {
            ++ks;
        }
        // This wave's corpus pieces of K-step s+1 (issued one step ago) are in LDS; the youngest ops -- the corpus
        // pieces of s+2 and the 8 query loads of s+1 -- stay in flight.
        wait_but_youngest<8 + NP>();
#ifndef INNR_GEMM_PROBE_NOBAR  // tools/gemm_probe.hip
        __syncthreads();  // ... so are everyone else's, and the stage just consumed may be overwritten next step
#endif
        st = (st == 2) ? 0u : st + 1;
    }
    wait_all();  // the trailing (unused) DMA and loads must land before this block's LDS and registers are released
    __syncthreads();

#ifdef INNR_GEMM_PROBE_NOEPI  // keep the accumulators alive
    if (KP == 0xFFFFFFFFu) {
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int g = 0; g < 16; ++g) reinterpret_cast<float*>(lists)[threadIdx.x + 256 * (g + 16 * (ct + 2 * rt))] = acc[rt][ct][g];
    }
#endif
    if (MODE == 0) {
        // leave <= KP entries per list and publish the counts (every query of the tile, padded ones too)
        const uint32_t c = __hip_atomic_load(&s.cnt[64 * w + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        unsigned long long need = __ballot(c > KP);
        uint32_t mine = c;
        while (need) {
            const int j = __builtin_ctzll(need);
            need &= need - 1;
            const uint32_t cj = __builtin_amdgcn_readlane(c, j);
            uint32_t t;
            const uint32_t keep = wave_compact<R>(my_lists + (size_t)(64 * w + j) * cap, cj, KP, &t);
            if (lane == j) mine = keep;
        }
        counts[(size_t)slice * Qpad + q0 + 64 * w + lane] = mine;
    }
}

// ---- helpers around the GEMM ------------------------------------------------------------------------
// Qt[d*Qpad + j] = Q[j*D + d] (zero padded to Dpad x Qpad)
__global__ __launch_bounds__(256) void transpose_queries_kernel(const float* __restrict__ Qm, uint32_t Q, uint32_t D,
                                                                 float* __restrict__ Qt, size_t Qpad, uint32_t Dpad) {
    __shared__ float tile[32][33];
    const uint32_t tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const uint32_t jb = blockIdx.x * 32, db = blockIdx.y * 32;
#pragma unroll
    for (int r = 0; r < 32; r += 8) {
        const uint32_t j = jb + ty + r, d = db + tx;
        tile[ty + r][tx] = (j < Q && d < D) ? Qm[(size_t)j * D + d] : 0.0f;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 32; r += 8) {
        const uint32_t d = db + ty + r, j = jb + tx;
        if (d < Dpad && j < Qpad) Qt[(size_t)d * Qpad + j] = tile[tx][ty + r];
    }
}

// inv[i] = x > eps ? 1/x : 0   (scale factors for the approximate cosine used only for candidate selection)
__global__ void inv_norms_kernel(const float* __restrict__ x, size_t n, size_t n_valid, float* __restrict__ inv) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = (i < n_valid) ? x[i] : 0.0f;
    inv[i] = (v > INNR_NORM_EPSILON) ? 1.0f / v : 0.0f;
}
__global__ void inv_qnorms_kernel(const float* __restrict__ x, size_t n, size_t n_valid, float* __restrict__ inv) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = (i < n_valid) ? x[i] : 0.0f;
    inv[i] = (v >= INNR_NORM_EPSILON) ? 1.0f / v : 0.0f;  // batch.rs:716: whole output 0 when ||q|| < eps
}

// Exact re-score + final ordering + margin proof. One wave per query.
//   sel[q][0..KP): best-first composites by APPROXIMATE (MFMA) score; sel_cnt[q] of them valid.
//   Each candidate is re-scored in the reference's arithmetic order (ex::mad2 over d ascending; cosine
//   epilogue batch.rs:721-727), candidates are ranked by (exact score total order desc, index asc) and the
//   best kout are written. Proof obligation for "no vector outside the candidate set can belong to the true
//   top-k": every outsider has approx <= T (the KP-th approximate score) and |approx - exact| <= E, so it is
//   enough that exact(k-th best candidate) > T + E. Failing queries are flagged for the exact engine.
// MET: 0 dot, 1 cosine, 2 squared L2 (direct differences, batch.rs:262-263; smaller is better; qaux[q] = C_q of the
// GEMM epilogue, err_scale * C_q bounds |C_q - approx - exact distance|).
// Progressive (early = true; the candidates are sorted best-first by approximate score): the wave re-scores them in
// rounds -- the best 32, the next 32, then 64, then 128 -- and stops at the first round after which the proof holds with
// T = the approximate score of the best candidate NOT yet re-scored (every such candidate, and every outsider, has approx <= T).
// The column gathers of the re-score are sector traffic (4 useful bytes per line fetched): at C2 the low-precision filters
// kept 128 candidates per query and spent 2.0 ms re-scoring them, although the answer was usually proven by the first 32.
template <int MET, int RK>
__global__ __launch_bounds__(64) void rescore_kernel(const float* __restrict__ V, size_t ldN, uint32_t D,
                                                     const float* __restrict__ Qm, const float* __restrict__ norms,
                                                     const float* __restrict__ qnorm, const float* __restrict__ qaux,
                                                     const uint64_t* __restrict__ sel,
                                                     const uint32_t* __restrict__ sel_cnt, uint32_t KP, uint32_t kout,
                                                     float err_scale, uint64_t index_base,
                                                     uint64_t* __restrict__ out_idx, float* __restrict__ out_score,
                                                     uint32_t* __restrict__ fallback, const float* __restrict__ eq = nullptr,
                                                     bool early = false, const uint32_t* __restrict__ gthr = nullptr) {
    constexpr bool COS = MET == 1, L2 = MET == 2;
    const uint32_t q = blockIdx.x;
    const int lane = threadIdx.x;
    const uint32_t cnt = sel_cnt[q];
    // the query's final chip-wide bound (topk_dev.h; 0 = none): every site the filter rejected had an approximate score below it
    const uint32_t G = gthr ? gthr[q] : 0u;
    const float* qv = Qm + (size_t)q * D;
    const float qn = L2 ? 0.0f : qnorm[q];
    uint64_t e[RK];
#pragma unroll
    for (int r = 0; r < RK; ++r) e[r] = 0;
    auto exact = [&](uint32_t c) -> uint64_t {  // candidate c of this query, re-scored in the reference's order
        const uint32_t i = cand_idx(sel[(size_t)q * KP + c]);
        const float* col = V + i;
        float acc = 0.0f;
        // (32 column loads in flight per lane: the sum is sequential, its operands are not -- at 8 a one-query call spent 0.10 ms here)
        if (L2) {
#pragma unroll 32
            for (uint32_t d = 0; d < D; ++d) {
                const float diff = ex::sub_keepnan(qv[d], col[(size_t)d * ldN]);
                acc = ex::mad2(acc, diff, diff);
            }
        } else {
#pragma unroll 32
            for (uint32_t d = 0; d < D; ++d) acc = ex::mad2(acc, qv[d], col[(size_t)d * ldN]);
        }
        if (COS) {
            const float vn = norms[i];
            acc = (qn < INNR_NORM_EPSILON) ? 0.0f : ((vn > INNR_NORM_EPSILON) ? ex::div(acc, ex::mul(qn, vn)) : 0.0f);
        }
        return cand_make(score_pref<L2>(acc), i);
    };
    // rounds [0,32) [32,64) [64,128) [128,256): candidate c lives in slot c / 64 of lane c % 64
    constexpr int NR = RK == 1 ? 2 : (RK == 2 ? 3 : 4);
#pragma unroll
    for (int round = 0; round < NR; ++round) {
        const uint32_t lo = round == 0 ? 0u : (32u << (round - 1)), hi = 32u << round;
        if (lo >= cnt && round > 0) break;  // (wave-uniform)
        if (round < 2) {
            const uint32_t c = (uint32_t)lane;
            if (c >= lo && c < hi && c < cnt) e[0] = exact(c);
        } else {
#pragma unroll
            for (int r = (round == 2 ? 1 : 2); r < (round == 2 ? 2 : 4); ++r)
                if (r < RK) {
                    const uint32_t c = (uint32_t)(r * 64 + lane);
                    if (c < cnt) e[r < RK ? r : 0] = exact(c);
                }
        }
        const uint32_t done = hi < cnt ? hi : cnt;
        const bool last = done == cnt;
        if (!last && (!early || done < kout)) continue;  // nothing to decide yet
        uint32_t rank[RK];
        rescore_rank<RK>(e, done, rank);
        uint32_t kth_bits = 0;
        bool have_kth = false;
#pragma unroll
        for (int r = 0; r < RK; ++r) {
            const uint32_t c = r * 64 + lane;
            if (c < done && rank[r] == kout - 1) {
                kth_bits = cand_pref(e[r]);
                have_kth = true;
            }
        }
        // margin proof (one lane holds the k-th best exact score)
        bool bad = false;
        // T: no candidate that has not been re-scored, and no vector outside the lists, has a better approximate score -- outsiders
        // were either dropped from a full list (<= the KP-th approximate score) or rejected by a threshold (< the final bound G).
        // Last round with cnt < KP and no bound: every corpus vector is a candidate, nothing to prove.
        uint32_t tp = G;
        bool prove = G != 0u;
        if (!last || cnt == KP) {
            const uint32_t lp = cand_pref(sel[(size_t)q * KP + (last ? KP - 1 : done)]);
            tp = (!prove || lp > tp) ? lp : tp;
            prove = true;
        }
        if (last && G != 0u && cnt < kout) bad = true;  // (cannot happen: at least k sites clear a bound of either rule)
        if (have_kth && prove) {
            const float exact_k = pref_score(kth_bits, L2);
            const float T = ord_f32(tp);
            if (L2) {
                // outsiders: approx <= T, i.e. their approximate distance C - approx >= C - T, exact >= C - T - E
                const float Cq = qaux[q];
                bad = !(exact_k < (Cq - T) - (eq ? eq[q] : err_scale * Cq));
            } else {
                const float E = eq ? eq[q] : (COS ? err_scale : err_scale * qn);  // eq: a per-query bound (int8 filter of an f32 corpus)
                bad = !(exact_k > T + E);  // also true for NaN / inf arithmetic: those queries go to the exact engine
            }
        }
        const bool failed = __any(bad);
        if (failed && !last) continue;  // not proven yet: the next round brings more candidates
#pragma unroll
        for (int r = 0; r < RK; ++r) {
            const uint32_t c = r * 64 + lane;
            if (c < done && rank[r] < kout) {
                out_idx[(size_t)q * kout + rank[r]] = index_base + cand_idx(e[r]);
                out_score[(size_t)q * kout + rank[r]] = pref_score(cand_pref(e[r]), L2);
            }
        }
        if (failed && lane == 0) {
            fallback[q] = 1;
            // the completion pass (knn_complete) reads the k-th score of a failed query as a lower bound of the true one:
            // with fewer than k candidates there is none
            if (done < kout) out_score[(size_t)q * kout + kout - 1] = __builtin_nanf("");
        }
        return;
    }
}

// Re-rank support: composites for caller-given candidate indices (any order; duplicates allowed but pointless),
// sel[q][c] = (0 | ~idx) for c < kc, count kc. Out-of-range indices are clamped into the batch and flagged.
__global__ void rerank_prepare_kernel(const uint64_t* __restrict__ cand, uint32_t Q, uint32_t kc, uint32_t KP, uint32_t N,
                                      uint64_t index_base, uint64_t* __restrict__ sel, uint32_t* __restrict__ sel_cnt,
                                      uint32_t* __restrict__ bad) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= Q * KP) return;
    const uint32_t q = t / KP, c = t % KP;
    if (c == 0) sel_cnt[q] = kc;
    uint64_t v = 0;
    if (c < kc) {
        const uint64_t g = cand[(size_t)q * kc + c];
        uint64_t i = g - index_base;
        if (g < index_base || i >= N) {
            atomicOr(bad, 1u);
            i = 0;
        }
        v = cand_make(0u, (uint32_t)i);
    }
    sel[t] = v;
}

// Re-rank with more candidates per query than a candidate list holds (innr_batch_rerank*, kc > 256): one thread per
// (query, candidate) computes the exact score in the reference's order and writes its composite; the caller sorts every
// query's kc composites. Out-of-range indices are clamped into the batch and flagged, as in rerank_prepare_kernel.
template <int MET>
__global__ __launch_bounds__(256) void rerank_scores_kernel(const float* __restrict__ V, size_t ldN, uint32_t N, uint32_t D,
                                                             const float* __restrict__ Qm, const float* __restrict__ norms,
                                                             const float* __restrict__ qnorm, const uint64_t* __restrict__ cand,
                                                             uint32_t Q, uint32_t kc, uint64_t index_base,
                                                             uint64_t* __restrict__ keys, uint32_t* __restrict__ bad) {
    constexpr bool COS = MET == 1, L2 = MET == 2;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)Q * kc) return;
    const uint32_t q = (uint32_t)(t / kc);
    const uint64_t g = cand[t];
    uint64_t i64 = g - index_base;
    if (g < index_base || i64 >= N) {
        atomicOr(bad, 1u);
        i64 = 0;
    }
    const uint32_t i = (uint32_t)i64;
    const float* qv = Qm + (size_t)q * D;
    const float* col = V + i;
    float acc = 0.0f;
    if (L2) {
#pragma unroll 8
        for (uint32_t d = 0; d < D; ++d) {
            const float diff = ex::sub_keepnan(qv[d], col[(size_t)d * ldN]);
            acc = ex::mad2(acc, diff, diff);
        }
    } else {
#pragma unroll 8
        for (uint32_t d = 0; d < D; ++d) acc = ex::mad2(acc, qv[d], col[(size_t)d * ldN]);
    }
    if (COS) {
        const float qn = qnorm[q], vn = norms[i];
        acc = (qn < INNR_NORM_EPSILON) ? 0.0f : ((vn > INNR_NORM_EPSILON) ? ex::div(acc, ex::mul(qn, vn)) : 0.0f);
    }
    keys[t] = cand_make(score_pref<L2>(acc), i);
}

// Threshold seeding. Without it every slice appends its whole first tile (no list has a threshold yet, the chip-wide
// bound is still 0): 128 x 128 slices = 16 384 appends per query, 94 % of all appends of a C2 launch and ~4 % of its
// time. The exact engine first finds the KP best of a corpus PREFIX per query; their KP-th exact score, lowered by
// the approximation error bound E (so that all KP of them are certain to clear it with their APPROXIMATE scores), is a
// valid chip-wide bound from the first tile on. kind: 0 dot (E = err_scale*|q|), 1 cosine (E = err_scale),
// 2 squared L2 in the epilogue's score space s = C - dist (E = err_scale*C). seed[j] = 0 ("no bound") when not finite.
// pos: which of the prefix's exact scores seeds the bound -- KP - 1 (the KP rule of topk_dev.h: all KP of them clear it with their
// approximate scores) or k - 1 (the k rule: the k-th best exact score of the WHOLE corpus is at least the prefix's, so a vector
// whose approximate score is below it by more than E is not in the top k); one key lower, so that the re-score's proof is strict.
__global__ void seed_thresholds_kernel(const float* __restrict__ kth_scores /*[Q][KP], best first*/, uint32_t Q, uint32_t KP,
                                       int kind, float err_scale, const float* __restrict__ qnorm, const float* __restrict__ Cj,
                                       uint32_t* __restrict__ seed /*[Qpad]*/, uint32_t Qpad, uint32_t pos) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= Qpad) return;
    uint32_t o = 0;
    if (j < Q) {
        const float x = kth_scores[(size_t)j * KP + pos];
        float t;
        if (kind == 2) t = (Cj[j] - x) - err_scale * Cj[j] * 1.0001f;
        else if (kind == 1) t = x - err_scale * 1.0001f;
        else t = x - err_scale * qnorm[j] * 1.0001f;
        if (t - t == 0.0f) o = f32_ord(t) - 1u;
    }
    seed[j] = o;
}

// L2 on the GEMM engine: per query C_j = (|q_j| + max|v|)^2 and the epilogue constant C_j - |q_j|^2 (padded queries: 0)
__global__ void l2_query_consts_kernel(const float* __restrict__ qnorm, size_t Qpad, size_t Q, float max_norm,
                                       float* __restrict__ cq, float* __restrict__ Cj) {
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= Qpad) return;
    float C = 0.0f, c = 0.0f;
    if (j < Q) {
        const float qn = qnorm[j], s = qn + max_norm;
        C = s * s;
        c = C - qn * qn;
    }
    cq[j] = c;
    if (j < Q) Cj[j] = C;
}

// |v_i|^2 from the cached exact norms (padding rows: 0)
__global__ void sq_norms_kernel(const float* __restrict__ norms, size_t ldN, size_t N, float* __restrict__ sq) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ldN) return;
    const float v = (i < N) ? norms[i] : 0.0f;
    sq[i] = v * v;
}

}  // namespace innr
