// kernels_gemm_bf16.h -- the GEMM engine's FILTER on the bf16 pipe (v_mfma_f32_32x32x16_bf16, 16x the f32 MFMA rate).
//
// Same contract as gemm_filter_kernel (kernels_gemm.h): approximate scores in, per-(slice, query) candidate lists out;
// the caller re-scores the candidates in the reference's f32 order and proves the answer, so the results stay the
// reference's bit for bit (batch.rs:742-764) -- only the approximation bound E changes: both operands are rounded to
// bf16 (relative 2^-9 each), so |score_bf16 - score| <= (2^-8 + 2^-16) sum|q_d v_d| + the f32 accumulation term.
//
// Layouts (built once per corpus / once per call by the pack kernels below):
//   corpus  Ab[tile][ks][kg 0..3][rt 0..3][i 0..31][8 bf16]   corpus row = 128 tile + 4 i + rt, dimension = 32 ks + 8 kg + e
//           one K-step (32 dimensions) of a 128-row tile is 8 KiB contiguous: eight 1-KiB LDS-DMA pieces, one per wave;
//           an A fragment (row i of row tile rt, 8 consecutive k) is one conflict-free ds_read_b128.
//   queries Bb[ks][kg 0..3][position 0..Qpad)[8 bf16]         position 64 w + 32 ct + j holds query 64 w + 2 j + ct
//           (the accumulator column mapping of gemm_filter_kernel, so the shared epilogue applies unchanged); a B
//           fragment is one 16-byte load per lane, straight from L2 into registers NLEAD K-steps ahead.
// Block = 8 waves, tile 128 corpus rows x 512 queries; LDS ring of kBfStages stages, DMA kBfStages-2 steps ahead;
// every step issues the same VMEM sequence (1 DMA, then 2 + 2 query loads), so all vmcnt waits are constants.
// tools/bf16_filter_probe.hip is this K-loop without the epilogue: 8.9 ms for the C2 GEMM (1.77 PFLOP/s).
#pragma once

#include "kernels_gemm.h"

namespace innr {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

constexpr int kBfStages = 8, kBfLead = 2, kBfStageBytes = 8192, kBfWaves = 8, kBfK = 32;
static_assert(kBfStages == 8 && kBfLead == 2, "the one-barrier-per-two-steps schedule is derived for an 8-stage ring and a 2-step register ring");

// round-to-nearest-even f32 -> bf16 (NaN stays NaN, quieted)
__device__ __forceinline__ uint16_t f32_to_bf16_rne(float x) {
    uint32_t b = __float_as_uint(x);
    if ((b & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((b >> 16) | 0x40u);
    b += 0x7fffu + ((b >> 16) & 1u);
    return (uint16_t)(b >> 16);
}

// limb l of x as a sum of bf16 values: x = limb0 + limb1 + limb2 up to 2^-24 |x| (each difference below is exact in f32)
__device__ __forceinline__ uint16_t bf16_limb(float x, int l) {
    uint16_t h = f32_to_bf16_rne(x);
    for (int t = 0; t < l; ++t) {
        x -= __uint_as_float((uint32_t)h << 16);
        h = f32_to_bf16_rne(x);
    }
    return h;
}
constexpr uint32_t kBfL2Extra = 6;  // the squared-L2 copy's additional K columns

// one thread per 16-byte output unit (8 dimensions of one corpus row)
// rowscale (nullable): 1/||v|| per row (0 for zero-norm rows) -- the COSINE copy holds the normalised rows, so that the plain
// dot of the filter kernel IS the approximate cosine (with the queries normalised the same way) and no norm is loaded per tile
// sqn (nullable): |v|^2 per row -- the SQUARED-L2 copy carries six more K columns per row, [three bf16 limbs of |v|^2, 1, 1, 1],
// against [-1, -1, -1, three limbs of c_j] and DOUBLED queries on the other side: the plain dot of the filter kernel is then
// 2 q.v - |v|^2 + c_j = C_j - |q - v|^2 up to rounding (c_j = C_j - |q_j|^2 >= 0 as for kGemmL2, kernels_gemm.h), no norm is
// loaded per tile and the score is non-negative where it matters (the raw-bit fast reject of the epilogue)
__global__ __launch_bounds__(256) void pack_corpus_bf16_kernel(const float* __restrict__ V, size_t ldN, uint32_t N, uint32_t D,
                                                                uint32_t nk, size_t units, uint4* __restrict__ Ab,
                                                                const float* __restrict__ rowscale = nullptr,
                                                                const float* __restrict__ sqn = nullptr) {
    const size_t u = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (u >= units) return;
    const uint32_t i = (uint32_t)(u & 31), rt = (uint32_t)((u >> 5) & 3), kg = (uint32_t)((u >> 7) & 3);
    const size_t tk = u >> 9;  // tile * nk + ks
    const uint32_t ks = (uint32_t)(tk % nk);
    const size_t row = (tk / nk) * 128 + 4 * i + rt;
    const float rs = (rowscale && row < N) ? rowscale[row] : 1.0f;
    uint16_t h[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const uint32_t d = ks * 32 + kg * 8 + e;
        h[e] = (row < N && d < D) ? f32_to_bf16_rne(V[(size_t)d * ldN + row] * rs) : (uint16_t)0;
        if (sqn && row < N && d >= D && d < D + kBfL2Extra)
            h[e] = (d - D < 3) ? bf16_limb(sqn[row], (int)(d - D)) : (uint16_t)0x3F80u;  // limbs of |v|^2, then 1.0 three times
    }
    uint4 o;
    o.x = h[0] | ((uint32_t)h[1] << 16); o.y = h[2] | ((uint32_t)h[3] << 16);
    o.z = h[4] | ((uint32_t)h[5] << 16); o.w = h[6] | ((uint32_t)h[7] << 16);
    Ab[u] = o;
}

// queries row-major [Q][D] -> Bb; one thread per 16-byte unit
// l2c (nullable; [Qpad] c_j = C_j - |q_j|^2): the squared-L2 packing, see pack_corpus_bf16_kernel
__global__ __launch_bounds__(256) void pack_queries_bf16_kernel(const float* __restrict__ Qm, uint32_t Q, uint32_t D, uint32_t nk,
                                                                 uint32_t Qpad, uint4* __restrict__ Bb,
                                                                 const float* __restrict__ qscale = nullptr,
                                                                 const float* __restrict__ l2c = nullptr) {
    const size_t u = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (u >= (size_t)nk * 4 * Qpad) return;
    const uint32_t pos = (uint32_t)(u % Qpad);
    const uint32_t kk = (uint32_t)(u / Qpad);  // ks * 4 + kg
    const uint32_t q = (pos & ~63u) + 2 * (pos & 31) + ((pos >> 5) & 1);
    const float qs = l2c ? 2.0f : ((qscale && q < Q) ? qscale[q] : 1.0f);  // cosine: 1/||q|| (0 below the reference's epsilon)
    uint16_t h[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const uint32_t d = kk * 8 + e;
        h[e] = (q < Q && d < D) ? f32_to_bf16_rne(Qm[(size_t)q * D + d] * qs) : (uint16_t)0;
        if (l2c && q < Q && d >= D && d < D + kBfL2Extra)
            h[e] = (d - D < 3) ? (uint16_t)0xBF80u : bf16_limb(l2c[q], (int)(d - D - 3));  // -1.0 three times, then limbs of c_j
    }
    uint4 o;
    o.x = h[0] | ((uint32_t)h[1] << 16); o.y = h[2] | ((uint32_t)h[3] << 16);
    o.z = h[4] | ((uint32_t)h[5] << 16); o.w = h[6] | ((uint32_t)h[7] << 16);
    Bb[u] = o;
}

__device__ __forceinline__ const char* uniform_ptr(const char* p) {  // pin a wave-uniform pointer into SGPRs
    const uint64_t v = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (const char*)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ void gload4(u32x4_t& dst, const char* base_uniform, uint32_t lane_off) {
    uint64_t base;  // s_mov_b64 first: see gload2 (kernels_gemm.h)
    asm volatile("s_mov_b64 %1, %3\n\tglobal_load_dwordx4 %0, %2, %1" : "=v"(dst), "=&s"(base) : "v"(lane_off), "s"(base_uniform) : "memory");
}
template <int N> __device__ __forceinline__ void use_after(u32x4_t& a, u32x4_t& b) {
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N));
}

struct alignas(16) GemmBf16Lds {
    alignas(16) char A[kBfStages * kBfStageBytes];
    uint32_t cnt[64 * kBfWaves];
    uint32_t thr[64 * kBfWaves];
};

// MODE 0: fused top-k filter. MODE 1: dump the dense score matrix (layout test).
template <int R, int MODE>
__global__ __launch_bounds__(64 * kBfWaves, 1) void gemm_bf16_filter_kernel(
    const char* __restrict__ Ab, const char* __restrict__ Bb, uint32_t ntiles, uint32_t N, uint32_t nk, size_t Qpad, uint32_t nqt,
    uint32_t qtg, uint32_t tiles_per_slice, uint64_t* __restrict__ lists, uint32_t* __restrict__ counts, uint32_t KP, uint32_t kk,
    uint32_t* __restrict__ errflag, uint32_t* gslots, uint32_t* gthr, float* __restrict__ dump, size_t ld_dump) {
    const float* const kmargin = reinterpret_cast<const float*>(gthr + Qpad);  // 2E per query: the k rule of topk_dev.h
    constexpr bool COS = false, U8 = false, L2K = false;  // kind flags of the shared epilogue: plain dot scores
    const float* const invn = nullptr;
    const float* const invq = nullptr;
    const float scale = 1.0f;
    const float iq_lane[2] = {1.0f, 1.0f};
    constexpr int kBQ = 64 * kBfWaves;
    constexpr int kEpiTgWait = 5 * (kBfStages - 3) + 4 + 5;  // gemm_epilogue.inc: nothing this wave still prefetches is waited for
    __shared__ GemmBf16Lds s;
    constexpr uint32_t cap = 64 * R;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wu = __builtin_amdgcn_readfirstlane(w);
    // block -> (slice, query tile): as in gemm_filter_kernel
    const uint32_t b = blockIdx.x;
    const uint32_t xcd = b & 7, lb = b >> 3, groups = nqt / qtg;
    const uint32_t qt = (xcd % groups) * qtg + lb % qtg;
    const uint32_t slice = (lb / qtg) * (8 / groups) + xcd / groups;
    const size_t q0 = (size_t)qt * kBQ;
    uint32_t t0 = slice * tiles_per_slice, t1 = t0 + tiles_per_slice;
    if (t1 > ntiles) t1 = ntiles;
    if (t0 > t1) t0 = t1;
    const uint32_t total = (t1 - t0) * nk;

    s.cnt[threadIdx.x] = 0;
    s.thr[threadIdx.x] = 0;
    uint64_t* my_lists = lists + ((size_t)slice * Qpad + q0) * cap;

    f32x16 acc[4][2];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[rt][ct][g] = 0.0f;

    // operand addresses: wave-uniform base + constant lane offset
    const char* sa = Ab + (size_t)t0 * nk * kBfStageBytes + (size_t)wu * 1024;  // this wave's 1-KiB piece of step 0
    const uint32_t va = (uint32_t)lane * 16u;
    const uint32_t lds0 = lds_addr_uniform(&s.A[0]) + (uint32_t)wu * 1024u;
    const size_t b_step = (size_t)4 * Qpad * 16, b_depth = (size_t)2 * Qpad * 16, b_ct = 32 * 16;
    const char* sb = Bb + (q0 + (size_t)wu * 64) * 16;
    const uint32_t vb = ((uint32_t)(lane >> 5) * (uint32_t)Qpad + (uint32_t)(lane & 31)) * 16u;
    uint32_t a_issued = 0, b_ks = 0;
    const uint32_t last = total ? total - 1 : 0;
    auto issue_a = [&]() {  // DMA of step a_issued; past the end of the slice: the last step again, into a stage nobody reads
        const uint32_t st = a_issued < total ? a_issued : last;
        glds16(uniform_ptr(sa + (size_t)st * kBfStageBytes), va, lds0 + (a_issued % kBfStages) * kBfStageBytes);
        ++a_issued;
    };
    auto issue_b = [&](u32x4_t& d0, u32x4_t& d1, int m) {
        const char* p = uniform_ptr(sb + (size_t)b_ks * b_step + (size_t)m * b_depth);
        gload4(d0, p, vb);
        gload4(d1, uniform_ptr(p + b_ct), vb);
    };
    u32x4_t breg[kBfLead][4];  // [step % kBfLead][2 m + ct]
#pragma unroll
    for (int r = 0; r < kBfLead; ++r)
#pragma unroll
        for (int x = 0; x < 4; ++x) breg[r][x] = u32x4_t{0u, 0u, 0u, 0u};
    if (total) {
        for (int i = 0; i < kBfStages - 2; ++i) issue_a();
#pragma unroll
        for (int r = 0; r < kBfLead; ++r) {
            issue_b(breg[r][0], breg[r][1], 0);
            issue_b(breg[r][2], breg[r][3], 1);
            b_ks = (b_ks + 1 == nk) ? 0 : b_ks + 1;
        }
    }
    wait_all();
#pragma unroll
    for (int r = 0; r < kBfLead; ++r) {
        use_after<0>(breg[r][0], breg[r][1]);
        use_after<0>(breg[r][2], breg[r][3]);
    }
    __syncthreads();

    uint32_t tg_next[2] = {0u, 0u};
    if (MODE == 0) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) gload1_agent(tg_next[ct], gthr + q0 + 64 * wu + ct, 8u * (uint32_t)(lane & 31));
    }
    uint32_t tile = t0, ks = 0;
    for (uint32_t step0 = 0; step0 < total; step0 += kBfLead) {
#pragma unroll
        for (int r = 0; r < kBfLead; ++r) {  // register ring position = step % kBfLead: static. nk is even, so total is too.
            const uint32_t step = step0 + r;
            const char* stage = s.A + (step % kBfStages) * kBfStageBytes;
            issue_a();  // step + kBfStages - 2
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                bf16x8_t a[4];
#pragma unroll
                for (int rt = 0; rt < 4; ++rt)
                    a[rt] = *reinterpret_cast<const bf16x8_t*>(stage + ((2 * m + (lane >> 5)) * 128 + rt * 32 + (lane & 31)) * 16);
                // ops younger than this depth's operands (loaded kBfLead steps ago, right after the same depth's MFMAs):
                // the rest of that step, kBfLead - 1 whole steps of 5, this step's DMA and (depth 1) depth 0's reload
                constexpr int kYounger = 5 * kBfLead - 2;
                if (m == 0) use_after<kYounger>(breg[r][0], breg[r][1]);
                else use_after<kYounger>(breg[r][2], breg[r][3]);
                const bf16x8_t b0 = __builtin_bit_cast(bf16x8_t, breg[r][2 * m]), b1 = __builtin_bit_cast(bf16x8_t, breg[r][2 * m + 1]);
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) {
                    acc[rt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[rt], b0, acc[rt][0], 0, 0, 0);
                    acc[rt][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[rt], b1, acc[rt][1], 0, 0, 0);
                }
                issue_b(breg[r][2 * m], breg[r][2 * m + 1], m);  // the same registers, kBfLead steps ahead
                __builtin_amdgcn_sched_barrier(0);
            }
            b_ks = (b_ks + 1 == nk) ? 0 : b_ks + 1;
            // nk is a multiple of kBfLead, so a tile always ends on the last ring position: ONE copy of the epilogue in
            // the unrolled loop (two copies made the kernel 81 KB, more than the 64 KB instruction cache)
#ifdef INNR_GEMM_PROBE_SKIPEPI  // tools/bf16_epi_probe.hip: the same code, the epilogue never taken (dump == nullptr at run time)
            if (r == kBfLead - 1 && ks + 1 == nk && dump != nullptr) {
#else
            if (r == kBfLead - 1 && ks + 1 == nk) {
#endif
#include "gemm_epilogue.inc"
                ks = 0;
                ++tile;
            } else {
                ++ks;
            }
            // ONE barrier per kLead = 2 K-steps: between two barriers the block reads stages s, s + 1 and its DMAs write the
            // stages of steps s + 6, s + 7 -- last read two steps before the previous barrier, never one of the two in use.
            // At the barrier this wave's pieces of steps s + 2 and s + 3 must have landed: the younger one was issued at the
            // top of step s - 3, with 4 + 5 (STAGES - 4) VMEM ops of this wave behind it (more after an epilogue: the wait
            // is then only stricter). The waves of a SIMD drift apart inside the two-step window instead of meeting at a
            // barrier every 16 MFMAs.
            if (r == kBfLead - 1) {
                wait_but_youngest<5 * (kBfStages - 4) + 4>();
                __syncthreads();
            }
        }
    }
    wait_all();
    __syncthreads();
    if (MODE == 0) {
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

}  // namespace innr
