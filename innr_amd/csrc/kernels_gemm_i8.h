// kernels_gemm_i8.h -- scalar::batch_knn_u8 (src/scalar.rs:370-393) with the FILTER on the integer matrix pipe:
// v_mfma_i32_32x32x32_i8, twice the bf16 MFMA rate, 32x the f32 MFMA rate the "path B" kernel (kernels_gemm.h, kGemmU8) runs at.
//
// Reference hot loop: mixed_dot_u8_f32_portable (scalar.rs:353-358), sum_d q_d * (c_d as f32), driven per document by
// asymmetric_dot_u8_precomputed (:284-300) inside batch_knn_u8 (:370-393).
//
// The corpus side is already 8-bit: c' = c - 128 (one XOR) is an exact i8. The f32 query becomes a 16-bit fixed-point
// number per dimension, t_d = round(q_d / s_j) with one scale s_j = max|q_d| / T per query, split into two i8 LIMBS
// t = 256 r1 + r2 (r2 in [-128,127], |r1| <= R1). Two integer GEMMs over the same corpus tile, hi = sum c' r1 and
// lo = sum c' r2, give V = 256 hi + lo = sum_d c'_d t_d EXACTLY (T is chosen so that |V| < 2^31), and
//     approx(j, i) = A_j V + B_j,   A_j = (alpha/255) s_j,   B_j = (128 alpha/255 + offset) sum(q_j)
// differs from the reference's score only by the query's quantisation, |alpha/255| (s_j/2) sum_d |c'_d| <= |alpha/255| s_j 64 D,
// and float rounding: ~1e-3 of the gap between neighbouring top scores at C3. The approximate scores are only a FILTER
// (like gemm_filter_kernel's): candidates go into the same per-(slice, query) lists, the caller re-scores them in the
// reference's f32 order (rescore_u8_kernel) and proves the answer against that bound; unproven queries are redone exactly.
//
// Layouts (built once per corpus / once per call by the pack kernels below):
//   corpus  Ai8[tile][ks][kg 0..3][rt 0..3][i 0..31][16 i8]   corpus row = 128 tile + 4 i + rt, dimension = 64 ks + 16 kg + e
//           one K-step (64 dimensions) of a 128-row tile is 8 KiB contiguous = eight 1-KiB LDS-DMA pieces, one per wave;
//           an A fragment (row i of row tile rt, 16 consecutive k) is one conflict-free ds_read_b128.
//   queries Bq[ks][kg 0..3][limb 0..1][query 0..Qpad)[16 i8]   one 16-byte load per lane and fragment, straight from L2.
//   (The hardware's k order inside a 32-deep MFMA step is the same for the A and the B operand, and the sum over k does
//    not depend on it: both sides are packed "lane half h holds dimensions 16 (2 m + h) .. + 15 of depth m".)
// Block = 8 waves; tile 128 corpus rows x 256 queries: wave w owns queries [32 w, 32 w + 32) x 128 rows x both limbs =
// 4 x 2 accumulator tiles of 32 x 32 i32 (128 VGPRs), so a lane holds 64 corpus rows of ONE query. The K-loop is
// gemm_bf16_filter_kernel's: LDS ring of 8 stages, DMA six steps ahead, query fragments two steps ahead in a register
// ring, one barrier per K-step, every step the same VMEM sequence (1 DMA + 2 x 2 fragment loads) with hand-counted waits.
// Roofline: integer MFMA; algorithmic ops 2 Q N D (x 2 limbs executed); corpus bytes N D per 256 queries.
#pragma once

#include "kernels_gemm_bf16.h"
#include "kernels_u8.h"

namespace innr {

typedef int i32x4_t __attribute__((ext_vector_type(4)));
typedef int i32x16_t __attribute__((ext_vector_type(16)));

// The one-limb kernel re-derives a query's chip-wide bound for one admitted candidate in 32 (kPubEvery, topk_dev.h, says 16 for
// the other kernels): a re-derivation loads the query's slots, and that wait sits out everything the wave has in flight. C2
// shape, kernel ms (tools/lib_ab.py over builds with -DINNR_I8H_PUB_EVERY): 1: 12.97, 2: 11.99, 4: 11.3-11.5, 8: 11.19, 16: 11.00,
// 32: 10.73, 64: 10.89; C3: 43.2 -> 42.8.
#ifndef INNR_I8H_PUB_EVERY
#define INNR_I8H_PUB_EVERY 32
#endif
constexpr uint32_t kI8hPubEvery = INNR_I8H_PUB_EVERY;
constexpr int kI8Stages = 8, kI8Lead = 2, kI8StageBytes = 8192, kI8Waves = 8, kI8K = 64, kI8BQ = 256;
static_assert(kI8Stages == 8 && kI8Lead == 2, "the one-barrier-per-two-steps schedule is derived for an 8-stage ring and a 2-step register ring");

// The query value is t = (r1 << S) + r2, r2 in [-2^(S-1), 2^(S-1) - 1], |r1| <= R1: the largest magnitude
// T = (R1 << S) + 2^(S-1) - 1 such that |V| <= D * 128 * T stays below 2^31. S = 8: two full int8 limbs, both on the matrix
// pipe (gemm_i8_filter_kernel); S = 6: a 14-bit value whose low limb is small enough to be bounded away in the filter and
// computed only for the survivors (gemm_i8h_filter_kernel).
__host__ __device__ inline uint32_t i8_limb_r1(uint32_t D, uint32_t S = 8) {
    const uint64_t tmax = (((uint64_t)1 << 31) - 1) / ((uint64_t)(D ? D : 1) * 128);
    const uint64_t lo_max = ((uint64_t)1 << (S - 1)) - 1;
    if (tmax < lo_max + ((uint64_t)1 << S)) return 0;  // not even one step of the high limb: the engine is off for this dimension
    const uint64_t r1 = (tmax - lo_max) >> S;
    return (uint32_t)(r1 > 127 ? 127 : r1);
}

// PDX codes C8[d*ldN + i] -> Ai8. One thread per (tile, ks, kg, i): 16 dimensions x 4 consecutive corpus rows (rows 4i .. 4i+3 of
// the tile are the four row tiles rt), read as 16 coalesced dwords, written as four 16-byte units.
__global__ __launch_bounds__(256) void pack_corpus_i8_kernel(const uint8_t* __restrict__ C8, size_t ldN, uint32_t N, uint32_t D,
                                                              uint32_t nk, size_t nthreads, uint4* __restrict__ Ai8) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= nthreads) return;
    const uint32_t i = (uint32_t)(t & 31), kg = (uint32_t)((t >> 5) & 3);
    const size_t tk = t >> 7;  // tile * nk + ks
    const uint32_t ks = (uint32_t)(tk % nk);
    const size_t row0 = (tk / nk) * 128 + 4 * (size_t)i;  // < ldN (ldN is a multiple of 1024)
    uint32_t w[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const uint32_t d = ks * 64 + kg * 16 + e;
        uint32_t v = 0x80808080u;  // code 128 -> c' = 0: padding dimensions contribute nothing
        if (d < D) v = *reinterpret_cast<const uint32_t*>(C8 + (size_t)d * ldN + row0);
        w[e] = v ^ 0x80808080u;  // c - 128 as i8, four rows at once
    }
    uint4* out = Ai8 + (tk * 4 + kg) * 128 + i;  // + rt * 32
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
        uint32_t o[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            uint32_t byte = (w[e] >> (8 * rt)) & 0xffu;
            if (row0 + rt >= N) byte = 0u;
            o[e >> 2] |= byte << (8 * (e & 3));
        }
        out[rt * 32] = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// An f32 PDX corpus as the int8 filter's operand: every value scalar-quantised with ONE (offset, alpha) for the whole corpus --
// quantize_u8 (scalar.rs:212-225) with the corpus' own range, i.e. the first stage of the two-stage pipeline the reference
// describes (scalar.rs:366-368), except that here the second stage PROVES the answer: |v_d - (offset + alpha c_d / 255)| <=
// alpha / 510 for every value inside the range, so the filter's score is within (alpha / 510) sum_d |q_d| of q.v.
// rowscale (nullable): 1/||v|| per row -- the cosine copy quantises the normalised rows (range [-1, 1]).
// sqn != null: the SQUARED-L2 copy (batch_knn, batch.rs:385-411, on this filter). C_j - |q - v|^2 = 2 q.v - |v|^2 + (C_j - |q|^2)
// is a plain DOT PRODUCT of augmented vectors -- [2q, -w1 (R times), -w2] . [v, z1 (R times), z2] + constants -- so the kernel and
// everything behind it stay the dot filter's: the row's |v|^2 = n is mapped into the corpus' own value range, z' = offset + alpha
// n / nmax, and carried as TWO 8-bit limbs in D' - D = R + 1 more dimensions: z1 = the quantised z' in R dimensions (each with
// query weight -w1 = -(nmax / alpha) / R, so that no single query entry dwarfs the 2 q_d and costs them their bits), z2 = the
// residual z' - dequant(z1), stretched by 255, in one (weight -w2 = -(nmax / alpha) / 255). |n - n^| <= nmax / 130050.
__global__ __launch_bounds__(256) void pack_corpus_f32_i8_kernel(const float* __restrict__ V, size_t ldN, uint32_t N, uint32_t D,
                                                                  uint32_t nk, size_t nthreads, float offset, float inv_alpha,
                                                                  const float* __restrict__ rowscale, uint4* __restrict__ Ai8,
                                                                  const float* __restrict__ sqn = nullptr, uint32_t R = 0,
                                                                  float inv_nmax = 0.0f) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= nthreads) return;
    const uint32_t i = (uint32_t)(t & 31), kg = (uint32_t)((t >> 5) & 3);
    const size_t tk = t >> 7;  // tile * nk + ks
    const uint32_t ks = (uint32_t)(tk % nk);
    const size_t row0 = (tk / nk) * 128 + 4 * (size_t)i;  // < ldN (a multiple of 256)
    float rs[4] = {1.0f, 1.0f, 1.0f, 1.0f};
    if (rowscale) {
        const float4 r4 = *reinterpret_cast<const float4*>(rowscale + row0);
        rs[0] = r4.x; rs[1] = r4.y; rs[2] = r4.z; rs[3] = r4.w;
    }
    uint32_t w[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const uint32_t d = ks * 64 + kg * 16 + e;
        uint32_t v = 0x80808080u;  // code 128 -> c' = 0: padding dimensions contribute nothing
        if (d < D) {
            const float4 x = *reinterpret_cast<const float4*>(V + (size_t)d * ldN + row0);
            v = (uint32_t)quantize_one(x.x * rs[0], offset, inv_alpha) | ((uint32_t)quantize_one(x.y * rs[1], offset, inv_alpha) << 8) |
                ((uint32_t)quantize_one(x.z * rs[2], offset, inv_alpha) << 16) | ((uint32_t)quantize_one(x.w * rs[3], offset, inv_alpha) << 24);
        } else if (sqn && d <= D + R) {  // the two limbs of |v|^2 (d < D + R: the first, d == D + R: the second)
            const float4 n4 = *reinterpret_cast<const float4*>(sqn + row0);
            const float nn[4] = {n4.x, n4.y, n4.z, n4.w};
            const float alpha = 255.0f / inv_alpha;
            v = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float z1 = offset + alpha * (nn[r] * inv_nmax);
                const uint32_t c1 = quantize_one(z1, offset, inv_alpha);
                uint32_t code = c1;
                if (d == D + R) {
                    const float res = z1 - (offset + alpha * ((float)c1 * (1.0f / 255.0f)));  // within +- alpha / 510
                    code = quantize_one(offset + 0.5f * alpha + res * 255.0f, offset, inv_alpha);
                }
                v |= code << (8 * r);
            }
        }
        w[e] = v ^ 0x80808080u;
    }
    uint4* out = Ai8 + (tk * 4 + kg) * 128 + i;
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
        uint32_t o[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            uint32_t byte = (w[e] >> (8 * rt)) & 0xffu;
            if (row0 + rt >= N) {
                // padding rows: c' = 0 (approximation = B_j: far below any bound of a dot / cosine call). Squared L2: c' = 0 in
                // the |v|^2 limbs would read as a vector of norm^2 nmax / 2 AT the range's centre -- closer to every query
                // than the corpus, survivors all (never appended: i >= N; at C2 the block holding them ran 23.6 M cycles against
                // the others' 15.9 M, 12.9 ms against 8.9 for the kernel): the largest |v|^2 the limbs can say instead.
                const uint32_t d = ks * 64 + kg * 16 + (uint32_t)e;
                byte = (sqn && d >= D && d <= D + R) ? 0x7fu : 0u;
            }
            o[e >> 2] |= byte << (8 * (e & 3));
        }
        out[rt * 32] = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// queries of the f32-corpus filter: (cosine: normalised copy,) sum and L1 norm per query
__global__ __launch_bounds__(64) void f32i8_query_prep_kernel(const float* __restrict__ Qm, uint32_t Q, uint32_t D, const float* __restrict__ qscale,
                                                              float* __restrict__ qhat, float* __restrict__ qsum, float* __restrict__ ql1) {
    // one wave per query (grid = Q); the sums are order-free (the bound below covers any order of D - 1 additions)
    const uint32_t j = blockIdx.x;
    if (j >= Q) return;
    const int lane = threadIdx.x;
    const float sc = qscale ? qscale[j] : 1.0f;
    float s1 = 0.0f, l1 = 0.0f;
    for (uint32_t d = lane; d < D; d += 64) {
        const float x = Qm[(size_t)j * D + d] * sc;
        if (qhat) qhat[(size_t)j * D + d] = x;
        s1 += x;
        l1 += fabsf(x);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        s1 += __shfl_xor(s1, off, 64);
        l1 += __shfl_xor(l1, off, 64);
    }
    if (lane == 0) {
        qsum[j] = s1;
        ql1[j] = l1 * (1.0f + 1.2e-7f * (float)D);  // an upper bound of the true L1 norm despite the rounding of the sum
    }
}

// the proof's bound per query: the int8 engine's own share (qc[3]: the query's quantisation, float rounding) + the corpus
// quantisation (alpha / 510 per value, clamp slack) + the reference's own f32 accumulation against the true dot
// centre = 128 alpha/255 + offset multiplies the query's SUM in B_j; that sum is accumulated in f32 (f32i8_query_prep_kernel), off
// the true one by up to (D - 1) u |q|_1 -- nothing for a corpus centred on zero, the dominant term for an off-centre range (ReLU
// embeddings, tf-idf in [0, 1]: |centre| / alpha = 1/2)
__global__ void f32i8_finish_bound_kernel(float* __restrict__ qc, uint32_t Qpad, uint32_t Q, const float* __restrict__ ql1,
                                          const float* __restrict__ qnorm, float alpha, float ref_scale, int cosine, float centre,
                                          float D, float l1_scale = 1.0f, const float* __restrict__ ql1_sum = nullptr) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= Q) return;
    const float ref = cosine ? ref_scale : ref_scale * qnorm[j];
    // (squared L2: ql1 = |q|_1 of the query, twice of which meets quantised corpus values; ql1_sum = that of the augmented query)
    const float sum_err = fabsf(centre) * D * 6.0e-8f * (ql1_sum ? ql1_sum[j] : ql1[j]);
    qc[3 * (size_t)Qpad + j] = (qc[3 * (size_t)Qpad + j] + (alpha / 510.0f) * 1.002f * l1_scale * ql1[j] + sum_err + ref) * 1.0001f;
}

// the augmented query of the squared-L2 copy: [2 q_d, -w1 (R times), -w2]
__global__ void f32i8_l2_queries_kernel(const float* __restrict__ Qm, uint32_t Q, uint32_t D, uint32_t R, float w1, float w2,
                                        float* __restrict__ Qaug /*[Q][D + R + 1]*/) {
    const uint32_t Da = D + R + 1;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)Q * Da) return;
    const uint32_t j = (uint32_t)(t / Da), d = (uint32_t)(t % Da);
    Qaug[t] = d < D ? 2.0f * Qm[(size_t)j * D + d] : (d < D + R ? -w1 : -w2);
}
// ... and its constants: B_j += K0 + (C_j - |q_j|^2), E_j += the encoding of |v|^2 + the f32 squared-L2 engine's own terms
__global__ void f32i8_l2_finish_kernel(float* __restrict__ qc, uint32_t Qpad, uint32_t Q, const float* __restrict__ cq /*C_j - |q_j|^2*/,
                                       const float* __restrict__ Cj, float K0, float enc_err, float l2_scale) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= Q) return;
    const float B = qc[(size_t)Qpad + j] + K0 + cq[j];
    qc[(size_t)Qpad + j] = B;
    qc[3 * (size_t)Qpad + j] = (qc[3 * (size_t)Qpad + j] + enc_err + l2_scale * Cj[j] + 4.8e-7f * (fabsf(B) + fabsf(K0) + fabsf(cq[j]))) * 1.0001f;
}

// Cj != null: squared L2 -- kth_scores are distances, the bound lives in the score space C_j - distance
__global__ void seed_thresholds_eq_kernel(const float* __restrict__ kth_scores /*[Q][KP], best first*/, uint32_t Q, uint32_t KP,
                                          const float* __restrict__ eq, uint32_t* __restrict__ seed, uint32_t Qpad, uint32_t pos,
                                          const float* __restrict__ Cj = nullptr) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= Qpad) return;
    uint32_t o = 0;
    if (j < Q) {
        const float x = kth_scores[(size_t)j * KP + pos];  // pos: see seed_thresholds_kernel
        const float t = (Cj ? Cj[j] - x : x) - eq[j] * 1.0001f - 1e-30f;
        if (t - t == 0.0f) o = f32_ord(t) - 1u;
    }
    seed[j] = o;
}

// One wave per query: scale, two limbs per dimension, per-query constants.
//   qc[0][j] = A_j, qc[1][j] = B_j, qc[2][j] = 1 / A_j, qc[3][j] = the query's share of the proof's error bound:
//   |approx - (alpha/255 * true mixed dot + offset * sum q)| <= |alpha/255| s_j 64 D (quantisation, sum|c'| <= 128 D) + float
//   rounding of A V + B (three roundings of values <= |A V| + |B|), with 2 % to spare; +inf when no finite scale exists
//   (the proof then fails and the query takes the exact engine).
//   qc[4][j] (as integer bits) = LOB_j = 128 * sum_d |r2_d| >= |sum_d c'_d r2_d|: the low limb's reach (gemm_i8h_filter_kernel)
__global__ __launch_bounds__(64) void pack_queries_i8_kernel(const float* __restrict__ Qm, const float* __restrict__ qsum, uint32_t Q,
                                                              uint32_t D, uint32_t nk, uint32_t Qpad, uint32_t R1, uint32_t S, float a255,
                                                              float offset, uint4* __restrict__ Bq, float* __restrict__ qc) {
    const uint32_t j = blockIdx.x;
    const int lane = threadIdx.x;
    const int half_lo = 1 << (S - 1);
    const float T = (float)((R1 << S) + (uint32_t)half_lo - 1u);
    float mx = 0.0f;
    bool finite = true;
    if (j < Q)
        for (uint32_t d = lane; d < D; d += 64) {
            const float x = Qm[(size_t)j * D + d];
            finite = finite && (x - x == 0.0f);
            mx = fmaxf(mx, fabsf(x));
        }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    finite = __all(finite);
    float s = mx / T;
    const bool usable = j < Q && finite && mx > 0.0f && s > 0.0f && (1.0f / s) - (1.0f / s) == 0.0f;
    if (!usable) s = 1.0f;
    const float inv_s = 1.0f / s;
    const uint32_t nchunks = nk * 4;  // 16-dimension chunks
    uint32_t abs_lo = 0;
    for (uint32_t c = lane; c < nchunks; c += 64) {
        uint32_t hi[4] = {0u, 0u, 0u, 0u}, lo[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const uint32_t d = c * 16 + e;
            int t = 0;
            if (usable && d < D) {
                float x = rintf(Qm[(size_t)j * D + d] * inv_s);
                x = fminf(fmaxf(x, -T), T);
                t = (int)x;
            }
            const int r1 = (t + half_lo) >> S;  // floor((t + 2^(S-1)) / 2^S): |r1| <= R1
            const int r2 = t - (r1 << S);       // [-2^(S-1), 2^(S-1) - 1]
            abs_lo += (uint32_t)(r2 < 0 ? -r2 : r2);
            hi[e >> 2] |= ((uint32_t)r1 & 0xffu) << (8 * (e & 3));
            lo[e >> 2] |= ((uint32_t)r2 & 0xffu) << (8 * (e & 3));
        }
        // c = ks * 4 + kg
        Bq[((size_t)c * 2 + 0) * Qpad + j] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
        Bq[((size_t)c * 2 + 1) * Qpad + j] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) abs_lo += (uint32_t)__shfl_xor((int)abs_lo, off, 64);
    if (lane == 0) {
        float A = 1.0f, B = 0.0f, E = 0.0f;
        if (j < Q) {
            const float qs = qsum[j];
            A = a255 * s;
            B = (128.0f * a255 + offset) * qs;
            const float vmax = 128.0f * (float)D * T;  // |V| <= this
            E = usable ? 1.02f * (fabsf(a255) * s * 64.0f * (float)D + 2.4e-7f * (fabsf(A) * vmax + fabsf(B)) +
                                  2.4e-7f * (fabsf(128.0f * a255 * qs) + fabsf(offset * qs)))
                       : __builtin_inff();
            if (!usable && mx == 0.0f && finite) E = 0.0f;  // the zero query: V = 0 and approx = B = exact for every document
            if (!(A > 0.0f) || !(E - E == 0.0f)) {
                A = 1.0f;  // never used for ranking claims: E = +inf sends the query to the exact engine
                E = __builtin_inff();
            }
        }
        qc[j] = A;
        qc[Qpad + j] = B;
        qc[2 * (size_t)Qpad + j] = 1.0f / A;
        qc[3 * (size_t)Qpad + j] = E;
        qc[4 * (size_t)Qpad + j] = __int_as_float((int)(128u * abs_lo));  // < 2^31: abs_lo <= 2^(S-1) D, D <= 65535
    }
}

// seed[j] = the KP-th best exact score of a corpus prefix, lowered by the query's whole error bound (rescore_u8_kernel's E):
// at least KP documents have an APPROXIMATE score above it, so it is a valid chip-wide bound. 0 = "no bound".
__global__ void seed_thresholds_u8_kernel(const float* __restrict__ kth_scores /*[Q][KP], best first*/, uint32_t Q, uint32_t KP,
                                          float err_scale, const float* __restrict__ qnorm, const float* __restrict__ qsum,
                                          float offset, const float* __restrict__ eq, uint32_t* __restrict__ seed, uint32_t Qpad,
                                          uint32_t pos) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= Qpad) return;
    uint32_t o = 0;
    if (j < Q) {
        const float x = kth_scores[(size_t)j * KP + pos];  // pos: see seed_thresholds_kernel
        const float E = err_scale * qnorm[j] + 4.8e-7f * fabsf(offset * qsum[j]) + eq[j];
        const float t = x - E * 1.0001f - 1e-30f;
        if (t - t == 0.0f) o = f32_ord(t) - 1u;
    }
    seed[j] = o;
}

template <int N> __device__ __forceinline__ void use_after1(uint32_t& a) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(a) : "n"(N)); }

struct alignas(16) GemmI8Lds {
    alignas(16) char A[kI8Stages * kI8StageBytes];
    uint32_t cnt[kI8BQ];
    uint32_t thr[kI8BQ];
};

// MODE 0: fused top-k filter. MODE 1: dump the dense approximate score matrix (layout test).
template <int R, int MODE>
__global__ __launch_bounds__(64 * kI8Waves, 1) void gemm_i8_filter_kernel(
    const char* __restrict__ Ai8, const char* __restrict__ Bq, uint32_t ntiles, uint32_t N, uint32_t nk, size_t Qpad, uint32_t nqt,
    uint32_t qtg, uint32_t tiles_per_slice, const float* __restrict__ qc, uint64_t* __restrict__ lists, uint32_t* __restrict__ counts,
    uint32_t KP, uint32_t kk, uint32_t* __restrict__ errflag, uint32_t* gslots, uint32_t* gthr, float* __restrict__ dump, size_t ld_dump) {
    const float* const kmargin = reinterpret_cast<const float*>(gthr + Qpad);  // 2E per query: the k rule of topk_dev.h
    constexpr int kEpiTgWait = 5 * (kI8Stages - 3) + 4 + 5;
    __shared__ GemmI8Lds s;
    constexpr uint32_t cap = 64 * R;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wu = __builtin_amdgcn_readfirstlane(w);
    // block -> (slice, query tile): as in gemm_filter_kernel
    const uint32_t b = blockIdx.x;
    const uint32_t xcd = b & 7, lb = b >> 3, groups = nqt / qtg;
    const uint32_t qt = (xcd % groups) * qtg + lb % qtg;
    const uint32_t slice = (lb / qtg) * (8 / groups) + xcd / groups;
    const size_t q0 = (size_t)qt * kI8BQ;
    uint32_t t0 = slice * tiles_per_slice, t1 = t0 + tiles_per_slice;
    if (t1 > ntiles) t1 = ntiles;
    if (t0 > t1) t0 = t1;
    const uint32_t total = (t1 - t0) * nk;

    if (threadIdx.x < kI8BQ) {
        s.cnt[threadIdx.x] = 0;
        s.thr[threadIdx.x] = 0;
    }
    uint64_t* my_lists = lists + ((size_t)slice * Qpad + q0) * cap;

    i32x16_t acc[4][2];  // [row tile][limb: 0 = high, 1 = low]
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int L = 0; L < 2; ++L)
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[rt][L][g] = 0;

    const int half = lane >> 5, C = lane & 31;
    const int ql = 32 * w + C;  // this lane's query inside the block tile (lanes l and l + 32 share it)
    const float Aj = qc[q0 + ql], Bj = qc[Qpad + q0 + ql], invAj = qc[2 * Qpad + q0 + ql];

    // operand addresses: wave-uniform base + constant lane offset
    const char* sa = Ai8 + (size_t)t0 * nk * kI8StageBytes + (size_t)wu * 1024;  // this wave's 1-KiB piece of step 0
    const uint32_t va = (uint32_t)lane * 16u;
    const uint32_t lds0 = lds_addr_uniform(&s.A[0]) + (uint32_t)wu * 1024u;
    // Bq[ks][kg][limb][Qpad][16]: depth m of a lane half h is kg = 2 m + h
    const size_t b_step = (size_t)8 * Qpad * 16, b_depth = (size_t)4 * Qpad * 16, b_limb = Qpad * 16;
    const char* sb = Bq + (q0 + (size_t)wu * 32) * 16;
    const uint32_t vb = ((uint32_t)half * 2u * (uint32_t)Qpad + (uint32_t)C) * 16u;
    uint32_t a_issued = 0, b_ks = 0;
    const uint32_t last = total ? total - 1 : 0;
    auto issue_a = [&]() {  // DMA of step a_issued; past the end of the slice: the last step again, into a stage nobody reads
        const uint32_t st = a_issued < total ? a_issued : last;
        glds16(uniform_ptr(sa + (size_t)st * kI8StageBytes), va, lds0 + (a_issued % kI8Stages) * kI8StageBytes);
        ++a_issued;
    };
    auto issue_b = [&](u32x4_t& dhi, u32x4_t& dlo, int m) {
        const char* p = uniform_ptr(sb + (size_t)b_ks * b_step + (size_t)m * b_depth);
        gload4(dhi, p, vb);
        gload4(dlo, uniform_ptr(p + b_limb), vb);
    };
    u32x4_t breg[kI8Lead][4];  // [step % kI8Lead][2 m + limb]
#pragma unroll
    for (int r = 0; r < kI8Lead; ++r)
#pragma unroll
        for (int x = 0; x < 4; ++x) breg[r][x] = u32x4_t{0u, 0u, 0u, 0u};
    if (total) {
        for (int i = 0; i < kI8Stages - 2; ++i) issue_a();
#pragma unroll
        for (int r = 0; r < kI8Lead; ++r) {
            issue_b(breg[r][0], breg[r][1], 0);
            issue_b(breg[r][2], breg[r][3], 1);
            b_ks = (b_ks + 1 == nk) ? 0 : b_ks + 1;
        }
    }
    wait_all();
#pragma unroll
    for (int r = 0; r < kI8Lead; ++r) {
        use_after<0>(breg[r][0], breg[r][1]);
        use_after<0>(breg[r][2], breg[r][3]);
    }
    __syncthreads();

    uint32_t tg_next = 0u;
    if (MODE == 0) gload1_agent(tg_next, gthr + q0 + 32 * wu, 4u * (uint32_t)C);
    uint32_t tile = t0, ks = 0;
    for (uint32_t step0 = 0; step0 < total; step0 += kI8Lead) {
#pragma unroll
        for (int r = 0; r < kI8Lead; ++r) {  // register ring position = step % kI8Lead: static. nk is even, so total is too.
            const uint32_t step = step0 + r;
            const char* stage = s.A + (step % kI8Stages) * kI8StageBytes;
            issue_a();  // step + kI8Stages - 2
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                i32x4_t a[4];
#pragma unroll
                for (int rt = 0; rt < 4; ++rt)
                    a[rt] = *reinterpret_cast<const i32x4_t*>(stage + ((2 * m + half) * 128 + rt * 32 + C) * 16);
                // ops younger than this depth's operands: see gemm_bf16_filter_kernel (the same VMEM sequence per step)
                constexpr int kYounger = 5 * kI8Lead - 2;
                if (m == 0) use_after<kYounger>(breg[r][0], breg[r][1]);
                else use_after<kYounger>(breg[r][2], breg[r][3]);
                const i32x4_t bhi = __builtin_bit_cast(i32x4_t, breg[r][2 * m]), blo = __builtin_bit_cast(i32x4_t, breg[r][2 * m + 1]);
                // A tile's first MFMAs take the constant 0 as their C operand instead of 128 zeroed accumulator registers per
                // tile (a tile starts on ring position 0: nk is even): the epilogue leaves the registers as they are.
                if (r == 0 && m == 0 && ks == 0) {
                    const i32x16_t zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                    for (int rt = 0; rt < 4; ++rt) {
                        acc[rt][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[rt], bhi, zero, 0, 0, 0);
                        acc[rt][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[rt], blo, zero, 0, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int rt = 0; rt < 4; ++rt) {
                        acc[rt][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[rt], bhi, acc[rt][0], 0, 0, 0);
                        acc[rt][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[rt], blo, acc[rt][1], 0, 0, 0);
                    }
                }
                issue_b(breg[r][2 * m], breg[r][2 * m + 1], m);  // the same registers, kI8Lead steps ahead
                __builtin_amdgcn_sched_barrier(0);
            }
            b_ks = (b_ks + 1 == nk) ? 0 : b_ks + 1;
            if (r == kI8Lead - 1 && ks + 1 == nk) {
                // ---------------- epilogue for corpus tile `tile`: one query per lane, 64 corpus rows ----------------
                const size_t tb = (size_t)tile * 128;
                // V = 256 hi + lo (exact, |V| < 2^31), kept in the high limb's registers
#pragma unroll
                for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                    for (int g = 0; g < 16; ++g) acc[rt][0][g] = (int)(((uint32_t)acc[rt][0][g] << 8) + (uint32_t)acc[rt][1][g]);
                if (MODE == 1) {
                    const size_t q = q0 + ql;
#pragma unroll
                    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                        for (int g = 0; g < 16; ++g) {
                            const size_t i = tb + 4 * ((g & 3) + 8 * (g >> 2) + 4 * half) + rt;
                            dump[q * ld_dump + i] = __builtin_fmaf(Aj, (float)acc[rt][0][g], Bj);
                        }
                } else {
                    use_after1<kEpiTgWait>(tg_next);  // requested a whole tile ago (or before the first K-step)
                    const uint32_t tl = __hip_atomic_load(&s.thr[ql], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    const uint32_t thr = tl > tg_next ? tl : tg_next;
                    // The fast reject runs on the integers: approx >= thr  =>  V >= Tint, with Tint derived from the float
                    // threshold conservatively (a lower Tint only sends more sites to the exact float test below).
                    int32_t Tint = INT32_MIN;  // thr == 0: no bound yet
                    const float thr_f = ord_f32(thr);
                    if (thr != 0u) {
                        const float x = (thr_f - Bj) * invAj;
                        if (x >= 2147483520.0f || thr == 0xFFFFFFFFu) Tint = INT32_MAX;  // beyond every legal V (|V| < 2^31; all ones: a padding query, closed by the host)
                        else if (x > -2.0e9f) Tint = (int32_t)__builtin_floorf(x) - 2 - (int32_t)(fabsf(x) * 4.8e-7f);
                        // x <= -2e9 or NaN: everything passes
                    }
                    int32_t gbest[4];
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        int32_t m4 = INT32_MIN;
#pragma unroll
                        for (int g3 = 0; g3 < 4; ++g3)
#pragma unroll
                            for (int rt = 0; rt < 4; ++rt) m4 = m4 > acc[rt][0][4 * gq + g3] ? m4 : acc[rt][0][4 * gq + g3];
                        gbest[gq] = m4;
                    }
                    const int32_t b01 = gbest[0] > gbest[1] ? gbest[0] : gbest[1], b23 = gbest[2] > gbest[3] ? gbest[2] : gbest[3];
                    const bool hit = (b01 > b23 ? b01 : b23) >= Tint;
                    if (__any(hit)) {
                        uint64_t* lq = my_lists + (size_t)ql * cap;
                        bool admitted = false;
#pragma unroll
                        for (int gq = 0; gq < 4; ++gq) {
                            const bool ghit = hit && gbest[gq] >= Tint;
                            if (!__any(ghit)) continue;
                            if (ghit) {
#pragma unroll
                                for (int g3 = 0; g3 < 4; ++g3)
#pragma unroll
                                    for (int rt = 0; rt < 4; ++rt) {
                                        const int32_t V = acc[rt][0][4 * gq + g3];
                                        if (V < Tint) continue;
                                        const uint32_t o = f32_ord(__builtin_fmaf(Aj, (float)V, Bj));
                                        const size_t i = tb + 4 * (g3 + 8 * gq + 4 * half) + rt;
                                        if (o >= thr && i < N) {
                                            admitted = admitted || (((uint32_t)i & (kPubEvery - 1)) == 0);
                                            cand_append(lq, &s.cnt[ql], cap, cand_make(o, (uint32_t)i), errflag);
                                            gthr_raise(gslots + (q0 + ql) * (size_t)(kSlotMul * KP), kSlotMul * KP, o, (uint32_t)i);
                                        }
                                    }
                            }
                        }
                        // re-derive the chip-wide bound of the queries that asked for it (lanes l and l + 32 hold the same query)
                        unsigned long long m = __ballot(admitted);
                        m = (m | (m >> 32)) & 0xffffffffull;
                        while (m) {
                            const int L = __builtin_ctzll(m);
                            m &= m - 1;
                            const size_t qg = q0 + 32 * wu + L;  // wave-uniform
                            gthr_publish_select<(kSlotMul * (16 * R - 64) + 63) / 64>(gslots + qg * (size_t)(kSlotMul * KP), gthr + qg, KP, lane, kk, kmargin[qg]);
                        }
                        __builtin_amdgcn_wave_barrier();
                        // compact the lists of this wave's 32 queries that are running out of room
                        const uint32_t c = lane < 32 ? __hip_atomic_load(&s.cnt[32 * w + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) : 0u;
                        unsigned long long need = __ballot(c > cap - kGemmBurst);
                        while (need) {
                            const int j = __builtin_ctzll(need);
                            need &= need - 1;
                            const int qj = 32 * w + j;
                            const uint32_t cj = __builtin_amdgcn_readlane(c, j);
                            uint32_t t;
                            const uint32_t keep = wave_compact<R>(my_lists + (size_t)qj * cap, cj, KP, &t);
                            if (lane == 0) {
                                s.cnt[qj] = keep;
                                s.thr[qj] = t;
                                if (t > __hip_atomic_load(&gthr[q0 + qj], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                                    __hip_atomic_fetch_max(&gthr[q0 + qj], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            }
                        }
                        __builtin_amdgcn_wave_barrier();
                    }
                }
                if (MODE == 0) gload1_agent(tg_next, gthr + q0 + 32 * wu, 4u * (uint32_t)C);
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
            if (r == kI8Lead - 1) {
                wait_but_youngest<5 * (kI8Stages - 4) + 4>();
                __syncthreads();
            }
        }
    }
    wait_all();
    __syncthreads();
    if (MODE == 0) {
        const uint32_t c = lane < 32 ? __hip_atomic_load(&s.cnt[32 * w + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) : 0u;
        unsigned long long need = __ballot(c > KP);
        uint32_t mine = c;
        while (need) {
            const int j = __builtin_ctzll(need);
            need &= need - 1;
            const uint32_t cj = __builtin_amdgcn_readlane(c, j);
            uint32_t t;
            const uint32_t keep = wave_compact<R>(my_lists + (size_t)(32 * w + j) * cap, cj, KP, &t);
            if (lane == j) mine = keep;
        }
        if (lane < 32) counts[(size_t)slice * Qpad + q0 + 32 * w + lane] = mine;
    }
}


// =====================================================================================================================
// gemm_i8h_filter_kernel -- the same filter with ONE limb on the matrix pipe.
//
// The int8 filter is clock-limited: its time follows the MFMA work it executes, and the two-limb kernel
// executes twice the algorithmic 2 Q N D. Here the query value is t = (r1 << 6) + r2 with a 6-bit low limb: only
// hi = sum c' r1 runs on v_mfma_i32_32x32x32_i8; the low limb is BOUNDED in the fast reject,
//     V = (hi << 6) + lo >= Tint   =>   hi >= (Tint - LOB_j) >> 6,     LOB_j = 128 sum_d |r2_d| >= |lo|,
// and computed exactly (v_dot4_i32_i8 over the site's corpus row and the query's low limb, both read back from the packed
// arrays) only for the sites that survive that test: ~1.5x the sites the full V would let through -- queued in an LDS list per
// wave and finished four per memory round trip on the K-loop's own operand-ring registers (see the epilogue). V is
// then exact as before, so the candidates, the re-score and the proof are unchanged; the 14-bit query costs a 4x larger
// quantisation bound than the 16-bit one (still ~1e-2 of the gap between neighbouring top scores at C3).
// With one accumulator set per (row tile, query tile) a wave owns 64 queries again: block tile 128 rows x 512 queries, the
// corpus streamed once per 512 queries, the bf16 kernel's operand reuse (an LDS fragment feeds 2 MFMAs, an L2 fragment 4).
// =====================================================================================================================
#ifndef INNR_I8H_S
#define INNR_I8H_S 6
#endif
// Timing probes of the one-limb kernel are COMPILE-TIME switches (builds with -DINNR_I8H_PROBE=<bits>, tools/i8h_probe.py;
// the product library is built without them: no probe branch exists in its K-loop). Bits -- 1: the epilogue never visits (what the
// K-loop and the fast reject cost; the hits are counted so that the reject is not dead code); 8 / 16: the query fragments of every
// K-step come from step 0 (L1) / every DMA re-reads the slice's first stage (L2); 32: no fast reject either (the accumulators are
// folded into one word per lane: a probe build whose accumulators nobody reads loses its MFMAs to dead-code elimination -- the
// first compile-time version of bits 1 and 32 "measured" a 4.1 ms K-loop that multiplied nothing); 4: count visiting wave
// epilogues / survivors / bound re-derivations and the cycles they take into errflag[8..17]. Builds with bits 1, 8, 16 or 32 give
// wrong answers and their calls fail after filling the stats. 64 / 128 (with 1): no barrier in the K-loop / no LDS fragment reads.
#ifndef INNR_I8H_PROBE
#define INNR_I8H_PROBE 0
#endif
constexpr uint32_t kI8hProbe = INNR_I8H_PROBE;
constexpr int kI8hBQ = 512, kI8hS = INNR_I8H_S;

constexpr int kI8hSurvCap = 128;  // survivors a wave queues before it must finish them (at least one site's worth: 64)
// The queued survivors are finished every kI8hFlushEvery-th tile of a slice (every wave of the block at the same tile), or as soon as
// a wave's list holds kI8hFlushAt of them. C2 shape, kernel ms (tools/i8_ab.py over builds with -DINNR_I8H_FLUSH_EVERY=1 / 2 / 4 / 8 /
// 16 / 64): 8.29 / 8.14 / 8.13 / 8.15 / 8.45 / 8.77 -- a survivor that waits keeps its slot from raising the chip-wide bound, and
// past a few tiles that costs more survivors than the rarer visits save (profiles/r03_i8_flush_ab.txt).
#ifndef INNR_I8H_FLUSH_EVERY
#define INNR_I8H_FLUSH_EVERY 4
#endif
#ifndef INNR_I8H_FLUSH_AT
#define INNR_I8H_FLUSH_AT 48
#endif
constexpr int kI8hFlushEvery = INNR_I8H_FLUSH_EVERY, kI8hFlushAt = INNR_I8H_FLUSH_AT;
// survivors a visit finishes per memory round trip (64 / kI8hPer lanes each): a visit's length is its number of round trips to HBM
#ifndef INNR_I8H_PER
#define INNR_I8H_PER 4
#endif
constexpr int kI8hPer = INNR_I8H_PER;
static_assert(kI8hPer == 4 || kI8hPer == 8, "4 survivors x 16 lanes x 3 chunks, or 8 x 8 x 6, per pass of 48 chunks");
static_assert(kI8hFlushAt + 64 <= kI8hSurvCap + 64 && kI8hFlushEvery >= 1, "list geometry");
struct alignas(16) GemmI8hLds {
    alignas(16) char A[kI8Stages * kI8StageBytes];
    uint32_t cnt[kI8hBQ];
    uint32_t thr[kI8hBQ];
    uint32_t surv[kI8Waves][kI8hSurvCap][4];  // a visit's survivors: high limb, corpus row, lane | query column tile << 8
};

// lo(i, q) = sum_d c'_d r2_d of corpus row `row_in_tile` of tile `tile` and query column q, from the packed arrays
__device__ __forceinline__ int32_t i8_lo_dot(const char* __restrict__ Ai8, const char* __restrict__ Bq, size_t tile, uint32_t nk,
                                             uint32_t rt, uint32_t i_, size_t Qpad, size_t q) {
    const uint4* a = reinterpret_cast<const uint4*>(Ai8) + (tile * nk * 4) * 128 + rt * 32 + i_;  // + c * 128, c = ks * 4 + kg
    const uint4* b = reinterpret_cast<const uint4*>(Bq) + Qpad + q;                                  // + c * 2 * Qpad (limb 1)
    int32_t acc = 0;
    const uint32_t nchunks = nk * 4;
#pragma unroll 4
    for (uint32_t c = 0; c < nchunks; ++c) {
        const uint4 x = a[(size_t)c * 128], y = b[(size_t)c * 2 * Qpad];
        acc = __builtin_amdgcn_sdot4((int)x.x, (int)y.x, acc, false);
        acc = __builtin_amdgcn_sdot4((int)x.y, (int)y.y, acc, false);
        acc = __builtin_amdgcn_sdot4((int)x.z, (int)y.z, acc, false);
        acc = __builtin_amdgcn_sdot4((int)x.w, (int)y.w, acc, false);
    }
    return acc;
}

// MODE 0: fused top-k filter. MODE 1: dump the dense approximate score matrix (layout test; computes every low limb).
// MODE 2: COLLECT (the completion pass, knn_f32_i8 in api.hip): every query has a FIXED threshold (gthr[q], set by the host: its
//   k-th exact score so far, less its bound E) and every site whose exact V clears it is appended to the query's GLOBAL list
//   (`lists` = uint32 indices [Qpad][KP], `counts` = uint32 [Qpad] lengths, KP = the capacity; an overfull list keeps counting).
template <int R, int MODE>
__global__ __launch_bounds__(64 * kI8Waves, 1) void gemm_i8h_filter_kernel(
    const char* __restrict__ Ai8, const char* __restrict__ Bq, uint32_t ntiles, uint32_t N, uint32_t nk, size_t Qpad, uint32_t nqt,
    uint32_t qtg, uint32_t tiles_per_slice, const float* __restrict__ qc, uint64_t* __restrict__ lists, uint32_t* __restrict__ counts,
    uint32_t KP, uint32_t kk, uint32_t* __restrict__ errflag, uint32_t* gslots, uint32_t* gthr, float* __restrict__ dump, size_t ld_dump) {
    const float* const kmargin = reinterpret_cast<const float*>(gthr + Qpad);  // 2E per query: the k rule of topk_dev.h
    constexpr int kEpiTgWait = 5 * (kI8Stages - 3) + 4 + 5;
    constexpr int S = kI8hS;
    __shared__ GemmI8hLds s;
    constexpr uint32_t cap = 64 * R;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wu = __builtin_amdgcn_readfirstlane(w);
    const uint32_t b = blockIdx.x;
    const uint32_t xcd = b & 7, lb = b >> 3, groups = nqt / qtg;
    const uint32_t qt = (xcd % groups) * qtg + lb % qtg;
    const uint32_t slice = (lb / qtg) * (8 / groups) + xcd / groups;
    const size_t q0 = (size_t)qt * kI8hBQ;
    uint32_t t0 = slice * tiles_per_slice, t1 = t0 + tiles_per_slice;
    if (t1 > ntiles) t1 = ntiles;
    if (t0 > t1) t0 = t1;
    const uint32_t total = (t1 - t0) * nk;

    s.cnt[threadIdx.x] = 0;  // kI8hBQ == threads
    s.thr[threadIdx.x] = 0;
    uint64_t* my_lists = lists + ((size_t)slice * Qpad + q0) * cap;
    // (Tried: waves whose 64 queries are all padding skip their fragment loads, MFMAs and epilogue -- a batch smaller than the tile
    //  pays the MFMA work of 512 queries, 59 % of the int8 pipe on padding at one query. The wave-uniform branch around the K-step
    //  made the compiled loop slower for EVERY batch size: 2.67 -> 3.05 ms at 1 query, 4.20 -> 5.17 at 512; not kept.)
    i32x16_t acc[4][2];  // [row tile][query column tile]: the HIGH limb's sums
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[rt][ct][g] = 0;

    const int half = lane >> 5, C = lane & 31;
    // this lane's two queries inside the block tile: 64 w + 32 ct + C (lanes l and l + 32 share them)
    float Aj[2], Bj[2], invAj[2];
    int32_t lob[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        const size_t q = q0 + 64 * w + 32 * ct + C;
        Aj[ct] = qc[q];
        Bj[ct] = qc[Qpad + q];
        invAj[ct] = qc[2 * Qpad + q];
        lob[ct] = __float_as_int(qc[4 * Qpad + q]);
    }

    const char* sa = Ai8 + (size_t)t0 * nk * kI8StageBytes + (size_t)wu * 1024;
    const uint32_t va = (uint32_t)lane * 16u;
    const uint32_t lds0 = lds_addr_uniform(&s.A[0]) + (uint32_t)wu * 1024u;
    // Bq[ks][kg][limb][Qpad][16]: depth m of a lane half h is kg = 2 m + h; only limb 0 feeds the matrix pipe
    const size_t b_step = (MODE == 0 && (kI8hProbe & 8)) ? 0 : (size_t)8 * Qpad * 16, b_depth = (size_t)4 * Qpad * 16, b_ct = 32 * 16;
    const char* sb = Bq + (q0 + (size_t)wu * 64) * 16;
    const uint32_t vb = ((uint32_t)half * 2u * (uint32_t)Qpad + (uint32_t)C) * 16u;
    uint32_t a_issued = 0, b_ks = 0;
    const uint32_t last = total ? total - 1 : 0;
    auto issue_a = [&]() {
        const uint32_t st = (MODE == 0 && (kI8hProbe & 16)) ? 0u : (a_issued < total ? a_issued : last);
        glds16(uniform_ptr(sa + (size_t)st * kI8StageBytes), va, lds0 + (a_issued % kI8Stages) * kI8StageBytes);
        ++a_issued;
    };
    auto issue_b = [&](u32x4_t& d0, u32x4_t& d1, int m) {
        const char* p = uniform_ptr(sb + (size_t)b_ks * b_step + (size_t)m * b_depth);
        gload4(d0, p, vb);
        gload4(d1, uniform_ptr(p + b_ct), vb);
    };
    auto issue_b_at = [&](u32x4_t& d0, u32x4_t& d1, int m, uint32_t kidx) {
        const char* p = uniform_ptr(sb + (size_t)kidx * b_step + (size_t)m * b_depth);
        gload4(d0, p, vb);
        gload4(d1, uniform_ptr(p + b_ct), vb);
    };
    u32x4_t breg[kI8Lead][4];  // [step % kI8Lead][2 m + ct]
#pragma unroll
    for (int r = 0; r < kI8Lead; ++r)
#pragma unroll
        for (int x = 0; x < 4; ++x) breg[r][x] = u32x4_t{0u, 0u, 0u, 0u};
    if (total) {
        for (int i = 0; i < kI8Stages - 2; ++i) issue_a();
#pragma unroll
        for (int r = 0; r < kI8Lead; ++r) {
            issue_b(breg[r][0], breg[r][1], 0);
            issue_b(breg[r][2], breg[r][3], 1);
            b_ks = (b_ks + 1 == nk) ? 0 : b_ks + 1;
        }
    }
    wait_all();
#pragma unroll
    for (int r = 0; r < kI8Lead; ++r) {
        use_after<0>(breg[r][0], breg[r][1]);
        use_after<0>(breg[r][2], breg[r][3]);
    }
    __syncthreads();

    uint32_t tg_next[2] = {0u, 0u};
    if (MODE != 1) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) gload1_agent(tg_next[ct], gthr + q0 + 64 * wu + 32 * ct, 4u * (uint32_t)C);
    }
    uint32_t tile = t0, ks = 0;
    // probe bit 4 (tools/i8h_probe.py): per wave, flushed once at the end -- per-event atomics on one address slowed the kernel 5x
#ifdef INNR_I8H_SETPRIO  // A/B switch (MI355X_MICROARCH.md, two waves per SIMD, item 4): static priority for the block's younger half
    if (wu >= 4) __builtin_amdgcn_s_setprio(1);
#endif
    uint32_t ns = 0;  // survivors queued in this wave's LDS list (wave-uniform; carried from tile to tile)
    uint32_t pc_nvis = 0, pc_nsurv = 0, pc_npub = 0, pc_napp = 0, pc_nhit = 0;
    unsigned long long pc_visit = 0, pc_surv = 0, pc_tail = 0, pc_queue = 0;
    const unsigned long long pc_t0 = (kI8hProbe & 4) ? __builtin_readcyclecounter() : 0ull;  // cycles inside visits / inside the survivors' loops / publish + compaction
    for (uint32_t step0 = 0; step0 < total; step0 += kI8Lead) {
#pragma unroll
        for (int r = 0; r < kI8Lead; ++r) {
            const uint32_t step = step0 + r;
            const char* stage = s.A + (step % kI8Stages) * kI8StageBytes;
            issue_a();
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                i32x4_t a[4];
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) {
                    if ((kI8hProbe & 128) && step0 > 0) a[rt] = i32x4_t{(int)step, (int)lane, rt, m};  // timing only: no LDS fragment reads
                    else a[rt] = *reinterpret_cast<const i32x4_t*>(stage + ((2 * m + half) * 128 + rt * 32 + C) * 16);
                }
                constexpr int kYounger = 5 * kI8Lead - 2;
                if (m == 0) use_after<kYounger>(breg[r][0], breg[r][1]);
                else use_after<kYounger>(breg[r][2], breg[r][3]);
                const i32x4_t b0 = __builtin_bit_cast(i32x4_t, breg[r][2 * m]), b1 = __builtin_bit_cast(i32x4_t, breg[r][2 * m + 1]);
                if (r == 0 && m == 0 && ks == 0) {  // a tile's first MFMAs: C = 0 (no accumulator reset per tile)
                    const i32x16_t zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                    for (int rt = 0; rt < 4; ++rt) {
                        acc[rt][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[rt], b0, zero, 0, 0, 0);
                        acc[rt][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[rt], b1, zero, 0, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int rt = 0; rt < 4; ++rt) {
                        acc[rt][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[rt], b0, acc[rt][0], 0, 0, 0);
                        acc[rt][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[rt], b1, acc[rt][1], 0, 0, 0);
                    }
                }
                issue_b(breg[r][2 * m], breg[r][2 * m + 1], m);
                __builtin_amdgcn_sched_barrier(0);
            }
            b_ks = (b_ks + 1 == nk) ? 0 : b_ks + 1;
            if (r == kI8Lead - 1 && ks + 1 == nk) {
                // ---------------- epilogue for corpus tile `tile`: two queries per lane, 64 corpus rows each ----------------
                const size_t tb = (size_t)tile * 128;
                if (MODE == 1) {
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) {
                        const size_t q = q0 + 64 * w + 32 * ct + C;
#pragma unroll
                        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                            for (int g = 0; g < 16; ++g) {
                                const uint32_t i_ = (uint32_t)((g & 3) + 8 * (g >> 2) + 4 * half);
                                const size_t i = tb + 4 * i_ + rt;
                                const int32_t lo = i8_lo_dot(Ai8, Bq, tile, nk, (uint32_t)rt, i_, Qpad, q);
                                const int32_t V = (int32_t)(((uint32_t)acc[rt][ct][g] << S) + (uint32_t)lo);
                                dump[q * ld_dump + i] = __builtin_fmaf(Aj[ct], (float)V, Bj[ct]);
                            }
                    }
                } else if (kI8hProbe & 32) {  // timing only: no fast reject -- the accumulators are only folded into one word per lane, so that no MFMA is dead code
                    use_after<kEpiTgWait>(tg_next[0], tg_next[1]);
                    int32_t fold = 0;
#pragma unroll
                    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                            for (int g = 0; g < 16; ++g) fold ^= acc[rt][ct][g];
                    pc_nsurv += (uint32_t)fold;
                } else {
                    use_after<kEpiTgWait>(tg_next[0], tg_next[1]);
                    uint32_t thr[2];
                    int32_t Tint[2], Thi[2];
                    float thr_f[2];
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) {
                        const uint32_t tl = __hip_atomic_load(&s.thr[64 * w + 32 * ct + C], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                        thr[ct] = tl > tg_next[ct] ? tl : tg_next[ct];
                        thr_f[ct] = ord_f32(thr[ct]);
                        // approx >= thr  =>  V >= Tint (conservative float -> integer, as in gemm_i8_filter_kernel)
                        //               =>  (hi << S) >= Tint - LOB  =>  hi >= floor((Tint - LOB) / 2^S)
                        Tint[ct] = INT32_MIN;
                        Thi[ct] = INT32_MIN;
                        if (thr[ct] != 0u) {
                            const float x = (thr_f[ct] - Bj[ct]) * invAj[ct];
                            if (x >= 2147483520.0f || thr[ct] == 0xFFFFFFFFu) {  // beyond every legal V (|V| < 2^31), or a padding query
                                Tint[ct] = INT32_MAX;
                                Thi[ct] = INT32_MAX;
                            } else if (x > -2.0e9f) {
                                Tint[ct] = (int32_t)__builtin_floorf(x) - 2 - (int32_t)(fabsf(x) * 4.8e-7f);
                                const long long d = (long long)Tint[ct] - (long long)lob[ct];
                                Thi[ct] = d < -2147483000ll ? INT32_MIN : (int32_t)(d >> S);
                            }
                        }
                    }
                    // fast reject on the high limb: one maximum per group of 16 consecutive corpus rows and query column
                    int32_t gbest[2][4];
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                        for (int gq = 0; gq < 4; ++gq) {
                            int32_t m4 = INT32_MIN;
#pragma unroll
                            for (int g3 = 0; g3 < 4; ++g3)
#pragma unroll
                                for (int rt = 0; rt < 4; ++rt) m4 = m4 > acc[rt][ct][4 * gq + g3] ? m4 : acc[rt][ct][4 * gq + g3];
                            gbest[ct][gq] = m4;
                        }
                    bool hit[2];
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) {
                        const int32_t b01 = gbest[ct][0] > gbest[ct][1] ? gbest[ct][0] : gbest[ct][1];
                        const int32_t b23 = gbest[ct][2] > gbest[ct][3] ? gbest[ct][2] : gbest[ct][3];
                        hit[ct] = (b01 > b23 ? b01 : b23) >= Thi[ct];
                    }
                    constexpr bool probe = (kI8hProbe & 4) != 0;
                    // ---- survivors: QUEUED per tile, FINISHED per kI8hFlushEvery tiles, by all waves of the block at once ----
                    // Finishing a survivor needs memory (its corpus row and its query's low limb): a wave that does it waits for
                    // everything it has in flight and holds its block at the next barrier meanwhile -- measured (compile-time
                    // probes, C2): K-loop + fast reject 4.2 ms, with a visit in every wave epilogue that has a survivor 8.5 ms
                    // (380 K visits of 9.7 K cycles, two or three per block tile, each stalling eight waves). So a tile's
                    // epilogue only QUEUES its survivors in the wave's LDS list (no vector memory instruction, the operand
                    // ring stays in flight), and the lists are finished in a VISIT that every wave of the block makes at the
                    // same tile: every kI8hFlushEvery-th tile of the slice and its last one -- the block stalls once for all
                    // eight -- or earlier when a wave's list runs full (early tiles, weak bounds). A queued survivor reaches
                    // the candidate lists and the chip-wide bound a few tiles late: the bound only lags, nothing is lost (the
                    // exact test at the visit uses the thresholds of that moment).
                    // (probe bit 1: the hits are only counted -- the count keeps the fast reject, and with it every MFMA, alive in that build)
                    if ((kI8hProbe & 1) && __any(hit[0] || hit[1])) ++pc_nvis;
                    const bool any_hit = __any(hit[0] || hit[1]) && !(kI8hProbe & 1);
                    const bool sync_flush = (tile + 1 == t1) || ((tile - t0) % (uint32_t)kI8hFlushEvery == (uint32_t)kI8hFlushEvery - 1u);
                    if (any_hit || (sync_flush && ns)) {
                        bool admitted[2] = {false, false};
                        // Finish the queued survivors FOUR per memory round trip. A survivor needs lo = the dot of its corpus
                        // row with its query's low limb (nk * 4 chunks of 16 dimensions: a 16-byte load of each, four
                        // v_dot4_i32_i8); quarter g of the wave computes the one of entry g, each of its 16 lanes every 16th
                        // chunk with all its loads requested before the first is used, then a 4-step reduction; the lane that
                        // owns the query runs the exact test and the append.
                        auto flush = [&]() {
                            __builtin_amdgcn_wave_barrier();
                            const unsigned long long ps0 = probe ? __builtin_readcyclecounter() : 0ull;
                            constexpr int PER = kI8hPer, LPS = 64 / PER, CPL = 48 / LPS;  // lanes per survivor, chunks per lane and pass
                            const int g_mine = lane / LPS, l16 = lane % LPS;
                            for (uint32_t e0 = 0; e0 < ns; e0 += PER) {
                                const uint32_t nb = ns - e0 < (uint32_t)PER ? ns - e0 : (uint32_t)PER;
                                const uint32_t em = e0 + ((uint32_t)g_mine < nb ? (uint32_t)g_mine : 0u);
                                const uint32_t hi_m = s.surv[wu][em][0], row_m = s.surv[wu][em][1], lc_m = s.surv[wu][em][2];
                                const uint32_t rr = row_m & 127u;  // row inside its tile: 4 i_ + rt
                                const uint4* pa = reinterpret_cast<const uint4*>(Ai8) + ((size_t)(row_m >> 7) * nk * 4) * 128 + (rr & 3u) * 32 + (rr >> 2);
                                const uint4* pb = reinterpret_cast<const uint4*>(Bq) + Qpad + (q0 + 64 * wu + 32 * ((lc_m >> 8) & 1u) + (lc_m & 31u));
                                int32_t part = 0;
                                // (One pass of 64 chunks for 769 .. 1024 dimensions -- squared L2 at C2 is 842 -- beside this loop saved that
                                //  shape its second round trip, 13.05 -> 12.49 ms, and cost the dot shape 8.19 -> 8.71: not kept.)
                                for (uint32_t c0 = 0; c0 < nk * 4; c0 += 48) {  // 48 chunks (768 dimensions) per pass: three per lane
                                    uint4 x[CPL], y[CPL];
#pragma unroll
                                    for (int t = 0; t < CPL; ++t) {
                                        const uint32_t c = c0 + (uint32_t)l16 + (uint32_t)LPS * (uint32_t)t;
                                        const bool on = (uint32_t)g_mine < nb && c < nk * 4;
                                        x[t] = on ? pa[(size_t)c * 128] : uint4{0u, 0u, 0u, 0u};
                                        y[t] = on ? pb[(size_t)c * 2 * Qpad] : uint4{0u, 0u, 0u, 0u};
                                    }
#pragma unroll
                                    for (int t = 0; t < CPL; ++t) {
                                        part = __builtin_amdgcn_sdot4((int)x[t].x, (int)y[t].x, part, false);
                                        part = __builtin_amdgcn_sdot4((int)x[t].y, (int)y[t].y, part, false);
                                        part = __builtin_amdgcn_sdot4((int)x[t].z, (int)y[t].z, part, false);
                                        part = __builtin_amdgcn_sdot4((int)x[t].w, (int)y[t].w, part, false);
                                    }
                                }
#pragma unroll
                                for (int off = LPS / 2; off >= 1; off >>= 1) part += __shfl_xor(part, off, 64);
                                for (uint32_t g = 0; g < nb; ++g) {
                                    const int32_t lo = __builtin_amdgcn_readlane(part, LPS * (int)g);
                                    const int32_t hiL = (int32_t)__builtin_amdgcn_readlane((int)hi_m, LPS * (int)g);
                                    const uint32_t i = (uint32_t)__builtin_amdgcn_readlane((int)row_m, LPS * (int)g);
                                    const uint32_t lc = (uint32_t)__builtin_amdgcn_readlane((int)lc_m, LPS * (int)g);
                                    const int L = (int)(lc & 63u);
                                    const bool c1 = ((lc >> 8) & 1u) != 0;  // query column tile (wave-uniform)
                                    const int32_t V = (int32_t)(((uint32_t)hiL << S) + (uint32_t)lo);
                                    if (lane == L && V >= (c1 ? Tint[1] : Tint[0])) {
                                        const uint32_t o = f32_ord(__builtin_fmaf(c1 ? Aj[1] : Aj[0], (float)V, c1 ? Bj[1] : Bj[0]));
                                        if (o >= (c1 ? thr[1] : thr[0]) && i < N) {
                                            const int ql = 64 * w + 32 * (c1 ? 1 : 0) + C;
                                            if (probe) ++pc_napp;
                                            if (MODE == 2) {  // collect: the query's global list
                                                const uint32_t pos = atomicAdd(counts + q0 + ql, 1u);
                                                if (pos < KP) reinterpret_cast<uint32_t*>(lists)[(q0 + ql) * (size_t)KP + pos] = i;
                                            } else {
                                                const bool pub = (i & (kI8hPubEvery - 1)) == 0;
                                                admitted[0] = admitted[0] || (pub && !c1);
                                                admitted[1] = admitted[1] || (pub && c1);
                                                cand_append(my_lists + (size_t)ql * cap, &s.cnt[ql], cap, cand_make(o, i), errflag);
                                                gthr_raise(gslots + (q0 + ql) * (size_t)(kSlotMul * KP), kSlotMul * KP, o, i);
                                            }
                                        }
                                    }
                                }
                            }
                            ns = 0;
                            __builtin_amdgcn_wave_barrier();
                            if (probe) pc_surv += __builtin_readcyclecounter() - ps0;
                        };
                        // A visit borrows the K-loop's operand ring: once everything the wave has in flight has landed (the wait
                        // a survivor's first load would sit out anyway), the ring's 32 registers hold nothing a visit needs -- its
                        // fragments are requested again at the end -- and give the survivors' loads room to go out four
                        // survivors at a time. (With registers of their own those loads pushed the ring into scratch, and
                        // tools/check_gemm_asm.py refused the build.)
                        auto ring_release = [&]() {
#pragma unroll
                            for (int r = 0; r < kI8Lead; ++r) {
                                use_after<0>(breg[r][0], breg[r][1]);
                                use_after<0>(breg[r][2], breg[r][3]);
                            }
                            asm volatile("; innr operand ring released" ::: "memory");
                        };
                        // finish what is queued, re-derive the chip-wide bounds that asked for it, compact lists that run short of room
                        auto finish = [&]() {
                            flush();
                            if (MODE == 2) return;  // collect: no bound moves, no list to compact
                            unsigned long long admitted_by[2] = {__ballot(admitted[0]), __ballot(admitted[1])};
                            pc_npub += probe ? (uint32_t)(__popcll(admitted_by[0]) + __popcll(admitted_by[1])) : 0u;
                            const unsigned long long pt1 = probe ? __builtin_readcyclecounter() : 0ull;
#pragma unroll
                            for (int ct = 0; ct < 2; ++ct) {
                                unsigned long long m = admitted_by[ct];
                                m = (m | (m >> 32)) & 0xffffffffull;
                                while (m) {
                                    const int L = __builtin_ctzll(m);
                                    m &= m - 1;
                                    const size_t qg = q0 + 64 * wu + 32 * ct + L;  // wave-uniform
                                    gthr_publish_select<(kSlotMul * (16 * R - 64) + 63) / 64>(gslots + qg * (size_t)(kSlotMul * KP), gthr + qg, KP, lane, kk, kmargin[qg]);
                                }
                            }
                            __builtin_amdgcn_wave_barrier();
                            const uint32_t c = __hip_atomic_load(&s.cnt[64 * w + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                            unsigned long long need = __ballot(c > cap - kGemmBurst);
                            while (need) {
                                const int j = __builtin_ctzll(need);
                                need &= need - 1;
                                const int qj = 64 * w + j;
                                const uint32_t cj = __builtin_amdgcn_readlane(c, j);
                                uint32_t t;
                                const uint32_t keep = wave_compact<R>(my_lists + (size_t)qj * cap, cj, KP, &t);
                                if (lane == 0) {
                                    s.cnt[qj] = keep;
                                    s.thr[qj] = t;
                                    if (t > __hip_atomic_load(&gthr[q0 + qj], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                                        __hip_atomic_fetch_max(&gthr[q0 + qj], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                }
                            }
                            __builtin_amdgcn_wave_barrier();
                            if (probe) pc_tail += __builtin_readcyclecounter() - pt1;
                        };
                        // give the ring back: the fragments of the next two K-steps again (ring slot r holds K-step b_ks - 2 + r),
                        // and nothing may be in flight when the K-loop's counted waits resume
                        auto ring_reclaim = [&]() {
                            asm volatile("; innr operand ring reclaimed" ::: "memory");
#pragma unroll
                            for (int r = 0; r < kI8Lead; ++r) {
                                const uint32_t kidx = (b_ks + 2u * nk - (uint32_t)kI8Lead + (uint32_t)r) % nk;
                                issue_b_at(breg[r][0], breg[r][1], 0, kidx);
                                issue_b_at(breg[r][2], breg[r][3], 1, kidx);
                            }
                            wait_all();
#pragma unroll
                            for (int r = 0; r < kI8Lead; ++r) {
                                use_after<0>(breg[r][0], breg[r][1]);
                                use_after<0>(breg[r][2], breg[r][3]);
                            }
                        };
                        // How many survivors does this tile bring? Usually a handful per wave; a launch without seeded bounds (a small
                        // corpus) passes every site of its first tiles. A tile that fits is queued as it stands; one that does not
                        // is walked in ROUNDS with the list emptied in between (round r takes each lane's r-th survivor of a group
                        // of 16 sites: at most one entry per lane and round) -- which needs the visit's registers, so the visit
                        // starts before the walk.
                        bool rounds = false;
                        const unsigned long long pq0 = probe ? __builtin_readcyclecounter() : 0ull;
                        if (any_hit) {
                            pc_nhit += probe ? 1u : 0u;
                            uint32_t mine = 0;
#pragma unroll
                            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                                for (int gq = 0; gq < 4; ++gq) {
                                    const bool ghit = hit[ct] && gbest[ct][gq] >= Thi[ct];
                                    if (!__any(ghit)) continue;
#pragma unroll
                                    for (int g3 = 0; g3 < 4; ++g3)
#pragma unroll
                                        for (int rt = 0; rt < 4; ++rt) mine += (ghit && acc[rt][ct][4 * gq + g3] >= Thi[ct]) ? 1u : 0u;
                                }
#pragma unroll
                            for (int off = 32; off >= 1; off >>= 1) mine += (uint32_t)__shfl_xor((int)mine, off, 64);
                            const uint32_t total_new = (uint32_t)__builtin_amdgcn_readfirstlane((int)mine);
                            pc_nsurv += probe ? total_new : 0u;
                            rounds = ns + total_new > (uint32_t)kI8hSurvCap;
                            if (!rounds) {
#pragma unroll
                                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                                    for (int gq = 0; gq < 4; ++gq) {
                                        const bool ghit = hit[ct] && gbest[ct][gq] >= Thi[ct];
                                        if (!__any(ghit)) continue;  // wave-uniform: most tiles touch one group of one query column
#pragma unroll
                                        for (int g3 = 0; g3 < 4; ++g3)
#pragma unroll
                                            for (int rt = 0; rt < 4; ++rt) {
                                                const int32_t hi = acc[rt][ct][4 * gq + g3];
                                                const bool surv = ghit && hi >= Thi[ct];
                                                const unsigned long long mm = __ballot(surv);
                                                if (!mm) continue;
                                                if (surv) {  // the lanes of a site write side by side: slot = length + rank among its survivors
                                                    const uint32_t slot = ns + __builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm, 0u));
                                                    s.surv[wu][slot][0] = (uint32_t)hi;
                                                    s.surv[wu][slot][1] = (uint32_t)tb + 4u * ((uint32_t)g3 + 8u * (uint32_t)gq + 4u * (uint32_t)half) + (uint32_t)rt;
                                                    s.surv[wu][slot][2] = (uint32_t)lane | ((uint32_t)ct << 8);
                                                }
                                                ns += (uint32_t)__popcll(mm);
                                            }
                                    }
                            }
                        }
                        if (probe) pc_queue += __builtin_readcyclecounter() - pq0;
                        if (rounds || sync_flush || ns >= (uint32_t)kI8hFlushAt) {
                            const unsigned long long pt0 = probe ? __builtin_readcyclecounter() : 0ull;
                            pc_nvis += probe ? 1u : 0u;
                            ring_release();
                            if (rounds) {
#pragma unroll
                                for (int ct = 0; ct < 2; ++ct) {
#pragma unroll
                                    for (int gq = 0; gq < 4; ++gq) {
                                        const bool ghit = hit[ct] && gbest[ct][gq] >= Thi[ct];
                                        if (!__any(ghit)) continue;
                                        int taken = 0;
                                        while (true) {
                                            int32_t sel_hi = 0;
                                            int sel_site = -1, seen = 0;
#pragma unroll
                                            for (int g3 = 0; g3 < 4; ++g3)
#pragma unroll
                                                for (int rt = 0; rt < 4; ++rt) {
                                                    const int32_t hi = acc[rt][ct][4 * gq + g3];
                                                    const bool surv = ghit && hi >= Thi[ct];
                                                    const bool pick = surv && seen == taken;
                                                    sel_hi = pick ? hi : sel_hi;
                                                    sel_site = pick ? (4 * g3 + rt) : sel_site;
                                                    seen += surv ? 1 : 0;
                                                }
                                            const bool mine = sel_site >= 0;
                                            const unsigned long long mm = __ballot(mine);
                                            if (!mm) break;
                                            taken += mine ? 1 : 0;
                                            const uint32_t nm = (uint32_t)__popcll(mm);
                                            if (ns + nm > (uint32_t)kI8hSurvCap) flush();
                                            if (mine) {
                                                const uint32_t slot = ns + (uint32_t)__popcll(mm & ((1ull << lane) - 1ull));
                                                const uint32_t site = (uint32_t)sel_site;
                                                s.surv[wu][slot][0] = (uint32_t)sel_hi;
                                                s.surv[wu][slot][1] = (uint32_t)tb + 4u * ((site >> 2) + 8u * (uint32_t)gq + 4u * (uint32_t)half) + (site & 3u);
                                                s.surv[wu][slot][2] = (uint32_t)lane | ((uint32_t)ct << 8);
                                            }
                                            ns += nm;
                                        }
                                    }
                                }
                            }
                            finish();
                            ring_reclaim();
                            if (probe) pc_visit += __builtin_readcyclecounter() - pt0;
                        }
                    }
                }
                if (MODE != 1) {
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) gload1_agent(tg_next[ct], gthr + q0 + 64 * wu + 32 * ct, 4u * (uint32_t)C);
                }
                ks = 0;
                ++tile;
            } else {
                ++ks;
            }
            if (r == kI8Lead - 1) {  // one barrier per two K-steps: see gemm_bf16_filter_kernel
                wait_but_youngest<5 * (kI8Stages - 4) + 4>();
                if (!(kI8hProbe & 64)) __syncthreads();  // (probe bit 64, timing only: no barrier in the K-loop)
            }
        }
    }
    wait_all();
    __syncthreads();
    if (kI8hProbe & 4)
        for (int off = 32; off >= 1; off >>= 1) pc_napp += (uint32_t)__shfl_xor((int)pc_napp, off, 64);  // (counted by the lane that appends)
    if (MODE == 0 && (kI8hProbe & (4 | 1)) && lane == 0) {
        atomicAdd(errflag + 8, pc_nvis);
        atomicAdd(errflag + 9, pc_nsurv);
        atomicAdd(errflag + 10, pc_npub);
        atomicAdd(errflag + 11, pc_napp);
        atomicAdd(errflag + 18, pc_nhit);
        atomicAdd(reinterpret_cast<unsigned long long*>(errflag + 20), pc_queue);
        if (kI8hProbe & 4) {  // the longest and the shortest wave of the launch (cycles / 16): what the slices' imbalance costs
            const uint32_t dur = (uint32_t)((__builtin_readcyclecounter() - pc_t0) >> 4);
            atomicMax(errflag + 22, dur);
            atomicMax(errflag + 23, ~dur);
            if (w == 0 && blockIdx.x < 512) errflag[128 + blockIdx.x] = dur;  // per block (the flags buffer holds 1024 words)
        }
        atomicAdd(reinterpret_cast<unsigned long long*>(errflag + 12), pc_visit);
        atomicAdd(reinterpret_cast<unsigned long long*>(errflag + 14), pc_surv);
        atomicAdd(reinterpret_cast<unsigned long long*>(errflag + 16), pc_tail);
    }
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

// ---- gemm_i8s_filter_kernel: the one-limb filter for a SMALL query batch (<= 64 queries) ----------------------------------------
// gemm_i8h_filter_kernel's block tile is 128 corpus rows x 512 queries: a batch below 512 queries pays the matrix work of 512
// (2.6 ms at ONE query at C2 -- 59 % of the int8 pipe on padding -- where streaming the 7.7 GB copy takes 1.3 ms). Here the roles
// are swapped: the queries' high limbs (nk x 4 KB: every B fragment of the launch) sit in LDS for the whole kernel, every WAVE
// streams corpus rows of its own -- QUARTER tiles of 32 rows (row tile q & 3 of tile q >> 2) -- and nothing couples the waves: no
// LDS stage, no barrier after the prologue, a wave that finishes its survivors stalls nobody. The K-step count is a template
// parameter and the loop over it fully unrolled, every load plain C (as in maxsim_mfma_tile_kernel): a quarter tile's NK x 2
// fragments have registers of their own, fragment ks of the NEXT quarter tile is requested as soon as fragment ks of this one has
// been multiplied, and hipcc's own s_waitcnt vmcnt(N) is then the exact number of younger requests -- a whole quarter tile (NK KB)
// in flight per wave. (A first version with a four-stage ring, a run-time K loop and hand-issued asm loads was compiled into
// copies of in-flight registers at every phi; with plain loads in that loop the back-edge made every wait a vmcnt(0).)
// HBM-bound by design: 4 MFMAs of 32 cycles per 2 KB of fragments. Each wave is a slice of its own (lists, counts:
// [nblocks * 8][64]); thresholds, slots, the k rule and the list discipline are gemm_i8h_filter_kernel's. Needs seeded bounds
// (api.hip picks it only then): an unseeded first tile passes all its sites through the survivors' path, four per round trip.
// CT = 4 (65 .. 128 queries: four column tiles per wave, 64 accumulator registers): a quarter tile's fragments are requested in two
// K halves (unit = quarter tile x K half; the accumulators carry over), so that the fragment registers stay at NK x 4.
constexpr int kI8sBQ = 64, kI8sWaves = 8, kI8sSurvCap = 128;
template <int CT>
struct alignas(16) GemmI8sLds {
    uint32_t cnt[kI8sWaves][32 * CT];
    uint32_t thr[kI8sWaves][32 * CT];
    uint32_t surv[kI8sWaves][kI8sSurvCap][4];  // high limb, corpus row, lane | query column tile << 8
};
__host__ __device__ inline size_t i8s_dyn_lds_bytes(uint32_t nk, uint32_t ct) { return (size_t)nk * 2048 * ct; }
template <int CT, typename T> __device__ __forceinline__ T i8s_pick(const T (&v)[CT], uint32_t ct) {  // v[ct] without a register-indexed array
    T r = v[0];
#pragma unroll
    for (int x = 1; x < CT; ++x) r = ct == (uint32_t)x ? v[x] : r;
    return r;
}

// MODE 0: fused top-k filter; MODE 2: collect (fixed thresholds in gthr, global uint32 lists [Qpad][KP], lengths in counts) -- as in
// gemm_i8h_filter_kernel.
template <int R, int NK, int CT, int MODE>
__global__ __launch_bounds__(64 * kI8sWaves, 1) void gemm_i8s_filter_kernel(
    const char* __restrict__ Ai8, const char* __restrict__ Bq, uint32_t nquarter /*quarter tiles: 4 * ntiles*/, uint32_t N,
    size_t Qpad /*= 32 CT nqt*/, uint32_t nqt /*query groups of 32 CT*/, uint32_t quarters_per_wave, const float* __restrict__ qc,
    uint64_t* __restrict__ lists, uint32_t* __restrict__ counts, uint32_t KP, uint32_t kk, uint32_t* __restrict__ errflag,
    uint32_t* gslots, uint32_t* gthr, uint32_t* progress /*[wave slices][nqt], zeroed; null: the groups run free*/) {
    static_assert(CT == 2 || CT == 4, "two or four column tiles of 32 queries");
    static_assert(MODE == 0 || MODE == 2, "filter or collect");
    constexpr int KS = CT == 4 ? 2 : 1, NKH = NK / KS;  // K halves per quarter tile, K-steps per unit
    static_assert(NK % KS == 0, "the K-step count splits evenly");
    extern __shared__ __attribute__((aligned(16))) char i8s_b[];  // [NK][m 2][ct CT][64 lanes] x 16 B: the high limbs as B fragments
    __shared__ GemmI8sLds<CT> s;
    constexpr int S = kI8hS;
    constexpr uint32_t cap = 64 * R, nk = NK;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wu = __builtin_amdgcn_readfirstlane(w);
    const int half = lane >> 5, C = lane & 31;
    // Several query groups (nqt > 1: more than 32 CT queries): the blocks of ONE corpus slice -- one per group -- sit on the same XCD
    // (block b runs on XCD b % 8) next to each other in launch order, so that the slice is fetched from HBM once and served to the
    // other groups by that XCD's L2.
    const uint32_t xcd = blockIdx.x & 7u, lb = blockIdx.x >> 3, grp = lb % nqt, slice_blk = xcd * (gridDim.x / (8u * nqt)) + lb / nqt;
    const size_t q0 = (size_t)grp * (32 * CT);
    Bq += q0 * 16;          // (every [..][Qpad][16 B] plane of the packed queries)
    qc += q0;
    gslots += q0 * (size_t)(kSlotMul * KP);
    counts += q0;
    const float* const kmargin = reinterpret_cast<const float*>(gthr + Qpad) + q0;  // (behind the bounds of ALL groups)
    gthr += q0;
    for (uint32_t idx = threadIdx.x; idx < nk * 2 * CT * 64; idx += 64 * kI8sWaves) {
        const uint32_t l = idx & 63, x = idx >> 6, ct = x % CT, m = (x / CT) & 1, ks = x / (2 * CT);
        reinterpret_cast<uint4*>(i8s_b)[idx] =
            reinterpret_cast<const uint4*>(Bq)[((size_t)(ks * 4 + 2 * m + (l >> 5)) * 2) * Qpad + ct * 32 + (l & 31)];
    }
    for (int x = lane; x < 32 * CT; x += 64) {
        s.cnt[w][x] = 0;
        s.thr[w][x] = 0;
    }
    __syncthreads();  // (the only barrier)

    const uint32_t slice = slice_blk * kI8sWaves + (uint32_t)wu;
    uint32_t qt0 = slice * quarters_per_wave, qt1 = qt0 + quarters_per_wave;
    if (qt1 > nquarter) qt1 = nquarter;
    if (qt0 > qt1) qt0 = qt1;
    uint64_t* my_lists = lists + ((size_t)slice * Qpad + q0) * cap;
    float Aj[CT], Bj[CT], invAj[CT];
    int32_t lob[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const size_t q = 32 * ct + C;
        Aj[ct] = qc[q];
        Bj[ct] = qc[Qpad + q];
        invAj[ct] = qc[2 * Qpad + q];
        lob[ct] = __float_as_int(qc[4 * Qpad + q]);
    }
    // fragment m of quarter tile q at K-step ks: 16 B of row C of row tile q & 3, k-group 2 m + half
    const i32x4_t* const frag0 = reinterpret_cast<const i32x4_t*>(Ai8) + (size_t)half * 128 + C;
    auto frag = [&](uint32_t q, uint32_t ks, int m) -> i32x4_t {
        return frag0[(((size_t)(q >> 2) * nk + ks) * 4 + 2u * (uint32_t)m) * 128 + (q & 3u) * 32u];
    };
    const uint32_t u0 = qt0 * KS, u1 = qt1 * KS;  // units: (quarter tile, K half)
    i32x4_t A[NKH][2];
    if (u0 < u1) {
#pragma unroll
        for (int i = 0; i < NKH; ++i) {
            A[i][0] = frag(qt0, (uint32_t)i, 0);
            A[i][1] = frag(qt0, (uint32_t)i, 1);
        }
    }
    uint32_t ns = 0;
    uint32_t tg_cur[CT];
    i32x16_t acc[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        tg_cur[ct] = 0u;
#pragma unroll
        for (int g = 0; g < 16; ++g) acc[ct][g] = 0;
    }
    // Soft lockstep of the groups (nqt > 1): wave w of every group walks the SAME quarter tiles; if they pass a tile within a few tiles
    // of each other the first one fetches it from HBM and the others read it from the XCD's L2 (4 MB: 4 slices x 8 waves x the
    // window). Left alone they drift -- unequal groups at once (200 queries: 3.4 ms against 2.6 for 256), equal ones beyond two
    // groups (every further group cost the copy at the streaming rate again). Every kI8sSyncEvery quarter tiles a wave publishes its
    // position and waits until no group is more than kI8sSyncWindow tiles behind: a BOUNDED number of polls -- a group that never
    // shows up (its block not resident) switches the waiting off for good instead of hanging the launch.
    constexpr uint32_t kI8sSyncEvery = 2, kI8sSyncWindow = 4, kI8sSyncPolls = 4096;
    uint32_t* const prog = progress ? progress + (size_t)slice * nqt : nullptr;
    bool sync_on = prog != nullptr && nqt > 1;
    for (uint32_t u = u0; u < u1; ++u) {
        const uint32_t h = u / KS, kh = u % KS;  // (h: the quarter tile)
        if (kh == 0 && sync_on && ((h - qt0) % kI8sSyncEvery) == 0) {
            const uint32_t r = h - qt0;
            if (lane == 0) __hip_atomic_store(prog + grp, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (r > kI8sSyncWindow) {
                uint32_t polls = 0;
                while (true) {
                    const uint32_t v = (uint32_t)lane < nqt ? __hip_atomic_load(prog + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xFFFFFFFFu;
                    uint32_t mn = v;
#pragma unroll
                    for (int off = 32; off >= 1; off >>= 1) {
                        const uint32_t o = (uint32_t)__shfl_xor((int)mn, off, 64);
                        mn = o < mn ? o : mn;
                    }
                    if (mn + kI8sSyncWindow >= r) break;
                    if (++polls >= kI8sSyncPolls) {
                        sync_on = false;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(8);
                }
            }
        }
        if (kh == 0) {
            // the chip-wide bounds of this quarter tile, requested before its successor's fragments: the oldest request in flight
            // when the epilogue needs it
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                tg_cur[ct] = __hip_atomic_load(gthr + 32 * ct + C, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int g = 0; g < 16; ++g) acc[ct][g] = 0;
            }
        }
        const uint32_t un = u + 1 < u1 ? u + 1 : u;  // (the last unit requests its own fragments again: no branch in the K loop)
        const uint32_t qn = un / KS, ksn = (un % KS) * NKH, ks0 = kh * NKH;
#pragma unroll
        for (int i = 0; i < NKH; ++i) {
            const i32x4_t* bl = reinterpret_cast<const i32x4_t*>(i8s_b) + (size_t)(ks0 + (uint32_t)i) * (2 * CT * 64) + lane;  // [m][ct][64]
#pragma unroll
            for (int cp = 0; cp < CT; cp += 2) {  // two column tiles at a time: four operand registers each
                const i32x4_t b00 = bl[(0 * CT + cp) * 64], b01 = bl[(0 * CT + cp + 1) * 64];
                const i32x4_t b10 = bl[(1 * CT + cp) * 64], b11 = bl[(1 * CT + cp + 1) * 64];
                acc[cp] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[i][0], b00, acc[cp], 0, 0, 0);
                acc[cp + 1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[i][0], b01, acc[cp + 1], 0, 0, 0);
                acc[cp] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[i][1], b10, acc[cp], 0, 0, 0);
                acc[cp + 1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[i][1], b11, acc[cp + 1], 0, 0, 0);
            }
            A[i][0] = frag(qn, ksn + (uint32_t)i, 0);
            A[i][1] = frag(qn, ksn + (uint32_t)i, 1);
        }
        if (kh != KS - 1) continue;
        // ---------------- epilogue of quarter tile h: CT queries per lane, 16 corpus rows each ----------------
        const uint32_t tb = (h >> 2) * 128u, rt0 = h & 3u;
        uint32_t thr[CT];
        int32_t Tint[CT], Thi[CT];
        bool hit[CT], any = false;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const uint32_t tl = MODE == 2 ? 0u : __hip_atomic_load(&s.thr[wu][32 * ct + C], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            thr[ct] = tl > tg_cur[ct] ? tl : tg_cur[ct];
            Tint[ct] = INT32_MIN;
            Thi[ct] = INT32_MIN;
            if (thr[ct] != 0u) {  // (as in gemm_i8h_filter_kernel)
                const float x = (ord_f32(thr[ct]) - Bj[ct]) * invAj[ct];
                if (x >= 2147483520.0f || thr[ct] == 0xFFFFFFFFu) {
                    Tint[ct] = INT32_MAX;
                    Thi[ct] = INT32_MAX;
                } else if (x > -2.0e9f) {
                    Tint[ct] = (int32_t)__builtin_floorf(x) - 2 - (int32_t)(fabsf(x) * 4.8e-7f);
                    const long long d = (long long)Tint[ct] - (long long)lob[ct];
                    Thi[ct] = d < -2147483000ll ? INT32_MIN : (int32_t)(d >> S);
                }
            }
            int32_t m4 = INT32_MIN;
#pragma unroll
            for (int g = 0; g < 16; ++g) m4 = m4 > acc[ct][g] ? m4 : acc[ct][g];
            hit[ct] = m4 >= Thi[ct];
            any = any || hit[ct];
        }
        if (!__any(any)) continue;
        bool admitted[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) admitted[ct] = false;
        // finish the queued survivors, four per memory round trip (gemm_i8h_filter_kernel's flush: 16 lanes per survivor)
        auto flush = [&]() {
            __builtin_amdgcn_wave_barrier();
            constexpr int PER = 4, LPS = 64 / PER, CPL = 48 / LPS;
            const int g_mine = lane / LPS, l16 = lane % LPS;
            for (uint32_t e0 = 0; e0 < ns; e0 += PER) {
                const uint32_t nb = ns - e0 < (uint32_t)PER ? ns - e0 : (uint32_t)PER;
                const uint32_t em = e0 + ((uint32_t)g_mine < nb ? (uint32_t)g_mine : 0u);
                const uint32_t hi_m = s.surv[wu][em][0], row_m = s.surv[wu][em][1], lc_m = s.surv[wu][em][2];
                const uint32_t rr = row_m & 127u;
                const uint4* pa = reinterpret_cast<const uint4*>(Ai8) + ((size_t)(row_m >> 7) * nk * 4) * 128 + (rr & 3u) * 32 + (rr >> 2);
                const uint4* pb = reinterpret_cast<const uint4*>(Bq) + Qpad + (32 * ((lc_m >> 8) & 3u) + (lc_m & 31u));
                int32_t part = 0;
                for (uint32_t c0 = 0; c0 < nk * 4; c0 += 48) {
                    uint4 x[CPL], y[CPL];
#pragma unroll
                    for (int t = 0; t < CPL; ++t) {
                        const uint32_t c = c0 + (uint32_t)l16 + (uint32_t)LPS * (uint32_t)t;
                        const bool on = (uint32_t)g_mine < nb && c < nk * 4;
                        x[t] = on ? pa[(size_t)c * 128] : uint4{0u, 0u, 0u, 0u};
                        y[t] = on ? pb[(size_t)c * 2 * Qpad] : uint4{0u, 0u, 0u, 0u};
                    }
#pragma unroll
                    for (int t = 0; t < CPL; ++t) {
                        part = __builtin_amdgcn_sdot4((int)x[t].x, (int)y[t].x, part, false);
                        part = __builtin_amdgcn_sdot4((int)x[t].y, (int)y[t].y, part, false);
                        part = __builtin_amdgcn_sdot4((int)x[t].z, (int)y[t].z, part, false);
                        part = __builtin_amdgcn_sdot4((int)x[t].w, (int)y[t].w, part, false);
                    }
                }
#pragma unroll
                for (int off = LPS / 2; off >= 1; off >>= 1) part += __shfl_xor(part, off, 64);
                for (uint32_t g = 0; g < nb; ++g) {
                    const int32_t lo = __builtin_amdgcn_readlane(part, LPS * (int)g);
                    const int32_t hiL = (int32_t)__builtin_amdgcn_readlane((int)hi_m, LPS * (int)g);
                    const uint32_t i = (uint32_t)__builtin_amdgcn_readlane((int)row_m, LPS * (int)g);
                    const uint32_t lc = (uint32_t)__builtin_amdgcn_readlane((int)lc_m, LPS * (int)g);
                    const int L = (int)(lc & 63u);
                    const uint32_t c1 = (lc >> 8) & 3u;  // query column tile (wave-uniform)
                    const int32_t V = (int32_t)(((uint32_t)hiL << S) + (uint32_t)lo);
                    if (lane == L && V >= i8s_pick<CT>(Tint, c1)) {
                        const uint32_t o = f32_ord(__builtin_fmaf(i8s_pick<CT>(Aj, c1), (float)V, i8s_pick<CT>(Bj, c1)));
                        if (o >= i8s_pick<CT>(thr, c1) && i < N) {
                            const int ql = 32 * (int)c1 + C;
                            if (MODE == 2) {  // collect: the query's global list
                                const uint32_t pos = atomicAdd(counts + ql, 1u);
                                if (pos < KP) reinterpret_cast<uint32_t*>(lists)[(q0 + (size_t)ql) * KP + pos] = i;
                            } else {
                                const bool pub = (i & (kI8hPubEvery - 1)) == 0;
#pragma unroll
                                for (int ct = 0; ct < CT; ++ct) admitted[ct] = admitted[ct] || (pub && c1 == (uint32_t)ct);
                                cand_append(my_lists + (size_t)ql * cap, &s.cnt[wu][ql], cap, cand_make(o, i), errflag);
                                gthr_raise(gslots + (size_t)ql * (kSlotMul * KP), kSlotMul * KP, o, i);
                            }
                        }
                    }
                }
            }
            ns = 0;
            __builtin_amdgcn_wave_barrier();
        };
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            if (!__any(hit[ct])) continue;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int32_t hi = acc[ct][g];
                const bool surv = hit[ct] && hi >= Thi[ct];
                const unsigned long long mm = __ballot(surv);
                if (!mm) continue;
                const uint32_t nm = (uint32_t)__popcll(mm);
                if (ns + nm > (uint32_t)kI8sSurvCap) flush();
                if (surv) {
                    const uint32_t slot = ns + __builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm, 0u));
                    s.surv[wu][slot][0] = (uint32_t)hi;
                    s.surv[wu][slot][1] = tb + 4u * ((uint32_t)(g & 3) + 8u * (uint32_t)(g >> 2) + 4u * (uint32_t)half) + rt0;
                    s.surv[wu][slot][2] = (uint32_t)lane | ((uint32_t)ct << 8);
                }
                ns += nm;
            }
        }
        flush();
        if (MODE == 2) continue;  // collect: no bound moves, no list to compact
        // re-derive the chip-wide bounds that asked for it; compact lists that run short of room
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            unsigned long long m = __ballot(admitted[ct]);
            m = (m | (m >> 32)) & 0xffffffffull;
            while (m) {
                const int L = __builtin_ctzll(m);
                m &= m - 1;
                const size_t qg = 32 * ct + L;  // wave-uniform
                gthr_publish_select<(kSlotMul * (16 * R - 64) + 63) / 64>(gslots + qg * (size_t)(kSlotMul * KP), gthr + qg, KP, lane, kk, kmargin[qg]);
            }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int qb = 0; qb < 32 * CT; qb += 64) {
            const uint32_t c = __hip_atomic_load(&s.cnt[wu][qb + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            unsigned long long need = __ballot(c > cap - 32u);  // (a quarter tile appends at most 32 rows to one query's list)
            while (need) {
                const int j = __builtin_ctzll(need);
                need &= need - 1;
                const uint32_t cj = __builtin_amdgcn_readlane(c, j);
                uint32_t t;
                const uint32_t keep = wave_compact<R>(my_lists + (size_t)(qb + j) * cap, cj, KP, &t);
                if (lane == 0) {
                    s.cnt[wu][qb + j] = keep;
                    s.thr[wu][qb + j] = t;
                    if (t > __hip_atomic_load(&gthr[qb + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                        __hip_atomic_fetch_max(&gthr[qb + j], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (prog && nqt > 1 && lane == 0) __hip_atomic_store(prog + grp, 0xFFFFFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (done: nobody waits for this wave)
    if (MODE == 2) return;
#pragma unroll
    for (int qb = 0; qb < 32 * CT; qb += 64) {
        const uint32_t c = __hip_atomic_load(&s.cnt[wu][qb + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        unsigned long long need = __ballot(c > KP);
        uint32_t mine = c;
        while (need) {
            const int j = __builtin_ctzll(need);
            need &= need - 1;
            const uint32_t cj = __builtin_amdgcn_readlane(c, j);
            uint32_t t;
            const uint32_t keep = wave_compact<R>(my_lists + (size_t)(qb + j) * cap, cj, KP, &t);
            if (lane == j) mine = keep;
        }
        counts[(size_t)slice * Qpad + qb + lane] = mine;
    }
}

}  // namespace innr
