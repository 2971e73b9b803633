// kernels_u8.h -- scalar-quantised corpus: asymmetric f32-query x u8-code scan (src/scalar.rs).
//
// Reference per document (scalar.rs:284-300, portable inner loop :353-358):
//   mixed = sum_d q[d] * (c[d] as f32)          sequential, folded from -0.0, fl(acc + fl(q*c))
//   score = (alpha / 255.0) * mixed + offset * sum(q)
// and batch_knn_u8 (scalar.rs:370-393) = that for every document + stable sort descending + truncate(k).
//
// Device layout: the reference keeps N separately allocated Vec<u8> (QuantizedU8, scalar.rs:171-174); here the
// codes are one PDX array C[d*ldN + i] (u8, ldN = N rounded up to 1024), so that lane i's byte of dimension d is
// adjacent to lane i+1's, exactly like the f32 batch.
// Roofline: HBM, 1*N*D bytes per corpus pass (C3: 38.4 GB).
#pragma once

#include "common.h"
#include "kernels_prep.h"
#include "topk_dev.h"

namespace innr {

#ifndef INNR_U8_UNROLL
#define INNR_U8_UNROLL 8
#endif
constexpr int kU8Unroll = INNR_U8_UNROLL;
#ifndef INNR_U8_DIMBARRIER
#define INNR_U8_DIMBARRIER 1
#endif
constexpr int kU8Chunk = 64 * 16;  // vectors per wave step (64 lanes x 16 codes)

// quantize_u8 (scalar.rs:212-225): clamp(round((v - offset) * (255/alpha)), 0, 255); round = half away from zero
__device__ __forceinline__ uint8_t quantize_one(float v, float offset, float inv_alpha) {
    const float r = __builtin_roundf(ex::mul(ex::sub(v, offset), inv_alpha));
    if (r != r) return 0;  // NaN `as u8` == 0
    return (uint8_t)(r < 0.0f ? 0.0f : (r > 255.0f ? 255.0f : r));
}

// codes[(i0+i)*D + d] (row-major packed, what a &[QuantizedU8] holds) -> C[d*ldN + i0 + i]
__global__ __launch_bounds__(256) void transpose_rows_u8_kernel(const uint8_t* __restrict__ rows, uint32_t nrows,
                                                                 uint32_t D, uint8_t* __restrict__ C, size_t ldN,
                                                                 size_t i0) {
    __shared__ uint8_t tile[32][33];
    const uint32_t tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const uint32_t ib = blockIdx.x * 32, db = blockIdx.y * 32;
#pragma unroll
    for (int r = 0; r < 32; r += 8) {
        const uint32_t i = ib + ty + r, d = db + tx;
        tile[ty + r][tx] = (i < nrows && d < D) ? rows[(size_t)i * D + d] : 0;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 32; r += 8) {
        const uint32_t d = db + ty + r, i = ib + tx;
        if (d < D && i < nrows) C[(size_t)d * ldN + i0 + i] = tile[tx][ty + r];
    }
}

// synthetic codes: quantize_u8(uniform row, params) generated in place (one thread = 16 vectors of one dimension)
__global__ void generate_u8_pdx_kernel(uint8_t* __restrict__ C, size_t ldN, uint32_t N, uint32_t D, uint64_t seed,
                                       uint64_t row0, float offset, float inv_alpha) {
    const size_t i16 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
    const uint32_t d = blockIdx.y;
    if (i16 >= ldN || d >= D) return;
    uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const size_t i = i16 + c;
        const uint32_t q = (i < N) ? quantize_one(uniform_embedding(seed, row0 + i, D, d), offset, inv_alpha) : 0u;
        w[c >> 2] |= q << (8 * (c & 3));
    }
    *reinterpret_cast<uint4*>(C + (size_t)d * ldN + i16) = make_uint4(w[0], w[1], w[2], w[3]);
}

// quantize a resident f32 batch: C[d*ldN + i] = quantize_u8(V[d*ldN + i]) (16 codes per thread); padding columns 0
__global__ __launch_bounds__(256) void quantize_pdx_kernel(const float* __restrict__ V, size_t ldV, uint32_t N, uint32_t D,
                                                            float offset, float inv_alpha, uint8_t* __restrict__ C,
                                                            size_t ldN) {
    const size_t i16 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
    const uint32_t d = blockIdx.y;
    if (i16 >= ldN || d >= D) return;
    uint32_t w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);  // the two layouts pad N differently (256 vs 1024)
        if (i16 + 4 * c4 < ldV) v = *reinterpret_cast<const float4*>(V + (size_t)d * ldV + i16 + 4 * c4);
        const float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const size_t i = i16 + 4 * c4 + c;
            const uint32_t q = (i < N) ? quantize_one(x[c], offset, inv_alpha) : 0u;
            w[c4] |= q << (8 * c);
        }
    }
    *reinterpret_cast<uint4*>(C + (size_t)d * ldN + i16) = make_uint4(w[0], w[1], w[2], w[3]);
}

// QuantizationParams::fit (scalar.rs:68-87): global min and max of the stored values, NaN ignored (both of the
// reference's comparisons are false for NaN). Keys: f32_ord (total order), so -0.0 < +0.0 -- the reference keeps
// whichever zero it met first, the one point where a parallel reduction cannot follow a sequential scan.
// out[0] = max over values of ~ord (i.e. min), out[1] = max of ord; both start at 0 = "nothing seen".
// rowscale (nullable): the range of V[d][i] * rowscale[i] (the normalised rows: the cosine copy of the int8 filter)
__global__ __launch_bounds__(256) void minmax_pdx_kernel(const float* __restrict__ V, size_t ldN, uint32_t N, uint32_t D,
                                                          uint32_t* __restrict__ out, const float* __restrict__ rowscale = nullptr) {
    // A bounded grid that strides over the columns: one atomic pair per BLOCK on the two result words. (One pair per wave of a
    // grid sized by the corpus was 5 M atomics on two addresses -- ~88 per microsecond and address: 57 ms for a 30.7 GB corpus
    // that streams in 6.)
    __shared__ uint32_t red[2][4];
    uint32_t kmin = 0, kmax = 0;
    for (size_t i4 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i4 < ldN; i4 += (size_t)gridDim.x * blockDim.x * 4) {
        float4 rs = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
        if (rowscale) rs = *reinterpret_cast<const float4*>(rowscale + i4);
        for (uint32_t d = blockIdx.y; d < D; d += gridDim.y) {
            const float4 v = *reinterpret_cast<const float4*>(V + (size_t)d * ldN + i4);
            const float x[4] = {v.x * rs.x, v.y * rs.y, v.z * rs.z, v.w * rs.w};
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (i4 + c < N && x[c] == x[c]) {
                    const uint32_t o = f32_ord(x[c]);
                    kmax = kmax > o ? kmax : o;
                    kmin = kmin > ~o ? kmin : ~o;
                }
        }
    }
    for (int off = 32; off >= 1; off >>= 1) {
        kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, off, 64));
        kmin = max(kmin, (uint32_t)__shfl_xor((int)kmin, off, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = kmin;
        red[1][threadIdx.x >> 6] = kmax;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        kmin = max(max(red[0][0], red[0][1]), max(red[0][2], red[0][3]));
        kmax = max(max(red[1][0], red[1][1]), max(red[1][2], red[1][3]));
        if (kmin) atomicMax(out + 0, kmin);
        if (kmax) atomicMax(out + 1, kmax);
    }
}

// One pass of the radix select behind QuantizationParams::fit_quantile (scalar.rs:104-139: sort the FINITE values by
// total_cmp, take the values at two ranks): digit histograms of the keys f32_ord(x) whose higher digits equal prefA / prefB
// (himask selects those digits; pass 0: himask = 0). hist[0][256] for the low rank, hist[1][256] for the high one.
__global__ __launch_bounds__(256) void quantile_hist_kernel(const float* __restrict__ V, size_t ldN, uint32_t N, uint32_t D,
                                                             uint32_t shift, uint32_t himask, uint32_t prefA, uint32_t prefB,
                                                             unsigned long long* __restrict__ hist) {
    __shared__ uint32_t h[2][256];
    h[0][threadIdx.x] = 0;
    h[1][threadIdx.x] = 0;
    __syncthreads();
    const size_t i4 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i4 < ldN) {
        for (uint32_t d = blockIdx.y; d < D; d += gridDim.y) {
            const float4 v = *reinterpret_cast<const float4*>(V + (size_t)d * ldN + i4);
            const float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (i4 + c < N && x[c] - x[c] == 0.0f) {  // is_finite
                    const uint32_t key = f32_ord(x[c]), dig = (key >> shift) & 0xffu;
                    if ((key & himask) == prefA) atomicAdd(&h[0][dig], 1u);
                    if ((key & himask) == prefB) atomicAdd(&h[1][dig], 1u);
                }
        }
    }
    __syncthreads();
    if (h[0][threadIdx.x]) atomicAdd(hist + threadIdx.x, (unsigned long long)h[0][threadIdx.x]);
    if (h[1][threadIdx.x]) atomicAdd(hist + 256 + threadIdx.x, (unsigned long long)h[1][threadIdx.x]);
}

// query_context (scalar.rs:236-240): sum(q) folded from -0.0, and ||q|| for the GEMM engine's error bound
__global__ void query_sums_kernel(const float* __restrict__ Qm, uint32_t Q, uint32_t D, size_t ldq,
                                  float* __restrict__ qsum, float* __restrict__ qnorm) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= Q) return;
    float s = -0.0f, n2 = -0.0f;
    for (uint32_t d = 0; d < D; ++d) {
        const float x = Qm[(size_t)j * ldq + d];
        s = ex::add(s, x);
        n2 = ex::mad2(n2, x, x);
    }
    qsum[j] = s;
    qnorm[j] = ex::sqrt(n2);
}

// out[i] = i < n_valid ? x[i] * a : 0
__global__ void scale_kernel(const float* __restrict__ x, float a, size_t n, size_t n_valid, float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (i < n_valid) ? ex::mul(a, x[i]) : 0.0f;
}

__device__ __forceinline__ float u8_score(float a255, float mixed, float offset, float qsum) {
    return ex::add(ex::mul(a255, mixed), ex::mul(offset, qsum));  // scalar.rs:299, two roundings + one add
}

// mixed[j][c] for QB queries x 16 vectors starting at column col (col % 16 == 0)
template <int QB>
__device__ __forceinline__ void scan_u8_accumulate(const uint8_t* __restrict__ C, size_t ldN, uint32_t D, size_t col,
                                                   const float* __restrict__ Qm, size_t ldq, float (&acc)[QB][16]) {
#pragma unroll
    for (int j = 0; j < QB; ++j)
#pragma unroll
        for (int c = 0; c < 16; ++c) acc[j][c] = -0.0f;  // <f32 as Sum>::sum starts at -0.0 (scalar.rs:357)
    const uint4* p = reinterpret_cast<const uint4*>(C + col);
    const size_t stride = ldN / 16;
    // kU8Unroll dimensions per iteration, ALL their loads issued at its top (the sched_barrier keeps the scheduler from sinking
    // each load next to its use, which serialises load -> wait -> multiply; seen with an 8-fold unroll) and awaited one by
    // one (the compiler counts them down within an iteration). Loads carried ACROSS the back-edge it awaits with vmcnt(0), and
    // hand-placed asm loads + waits made the allocator copy registers whose data had not arrived -- both tried. What hides
    // the first load's round trip is the other waves of the SIMD, so an iteration must be long: 8 dimensions.
#if INNR_U8_UNROLL == 0  // (tools/u8_scan_probe.hip: round 1's loop, for comparison)
    uint32_t d = 0;
#pragma unroll 4
    for (; d < D; ++d) {
        const uint4 v = p[(size_t)d * stride];
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        float f[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) f[c] = (float)((w[c >> 2] >> (8 * (c & 3))) & 0xffu);
#pragma unroll
        for (int j = 0; j < QB; ++j) {
            const float q = Qm[(size_t)j * ldq + d];
#pragma unroll
            for (int c = 0; c < 16; ++c) acc[j][c] = ex::mad2(acc[j][c], q, f[c]);
        }
    }
#else
    constexpr int U = QB <= 4 ? kU8Unroll : (kU8Unroll < 4 ? kU8Unroll : 4);
    uint32_t d = 0;
#pragma unroll 1
    for (; d + U <= D; d += U) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = p[(size_t)(d + u) * stride];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t w[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
            float f[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) f[c] = (float)((w[c >> 2] >> (8 * (c & 3))) & 0xffu);  // u8 -> f32, exact
#pragma unroll
            for (int j = 0; j < QB; ++j) {
                const float q = Qm[(size_t)j * ldq + d + u];
#pragma unroll
                for (int c = 0; c < 16; ++c) acc[j][c] = ex::mad2(acc[j][c], q, f[c]);
            }
#if INNR_U8_DIMBARRIER
            __builtin_amdgcn_sched_barrier(0);  // one dimension's 16 widened codes live at a time, not the iteration's 128
#endif
        }
    }
    for (; d < D; ++d) {
        const uint4 v = p[(size_t)d * stride];
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        float f[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) f[c] = (float)((w[c >> 2] >> (8 * (c & 3))) & 0xffu);
#pragma unroll
        for (int j = 0; j < QB; ++j) {
            const float q = Qm[(size_t)j * ldq + d];
#pragma unroll
            for (int c = 0; c < 16; ++c) acc[j][c] = ex::mad2(acc[j][c], q, f[c]);
        }
    }
#endif
}

// every document's score: out[j*ldo + i] (the map inside batch_knn_u8, scalar.rs:384-388)
template <int QB>
__global__ __launch_bounds__(256) void scan_u8_scores_kernel(const uint8_t* __restrict__ C, size_t ldN, uint32_t D,
                                                             const float* __restrict__ Qm, size_t ldq,
                                                             const float* __restrict__ qsum, float a255, float offset,
                                                             float* __restrict__ out, size_t ldo) {
    const size_t nchunks = ldN / kU8Chunk;
    const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = ((size_t)gridDim.x * 256) >> 6;
    const int lane = threadIdx.x & 63;
    for (size_t ch = wave; ch < nchunks; ch += nwaves) {
        const size_t col = ch * kU8Chunk + (size_t)lane * 16;
        float acc[QB][16];
        scan_u8_accumulate<QB>(C, ldN, D, col, Qm, ldq, acc);
#pragma unroll
        for (int j = 0; j < QB; ++j) {
            const float qs = qsum[j];
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
                float4 o;
                o.x = u8_score(a255, acc[j][4 * c4 + 0], offset, qs);
                o.y = u8_score(a255, acc[j][4 * c4 + 1], offset, qs);
                o.z = u8_score(a255, acc[j][4 * c4 + 2], offset, qs);
                o.w = u8_score(a255, acc[j][4 * c4 + 3], offset, qs);
                *reinterpret_cast<float4*>(out + (size_t)j * ldo + col + 4 * c4) = o;
            }
        }
    }
}

// Exact re-score + final ordering + margin proof for the GEMM engine on a u8 corpus (see rescore_kernel in
// kernels_gemm.h). One wave per query. err[q] bounds |approximate score - exact score| for that query.
template <int RK>
__global__ __launch_bounds__(64) void rescore_u8_kernel(const uint8_t* __restrict__ C, size_t ldN, uint32_t D,
                                                        const float* __restrict__ Qm, const float* __restrict__ qsum,
                                                        const float* __restrict__ qnorm, float a255, float offset,
                                                        const uint64_t* __restrict__ sel, const uint32_t* __restrict__ sel_cnt,
                                                        uint32_t KP, uint32_t kout, float err_scale, uint64_t index_base,
                                                        uint64_t* __restrict__ out_idx, float* __restrict__ out_score,
                                                        uint32_t* __restrict__ fallback, const float* __restrict__ eq = nullptr,
                                                        bool early = false, const uint4* __restrict__ Ai8 = nullptr, uint32_t nk = 0,
                                                        const uint32_t* __restrict__ gthr = nullptr) {
    const uint32_t q = blockIdx.x;
    const int lane = threadIdx.x;
    const uint32_t cnt = sel_cnt[q];
    const uint32_t G = gthr ? gthr[q] : 0u;  // the query's final chip-wide bound (see rescore_kernel)
    const float* qv = Qm + (size_t)q * D;
    const float qs = qsum[q];
    uint64_t e[RK];
#pragma unroll
    for (int r = 0; r < RK; ++r) e[r] = 0;
    auto exact = [&](uint32_t c) -> uint64_t {
        const uint32_t i = cand_idx(sel[(size_t)q * KP + c]);
        float acc = -0.0f;
        if (Ai8) {
            // The int8 engine's K-packed copy (kernels_gemm_i8.h: row 128 tile + 4 i_ + rt, 16 consecutive dimensions per 16-byte
            // unit, c - 128 as i8) holds the same codes ROW-WISE: 16 codes per memory transaction instead of one -- a column
            // gather of 768 single bytes is 768 sectors of 32 B per candidate (C3: 2.1 ms for 128 x 1024 candidates).
            const uint4* rowp = Ai8 + ((size_t)(i >> 7) * nk * 4) * 128 + (i & 3u) * 32 + ((i & 127u) >> 2);
            for (uint32_t ch = 0; ch * 16 < D; ++ch) {
                const uint4 v = rowp[(size_t)ch * 128];
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const uint32_t d = ch * 16 + e;
                    if (d < D) acc = ex::mad2(acc, qv[d], (float)(((w[e >> 2] >> (8 * (e & 3))) & 0xffu) ^ 0x80u));
                }
            }
        } else {
            const uint8_t* col = C + i;
#pragma unroll 8
            for (uint32_t d = 0; d < D; ++d) acc = ex::mad2(acc, qv[d], (float)col[(size_t)d * ldN]);
        }
        return cand_make(f32_ord(u8_score(a255, acc, offset, qs)), i);
    };
    // progressive rounds [0,32) [32,64) [64,128) [128,256), as in rescore_kernel (kernels_gemm.h)
    constexpr int NR = RK == 1 ? 2 : (RK == 2 ? 3 : 4);
#pragma unroll
    for (int round = 0; round < NR; ++round) {
        const uint32_t lo = round == 0 ? 0u : (32u << (round - 1)), hi = 32u << round;
        if (lo >= cnt && round > 0) break;
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
        if (!last && (!early || done < kout)) continue;
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
        bool bad = false;
        uint32_t tp = G;
        bool prove = G != 0u;
        if (!last || cnt == KP) {
            const uint32_t lp = cand_pref(sel[(size_t)q * KP + (last ? KP - 1 : done)]);
            tp = (!prove || lp > tp) ? lp : tp;
            prove = true;
        }
        if (last && G != 0u && cnt < kout) bad = true;
        if (have_kth && prove) {
            const float exact_k = ord_f32(kth_bits);
            const float T = ord_f32(tp);
            // |approx - exact| <= a255 * (2D+12) u * ||q|| * max||c||  +  8u * |offset * sum(q)|
            // eq[q] (int8 filter engine): the query's own share of the bound (its 16-bit quantisation), +inf = unprovable
            const float E = err_scale * qnorm[q] + 4.8e-7f * fabsf(ex::mul(offset, qs)) + (eq ? eq[q] : 0.0f);
            bad = !(exact_k > T + E);
        }
        const bool failed = __any(bad);
        if (failed && !last) continue;
#pragma unroll
        for (int r = 0; r < RK; ++r) {
            const uint32_t c = r * 64 + lane;
            if (c < done && rank[r] < kout) {
                out_idx[(size_t)q * kout + rank[r]] = index_base + cand_idx(e[r]);
                out_score[(size_t)q * kout + rank[r]] = ord_f32(cand_pref(e[r]));
            }
        }
        if (failed && lane == 0) fallback[q] = 1;
        return;
    }
}

// fused top-k variant (see kernels_scan.h scan_filter_kernel)
template <int QB, int R>
__global__ __launch_bounds__(256) void scan_u8_filter_kernel(const uint8_t* __restrict__ C, size_t ldN, uint32_t N,
                                                             uint32_t D, const float* __restrict__ Qm, size_t ldq,
                                                             const float* __restrict__ qsum, float a255, float offset,
                                                             uint64_t* __restrict__ lists, uint32_t* __restrict__ counts,
                                                             uint32_t qstride, uint32_t KP, uint32_t chunks_per_slot,
                                                             uint32_t* __restrict__ errflag, uint32_t nvalid = 0xFFFFFFFFu) {
    constexpr uint32_t cap = 64 * R;
    __shared__ uint32_t s_cnt[4][QB];
    __shared__ uint32_t s_thr[4][QB];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t slot = (size_t)blockIdx.x * 4 + w;
    if (lane < QB) {
        s_cnt[w][lane] = 0;
        s_thr[w][lane] = 0;
    }
    __builtin_amdgcn_wave_barrier();
    // blockIdx.y = query group (QB consecutive queries each), as in scan_filter_kernel: small corpora put all their
    // groups into one launch; lists / counts are indexed [slot][query of the launch] (qstride = queries in the launch)
    const uint32_t qoff = blockIdx.y * QB;
    Qm += (size_t)qoff * ldq;
    qsum += qoff;
    const size_t nchunks = ldN / kU8Chunk;
    size_t ch0 = slot * chunks_per_slot, ch1 = ch0 + chunks_per_slot;
    if (ch1 > nchunks) ch1 = nchunks;
    uint64_t* my_lists = lists + (slot * (size_t)qstride + qoff) * cap;
    for (size_t ch = ch0; ch < ch1; ++ch) {
        const size_t col = ch * kU8Chunk + (size_t)lane * 16;
        float acc[QB][16];
        scan_u8_accumulate<QB>(C, ldN, D, col, Qm, ldq, acc);
        // admit in four quarter-steps of 4 codes per lane (256 candidates per query at most, = kBurst), checking
        // for compaction after each: keeps the lists as short as the f32 scan's (cap = 4*KP + 256)
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
#pragma unroll
            for (int j = 0; j < QB; ++j) {
                const uint32_t thr = __hip_atomic_load(&s_thr[w][j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                const float qs = qsum[j];
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
                    const int c = 4 * c4 + cc;
                    const size_t i = col + c;
                    const uint32_t pref = f32_ord(u8_score(a255, acc[j][c], offset, qs));
                    if (i < N && pref >= thr && qoff + j < nvalid)  // (beyond nvalid: zero rows padding a ragged query tail)
                        cand_append(my_lists + (size_t)j * cap, &s_cnt[w][j], cap, cand_make(pref, (uint32_t)i), errflag);
                }
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll 1
            for (int j = 0; j < QB; ++j) {
                const uint32_t c = __builtin_amdgcn_readfirstlane(
                    __hip_atomic_load(&s_cnt[w][j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT));
                if (c > cap - kBurst) {
                    uint32_t t;
                    const uint32_t keep = wave_compact<R>(my_lists + (size_t)j * cap, c, KP, &t);
                    if (lane == 0) {
                        s_cnt[w][j] = keep;
                        s_thr[w][j] = t;
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
#pragma unroll 1
    for (int j = 0; j < QB; ++j) {
        uint32_t c = __builtin_amdgcn_readfirstlane(
            __hip_atomic_load(&s_cnt[w][j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT));
        if (c > KP) {
            uint32_t t;
            c = wave_compact<R>(my_lists + (size_t)j * cap, c, KP, &t);
        }
        if (lane == 0) counts[slot * qstride + qoff + j] = c;
    }
}

}  // namespace innr
