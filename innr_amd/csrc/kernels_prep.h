// kernels_prep.h -- corpus ingest: synthetic generator, row-major -> PDX transpose, norms; query prep.
//
// Device layout of a VerticalBatch (src/batch.rs:88-95: data[d*N + i]): V[d * ldN + i], f32, with
//   ldN  = N rounded up to 256 (one wave-chunk of the scan kernels; every 16-byte access stays aligned)
//   Dpad = D rounded up to 32 (the GEMM's K-step); rows D..Dpad-1 and columns N..ldN-1 are zero.
// Zero padding is never reported: every consumer masks idx >= N, and zero rows add fma(0,0,acc) = acc.
#pragma once

#include "common.h"

namespace innr {

// examples/batch_demo.rs:233-242 generate_embedding(dim, seed)[d], bit-exact:
//   x = seed*6364136223846793005 + d*1442695040888963407 (wrapping u64)
//   v = ((x >> 33) as f32 / 2^31) * 2.0 - 1.0
__device__ __forceinline__ float lcg_embedding(uint64_t seed, uint32_t d) {
    const uint64_t x = seed * 6364136223846793005ull + (uint64_t)d * 1442695040888963407ull;
    const float f = __uint2float_rn((uint32_t)(x >> 33));  // < 2^31, round-to-nearest-even like `as f32`
    return ex::sub(ex::mul(ex::mul(f, 4.656612873077393e-10f /* 2^-31: exact divide */), 2.0f), 1.0f);
}

// i.i.d. uniform[-1,1): splitmix64 finaliser of the element index, top 24 bits -> k*2^-23 - 1 (exact in f32).
// Same stream as the CPU checker's orc_uniform_elem (distribution of benches/batch.rs:11-21).
__device__ __forceinline__ float uniform_embedding(uint64_t seed, uint64_t row, uint32_t D, uint32_t d) {
    uint64_t z = seed * 0xD1342543DE82EF95ull + (row * (uint64_t)D + d) + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return ex::sub(ex::mul(__uint2float_rn((uint32_t)(z >> 40)), 1.1920928955078125e-07f /* 2^-23 */), 1.0f);
}

// one thread = 4 consecutive vectors of one dimension row.
// GEN 0: row i = generate_embedding(D, seed + i) (the reference example's generator);
// GEN 1: row i = uniform stream `seed`, row index row0 + i.
template <int GEN>
__global__ void generate_pdx_kernel(float* __restrict__ V, size_t ldN, uint32_t N, uint32_t D, uint64_t seed,
                                    uint64_t row0) {
    const size_t i4 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const uint32_t d = blockIdx.y;
    if (i4 >= ldN || d >= D) return;
    float v[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const size_t i = i4 + c;
        v[c] = (i < N) ? (GEN == 0 ? lcg_embedding(seed + row0 + i, d) : uniform_embedding(seed, row0 + i, D, d)) : 0.0f;
    }
    *reinterpret_cast<float4*>(V + (size_t)d * ldN + i4) = make_float4(v[0], v[1], v[2], v[3]);
}

// VerticalBatch::from_flat (src/batch.rs:167-183) on the device: rows[(i0+i)*D + d] -> V[d*ldN + i0 + i].
// 32x32 tiles through LDS (+1 pad: conflict-free column reads), coalesced on both sides.
__global__ __launch_bounds__(256) void transpose_rows_kernel(const float* __restrict__ rows, uint32_t nrows,
                                                              uint32_t D, float* __restrict__ V, size_t ldN,
                                                              size_t i0) {
    __shared__ float tile[32][33];
    const uint32_t tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const uint32_t ib = blockIdx.x * 32, db = blockIdx.y * 32;
#pragma unroll
    for (int r = 0; r < 32; r += 8) {
        const uint32_t i = ib + ty + r, d = db + tx;
        tile[ty + r][tx] = (i < nrows && d < D) ? rows[(size_t)i * D + d] : 0.0f;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 32; r += 8) {
        const uint32_t d = db + ty + r, i = ib + tx;
        if (d < D && i < nrows) V[(size_t)d * ldN + i0 + i] = tile[tx][ty + r];
    }
}

// batch_norms_into (src/batch.rs:672-686): norm_i = sqrt(sum_d fl(v*v)), d ascending, no FMA.
// One lane = 4 vectors (float4 rows); also tracks max norm (for the GEMM engine's error bound).
__global__ __launch_bounds__(256) void norms_kernel(const float* __restrict__ V, size_t ldN, uint32_t N, uint32_t D,
                                                    float* __restrict__ norms, uint32_t* __restrict__ max_norm_bits) {
    const size_t i4 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i4 >= ldN) return;
    const float4* p = reinterpret_cast<const float4*>(V + i4);
    const size_t stride = ldN / 4;
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
#pragma unroll 8
    for (uint32_t d = 0; d < D; ++d) {
        const float4 v = p[(size_t)d * stride];
        a0 = ex::mad2(a0, v.x, v.x);
        a1 = ex::mad2(a1, v.y, v.y);
        a2 = ex::mad2(a2, v.z, v.z);
        a3 = ex::mad2(a3, v.w, v.w);
    }
    float4 o = make_float4(ex::sqrt(a0), ex::sqrt(a1), ex::sqrt(a2), ex::sqrt(a3));
    *reinterpret_cast<float4*>(norms + i4) = o;
    float m = 0.0f;  // padding columns are zero vectors -> norm 0; NaN norms poison max via the uint compare
    m = fmaxf(fmaxf(o.x, o.y), fmaxf(o.z, o.w));
    uint32_t bits = __float_as_uint(m);
    if (o.x != o.x || o.y != o.y || o.z != o.z || o.w != o.w) bits = 0x7fc00000u;
    // wave max then one atomic per wave (non-negative floats order like their bit patterns)
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        uint32_t ob = (uint32_t)__shfl_xor((int)bits, off, 64);
        bits = bits > ob ? bits : ob;
    }
    if ((threadIdx.x & 63) == 0) atomicMax(max_norm_bits, bits);
}

// Query prep: q_norm[j] = sqrt(sum_d fl(q*q)) sequential (src/batch.rs:714; the -0.0 start of
// <f32 as Sum>::sum is unobservable after sqrt + compare) and abs-norm for the error bound.
__global__ __launch_bounds__(64) void query_norms_kernel(const float* __restrict__ Qm, uint32_t Q, uint32_t D, size_t ldq,
                                                         float* __restrict__ qnorm) {
    // One WAVE per query (grid = Q): the lanes load 64 consecutive values and square them, the sum runs over them in index order
    // (wave-uniform: every lane adds the same 64 lane values) -- the reference's sequential order at a coalesced load per 64
    // dimensions. (One THREAD per query took 93 us for 64 x 768: 768 dependent strided loads.)
    const uint32_t j = blockIdx.x;
    if (j >= Q) return;
    const int lane = threadIdx.x;
    const float* row = Qm + (size_t)j * ldq;
    float s = -0.0f;
    for (uint32_t d0 = 0; d0 < D; d0 += 64) {
        const uint32_t d = d0 + (uint32_t)lane;
        const float x = d < D ? row[d] : 0.0f;
        const float p = ex::mul(x, x);
        const uint32_t n = D - d0 < 64u ? D - d0 : 64u;
        for (uint32_t l = 0; l < n; ++l) s = ex::add(s, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p), (int)l)));  // (l is wave-uniform)
    }
    if (lane == 0) qnorm[j] = ex::sqrt(s);
}

}  // namespace innr
