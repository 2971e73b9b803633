// mfma_rate_i8.hip -- ceiling of v_mfma_i32_32x32x32_i8 (the int8 filter engine's pipe, kernels_gemm_i8.h) with the engine's
// register shape (8 independent 32x32 i32 accumulators per wave), no memory traffic, random-ish operands; the bf16 form
// beside it on the same box. hipcc -O3 --offload-arch=gfx950 -o mfma_rate_i8 tools/mfma_rate_i8.hip && ./mfma_rate_i8
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4acc __attribute__((ext_vector_type(4)));
typedef float f32x4acc __attribute__((ext_vector_type(4)));
// the 16x16 shapes with the same accumulator footprint: 4 x NACC tiles of 16x16 (4 registers each)
template <int NACC, int BF>
__global__ __launch_bounds__(256, 2) void k16(int* out, int iters) {
    i32x4acc acc[4 * NACC];
    f32x4acc facc[4 * NACC];
    for (int a = 0; a < 4 * NACC; ++a)
        for (int g = 0; g < 4; ++g) { acc[a][g] = 0; facc[a][g] = 0.0f; }
    i32x4 x, y;
    for (int e = 0; e < 4; ++e) {
        x[e] = (int)(threadIdx.x * 2654435761u + e * 40503u);
        y[e] = (int)(blockIdx.x * 2246822519u - e * 97u + threadIdx.x);
    }
    const bf16x8 bx = __builtin_bit_cast(bf16x8, x), by = __builtin_bit_cast(bf16x8, y);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int a = 0; a < 4 * NACC; ++a) {
                if (BF) facc[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bx, by, facc[a], 0, 0, 0);
                else acc[a] = __builtin_amdgcn_mfma_i32_16x16x64_i8(x, y, acc[a], 0, 0, 0);
            }
        asm volatile("" : "+v"(x), "+v"(y));
    }
    int s = 0;
    for (int a = 0; a < 4 * NACC; ++a)
        for (int g = 0; g < 4; ++g) s += acc[a][g] + (int)facc[a][g];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC, int BF>
void run16(int blocks_per_cu) {
    int* out;
    (void)hipMalloc(&out, 256 * 3 * 256 * 4);
    const int iters = 2000, blocks = 256 * blocks_per_cu;
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    k16<NACC, BF><<<blocks, 256>>>(out, 10);
    float best = 1e9;
    for (int it = 0; it < 3; ++it) {
        (void)hipEventRecord(a);
        k16<NACC, BF><<<blocks, 256>>>(out, iters);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    const double n_mfma = (double)blocks * 4 * iters * 8 * 4 * NACC;
    const double ops = n_mfma * (2.0 * 16 * 16 * (BF ? 32 : 64));
    printf("%s: tiles/wave %d, waves/SIMD %d: %.3f ms -> %.1f T%s/s\n", BF ? "bf16 16x16x32" : "i8   16x16x64", 4 * NACC, blocks_per_cu,
           best, ops / best / 1e9, BF ? "FLOP" : "OP");
    (void)hipFree(out);
}
template <int NACC, int BF>
__global__ __launch_bounds__(256, 2) void k(int* out, int iters) {
    i32x16 acc[NACC];
    f32x16 facc[NACC];
    for (int a = 0; a < NACC; ++a)
        for (int g = 0; g < 16; ++g) { acc[a][g] = 0; facc[a][g] = 0.0f; }
    i32x4 x, y;
    for (int e = 0; e < 4; ++e) {
        x[e] = (int)(threadIdx.x * 2654435761u + e * 40503u);
        y[e] = (int)(blockIdx.x * 2246822519u - e * 97u + threadIdx.x);
    }
    const bf16x8 bx = __builtin_bit_cast(bf16x8, x), by = __builtin_bit_cast(bf16x8, y);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int a = 0; a < NACC; ++a) {
                if (BF) facc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx, by, facc[a], 0, 0, 0);
                else acc[a] = __builtin_amdgcn_mfma_i32_32x32x32_i8(x, y, acc[a], 0, 0, 0);
            }
        asm volatile("" : "+v"(x), "+v"(y));
    }
    int s = 0;
    for (int a = 0; a < NACC; ++a)
        for (int g = 0; g < 16; ++g) s += acc[a][g] + (int)facc[a][g];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC, int BF>
void run(int blocks_per_cu) {
    int* out;
    (void)hipMalloc(&out, 256 * 3 * 256 * 4);
    const int iters = 4000, blocks = 256 * blocks_per_cu;
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    k<NACC, BF><<<blocks, 256>>>(out, 10);
    float best = 1e9;
    for (int it = 0; it < 3; ++it) {
        (void)hipEventRecord(a);
        k<NACC, BF><<<blocks, 256>>>(out, iters);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    const double n_mfma = (double)blocks * 4 * iters * 8 * NACC;
    const double ops = n_mfma * (2.0 * 32 * 32 * (BF ? 16 : 32));
    // cycles per MFMA per SIMD at a nominal 2.4 GHz: time * clock / (MFMAs per SIMD)
    const double per_simd = n_mfma / (256.0 * 4);
    printf("%s: accumulators/wave %d, waves/SIMD %d: %.3f ms -> %.1f T%s/s, %.1f cycles per MFMA per SIMD at 2.4 GHz\n",
           BF ? "bf16 32x32x16" : "i8   32x32x32", NACC, blocks_per_cu, best, ops / best / 1e9, BF ? "FLOP" : "OP",
           best * 1e-3 * 2.4e9 / per_simd);
    (void)hipFree(out);
}
int main() {
    run<8, 0>(1); run<8, 0>(2); run<8, 1>(1); run<8, 1>(2);
    run16<8, 0>(2); run16<8, 1>(2);
    return 0;
}
