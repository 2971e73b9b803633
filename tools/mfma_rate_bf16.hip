// mfma_rate_bf16.hip -- ceiling of v_mfma_f32_32x32x16_bf16 (the candidate pipe for a low-precision filter stage,
// DESIGN.md section 7) with the GEMM kernel's register shape: 8 independent 32x32 f32 accumulators per wave, no
// memory traffic. hipcc -O3 --offload-arch=gfx950 -o mfma_rate_bf16 tools/mfma_rate_bf16.hip && ./mfma_rate_bf16
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int NACC>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters) {
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a)
        for (int g = 0; g < 16; ++g) acc[a][g] = 0.0f;
    bf16x8 x, y;
    for (int e = 0; e < 8; ++e) {
        x[e] = (__bf16)(threadIdx.x * 1e-3f + e);
        y[e] = (__bf16)(blockIdx.x * 1e-3f - e);
    }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[a], 0, 0, 0);
        asm volatile("" : "+v"(x), "+v"(y));
    }
    float s = 0.0f;
    for (int a = 0; a < NACC; ++a)
        for (int g = 0; g < 16; ++g) s += acc[a][g];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
void run(int blocks_per_cu) {
    float* out;
    (void)hipMalloc(&out, 256 * 3 * 256 * 4);
    const int iters = 4000, blocks = 256 * blocks_per_cu;
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    k<NACC><<<blocks, 256>>>(out, 10);
    float best = 1e9;
    for (int it = 0; it < 3; ++it) {
        (void)hipEventRecord(a);
        k<NACC><<<blocks, 256>>>(out, iters);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    const double flop = (double)blocks * 4 * iters * 8 * NACC * (2.0 * 32 * 32 * 16);
    printf("bf16 32x32x16: accumulators/wave %d, waves/SIMD %d: %.3f ms -> %.1f TFLOP/s (%.1f %% of 2516)\n", NACC, blocks_per_cu, best,
           flop / best / 1e9, flop / best / 1e9 / 2516 * 100);
    (void)hipFree(out);
}
int main() {
    run<8>(1); run<8>(2); run<4>(2);
    return 0;
}
