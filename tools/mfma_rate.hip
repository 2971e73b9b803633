// mfma_rate.hip -- ceiling of v_mfma_f32_32x32x2_f32 on this chip with the GEMM kernel's register shape
// (8 independent 32x32 accumulators per wave), no memory traffic: what "100 % MFMA issue" looks like in TFLOP/s.
// hipcc -O3 --offload-arch=gfx950 -o mfma_rate tools/mfma_rate.hip && ./mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters) {
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a)
        for (int g = 0; g < 16; ++g) acc[a][g] = 0.0f;
    float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-3f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
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
    hipMalloc(&out, 256 * 3 * 256 * 4);
    const int iters = 4000, blocks = 256 * blocks_per_cu;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k<NACC><<<blocks, 256>>>(out, 10);
    float best = 1e9;
    for (int it = 0; it < 3; ++it) {
        hipEventRecord(a);
        k<NACC><<<blocks, 256>>>(out, iters);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    const double flop = (double)blocks * 4 * iters * 8 * NACC * 4096.0;
    printf("accumulators/wave %d, blocks/CU %d (waves/SIMD %d): %.3f ms -> %.1f TFLOP/s (%.1f %% of 157.3)\n", NACC, blocks_per_cu,
           blocks_per_cu, best, flop / best / 1e9, flop / best / 1e9 / 157.3 * 100);
    hipFree(out);
}
int main() {
    run<8>(1); run<8>(2); run<4>(2); run<2>(2); run<1>(2);
    return 0;
}
