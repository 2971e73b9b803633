// valu_rate.hip -- issue cost (cycles per wave64 instruction per SIMD) of the VALU ops the maxsim kernel leans on.
// hipcc -O3 --offload-arch=gfx950 -o valu_rate tools/valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP8(x) x x x x x x x x
template <int OP>
__global__ __launch_bounds__(256) void k(float* out, int iters, long long* cyc) {
    f2 a0 = {1.f, 2.f}, a1 = {3.f, 4.f}, a2 = {5.f, 6.f}, a3 = {7.f, 8.f}, m = {1.0001f, 0.9999f};
    float s = threadIdx.x;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) asm volatile(REP8("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m));
        if (OP == 1) asm volatile(REP8("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m));
        if (OP == 2) asm volatile(REP8("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m));
        if (OP == 3) asm volatile(REP8("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4\n") : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x) : "v"(m.x));
        if (OP == 4) { int r0, r1, r2, r3; asm volatile(REP8("v_readlane_b32 %0, %4, 3\n v_readlane_b32 %1, %4, 5\n v_readlane_b32 %2, %4, 7\n v_readlane_b32 %3, %4, 9\n") : "=s"(r0), "=s"(r1), "=s"(r2), "=s"(r3) : "v"(s)); s += r0 + r1 + r2 + r3; }
        if (OP == 5) asm volatile(REP8("v_pk_mul_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m));
        if (OP == 6) asm volatile(REP8("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4\n") : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x) : "v"(m.x));
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = a0.x + a1.x + a2.x + a3.x + a0.y + s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int OP>
void run(const char* name, int waves_per_simd) {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 2048 * 4); hipMalloc(&cyc, 8);
    const int iters = 20000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<OP><<<256 * waves_per_simd, 256>>>(out, 100, cyc);
    hipEventRecord(a);
    k<OP><<<256 * waves_per_simd, 256>>>(out, iters, cyc);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    // instructions issued per SIMD = iters * 32 * waves_per_simd
    printf("%-14s waves/SIMD %d: %.3f ms -> %.2f ns per instr per SIMD (x2.4 GHz = %.2f cycles); s_memtime-ish cycles/instr %.2f\n", name, waves_per_simd,
           ms, ms * 1e6 / (iters * 32.0 * waves_per_simd), ms * 1e6 / (iters * 32.0 * waves_per_simd) * 2.4, (double)c / (iters * 32.0));
    hipFree(out); hipFree(cyc);
}
int main() {
    for (int w = 1; w <= 2; ++w) {
        run<0>("v_pk_mul_f32", w); run<1>("v_pk_add_f32", w); run<2>("v_pk_fma_f32", w); run<3>("v_mul_f32", w);
        run<6>("v_fma_f32", w); run<4>("v_readlane_b32", w); run<5>("pk mul/add mix", w);
    }
    return 0;
}
