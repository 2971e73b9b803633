// valu_rate.hip -- what the f32 VALU of one MI355X delivers for the two-rounding arithmetic (mul, then add) the exact engines
// must use to match the reference bit for bit, scalar and packed, next to the fused forms the quoted vector peak assumes:
//   hipcc -O3 --offload-arch=gfx950 -o valu_rate tools/valu_rate.hip && ./valu_rate
// Each wave runs 8 independent chains per instruction kind (no operand depends on the previous instruction of its kind), every
// CU holds 8 waves (two per SIMD). Reported: f32 results per second (a packed instruction produces two per lane).
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters, float a, float b) {
    float x[8];
    f32x2 y[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        x[i] = a + (float)(threadIdx.x + i);
        y[i] = f32x2{x[i], x[i] + 1.0f};
    }
    const f32x2 b2 = {b, b};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (KIND == 0) {  // v_mul_f32 + v_add_f32
                    float t;
                    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t) : "v"(x[i]), "v"(b));
                    asm volatile("v_add_f32 %0, %1, %2" : "=v"(x[i]) : "v"(t), "v"(a));
                } else if (KIND == 1) {  // v_pk_mul_f32 + v_pk_add_f32
                    f32x2 t;
                    asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t) : "v"(y[i]), "v"(b2));
                    asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(y[i]) : "v"(t), "v"(b2));
                } else if (KIND == 2) {  // v_fma_f32
                    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(x[i]) : "v"(x[i]), "v"(b), "v"(a));
                } else if (KIND == 3) {  // v_pk_fma_f32
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(y[i]) : "v"(y[i]), "v"(b2), "v"(b2));
                } else if (KIND == 4) {  // the multiplier from an SGPR pair (what maxsim_scan_kernel's query operands are)
                    f32x2 t;
                    asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t) : "s"(b2), "v"(y[i]));
                    asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(y[i]) : "v"(t), "v"(b2));
                } else if (KIND == 5) {  // unpacked, the multiplier from an SGPR
                    float t;
                    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t) : "s"(b), "v"(x[i]));
                    asm volatile("v_add_f32 %0, %1, %2" : "=v"(x[i]) : "v"(t), "v"(a));
                } else {  // KIND 6: packed, SGPR multiplier, four products then four sums (maxsim_group's order)
                    if ((i & 3) == 0) {
                        f32x2 t0, t1, t2, t3;
                        asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t0) : "s"(b2), "v"(y[i]));
                        asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t1) : "s"(b2), "v"(y[i + 1]));
                        asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t2) : "s"(b2), "v"(y[i + 2]));
                        asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t3) : "s"(b2), "v"(y[i + 3]));
                        asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(y[i]) : "v"(t0), "v"(b2));
                        asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(y[i + 1]) : "v"(t1), "v"(b2));
                        asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(y[i + 2]) : "v"(t2), "v"(b2));
                        asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(y[i + 3]) : "v"(t3), "v"(b2));
                    }
                }
            }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i] + y[i].x + y[i].y;
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int KIND>
static int run(const char* name, double results_per_instr_lane, double instr_per_iter, float* out) {
    const int iters = 20000, blocks = 256 * 2;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    rate_kernel<KIND><<<blocks, 256>>>(out, 100, 1.0f, 1.0000001f);
    CK(hipEventRecord(e0));
    rate_kernel<KIND><<<blocks, 256>>>(out, iters, 1.0f, 1.0000001f);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double instr = (double)blocks * 4 /*waves*/ * iters * instr_per_iter;       // wave instructions
    const double res = instr * 64 * results_per_instr_lane;
    printf("%-28s %8.2f ms  %6.2f cycles per wave instruction per SIMD at 2.4 GHz  %7.2f T results/s\n", name, ms,
           ms * 1e-3 * 2.4e9 / (instr / 1024.0), res / (ms * 1e-3) / 1e12);
    return 0;
}

int main() {
    float* out;
    CK(hipMalloc(&out, 4096));
    if (run<0>("v_mul_f32 + v_add_f32", 1, 64, out)) return 1;
    if (run<1>("v_pk_mul_f32 + v_pk_add_f32", 2, 64, out)) return 1;
    if (run<2>("v_fma_f32", 1, 32, out)) return 1;
    if (run<3>("v_pk_fma_f32", 2, 32, out)) return 1;
    if (run<4>("v_pk_mul_f32 (SGPR) + v_pk_add", 2, 64, out)) return 1;
    if (run<5>("v_mul_f32 (SGPR) + v_add_f32", 1, 64, out)) return 1;
    if (run<6>("4 x pk_mul (SGPR), 4 x pk_add", 2, 64, out)) return 1;
    return 0;
}
