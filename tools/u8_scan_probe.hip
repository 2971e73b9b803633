// u8_scan_probe.hip -- the exact u8 scan's inner loop (kernels_u8.h: scan_u8_accumulate inside scan_u8_scores_kernel) alone, to
// compare loop shapes in one GPU call:
//   for U in 2 4 8; do for B in 0 1; do hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -DINNR_U8_UNROLL=$U
//       -DINNR_U8_DIMBARRIER=$B -o /tmp/p_${U}_$B tools/u8_scan_probe.hip && /tmp/p_${U}_$B; done; done
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "../innr_amd/csrc/common.h"
#include "../innr_amd/csrc/topk_dev.h"
#include "../innr_amd/csrc/kernels_u8.h"
using namespace innr;
namespace innr { void set_error(const char*, ...) {} }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int QB> static int run(const uint8_t* C, size_t ldN, uint32_t D, const float* Q, float* out) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9f;
    for (int it = 0; it < 4; ++it) {
        hipEventRecord(a);
        scan_u8_scores_kernel<QB><<<256 * 8, 256>>>(C, ldN, D, Q, D, Q, 1.0f / 255.0f, -1.0f, out, ldN);
        hipEventRecord(b); CK(hipEventSynchronize(b));
        float ms; hipEventElapsedTime(&ms, a, b);
        if (it && ms < best) best = ms;
    }
    printf("  QB=%d: %7.3f ms = %6.0f GB/s of codes", QB, best, ldN * (double)D / best / 1e6);
    return 0;
}
int main(int argc, char** argv) {
    const size_t N = argc > 1 ? atol(argv[1]) : 16 * 1024 * 1024, D = 768;
    uint8_t* C; float *Q, *out;
    CK(hipMalloc(&C, N * D)); CK(hipMalloc(&Q, 8 * D * 4)); CK(hipMalloc(&out, 8 * N * 4));
    CK(hipMemset(C, 0x5a, N * D)); CK(hipMemset(Q, 0x3c, 8 * D * 4));
    printf("unroll %d, per-dimension barrier %d, %zu x %zu codes:", kU8Unroll, INNR_U8_DIMBARRIER, N, D);
    if (run<1>(C, N, (uint32_t)D, Q, out) || run<4>(C, N, (uint32_t)D, Q, out) || run<8>(C, N, (uint32_t)D, Q, out)) return 1;
    printf("\n");
    return 0;
}
