// bf16_epi_probe.hip -- the PRODUCT bf16 filter kernel (kernels_gemm_bf16.h + gemm_epilogue.inc) at the C2 shape, with the
// append-path counters of the shared epilogue switched on: how often is the append path visited, what does a visit cost,
// and what is left when the thresholds are perfect. Used to A/B epilogue variants in one GPU call:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -DINNR_GEMM_PROBE_COUNT [-DINNR_PUB_EVERY=1] [-DPROBE_R=12
//         -DPROBE_KP=128] -o bf16_epi_probe tools/bf16_epi_probe.hip && ./bf16_epi_probe [N]
// Launch 1 starts from empty thresholds (the product seeds them from an exact prefix scan: it sits between launch 1 and the
// "kept" launches); launches 2-4 start from the final thresholds of the previous one (perfect seeding).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../innr_amd/csrc/common.h"
#include "../innr_amd/csrc/topk_dev.h"
#include "../innr_amd/csrc/kernels_prep.h"
#include "../innr_amd/csrc/kernels_gemm_bf16.h"
using namespace innr;
namespace innr { void set_error(const char*, ...) {} }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#ifndef PROBE_R
#define PROBE_R 12
#define PROBE_KP 128
#endif
__global__ void gen_rows(float* q, uint32_t Q, uint32_t D, uint64_t seed) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < (size_t)Q * D) q[t] = uniform_embedding(seed, t / D, D, (uint32_t)(t % D));
}
int main(int argc, char** argv) {
    const size_t N = argc > 1 ? atol(argv[1]) : 10000000, D = 768, Q = 1024, ldN = (N + 255) / 256 * 256, Qpad = Q;
    const uint32_t KP = PROBE_KP, cap = 64 * PROBE_R, nqt = Q / 512, ns = 256 / nqt, ntiles = ldN / 128, tps = (ntiles + ns - 1) / ns;
    const uint32_t nk = (uint32_t)((D + 63) / 64 * 2);
    float *V, *Qm;
    char *Ab, *Bb;
    uint64_t* lists; uint32_t *counts, *gs, *err;
    const size_t units = (size_t)ntiles * nk * 512;
    CK(hipMalloc(&V, ldN * D * 4)); CK(hipMalloc(&Qm, Q * D * 4));
    CK(hipMalloc(&Ab, units * 16)); CK(hipMalloc(&Bb, (size_t)nk * 4 * Qpad * 16));
    CK(hipMalloc(&lists, (size_t)ns * Qpad * cap * 8)); CK(hipMalloc(&counts, (size_t)ns * Qpad * 4));
    CK(hipMalloc(&gs, (Qpad * kSlotMul * KP + 2 * Qpad) * 4 /* slots, bounds, k-rule margins */)); CK(hipMalloc(&err, 4096));
    generate_pdx_kernel<1><<<dim3((unsigned)((ldN / 4 + 255) / 256), (unsigned)D), 256>>>(V, ldN, (uint32_t)N, (uint32_t)D, 0, 0);
    gen_rows<<<(unsigned)((Q * D + 255) / 256), 256>>>(Qm, (uint32_t)Q, (uint32_t)D, 0xBE7C);
    pack_corpus_bf16_kernel<<<(unsigned)((units + 255) / 256), 256>>>(V, ldN, (uint32_t)N, (uint32_t)D, nk, units, (uint4*)Ab);
    pack_queries_bf16_kernel<<<(unsigned)(((size_t)nk * 4 * Qpad + 255) / 256), 256>>>(Qm, (uint32_t)Q, (uint32_t)D, nk, (uint32_t)Qpad, (uint4*)Bb);
    CK(hipDeviceSynchronize());
    CK(hipFree(V));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int it = 0; it < 4; ++it) {
        if (it == 0) CK(hipMemset(gs, 0, (Qpad * kSlotMul * KP + 2 * Qpad) * 4 /* slots, bounds, k-rule margins */));
        CK(hipMemset(err, 0, 4096));
        hipEventRecord(a);
        gemm_bf16_filter_kernel<PROBE_R, 0><<<nqt * ns, 64 * kBfWaves>>>(Ab, Bb, ntiles, (uint32_t)N, nk, Qpad, nqt, 1, tps, lists, counts, KP, 0u /* k rule off */, err, gs,
                                                                      gs + Qpad * kSlotMul * KP, nullptr, 0);
        hipEventRecord(b); CK(hipEventSynchronize(b));
        float ms; hipEventElapsedTime(&ms, a, b);
        uint32_t h[20];
        CK(hipMemcpy(h, err, sizeof(h), hipMemcpyDeviceToHost));
        unsigned long long cyc, c1, c2; memcpy(&cyc, h + 12, 8); memcpy(&c1, h + 14, 8); memcpy(&c2, h + 16, 8);
        const double wave_tiles = (double)ntiles * nqt * kBfWaves;
        printf("launch %d (%s thresholds): %.2f ms = %.0f TFLOP/s | errflag %u | visits %u (%.1f %% of %.0f wave-tiles), hits %u, appends %u "
               "(%.1f per query) | cycles per visit %.0f (scan+append %.0f, publish %.0f)\n",
               it, it == 0 ? "empty" : "kept", ms, 2.0 * N * D * Q / ms / 1e9, h[0], h[8], 100.0 * h[8] / wave_tiles, wave_tiles, h[9], h[10],
               h[10] / (double)Q, h[8] ? (double)cyc / h[8] : 0.0, h[8] ? (double)c1 / h[8] : 0.0, h[8] ? (double)c2 / h[8] : 0.0);
    }
    return 0;
}
