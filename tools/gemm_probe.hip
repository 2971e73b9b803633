// gemm_probe.hip -- where does gemm_filter_kernel's time go at C2 (10M x 768, 1024 queries)? Build variants:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off [-DINNR_GEMM_PROBE_NOEPI] [-DINNR_GEMM_PROBE_NODMA]
//         [-DINNR_GEMM_PROBE_NOBAR] -o gemm_probe tools/gemm_probe.hip && ./gemm_probe
// (NOEPI: K-loop only; NODMA: no LDS-DMA inside the loop; NOBAR: no per-step barrier. Results are meaningless in
//  the probe variants -- only the time is read.)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../innr_amd/csrc/common.h"
#include "../innr_amd/csrc/topk_dev.h"
#include "../innr_amd/csrc/kernels_prep.h"
#include "../innr_amd/csrc/kernels_gemm.h"
using namespace innr;
#ifndef INNR_GEMM_WAVES
#define INNR_GEMM_WAVES 8
#endif
constexpr int WAVES = INNR_GEMM_WAVES;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main(int argc, char** argv) {
    const size_t N = argc > 1 ? atol(argv[1]) : 10000000, D = 768, Q = 1024, ldN = (N + 255) / 256 * 256, Qpad = Q;
#ifndef PROBE_R
#define PROBE_R 6
#define PROBE_KP 32
#endif
    const uint32_t KP = PROBE_KP, cap = 64 * PROBE_R, nqt = Q / (64 * WAVES), ns = (8 / WAVES) * 256 / nqt, ntiles = ldN / kBC, tps = (ntiles + ns - 1) / ns;
    float *V, *Qt;
    uint64_t* lists; uint32_t *counts, *gs, *err;
    CK(hipMalloc(&V, ldN * D * 4)); CK(hipMalloc(&Qt, D * Qpad * 4));
    CK(hipMalloc(&lists, (size_t)ns * Qpad * cap * 8)); CK(hipMalloc(&counts, (size_t)ns * Qpad * 4));
    CK(hipMalloc(&gs, (Qpad * kSlotMul * KP + 2 * Qpad) * 4 /* slots, bounds, k-rule margins */)); CK(hipMalloc(&err, 4096));
    generate_pdx_kernel<1><<<dim3((unsigned)((ldN / 4 + 255) / 256), (unsigned)D), 256>>>(V, ldN, (uint32_t)N, (uint32_t)D, 0, 0);
    generate_pdx_kernel<1><<<dim3((unsigned)((Qpad / 4 + 255) / 256), (unsigned)D), 256>>>(Qt, Qpad, (uint32_t)Q, (uint32_t)D, 0xBE7C, 0);
    CK(hipMemset(err, 0, 4096));
    CK(hipDeviceSynchronize());
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9;
    for (int it = 0; it < 4; ++it) {
        if (!(getenv("PROBE_KEEP") && it > 0)) CK(hipMemset(gs, 0, (Qpad * kSlotMul * KP + 2 * Qpad) * 4 /* slots, bounds, k-rule margins */));  // PROBE_KEEP=1: later launches start from the final thresholds (perfect seeding)
        if (getenv("PROBE_KEEP") && it == 3) CK(hipMemset(err, 0, 4096));  // counters of the last (perfectly seeded) launch only
        hipEventRecord(a);
        gemm_filter_kernel<kGemmDot, PROBE_R, 0, WAVES><<<nqt * ns, 64 * WAVES>>>(V, ldN, (uint32_t)N, (uint32_t)D, Qt, Qpad, nqt, 1, tps, nullptr, nullptr,
                                                                       1.0f, lists, counts, KP, 0u /* k rule off */, err, gs, gs + Qpad * kSlotMul * KP, nullptr, 0);
        hipEventRecord(b); CK(hipEventSynchronize(b));
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
#ifdef INNR_GEMM_PROBE_COUNT
    {
        uint32_t h[20];
        CK(hipMemcpy(h, err, sizeof(h), hipMemcpyDeviceToHost));
        const double tiles_waves = 4.0 * (double)ntiles * nqt * (WAVES / 4.0);  // (tile, wave) epilogues per launch
        unsigned long long cyc; memcpy(&cyc, h + 12, 8);
        printf("cycles inside the append path: %.0f per entry; %.2f ms per wave per launch (s_memtime ticks at 100 MHz? raw %llu)\n",
               (double)cyc / h[8], (double)cyc / 4 / (4.0 * nqt * ns * WAVES / 4.0) / 1e5, cyc);
        unsigned long long c1, c2; memcpy(&c1, h + 14, 8); memcpy(&c2, h + 16, 8);
        printf("  of which scan + appends %.0f, bound re-derivation %.0f cycles per entry\n", (double)c1 / h[8], (double)c2 / h[8]);
        printf("over 4 launches: append path entered %u times (%.1f %% of %.0f wave-epilogues per launch), %u (lane,query) hits, %u appends (%.1f per query per launch)\n",
               h[8], 100.0 * h[8] / 4 / tiles_waves, tiles_waves, h[9], h[10], h[10] / 4.0 / Q);
    }
#endif
    printf("gemm_filter_kernel %zux%zu x %zu queries: %.2f ms -> %.1f TFLOP/s (%.1f %% of 157.3)\n", N, D, Q, best,
           2.0 * N * D * Q / best / 1e9, 2.0 * N * D * Q / best / 1e9 / 1.573);
    return 0;
}
