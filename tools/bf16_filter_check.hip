// bf16_filter_check.hip -- stand-alone check of kernels_gemm_bf16.h before it is wired into the library:
//   1. layout test: pack a small random corpus / query set, run MODE 1 (dense dump), compare every score with a host
//      dot product of the bf16-rounded inputs (double accumulation);
//   2. speed: MODE 0 (fused filter, unseeded thresholds) at the C2 shape.
// hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -o bf16_check tools/bf16_filter_check.hip && ./bf16_check
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../innr_amd/csrc/common.h"
#include "../innr_amd/csrc/topk_dev.h"
#include "../innr_amd/csrc/kernels_prep.h"
#include "../innr_amd/csrc/kernels_gemm_bf16.h"
using namespace innr;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

static float bf16_round_host(float x) {
    uint32_t b; memcpy(&b, &x, 4);
    b += 0x7fffu + ((b >> 16) & 1u);
    b &= 0xffff0000u;
    float y; memcpy(&y, &b, 4);
    return y;
}

static int layout_test(size_t N, size_t D, size_t Q) {
    const size_t ldN = (N + 255) / 256 * 256, Qpad = (Q + 511) / 512 * 512;
    const uint32_t nk = (uint32_t)((D + 63) / 64 * 2), ntiles = (uint32_t)(ldN / 128), nqt = (uint32_t)(Qpad / 512);
    std::vector<float> V(D * ldN, 0.f), Qm(Q * D);
    srand(7);
    for (size_t d = 0; d < D; ++d) for (size_t i = 0; i < N; ++i) V[d * ldN + i] = (float)rand() / RAND_MAX * 2 - 1;
    for (auto& x : Qm) x = (float)rand() / RAND_MAX * 2 - 1;
    float *dV, *dQ, *dump; char *Ab, *Bb; uint32_t* err;
    const size_t aunits = (size_t)ntiles * nk * 512, abytes = aunits * 16, bbytes = (size_t)nk * 4 * Qpad * 16;
    CK(hipMalloc(&dV, V.size() * 4)); CK(hipMalloc(&dQ, Qm.size() * 4)); CK(hipMalloc(&Ab, abytes)); CK(hipMalloc(&Bb, bbytes));
    CK(hipMalloc(&dump, Qpad * ldN * 4)); CK(hipMalloc(&err, 4096)); CK(hipMemset(err, 0, 4096)); CK(hipMemset(dump, 0xff, Qpad * ldN * 4));
    CK(hipMemcpy(dV, V.data(), V.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dQ, Qm.data(), Qm.size() * 4, hipMemcpyHostToDevice));
    pack_corpus_bf16_kernel<<<(unsigned)((aunits + 255) / 256), 256>>>(dV, ldN, (uint32_t)N, (uint32_t)D, nk, aunits, (uint4*)Ab);
    pack_queries_bf16_kernel<<<(unsigned)(((size_t)nk * 4 * Qpad + 255) / 256), 256>>>(dQ, (uint32_t)Q, (uint32_t)D, nk, (uint32_t)Qpad, (uint4*)Bb);
    const uint32_t ns = 8, tps = (ntiles + ns - 1) / ns;
    gemm_bf16_filter_kernel<6, 1><<<nqt * ns, 512>>>(Ab, Bb, ntiles, (uint32_t)N, nk, Qpad, nqt, 1, tps, nullptr, nullptr, 32, 0u, err, nullptr,
                                                     nullptr, dump, ldN);
    CK(hipDeviceSynchronize());
    std::vector<float> out(Qpad * ldN);
    CK(hipMemcpy(out.data(), dump, out.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0; size_t bad = 0;
    for (size_t q = 0; q < Q; ++q)
        for (size_t i = 0; i < N; ++i) {
            double ref = 0, mag = 0;
            for (size_t d = 0; d < D; ++d) {
                const double a = bf16_round_host(V[d * ldN + i]), b = bf16_round_host(Qm[q * D + d]);
                ref += a * b; mag += fabs(a * b);
            }
            const double e = fabs(out[q * ldN + i] - ref) / (mag + 1e-30);
            if (!(e < 1e-5)) { if (bad < 5) printf("  q=%zu i=%zu got %g want %g\n", q, i, out[q * ldN + i], ref); ++bad; }
            if (e > worst) worst = e;
        }
    printf("layout test N=%zu D=%zu Q=%zu: %zu of %zu scores off, worst relative (to sum|ab|) %.2e -> %s\n", N, D, Q, bad, Q * N, worst,
           bad ? "FAIL" : "ok");
    hipFree(dV); hipFree(dQ); hipFree(Ab); hipFree(Bb); hipFree(dump); hipFree(err);
    return bad ? 1 : 0;
}

static int speed_test(size_t N, int KPsel) {
    const size_t D = 768, Q = 1024, ldN = (N + 255) / 256 * 256, Qpad = Q;
    const uint32_t nk = (uint32_t)(D / 32), ntiles = (uint32_t)(ldN / 128), nqt = 2, ns = 128, tps = (ntiles + ns - 1) / ns;
    const uint32_t KP = KPsel, cap = 4 * KP + 256;
    float *dV, *dQ; char *Ab, *Bb; uint64_t* lists; uint32_t *counts, *gs, *err;
    const size_t aunits = (size_t)ntiles * nk * 512;
    CK(hipMalloc(&dV, ldN * D * 4)); CK(hipMalloc(&dQ, Q * D * 4)); CK(hipMalloc(&Ab, aunits * 16)); CK(hipMalloc(&Bb, (size_t)nk * 4 * Qpad * 16));
    CK(hipMalloc(&lists, (size_t)ns * Qpad * cap * 8)); CK(hipMalloc(&counts, (size_t)ns * Qpad * 4));
    const size_t gwords = Qpad * (size_t)kSlotMul * KP + 2 * Qpad;  // slots, bounds, k-rule margins
    CK(hipMalloc(&gs, gwords * 4)); CK(hipMalloc(&err, 4096)); CK(hipMemset(err, 0, 4096));
    generate_pdx_kernel<1><<<dim3((unsigned)((ldN / 4 + 255) / 256), (unsigned)D), 256>>>(dV, ldN, (uint32_t)N, (uint32_t)D, 0, 0);
    std::vector<float> Qm(Q * D); srand(3); for (auto& x : Qm) x = (float)rand() / RAND_MAX * 2 - 1;
    CK(hipMemcpy(dQ, Qm.data(), Qm.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a));
    pack_corpus_bf16_kernel<<<(unsigned)((aunits + 255) / 256), 256>>>(dV, ldN, (uint32_t)N, (uint32_t)D, nk, aunits, (uint4*)Ab);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float pack_ms; CK(hipEventElapsedTime(&pack_ms, a, b));
    pack_queries_bf16_kernel<<<(unsigned)(((size_t)nk * 4 * Qpad + 255) / 256), 256>>>(dQ, (uint32_t)Q, (uint32_t)D, nk, (uint32_t)Qpad, (uint4*)Bb);
    float best = 1e9;
    for (int it = 0; it < 4; ++it) {
        if (!(getenv("PROBE_KEEP") && it > 0)) CK(hipMemset(gs, 0, gwords * 4));  // PROBE_KEEP=1: later launches start from the final bounds
        CK(hipEventRecord(a));
        if (KP == 32) gemm_bf16_filter_kernel<6, 0><<<nqt * ns, 512>>>(Ab, Bb, ntiles, (uint32_t)N, nk, Qpad, nqt, 1, tps, lists, counts, KP, 0u /* k rule off */, err, gs, gs + Qpad * kSlotMul * KP, nullptr, 0);
        else gemm_bf16_filter_kernel<12, 0><<<nqt * ns, 512>>>(Ab, Bb, ntiles, (uint32_t)N, nk, Qpad, nqt, 1, tps, lists, counts, KP, 0u /* k rule off */, err, gs, gs + Qpad * kSlotMul * KP, nullptr, 0);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    uint32_t h[4]; CK(hipMemcpy(h, err, 16, hipMemcpyDeviceToHost));
    std::vector<uint32_t> cnt((size_t)ns * Qpad); CK(hipMemcpy(cnt.data(), counts, cnt.size() * 4, hipMemcpyDeviceToHost));
    unsigned long long tot = 0; for (auto c : cnt) tot += c;
    printf("C2 shape, KP=%u: corpus pack %.1f ms (once per corpus); filter kernel %.2f ms -> %.0f TFLOP/s (%.1f %% of 2516); errflag %u; "
           "%.1f candidates left per query\n", KP, pack_ms, best, 2.0 * N * D * Q / best / 1e9, 2.0 * N * D * Q / best / 1e9 / 25.16, h[0], (double)tot / Q);
    return 0;
}

int main(int argc, char** argv) {
    int rc = 0;
    rc |= layout_test(1000, 64, 70);
    rc |= layout_test(3000, 100, 513);
    rc |= layout_test(257, 768, 5);
    if (rc) return rc;
    if (argc > 1 && !strcmp(argv[1], "layout")) return 0;
    rc |= speed_test(10000000, 32);
    rc |= speed_test(10000000, 128);
    return rc;
}
