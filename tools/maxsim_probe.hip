// maxsim_probe.hip -- where does maxsim_scan_kernel's time go? Build three ways and compare:
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off [-DINNR_MS_PROBE_NOLOAD | -DINNR_MS_PROBE_NOMATH | -DINNR_MS_PROBE_NOSLOAD] \
//         -o maxsim_probe tools/maxsim_probe.hip && ./maxsim_probe [ndocs]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../innr_amd/csrc/common.h"
#include "../innr_amd/csrc/topk_dev.h"
#include "../innr_amd/csrc/kernels_prep.h"
#include "../innr_amd/csrc/kernels_maxsim.h"
using namespace innr;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main(int argc, char** argv) {
    const size_t ndocs = argc > 1 ? atol(argv[1]) : 100000, T = 64, dim = 128, NQ = 32;
    float *tok, *q, *out;
    CK(hipMalloc(&tok, ndocs * T * dim * 4));
    CK(hipMalloc(&q, NQ * dim * 4));
    CK(hipMalloc(&out, ndocs * 4));
    generate_tokens_kernel<<<(unsigned)((ndocs * T + 255) / 256), 256>>>(tok, ndocs * T, dim, 1, 0);
    generate_tokens_kernel<<<1, 64>>>(q, NQ, dim, 2, 0);
    CK(hipDeviceSynchronize());
    float* qpk;
    CK(hipMalloc(&qpk, NQ * dim * 4));
    maxsim_pack_query_kernel<<<(NQ * dim + 255) / 256, 256>>>(q, NQ, dim, qpk);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int blocks_per_cu = 1; blocks_per_cu <= 3; ++blocks_per_cu) {
        const unsigned blocks = 256 * blocks_per_cu;
        float best = 1e9;
        for (int it = 0; it < 5; ++it) {
            hipEventRecord(a);
            maxsim_scan_kernel<false, 32, false><<<blocks, kMsThreads>>>(tok, nullptr, (uint32_t)ndocs, T, 64, dim, q, qpk, NQ, nullptr, out, out, true, nullptr);
            hipEventRecord(b);
            CK(hipEventSynchronize(b));
            float ms; hipEventElapsedTime(&ms, a, b);
            if (ms < best) best = ms;
        }
        printf("blocks/CU %d: %.3f ms  (%.0f GB/s)\n", blocks_per_cu, best, ndocs * T * dim * 4 / best / 1e6);
    }
    return 0;
}
