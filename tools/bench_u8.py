"""C3 (BASELINE.json configs[2]): batch_knn_u8 asymmetric, 50M x 768 u8, 1024 queries, k=100, 1 GPU."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from innr_amd import KNN_EXACT, KNN_MFMA, KnnStats
from innr_amd import scalar as S

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
dim, nq, k = 768, 1024, 100
p = S.QuantizationParams.from_range(-1.0, 1.0)
qc = S.QuantizedCorpus.generate(n, dim, p, seed=0)
qs = oracle.generate_uniform(nq, dim, 0xBE7C)
best = None
for it in range(3):
    st = KnnStats()
    idx, sc = qc.knn_multi(qs, k, engine=KNN_MFMA, stats=st)
    if best is None or st.total_ms < best.total_ms:
        best = st
flop = 2.0 * nq * n * dim
st1 = KnnStats()
i1, s1 = qc.knn_multi(qs[:4], k, engine=KNN_EXACT, stats=st1)
assert np.array_equal(i1, idx[:4]) and np.array_equal(s1.view(np.uint32), sc[:4].view(np.uint32))
print(json.dumps({"workload": f"batch_knn_u8 {n}x{dim} u8, {nq} queries, k={k}", "engine": "f32 MFMA on widened u8 codes (path B)",
                  "total_ms": best.total_ms, "gemm_ms": best.gemm_ms, "vectors_per_s": nq * n / (best.total_ms * 1e-3),
                  "qps": nq / (best.total_ms * 1e-3), "gemm_tflops": flop / (best.gemm_ms * 1e-3) / 1e12,
                  "frac_f32_mfma_peak": flop / (best.gemm_ms * 1e-3) / 1e12 / 157.3, "corpus_GB": n * dim / 1e9,
                  "queries_fallback": best.queries_fallback, "candidates_kept": best.candidates_kept,
                  "exact_engine_4q_ms": st1.total_ms, "exact_engine_GBps": n * dim / (st1.total_ms * 1e-3) / 1e9,
                  "parity": "GEMM engine == exact engine on 4 queries (bitwise)"}))
