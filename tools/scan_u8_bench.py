"""Exact u8 engine (scalar::batch_knn_u8, one query per call = the reference's signature) at C3's corpus size."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from innr_amd import KNN_EXACT, KnnStats
from innr_amd import scalar as S
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
dim, k = 768, 100
p = S.QuantizationParams.from_range(-1.0, 1.0)
qc = S.QuantizedCorpus.generate(n, dim, p, seed=0)
qs = np.random.default_rng(0xBE7C).uniform(-1.0, 1.0, size=(16, dim)).astype(np.float32)
for nq in (1, 4, 8, 16):
    best = 1e9
    for _ in range(3):
        st = KnnStats()
        qc.knn_multi(qs[:nq], k, engine=KNN_EXACT, stats=st)
        best = min(best, st.total_ms)
    print(f"u8 exact knn N={n} D={dim} Q={nq:2d}: {best:8.3f} ms -> {n*dim/best/1e6:7.1f} GB/s corpus stream per pass-equivalent, {nq*n/best/1e3:9.1f} Mvec/s")
