"""Race screen for the int8 engine on code corpora: run batch_knn_u8 repeatedly and compare with the exact engine (bitwise).
    python tests/stress_u8.py [rounds]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import oracle
from innr_amd import KNN_EXACT, KNN_MFMA_I8, KnnStats
from innr_amd import scalar as S

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
bad = 0
p = S.QuantizationParams.from_range(-1.0, 1.0)
for (n, dim, nq, k) in [(300_000, 320, 600, 10), (1_000_000, 128, 520, 100), (200_000, 768, 1030, 33), (150_000, 64, 300, 120), (2_000_000, 256, 513, 10)]:
    qc = S.QuantizedCorpus.generate(n, dim, p, seed=11)
    qs = oracle.generate_uniform(nq, dim, 5)
    ei, es = qc.knn_multi(qs, k, engine=KNN_EXACT)
    for r in range(rounds):
        st = KnnStats()
        mi, ms = qc.knn_multi(qs, k, engine=KNN_MFMA_I8, stats=st)
        if not (np.array_equal(mi, ei) and np.array_equal(ms.view(np.uint32), es.view(np.uint32))):
            bad += 1
            rows = np.where((mi != ei).any(axis=1))[0]
            print(f"MISMATCH shape={n}x{dim} Q={nq} k={k} round={r} fallback={st.queries_fallback} bad_queries={rows[:8].tolist()}")
    print(f"shape {n}x{dim} Q={nq} k={k}: {rounds} rounds done, cumulative mismatches {bad}", flush=True)
print("TOTAL MISMATCHES", bad)
