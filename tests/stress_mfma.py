"""Race screen: run the GEMM engine repeatedly on small shapes and compare with the exact engine (bitwise)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import oracle
from innr_amd import KNN_EXACT, KNN_MFMA, KNN_MFMA_BF16, KNN_MFMA_I8, KnnStats
from innr_amd import batch as B

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
ENGINE = {"bf16": KNN_MFMA_BF16, "i8": KNN_MFMA_I8}.get(sys.argv[2] if len(sys.argv) > 2 else "", KNN_MFMA)  # python tests/stress_mfma.py 30 bf16 | i8
bad = 0
shapes = [(10_000, 48, 70, 10), (40_000, 48, 70, 10), (3_333, 64, 300, 33), (200_000, 32, 40, 10), (1_000_000, 128, 256, 10),
          (500_000, 96, 1024, 10), (300_000, 64, 600, 16), (400_000, 320, 700, 10)]  # the last three run on 8-wave blocks (Q > 256, k <= 16)
for (n, dim, nq, k) in shapes:
    vb = B.VerticalBatch.generate(n, dim, seed=7)
    qs = oracle.generate_uniform(nq, dim, 99)
    ei, es = B.batch_knn_dot_multi(qs, vb, k, engine=KNN_EXACT)
    for r in range(rounds):
        st = KnnStats()
        mi, ms = B.batch_knn_dot_multi(qs, vb, k, engine=ENGINE, stats=st)
        if not (np.array_equal(mi, ei) and np.array_equal(ms.view(np.uint32), es.view(np.uint32))):
            bad += 1
            rows = np.where((mi != ei).any(axis=1))[0]
            print(f"MISMATCH shape={n}x{dim} Q={nq} k={k} round={r} fallback={st.queries_fallback} bad_queries={rows[:8].tolist()}")
            j = int(rows[0])
            print("   mfma ", mi[j].tolist(), ms[j].tolist())
            print("   exact", ei[j].tolist(), es[j].tolist())
    print(f"shape {n}x{dim} Q={nq}: {rounds} rounds done, cumulative mismatches {bad}", flush=True)
    vb.close()
print("TOTAL MISMATCHES", bad)
