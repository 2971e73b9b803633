"""The three kNN kinds at the C2 shape (10M x 768 f32, 1024 queries, k = 10) on every engine that serves them.

    python tools/bench_metrics.py [N] [D] [Q] > profiles/r02_metrics_10Mx768.txt

Per row: ms of the whole call (device time, best of 3 after the call that builds the engine's corpus copy), the filter
kernel's share, queries redone exactly, and whether indices and score bits equal the f32 GEMM engine's answer for ALL queries
(which tests/ pins to the oracle at small sizes, and to the exact engine on a subset here)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from innr_amd import KNN_AUTO, KNN_EXACT, KNN_MFMA, KNN_MFMA_BF16, KNN_MFMA_I8, KnnStats
from innr_amd import batch as B

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 768
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
k = 10
vb = B.VerticalBatch.generate(n, dim, 0)
q = np.random.default_rng(0xBE7C).uniform(-1, 1, size=(nq, dim)).astype(np.float32)
names = {KNN_EXACT: "exact", KNN_MFMA: "f32 gemm", KNN_MFMA_BF16: "bf16 filter", KNN_MFMA_I8: "int8 filter", KNN_AUTO: "auto"}
print(f"# kNN k={k}, {nq} queries on {n} x {dim} f32 (uniform), one MI355X; device time of the whole call")
print(f"# {'metric':>6} {'asked':>11} {'ran':>11} {'ms':>9} {'filter ms':>9} {'redone':>6} {'Mvec/s':>10}  same answer as the f32 engine")
for metric, fn in (("dot", B.batch_knn_dot_multi), ("cosine", B.batch_knn_cosine_multi), ("l2", B.batch_knn_multi)):
    ref = None
    for engine in (KNN_MFMA, KNN_MFMA_BF16, KNN_MFMA_I8, KNN_AUTO):
        best = None
        for it in range(4):
            st = KnnStats()
            idx, sc = fn(q, vb, k, engine=engine, stats=st)
            if it and (best is None or st.total_ms < best.total_ms):
                best = st
        if ref is None:
            ref = (idx, sc)
            i4, s4 = fn(q[:4], vb, k, engine=KNN_EXACT)
            assert np.array_equal(i4, idx[:4]) and np.array_equal(s4.view(np.uint32), sc[:4].view(np.uint32))
        same = np.array_equal(idx, ref[0]) and np.array_equal(sc.view(np.uint32), ref[1].view(np.uint32))
        print(f"  {metric:>6} {names[engine]:>11} {names[best.engine]:>11} {best.total_ms:9.3f} {best.gemm_ms:9.3f} {best.queries_fallback:6d} "
              f"{nq * n / best.total_ms / 1e3:10.1f}  {same}", flush=True)
