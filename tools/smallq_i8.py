import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from innr_amd import KNN_MFMA_I8, KNN_EXACT, KnnStats
from innr_amd import batch as B
vb = B.VerticalBatch.generate(10_000_000, 768, 0)
rng = np.random.default_rng(0)
for nq in (1, 2, 4, 8, 16, 64, 128, 256, 512):
    q = rng.uniform(-1, 1, size=(nq, 768)).astype(np.float32)
    for metric, fn in (("dot", B.batch_knn_dot_multi), ("cos", B.batch_knn_cosine_multi)):
        best = None
        for it in range(4):
            st = KnnStats()
            i1, s1 = fn(q, vb, 10, engine=KNN_MFMA_I8, stats=st)
            if it and (best is None or st.total_ms < best.total_ms): best = st
        i0, s0 = fn(q[:2], vb, 10, engine=KNN_EXACT)
        print(f"Q={nq} {metric}: int8 call {best.total_ms:.3f} ms kernel {best.gemm_ms:.3f} ms engine {best.engine} redone {best.queries_fallback} same_as_exact {bool(np.array_equal(i0, i1[:2]) and np.array_equal(s0.view(np.uint32), s1[:2].view(np.uint32)))}", flush=True)
