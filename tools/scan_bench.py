"""Quick device-time probe of the exact scan engine (HBM-bound): GB/s = 4*N*D / t per corpus pass."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from innr_amd import batch as B, KnnStats, KNN_EXACT

n, dim = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000, int(sys.argv[2]) if len(sys.argv) > 2 else 768
vb = B.VerticalBatch.generate(n, dim, 0)
rng = np.random.default_rng(0)
for nq in (1, 2, 4, 8, 16):
    q = rng.uniform(-1, 1, size=(nq, dim)).astype(np.float32)
    for metric, fn in (("dot", B.batch_knn_dot_multi), ("cos", B.batch_knn_cosine_multi), ("l2", B.batch_knn_multi)):
        best = 1e9
        for it in range(3):
            st = KnnStats()
            fn(q, vb, 10, engine=KNN_EXACT, stats=st)
            best = min(best, st.total_ms)
        passes = (nq + 7) // 8
        print(f"exact knn {metric:3s} N={n} D={dim} Q={nq:2d}: {best:8.3f} ms  -> {4.0*n*dim*passes/best/1e6:8.1f} GB/s corpus stream,"
              f" {nq*n/best/1e3:10.1f} Mvec/s", flush=True)
t0 = time.time(); s = B.batch_dot(q[0], vb); t1 = time.time()
print(f"batch_dot (incl. D2H of {n*4/1e6:.0f} MB): {(t1-t0)*1e3:.2f} ms")
