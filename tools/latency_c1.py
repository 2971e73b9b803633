"""C1 shape (BASELINE.json configs[0]: 10K x 128 f32, 100 queries, k = 10): call latencies of the reference-shaped
API on one GPU -- one query per call (the reference's signature) and the whole batch in one call."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from innr_amd import KNN_EXACT, KNN_MFMA, KnnStats
from innr_amd import batch as B

n, dim, nq, k = 10_000, 128, 100, 10
vb = B.VerticalBatch.generate(n, dim, seed=0)
qs = np.random.default_rng(50_000).uniform(-1.0, 1.0, size=(nq, dim)).astype(np.float32)
for _ in range(3):
    B.batch_knn_dot(qs[0], vb, k)
t0 = time.perf_counter()
for j in range(nq):
    r = B.batch_knn_dot(qs[j], vb, k)
t_single = (time.perf_counter() - t0) / nq
t0 = time.perf_counter()
for j in range(nq):
    s = B.batch_dot(qs[j], vb)
t_scores = (time.perf_counter() - t0) / nq
out = {"workload": f"C1 {n}x{dim} f32, {nq} queries, k={k}", "batch_knn_dot_one_query_per_call_us": t_single * 1e6,
       "batch_dot_one_query_per_call_us": t_scores * 1e6}
for name, eng in (("exact", KNN_EXACT), ("mfma", KNN_MFMA)):
    B.batch_knn_dot_multi(qs, vb, k, engine=eng)
    best = 1e9
    for _ in range(5):
        st = KnnStats()
        t0 = time.perf_counter()
        B.batch_knn_dot_multi(qs, vb, k, engine=eng, stats=st)
        best = min(best, time.perf_counter() - t0)
    out[f"batch_knn_dot_multi_{name}_call_us"] = best * 1e6
    out[f"batch_knn_dot_multi_{name}_device_us"] = st.total_ms * 1e3
print(json.dumps(out))
