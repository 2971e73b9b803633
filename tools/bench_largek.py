"""k beyond the candidate lists (k > INNR_MAX_K): all scores + radix select + sort of the best k (sort_full.hip).
    python tools/bench_largek.py      10M x 128, 2 queries, k = 100 000 / 1000 / 10M (rank the whole corpus)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from innr_amd import KNN_AUTO, KnnStats
from innr_amd import batch as B
n, dim = 10_000_000, 128
vb = B.VerticalBatch.generate(n, dim, 0)
q = np.random.default_rng(5).uniform(-1, 1, size=(2, dim)).astype(np.float32)
for k in (1000, 100_000, n):
    best = None
    for it in range(3):
        st = KnnStats()
        idx, sc = B.batch_knn_dot_multi(q, vb, k, engine=KNN_AUTO, stats=st)
        if it and (best is None or st.total_ms < best):
            best = st.total_ms
    ok = bool(np.all(np.diff(sc[0].astype(np.float64)) <= 0)) and len(set(idx[0][:1000].tolist())) == min(k, 1000)
    print(f"10M x 128, 2 queries, k = {k}: {best:.2f} ms per call (device), sorted descending and distinct: {ok}", flush=True)
