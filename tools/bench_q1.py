"""The memory-bound single-query scans (the reference's own call shapes: one query per call) at the full corpus sizes:
batch_knn_dot / batch_knn_cosine / batch_knn on 10M x 768 f32, batch_knn_u8 on 50M x 768 codes (second argument), maxsim
scores of 1M documents x 64 tokens x 128 (third). Device time of the whole call and the corpus bytes it implies per second.

    python tools/bench_q1.py [f32 | u8 | maxsim]      (tools/profile_q1.sh runs it under rocprofv3 for the PMC byte counts)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from innr_amd import KNN_EXACT, KnnStats

which = sys.argv[1] if len(sys.argv) > 1 else "f32"
rng = np.random.default_rng(7)
if which == "f32":
    from innr_amd import batch as B
    n, dim = 10_000_000, 768
    vb = B.VerticalBatch.generate(n, dim, 0)
    q = rng.uniform(-1, 1, size=(1, dim)).astype(np.float32)
    for name, fn in (("batch_knn_dot", B.batch_knn_dot_multi), ("batch_knn_cosine", B.batch_knn_cosine_multi), ("batch_knn (L2)", B.batch_knn_multi)):
        best = 1e9
        for _ in range(5):
            st = KnnStats()
            fn(q, vb, 10, engine=KNN_EXACT, stats=st)
            best = min(best, st.total_ms)
        print(f"{name:18s} 1 query x {n} x {dim} f32: {best:7.3f} ms -> {4.0 * n * dim / best / 1e6:7.1f} GB/s of corpus (4ND bytes)", flush=True)
elif which == "u8":
    from innr_amd import scalar as S
    n, dim = 50_000_000, 768
    qc = S.QuantizedCorpus.generate(n, dim, S.QuantizationParams.from_range(-1.0, 1.0), seed=0)
    q = rng.uniform(-1, 1, size=(1, dim)).astype(np.float32)
    best = 1e9
    for _ in range(5):
        st = KnnStats()
        qc.knn_multi(q, 100, engine=KNN_EXACT, stats=st)
        best = min(best, st.total_ms)
    print(f"batch_knn_u8       1 query x {n} x {dim} u8:  {best:7.3f} ms -> {1.0 * n * dim / best / 1e6:7.1f} GB/s of corpus (ND bytes)", flush=True)
else:
    from innr_amd import maxsim as M
    docs, T, dim, Tq = 1_000_000, 64, 128, 32
    dc = M.DocumentCorpus.generate(docs, T, dim, seed=0)
    qt = rng.uniform(-1, 1, size=(Tq, dim)).astype(np.float32)
    best = 1e9
    for _ in range(5):
        st = KnnStats()
        dc.topk(qt, 100, stats=st)
        best = min(best, st.total_ms)
    print(f"maxsim top-100     {Tq}-token query x {docs} docs x {T} x {dim} f32: {best:7.3f} ms -> {4.0 * docs * T * dim / best / 1e6:7.1f} GB/s of corpus", flush=True)
