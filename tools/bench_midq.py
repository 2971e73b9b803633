"""Mid-size query batches at the C2 corpus (10M x 768 f32): device time of one kNN call per engine and batch size.

    python tools/bench_midq.py [N] [D] > profiles/r02_midq_10Mx768.txt

Per point: ms of the whole call (HIP events, best of 3), the corpus stream it implies (4*N*D bytes per corpus pass / t: the
HBM-bound figure for small Q) and the f32 MFMA rate (2*Q*N*D / t: the MFMA-bound figure for the GEMM engine), and which
engine INNR_KNN_AUTO picks. Metrics: dot (cosine / L2 share the kernels; one point each at Q = 64)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from innr_amd import KNN_AUTO, KNN_EXACT, KNN_MFMA, KNN_MFMA_BF16, KNN_MFMA_I8, KnnStats
from innr_amd import batch as B

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 768
vb = B.VerticalBatch.generate(n, dim, 0)
rng = np.random.default_rng(0)
names = {KNN_EXACT: "exact", KNN_MFMA: "gemm", KNN_MFMA_I8: "int8", KNN_MFMA_BF16: "bf16", KNN_AUTO: "AUTO"}
print(f"# kNN k=10 on {n} x {dim} f32 (uniform), one MI355X; ms = whole innr_batch_knn call, device time")
print(f"# {'Q':>4} {'engine':>6} {'ms':>9} {'GB/s (4ND per pass)':>20} {'TFLOP/s (2QND)':>15} {'Mvec/s':>10}  note")
qlist = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 2, 3, 4, 5, 8, 12, 16, 32, 64, 100, 128, 200, 256, 384, 512, 768, 1024]
for nq in qlist:
    q = rng.uniform(-1, 1, size=(nq, dim)).astype(np.float32)
    st = KnnStats()
    B.batch_knn_dot_multi(q, vb, 10, engine=KNN_AUTO, stats=st)
    auto = st.engine
    for engine in (KNN_EXACT, KNN_MFMA, KNN_MFMA_I8, KNN_MFMA_BF16, KNN_AUTO):
        if (engine == KNN_EXACT and nq > 64) or (engine == KNN_MFMA_BF16 and nq < 8) or (engine == KNN_MFMA and nq > 512):
            continue
        best, redo = 1e9, 0
        for it in range(3):
            st = KnnStats()
            B.batch_knn_dot_multi(q, vb, 10, engine=engine, stats=st)
            if st.total_ms < best:
                best, redo = st.total_ms, st.queries_fallback
        passes = (nq + 7) // 8 if engine == KNN_EXACT else 1
        # floor of the call: the corpus stream of the engine's operand (4 / 1 / 2 bytes per value at ~6 TB/s achievable) or its
        # matrix pipe (f32 157 T, int8 5.03 P, bf16 2.5 P), whichever is larger
        bpe, peak = {KNN_MFMA: (4, 157.3e12), KNN_MFMA_I8: (1, 5.03e15), KNN_MFMA_BF16: (2, 2.516e15)}.get(engine if engine != KNN_AUTO else auto, (4, 157.3e12))
        floor = max(bpe * n * dim * passes / 6.0e12, 2.0 * nq * n * dim / peak if engine != KNN_EXACT and auto != KNN_EXACT or engine not in (KNN_EXACT, KNN_AUTO) else 0.0) * 1e3
        print(f"  {nq:4d} {names[engine]:>6} {best:9.3f} {4.0 * n * dim * passes / best / 1e6:20.1f} {2.0 * nq * n * dim / best / 1e9:15.2f} "
              f"{nq * n / best / 1e3:10.1f}  floor {floor:6.2f} ms x{best / floor:5.2f}  {'(AUTO ran ' + names.get(auto, str(auto)) + ')' if engine == KNN_AUTO else ''}"
              f"{' redone ' + str(redo) if redo else ''}", flush=True)
q = rng.uniform(-1, 1, size=(64, dim)).astype(np.float32)
for name, fn in (("cosine", B.batch_knn_cosine_multi), ("l2", B.batch_knn_multi)):
    best = 1e9
    for it in range(3):
        st = KnnStats()
        fn(q, vb, 10, engine=KNN_MFMA, stats=st)
        best = min(best, st.total_ms)
    print(f"  {64:4d} {'gemm':>6} {best:9.3f} {4.0 * n * dim / best / 1e6:20.1f} {2.0 * 64 * n * dim / best / 1e9:15.2f} {64 * n / best / 1e3:10.1f}  {name}")
