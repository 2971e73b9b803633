"""Small query batches on a code corpus (C3's corpus: 50M x 768 u8): whole-call device time per engine and batch size.
    python tools/bench_u8_smallq.py [N] [k] > profiles/r03_u8_smallq_50Mx768.txt"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from innr_amd import KNN_AUTO, KNN_EXACT, KNN_MFMA_I8, KnnStats
from innr_amd import scalar as S

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dim = 768
p = S.QuantizationParams.from_range(-1.0, 1.0)
qc = S.QuantizedCorpus.generate(n, dim, p, seed=0)
rng = np.random.default_rng(0xBE7C)
names = {KNN_EXACT: "exact", KNN_MFMA_I8: "int8", KNN_AUTO: "AUTO"}
print(f"# batch_knn_u8 k={k} on {n} x {dim} u8 codes, one MI355X; ms = whole call, device time (best of 3); the corpus stream alone: {n * dim / 6.0e12 * 1e3:.2f} ms at 6 TB/s")
for nq in (1, 2, 4, 8, 16, 32, 64, 100, 128, 256):
    qs = rng.uniform(-1.0, 1.0, size=(nq, dim)).astype(np.float32)
    ref = None
    for engine in (KNN_EXACT, KNN_MFMA_I8, KNN_AUTO):
        if engine == KNN_EXACT and nq > 16:
            continue
        best, ran, kern = 1e9, None, 0.0
        for it in range(3):
            st = KnnStats()
            idx, sc = qc.knn_multi(qs, k, engine=engine, stats=st)
            if st.total_ms < best:
                best, ran, kern = st.total_ms, st.engine, abs(st.gemm_ms)
        if ref is None:
            ref = (idx, sc)
        else:
            assert np.array_equal(idx, ref[0]) and np.array_equal(sc.view(np.uint32), ref[1].view(np.uint32))
        print(f"  {nq:4d} {names[engine]:>6} {best:9.3f} ms  (kernel {kern:7.3f})  {'ran ' + names.get(ran, str(ran)) if engine == KNN_AUTO else ''}", flush=True)
