"""One process = the C2 call on one engine, a few times (the unit rocprofv3 wraps: tools/profile_c2_filters.sh).
    python3 tools/run_c2.py i8|bf16|f32 [k] [metric dot|cos|l2] [queries] [dimension]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from innr_amd import KNN_MFMA, KNN_MFMA_BF16, KNN_MFMA_I8, KnnStats
from innr_amd import batch as B

eng = {"i8": KNN_MFMA_I8, "bf16": KNN_MFMA_BF16, "f32": KNN_MFMA}[sys.argv[1] if len(sys.argv) > 1 else "i8"]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 10
metric = sys.argv[3] if len(sys.argv) > 3 else "dot"
nq = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
dim = int(sys.argv[5]) if len(sys.argv) > 5 else 768
fn = {"dot": B.batch_knn_dot_multi, "cos": B.batch_knn_cosine_multi, "l2": B.batch_knn_multi}[metric]
vb = B.VerticalBatch.generate(10_000_000, dim, 0)
q = np.random.default_rng(0xBE7C).uniform(-1, 1, size=(nq, dim)).astype(np.float32)
best = None
for it in range(5):
    st = KnnStats()
    fn(q, vb, k, engine=eng, stats=st)
    if it and (best is None or st.total_ms < best.total_ms):
        best = st
print(f"C2 {sys.argv[1:]} engine {best.engine}: kernel {abs(best.gemm_ms):.3f} ms, call {best.total_ms:.3f} ms, kept {best.candidates_kept}, redone {best.queries_fallback}")
