import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from innr_amd import KNN_MFMA, KnnStats, GEN_EXAMPLE_LCG, _lib
from innr_amd import batch as B
n = 10_000_000
vb = B.VerticalBatch.generate(n, 768, seed=0, generator=GEN_EXAMPLE_LCG)
qb = B.VerticalBatch.generate(64, 768, seed=n, generator=GEN_EXAMPLE_LCG)
q = np.ascontiguousarray(np.asarray(qb.data(), dtype=np.float32).reshape(768, 64).T)
_lib.default_context().set_option("trace", 1)
st = KnnStats()
idx, sc = B.batch_knn_dot_multi(q, vb, 10, engine=KNN_MFMA, stats=st)
print("redone", st.queries_fallback, "total ms", st.total_ms)
print(sc[0], idx[0])
sd = B.batch_dot(q[0], vb)
o = np.sort(sd)[::-1]
print("top scores", o[:12], "count within 0.025 of 10th:", int((sd >= o[9] - 0.025).sum()), "within 0.0025:", int((sd >= o[9] - 0.0025).sum()), "within 2.5e-4:", int((sd >= o[9]-2.5e-4).sum()))
