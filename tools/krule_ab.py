"""A/B of the chip-wide bound's k rule (topk_dev.h) against the KP rule alone, C2 shape (10M x 768, 1024 queries), every filter engine:
    python tools/krule_ab.py [k ...]        (default k = 10)
Each arm is a process of its own: the context reads INNR_NO_K_RULE once, at creation."""
import os
import subprocess
import sys

code = r'''
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from innr_amd import KNN_MFMA, KNN_MFMA_I8, KNN_MFMA_BF16, KnnStats
from innr_amd import batch as B
vb = B.VerticalBatch.generate(10_000_000, 768, 0)
q = np.random.default_rng(0xBE7C).uniform(-1, 1, size=(1024, 768)).astype(np.float32)
k = int(os.environ["AB_K"])
ref = None
for name, eng, fn in (("int8 dot", KNN_MFMA_I8, B.batch_knn_dot_multi), ("bf16 dot", KNN_MFMA_BF16, B.batch_knn_dot_multi),
                      ("int8 cos", KNN_MFMA_I8, B.batch_knn_cosine_multi), ("bf16 l2", KNN_MFMA_BF16, B.batch_knn_multi),
                      ("f32 dot", KNN_MFMA, B.batch_knn_dot_multi)):
    best = None
    for it in range(4 if eng != KNN_MFMA else 2):
        st = KnnStats()
        idx, sc = fn(q, vb, k, engine=eng, stats=st)
        if it and (best is None or st.total_ms < best.total_ms):
            best = st
    tag = ""
    if name.endswith("dot"):
        if ref is None:
            ref = (idx.copy(), sc.copy())
        else:
            tag = " same_as_int8" if (np.array_equal(idx, ref[0]) and np.array_equal(sc.view(np.uint32), ref[1].view(np.uint32))) else " DIFFERENT"
    print(f"k={k} no_k_rule={os.environ.get('INNR_NO_K_RULE', '0')} {name}: engine {best.engine} kernel {abs(best.gemm_ms):.3f} ms, call {best.total_ms:.3f} ms, "
          f"kept {best.candidates_kept}, redone {best.queries_fallback}{tag}", flush=True)
'''
ks = sys.argv[1:] or ["10"]
for k in ks:
    for arm in ("0", "1"):
        env = dict(os.environ, INNR_NO_K_RULE=arm, AB_K=k)
        subprocess.run([sys.executable, "-c", code], env=env, check=False)
