"""A/B of library builds on the int8 paths: C2 (f32 corpus, int8 filter, k = 10) and -- with --c3 -- C3 (50M x 768 codes, 1024 queries, k = 100).
    python tools/i8_ab.py [--c3] libdir ...     (default: innr_amd/lib)"""
import os
import subprocess
import sys

code = r'''
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from innr_amd import KNN_MFMA_I8, KNN_EXACT, KnnStats
from innr_amd import batch as B
from innr_amd import scalar as S
q = np.random.default_rng(0xBE7C).uniform(-1, 1, size=(1024, 768)).astype(np.float32)
def run(fn, label):
    best, res = None, None
    for it in range(5):
        st = KnnStats()
        res = fn(st)
        if it and (best is None or st.total_ms < best.total_ms):
            best = st
    print(f"{os.environ.get('INNR_HIP_LIB_PATH', 'default')} {label}: kernel {best.gemm_ms:.3f} ms, call {best.total_ms:.3f} ms, redone {best.queries_fallback}", flush=True)
    return res
vb = B.VerticalBatch.generate(10_000_000, 768, 0)
i1, s1 = run(lambda st: B.batch_knn_dot_multi(q, vb, 10, engine=KNN_MFMA_I8, stats=st), "C2 int8 dot k=10")
i0, s0 = B.batch_knn_dot_multi(q[:16], vb, 10, engine=KNN_EXACT)
print("   first 16 queries identical to the exact engine:", bool(np.array_equal(i0, i1[:16]) and np.array_equal(s0.view(np.uint32), s1[:16].view(np.uint32))), flush=True)
run(lambda st: B.batch_knn_cosine_multi(q, vb, 10, engine=KNN_MFMA_I8, stats=st), "C2 int8 cos k=10")
vb.close()
if os.environ.get("AB_C3"):
    p = S.QuantizationParams.from_range(-1.0, 1.0)
    qc = S.QuantizedCorpus.generate(50_000_000, 768, p, seed=0)
    i1, s1 = run(lambda st: qc.knn_multi(q, 100, engine=KNN_MFMA_I8, stats=st), "C3 u8 k=100")
    i0, s0 = qc.knn_multi(q[:8], 100, engine=KNN_EXACT)
    print("   first 8 queries identical to the exact engine:", bool(np.array_equal(i0, i1[:8]) and np.array_equal(s0.view(np.uint32), s1[:8].view(np.uint32))), flush=True)
'''
args = sys.argv[1:]
c3 = "--c3" in args
libs = [a for a in args if a != "--c3"] or ["innr_amd/lib"]
for d in libs:
    env = dict(os.environ, INNR_HIP_LIB_PATH=os.path.join(os.path.abspath(d), "libinnr_hip.so"))
    if c3:
        env["AB_C3"] = "1"
    subprocess.run([sys.executable, "-c", code], env=env, check=False)
