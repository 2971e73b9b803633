"""Threshold seeding A/B at the C2 shape: python tools/seed_ab.py 0 1024 2048 4096 8192   (0 = no seeding; INNR_GEMM_SEED_N)"""
import os, subprocess, sys
code = r'''
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from innr_amd import KNN_MFMA_I8, KNN_MFMA_BF16, KnnStats
from innr_amd import batch as B
vb = B.VerticalBatch.generate(10_000_000, 768, 0)
q = np.random.default_rng(0xBE7C).uniform(-1, 1, size=(1024, 768)).astype(np.float32)
for name, eng in (("int8", KNN_MFMA_I8), ("bf16", KNN_MFMA_BF16)):
    best = None
    for it in range(5):
        st = KnnStats()
        B.batch_knn_dot_multi(q, vb, 10, engine=eng, stats=st)
        if it and (best is None or st.total_ms < best.total_ms):
            best = st
    print(f"seed prefix {os.environ.get('INNR_GEMM_SEED_N', 'none' if os.environ.get('INNR_GEMM_NO_SEED') else '2048')} {name}: kernel {best.gemm_ms:.3f} ms, call {best.total_ms:.3f} ms, redone {best.queries_fallback}", flush=True)
'''
for n in sys.argv[1:] or ["2048"]:
    env = dict(os.environ)
    if n == "0":
        env["INNR_GEMM_NO_SEED"] = "1"
    else:
        env["INNR_GEMM_SEED_N"] = n
    subprocess.run([sys.executable, "-c", code], env=env, check=False)
