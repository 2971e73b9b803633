"""A/B of library builds (INNR_HIP_LIB_PATH) on the low-precision filters at the C2 shape: python tools/lib_ab.py lib1.so lib2.so ..."""
import os
import subprocess
import sys

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
    print(f"{os.path.basename(os.environ.get('INNR_HIP_LIB_PATH', 'default'))} {name}: kernel {best.gemm_ms:.3f} ms, call {best.total_ms:.3f} ms, redone {best.queries_fallback}", flush=True)
'''
for lib in ["default"] + sys.argv[1:]:
    env = dict(os.environ)
    if lib != "default":
        env["INNR_HIP_LIB_PATH"] = os.path.abspath(lib)
    subprocess.run([sys.executable, "-c", code], env=env, check=False)
