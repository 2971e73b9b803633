"""Where does the one-limb int8 filter kernel's time go? C2 shape, INNR_I8H_PROBE bits: 1 = the epilogue never visits (what the K-loop + fast reject cost; the call fills its
stats and then fails: a timing run hands out no results), 8 / 16 (with 1) = the K-loop with its query fragments served from L1 / its corpus stages from L2, 32 = no epilogue at all, 4 = count visiting wave epilogues / survivors / appends (slows the kernel):
    python tools/i8h_probe.py
"""
import os
import subprocess
import sys

code = r'''
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from innr_amd import KNN_MFMA_I8, KnnStats
from innr_amd import batch as B
vb = B.VerticalBatch.generate(10_000_000, 768, 0)
q = np.random.default_rng(0xBE7C).uniform(-1, 1, size=(1024, 768)).astype(np.float32)
best = None
for it in range(4):
    st = KnnStats()
    try:
        B.batch_knn_dot_multi(q, vb, 10, engine=KNN_MFMA_I8, stats=st)
    except Exception:  # bit 1: a timing run; the library fills the stats and refuses to return results
        pass
    if it and (best is None or st.gemm_ms < best.gemm_ms):
        best = st
print(f"INNR_I8H_PROBE={os.environ.get('INNR_I8H_PROBE', '0')}: kernel {best.gemm_ms:.3f} ms, call {best.total_ms:.3f} ms, redone {best.queries_fallback}")
'''
for bits in ("0", "1", "9", "17", "25", "33", "57", "4"):
    env = dict(os.environ, INNR_I8H_PROBE=bits)
    subprocess.run([sys.executable, "-c", code], env=env, check=False)
