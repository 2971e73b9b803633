"""Where does the one-limb int8 filter kernel's time go? C2 shape, library builds with -DINNR_I8H_PROBE=<bits> (compile-time: the
product library has no probe code) -- 1: the epilogue never visits (what the K-loop + fast reject cost; the call fills its stats
and then fails: a timing run hands out no results), 8 / 16 (with 1): the K-loop with its query fragments served from L1 / its
corpus stages from L2, 32: no epilogue at all, 4: count visiting wave epilogues / survivors / bound re-derivations (slows the kernel).
Build the variants on the build box, then run on the GPU box:
    for b in 1 4 33; do make -C innr_amd/csrc LIBDIR=../lib_probe$b EXTRA=-DINNR_I8H_PROBE=$b all; done
    python tools/i8h_probe.py [lib dirs ...]          (default: innr_amd/lib and every innr_amd/lib_probe*)
INNR_PROBE_METRIC=dot|cos|l2 picks the call (squared L2: the augmented copy, 14 K-steps at C2 instead of 12).
"""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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
        fn = {"dot": B.batch_knn_dot_multi, "cos": B.batch_knn_cosine_multi, "l2": B.batch_knn_multi}[os.environ.get("INNR_PROBE_METRIC", "dot")]
        fn(q, vb, 10, engine=KNN_MFMA_I8, stats=st)
    except Exception:  # a timing build: the library fills the stats and refuses to return results
        pass
    if it and (best is None or st.gemm_ms < best.gemm_ms):
        best = st
print(f"{os.environ.get('INNR_HIP_LIB_PATH', 'default')}: kernel {best.gemm_ms:.3f} ms, call {best.total_ms:.3f} ms, redone {best.queries_fallback}", flush=True)
'''
libs = sys.argv[1:] or ([os.path.join(ROOT, "innr_amd", "lib")] + sorted(glob.glob(os.path.join(ROOT, "innr_amd", "lib_probe*"))))
for d in libs:
    env = dict(os.environ, INNR_HIP_LIB_PATH=os.path.join(os.path.abspath(d), "libinnr_hip.so"), INNR_NO_COMPLETION="1")  # (kernel ms = the first pass alone)
    subprocess.run([sys.executable, "-c", code], env=env, check=False)
