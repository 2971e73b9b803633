"""Diagnostic: inspect the GEMM engine's candidate selection vs the oracle (why does a margin proof fail?)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from innr_amd import batch as B, KnnStats, KNN_MFMA, _lib

n, dim, nq, k = [int(x) for x in (sys.argv[1:5] + [None] * 4)[:4]] if len(sys.argv) >= 5 else (10000, 128, 100, 10)
rows = oracle.generate_corpus(n, dim, 0); data = oracle.from_rows(rows)
qs = np.stack([oracle.generate_embedding(dim, 50_000 + j) for j in range(nq)])
vb = B.VerticalBatch.from_rows(rows)
st = KnnStats()
idx, sc = B.batch_knn_dot_multi(qs, vb, k, engine=KNN_MFMA, stats=st)
print("fallback", st.queries_fallback, "kept", st.candidates_kept, "gemm_ms", st.gemm_ms)
KP = st.candidates_kept
import ctypes as _C, os as _os
L = _C.CDLL(_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "innr_amd", "lib", "libinnr_hip_testhooks.so"))  # make -C innr_amd/csrc hooks
sel = np.zeros((nq, KP), np.uint64); cnt = np.zeros(nq, np.uint32); qn = np.zeros(nq, np.float32); info = np.zeros(4, np.float32)
fn = L.innrdbg_last_selection; fn.restype = C.c_int
fn.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
_lib.check(fn(vb._h, nq, KP, sel.ctypes.data, cnt.ctypes.data, qn.ctypes.data, info.ctypes.data))
def unord(o):
    o = np.uint32(o); b = (o & np.uint32(0x7fffffff)) if (o & np.uint32(0x80000000)) else ~o
    return np.array([b], np.uint32).view(np.float32)[0]
for q in (nq - 1, nq - 2):
    pref = (sel[q] >> np.uint64(32)).astype(np.uint32); ids = (~sel[q].astype(np.uint32))
    ids = (~(sel[q] & np.uint64(0xFFFFFFFF)).astype(np.uint32)).astype(np.uint32)
    appr = np.array([unord(p) for p in pref])
    exact = oracle.batch_dot(qs[q], data)
    print("q", q, "cnt", cnt[q], "qn", qn[q], "max_norm", info[0])
    print(" ids   ", ids[:KP].tolist())
    print(" approx", np.round(appr[:KP], 4).tolist())
    print(" exact ", np.round(exact[ids[:KP] % n], 4).tolist())
    order = np.argsort(-exact, kind="stable")[:KP]
    print(" oracle top ids", order.tolist())
    print(" oracle top sc ", np.round(exact[order], 4).tolist())
