"""CPU ORACLE loader -- test infrastructure, NOT product code.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product path (innr_amd/) never does; innr_amd fails loudly when its HIP library is missing.

numpy/ctypes bindings over oracle/innr_oracle.c (a plain-C restatement of innr's portable CPU path;
each C function cites the reference file:line it follows).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libinnr_oracle.so")

_f32p = C.POINTER(C.c_float)
_u64p = C.POINTER(C.c_uint64)
_u32p = C.POINTER(C.c_uint32)
_u8p = C.POINTER(C.c_uint8)
_sz = C.c_size_t


class QParams(C.Structure):
    _fields_ = [("alpha", C.c_float), ("offset", C.c_float)]


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (oracle/Makefile). Building the checker is not using it."""
    src = os.path.join(_HERE, "innr_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib: Optional[C.CDLL] = None
_NATIVE_SO = os.path.join(_HERE, "_build", "libinnr_oracle_native.so")


def build_native() -> Optional[str]:
    """The oracle compiled -march=native ON THIS HOST (bench.py's cpu_baseline leg). Returns the path, or
    None if it cannot be built here (no gcc / make). Call it before the process initialises the GPU: it starts a child."""
    try:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "native"])  # -B: always for THIS host, never a copy from another
        return _NATIVE_SO
    except Exception:
        return None


def native_knn_dot():
    """batch_knn_dot of the -march=native build (None if build_native() has not produced it)"""
    if not os.path.exists(_NATIVE_SO):
        return None
    L = C.CDLL(_NATIVE_SO)
    f = L.orc_batch_knn_dot
    f.restype = _sz
    f.argtypes = [_f32p, _f32p, _sz, _sz, _sz, _u64p, _f32p]
    return lambda q, data, k: _knn(f, q, data, k)


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_SO)

    def sig(name, res, *args):
        f = getattr(L, name)
        f.restype = res
        f.argtypes = list(args)

    sig("orc_total_key", C.c_int32, C.c_float)
    sig("orc_vb_from_flat", None, _f32p, _sz, _sz, _f32p)
    sig("orc_vb_extract_vector", None, _f32p, _sz, _sz, _sz, _f32p)
    for n in ("orc_batch_dot", "orc_batch_l2_squared"):
        sig(n, None, _f32p, _f32p, _sz, _sz, _f32p)
    sig("orc_batch_norms", None, _f32p, _sz, _sz, _f32p)
    sig("orc_batch_cosine", None, _f32p, _f32p, _sz, _sz, _f32p, _f32p)
    for n in ("orc_batch_knn_dot", "orc_batch_knn_cosine", "orc_batch_knn", "orc_batch_knn_reordered"):
        sig(n, _sz, _f32p, _f32p, _sz, _sz, _sz, _u64p, _f32p)
    sig("orc_batch_knn_filtered", _sz, _f32p, _f32p, _sz, _sz, _sz, _u8p, _u64p, _f32p)
    sig("orc_batch_l2_squared_pruning", _sz, _f32p, _f32p, _sz, _sz, C.c_float, _u64p, _f32p)
    sig("orc_batch_knn_adaptive", _sz, _f32p, _f32p, _sz, _sz, _sz, _sz, _u64p, _f32p)
    sig("orc_batch_dimension_variance", None, _f32p, _sz, _sz, _f32p)
    sig("orc_topk_new", C.c_void_p, _sz)
    sig("orc_topk_free", None, C.c_void_p)
    sig("orc_topk_threshold", C.c_float, C.c_void_p)
    sig("orc_topk_insert", None, C.c_void_p, C.c_uint32, C.c_float)
    sig("orc_topk_len", _sz, C.c_void_p)
    sig("orc_topk_into_sorted", _sz, C.c_void_p, _u32p, _f32p)
    for n in ("orc_dot_portable", "orc_cosine_portable", "orc_l2_distance_squared_portable",
              "orc_l1_distance_portable", "orc_dist_cosine", "orc_dist_dot", "orc_dist_l2", "orc_dist_l1"):
        sig(n, C.c_float, _f32p, _f32p, _sz)
    for n in ("orc_maxsim", "orc_maxsim_cosine"):
        sig(n, C.c_float, _f32p, _sz, _f32p, _sz, _sz)
    sig("orc_qparams_from_range", QParams, C.c_float, C.c_float)
    sig("orc_qparams_fit", QParams, _f32p, _sz)
    sig("orc_qparams_fit_quantile", QParams, _f32p, _sz, C.c_float)
    sig("orc_quantize_u8", None, _f32p, _sz, QParams, _u8p)
    sig("orc_query_sum", C.c_float, _f32p, _sz)
    sig("orc_mixed_dot_u8_f32", C.c_float, _f32p, _u8p, _sz)
    sig("orc_asymmetric_dot_u8", C.c_float, _f32p, _u8p, _sz, QParams)
    sig("orc_batch_knn_u8", _sz, _f32p, _u8p, _sz, _sz, QParams, _sz, _u64p, _f32p)
    sig("orc_generate_embedding", None, _sz, C.c_uint64, _f32p)
    sig("orc_generate_normalized", None, _sz, C.c_uint64, _f32p)
    sig("orc_generate_rows", None, _sz, _sz, C.c_uint64, C.c_int, _f32p)
    sig("orc_generate_uniform_rows", None, _sz, _sz, C.c_uint64, C.c_uint64, _f32p)
    _lib = L
    return L


# ---------------------------------------------------------------------------------------------
# numpy helpers
# ---------------------------------------------------------------------------------------------
def _f(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a: np.ndarray, t=_f32p):
    return a.ctypes.data_as(t)


def total_key(x: float) -> int:
    return int(lib().orc_total_key(C.c_float(x)))


def from_flat(rows, n: int, dim: int) -> np.ndarray:
    """VerticalBatch::from_flat -> dimension-major data[d*N+i] as a (dim, n) array."""
    rows = _f(rows).reshape(-1)
    assert rows.size == n * dim
    out = np.empty((dim, n), dtype=np.float32)
    lib().orc_vb_from_flat(_p(rows), n, dim, _p(out))
    return out


def from_rows(vectors) -> np.ndarray:
    v = _f(vectors)
    if v.size == 0 and v.ndim < 2:
        return np.empty((0, 0), dtype=np.float32)
    n, dim = v.shape
    return from_flat(v, n, dim)


def _nd(data: np.ndarray) -> Tuple[int, int]:
    dim, n = data.shape
    return n, dim


def batch_dot(q, data) -> np.ndarray:
    data = _f(data); q = _f(q); n, dim = _nd(data)
    assert q.size == dim, "assert_eq!(query.len(), batch.dimension)"
    out = np.empty(n, dtype=np.float32)
    lib().orc_batch_dot(_p(q), _p(data), n, dim, _p(out))
    return out


def batch_l2_squared(q, data) -> np.ndarray:
    data = _f(data); q = _f(q); n, dim = _nd(data)
    assert q.size == dim
    out = np.empty(n, dtype=np.float32)
    lib().orc_batch_l2_squared(_p(q), _p(data), n, dim, _p(out))
    return out


def batch_norms(data) -> np.ndarray:
    data = _f(data); n, dim = _nd(data)
    out = np.empty(n, dtype=np.float32)
    lib().orc_batch_norms(_p(data), n, dim, _p(out))
    return out


def batch_cosine(q, data, norms) -> np.ndarray:
    data = _f(data); q = _f(q); norms = _f(norms); n, dim = _nd(data)
    assert norms.size == n and q.size == dim
    out = np.empty(n, dtype=np.float32)
    lib().orc_batch_cosine(_p(q), _p(data), n, dim, _p(norms), _p(out))
    return out


def _knn(fn, q, data, k, *extra):
    data = _f(data); q = _f(q); n, dim = _nd(data)
    assert q.size == dim
    cap = max(1, min(k, n) if n else 1)
    idx = np.empty(cap, dtype=np.uint64)
    sc = np.empty(cap, dtype=np.float32)
    r = fn(_p(q), _p(data), n, dim, k, *extra, _p(idx, _u64p), _p(sc))
    return idx[:r].copy(), sc[:r].copy()


def batch_knn_dot(q, data, k):
    return _knn(lib().orc_batch_knn_dot, q, data, k)


def batch_knn_cosine(q, data, k):
    return _knn(lib().orc_batch_knn_cosine, q, data, k)


def batch_knn(q, data, k):
    return _knn(lib().orc_batch_knn, q, data, k)


def batch_knn_reordered(q, data, k):
    return _knn(lib().orc_batch_knn_reordered, q, data, k)


def batch_knn_filtered(q, data, k, mask):
    mask = np.ascontiguousarray(mask, dtype=np.uint8)
    return _knn(lib().orc_batch_knn_filtered, q, data, k, _p(mask, _u8p))


def batch_knn_adaptive(q, data, k, warmup_dims):
    assert warmup_dims > 0, "warmup_dims must be > 0"
    data = _f(data); q = _f(q); n, dim = _nd(data)
    cap = max(1, min(k, n) if n else 1)
    idx = np.empty(cap, dtype=np.uint64); sc = np.empty(cap, dtype=np.float32)
    r = lib().orc_batch_knn_adaptive(_p(q), _p(data), n, dim, k, warmup_dims, _p(idx, _u64p), _p(sc))
    return idx[:r].copy(), sc[:r].copy()


def batch_l2_squared_pruning(q, data, threshold):
    data = _f(data); q = _f(q); n, dim = _nd(data)
    idx = np.empty(max(n, 1), dtype=np.uint64); ds = np.empty(max(n, 1), dtype=np.float32)
    r = lib().orc_batch_l2_squared_pruning(_p(q), _p(data), n, dim, C.c_float(threshold), _p(idx, _u64p), _p(ds))
    return idx[:r].copy(), ds[:r].copy()


def batch_dimension_variance(data) -> np.ndarray:
    data = _f(data); n, dim = _nd(data)
    out = np.empty(max(dim, 1), dtype=np.float32)
    lib().orc_batch_dimension_variance(_p(data), n, dim, _p(out))
    return out[:dim]


class TopK:
    """topk::TopK (topk.rs:47-187)."""

    def __init__(self, k: int):
        assert k > 0, "innr::TopK: k must be >= 1"
        self.k = k
        self._h = lib().orc_topk_new(k)

    def threshold(self) -> float:
        return float(lib().orc_topk_threshold(self._h))

    def insert(self, id_: int, distance: float) -> None:
        lib().orc_topk_insert(self._h, id_, C.c_float(distance))

    def __len__(self) -> int:
        return int(lib().orc_topk_len(self._h))

    def is_empty(self) -> bool:
        return len(self) == 0

    def into_sorted(self):
        ids = np.empty(self.k, dtype=np.uint32); ds = np.empty(self.k, dtype=np.float32)
        r = lib().orc_topk_into_sorted(self._h, _p(ids, _u32p), _p(ds))
        return [(int(ids[i]), float(ds[i])) for i in range(r)]

    def __del__(self):
        try:
            lib().orc_topk_free(self._h)
        except Exception:
            pass


def _pair(fn):
    def f(a, b) -> float:
        a = _f(a); b = _f(b)
        assert a.size == b.size, "slice length mismatch"
        return float(fn(_p(a), _p(b), a.size))
    return f


def dot_portable(a, b): return _pair(lib().orc_dot_portable)(a, b)
def cosine_portable(a, b): return _pair(lib().orc_cosine_portable)(a, b)
def l2_distance_squared_portable(a, b): return _pair(lib().orc_l2_distance_squared_portable)(a, b)
def l1_distance_portable(a, b): return _pair(lib().orc_l1_distance_portable)(a, b)
def dist_cosine(a, b): return _pair(lib().orc_dist_cosine)(a, b)
def dist_dot(a, b): return _pair(lib().orc_dist_dot)(a, b)
def dist_l2(a, b): return _pair(lib().orc_dist_l2)(a, b)
def dist_l1(a, b): return _pair(lib().orc_dist_l1)(a, b)


def _matryoshka(fn, a, b, prefix_len):
    a = np.ascontiguousarray(a, dtype=np.float32).reshape(-1)
    b = np.ascontiguousarray(b, dtype=np.float32).reshape(-1)
    fn.restype = C.c_float
    fn.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t]
    return float(fn(a.ctypes.data, a.size, b.ctypes.data, b.size, int(prefix_len)))


def matryoshka_dot(a, b, prefix_len): return _matryoshka(lib().orc_matryoshka_dot, a, b, prefix_len)
def matryoshka_cosine(a, b, prefix_len): return _matryoshka(lib().orc_matryoshka_cosine, a, b, prefix_len)


def dist_hamming(a, b) -> float:
    x = np.ascontiguousarray(a, dtype=np.uint8).reshape(-1); y = np.ascontiguousarray(b, dtype=np.uint8).reshape(-1)
    assert x.size == y.size
    f = lib().orc_dist_hamming
    f.restype, f.argtypes = C.c_float, [C.c_void_p, C.c_void_p, C.c_size_t]
    return float(f(x.ctypes.data, y.ctypes.data, x.size))


def dist_slot_u32(a, b) -> float:
    x = np.ascontiguousarray(a, dtype=np.uint32).reshape(-1); y = np.ascontiguousarray(b, dtype=np.uint32).reshape(-1)
    assert x.size == y.size
    f = lib().orc_dist_slot_u32
    f.restype, f.argtypes = C.c_float, [C.c_void_p, C.c_void_p, C.c_size_t]
    return float(f(x.ctypes.data, y.ctypes.data, x.size))


def _tok(t):
    t = _f(t)
    if t.size == 0:
        return t.reshape(0, 0)
    assert t.ndim == 2
    return t


def maxsim(q, d, cosine: bool = False) -> float:
    q = _tok(q); d = _tok(d)
    if q.shape[0] == 0 or d.shape[0] == 0:
        return 0.0
    assert q.shape[1] == d.shape[1], "dimension mismatch (doc)"
    fn = lib().orc_maxsim_cosine if cosine else lib().orc_maxsim
    return float(fn(_p(q), q.shape[0], _p(d), d.shape[0], q.shape[1]))


def maxsim_cosine(q, d) -> float:
    return maxsim(q, d, cosine=True)


def qparams_from_range(mn: float, mx: float) -> QParams:
    return lib().orc_qparams_from_range(C.c_float(mn), C.c_float(mx))


def qparams_fit(values) -> QParams:
    v = _f(values).reshape(-1)
    return lib().orc_qparams_fit(_p(v), v.size)


def qparams_fit_quantile(values, quantile: float) -> QParams:
    assert 0.0 < quantile <= 1.0, "quantile must be in (0.0, 1.0]"
    v = _f(values).reshape(-1)
    return lib().orc_qparams_fit_quantile(_p(v), v.size, C.c_float(quantile))


def quantize_u8(values, p: QParams) -> np.ndarray:
    v = _f(values)
    out = np.empty(v.shape, dtype=np.uint8)
    lib().orc_quantize_u8(_p(v), v.size, p, _p(out, _u8p))
    return out


def query_sum(q) -> float:
    q = _f(q)
    return float(lib().orc_query_sum(_p(q), q.size))


def mixed_dot_u8_f32(a, b) -> float:
    a = _f(a); b = np.ascontiguousarray(b, dtype=np.uint8)
    assert a.size == b.size, "mixed_dot_u8_f32: slice length mismatch"
    return float(lib().orc_mixed_dot_u8_f32(_p(a), _p(b, _u8p), a.size))


def asymmetric_dot_u8(q, codes, p: QParams) -> float:
    q = _f(q); codes = np.ascontiguousarray(codes, dtype=np.uint8)
    assert q.size == codes.size, "asymmetric_dot_u8: dimension mismatch"
    return float(lib().orc_asymmetric_dot_u8(_p(q), _p(codes, _u8p), q.size, p))


def batch_knn_u8(q, codes, p: QParams, k: int):
    """codes: (n, dim) uint8 packed rows."""
    q = _f(q); codes = np.ascontiguousarray(codes, dtype=np.uint8)
    if codes.size == 0 or k == 0:
        return np.empty(0, np.uint64), np.empty(0, np.float32)
    n, dim = codes.shape
    assert q.size == dim, "asymmetric_dot_u8_precomputed: dimension mismatch"
    cap = min(k, n)
    idx = np.empty(cap, dtype=np.uint64); sc = np.empty(cap, dtype=np.float32)
    r = lib().orc_batch_knn_u8(_p(q), _p(codes, _u8p), n, dim, p, k, _p(idx, _u64p), _p(sc))
    return idx[:r].copy(), sc[:r].copy()


def generate_embedding(dim: int, seed: int) -> np.ndarray:
    out = np.empty(dim, dtype=np.float32)
    lib().orc_generate_embedding(dim, C.c_uint64(seed & 0xFFFFFFFFFFFFFFFF), _p(out))
    return out


def generate_normalized(dim: int, seed: int) -> np.ndarray:
    out = np.empty(dim, dtype=np.float32)
    lib().orc_generate_normalized(dim, C.c_uint64(seed & 0xFFFFFFFFFFFFFFFF), _p(out))
    return out


def generate_corpus(n: int, dim: int, seed0: int = 0, normalized: bool = False) -> np.ndarray:
    """Row-major (n, dim) corpus, row i = generate_embedding(dim, seed0 + i) (batch_demo.rs:167)."""
    out = np.empty((n, dim), dtype=np.float32)
    lib().orc_generate_rows(n, dim, C.c_uint64(seed0 & 0xFFFFFFFFFFFFFFFF), 1 if normalized else 0, _p(out))
    return out


def generate_uniform(n: int, dim: int, seed: int = 0, row0: int = 0) -> np.ndarray:
    """Row-major (n, dim) i.i.d. uniform[-1,1) rows row0..row0+n of stream `seed` (bit-identical to the
    device generator INNR_GEN_UNIFORM; distribution of the reference's criterion benches)."""
    out = np.empty((n, dim), dtype=np.float32)
    lib().orc_generate_uniform_rows(n, dim, C.c_uint64(seed & 0xFFFFFFFFFFFFFFFF), C.c_uint64(row0), _p(out))
    return out
