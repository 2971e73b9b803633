"""Host-side mirror of innr's `scalar` module (reference: src/scalar.rs) over the HIP C ABI.

QuantizationParams (:44-163), QuantizedU8 (:171-208), quantize_u8 (:212-225), query_context (:236-240),
asymmetric_dot_u8[_precomputed] (:261-300), mixed_dot_u8_f32 (:314-358) and batch_knn_u8 (:370-393).
Fitting and quantising are one-time ingest steps and stay on the host; the pairwise dots are
host functions of libinnr_hip.so in the portable order; the corpus scan `batch_knn_u8` runs on the GPU.
Addition: `QuantizedCorpus` keeps the packed codes resident on the device across calls (the reference re-walks
N separate Vec<u8> allocations per call).
"""
from __future__ import annotations

import ctypes as C
import weakref
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import KNN_AUTO, InnrPanic, KnnStats, check, default_context, load

F = np.float32


def _vp(a: np.ndarray) -> C.c_void_p:
    return C.c_void_p(a.ctypes.data)


@dataclass
class QuantizationParams:
    """scalar.rs:44-49: alpha = range (max - min), offset = min."""
    alpha: float
    offset: float

    @staticmethod
    def from_range(mn: float, mx: float) -> "QuantizationParams":  # scalar.rs:54-60
        with np.errstate(over="ignore", invalid="ignore"):  # f32::MIN - f32::MAX = -inf, inf - inf = NaN: both "not > 0"
            alpha = F(mx) - F(mn)
        return QuantizationParams(float(alpha) if alpha > 0 else 1.0, float(F(mn)))

    @staticmethod
    def fit(values) -> "QuantizationParams":  # scalar.rs:68-87
        v = np.asarray(values, dtype=np.float32).reshape(-1)
        if v.size == 0:
            return QuantizationParams(1.0, 0.0)
        mn, mx = F(3.4028235e38), F(-3.4028235e38)
        # `if v < min` / `if v > max` skip NaNs exactly like the reference's comparisons
        with np.errstate(invalid="ignore"):
            fin = v[~np.isnan(v)]
        if fin.size:
            mn, mx = min(mn, fin.min()), max(mx, fin.max())
        return QuantizationParams.from_range(float(mn), float(mx))

    @staticmethod
    def fit_quantile(values, quantile: float) -> "QuantizationParams":  # scalar.rs:104-139
        if not (0.0 < quantile <= 1.0):
            raise InnrPanic("quantile must be in (0.0, 1.0]")
        v = np.asarray(values, dtype=np.float32).reshape(-1)
        if v.size == 0:
            return QuantizationParams(1.0, 0.0)
        if quantile >= 1.0:
            return QuantizationParams.fit(v)
        s = np.sort(v[np.isfinite(v)], kind="stable")
        if s.size == 0:
            return QuantizationParams(1.0, 0.0)
        tail = (F(1.0) - F(quantile)) / F(2.0)
        lo = int(np.floor(F(tail * F(s.size))))
        hi = min(int(np.ceil(F((F(1.0) - tail) * F(s.size)))), s.size - 1)
        return QuantizationParams.from_range(float(s[lo]), float(s[hi]))

    @staticmethod
    def fit_vectors(vectors: Sequence[Sequence[float]]) -> "QuantizationParams":  # scalar.rs:143-163
        flat = [np.asarray(v, dtype=np.float32).reshape(-1) for v in vectors]
        flat = np.concatenate(flat) if flat else np.empty(0, np.float32)
        fin = flat[~np.isnan(flat)]
        if fin.size == 0:
            return QuantizationParams(1.0, 0.0)
        return QuantizationParams.from_range(float(fin.min()), float(fin.max()))


class QuantizedU8:
    """scalar.rs:171-208."""

    def __init__(self, data, dimension: int):
        d = np.ascontiguousarray(data, dtype=np.uint8).reshape(-1)
        if d.size != dimension:
            raise InnrPanic(f"QuantizedU8: data length {d.size} doesn't match dimension {dimension}")
        self._data, self._dim = d, int(dimension)

    def data(self) -> np.ndarray: return self._data
    def dimension(self) -> int: return self._dim
    def memory_bytes(self) -> int: return int(self._data.size)


def quantize_u8(values, params: QuantizationParams) -> QuantizedU8:
    """scalar.rs:212-225 (round half away from zero, clamp to [0, 255])."""
    v = np.ascontiguousarray(values, dtype=np.float32).reshape(-1)
    out = np.empty(v.size, dtype=np.uint8)
    if v.size:
        load().innr_quantize_u8(_vp(v), v.size, C.c_float(params.alpha), C.c_float(params.offset), _vp(out))
    return QuantizedU8(out, v.size)


@dataclass
class QueryContext:
    query_sum: float


def query_context(query) -> QueryContext:
    """scalar.rs:236-240: sum(q), sequential, folded from -0.0."""
    s = F(-0.0)
    for x in np.asarray(query, dtype=np.float32).reshape(-1):
        s = F(s + x)
    return QueryContext(float(s))


def mixed_dot_u8_f32(a, b) -> float:
    """scalar.rs:314-326."""
    a = np.ascontiguousarray(a, dtype=np.float32).reshape(-1)
    b = np.ascontiguousarray(b, dtype=np.uint8).reshape(-1)
    if a.size != b.size:
        raise InnrPanic(f"mixed_dot_u8_f32: slice length mismatch ({a.size} vs {b.size})")
    return float(load().innr_mixed_dot_u8_f32(_vp(a), _vp(b), a.size))


def asymmetric_dot_u8_precomputed(query, quantized: QuantizedU8, params: QuantizationParams, ctx: QueryContext) -> float:
    """scalar.rs:284-300."""
    q = np.ascontiguousarray(query, dtype=np.float32).reshape(-1)
    if q.size != quantized.dimension():
        raise InnrPanic(f"asymmetric_dot_u8_precomputed: dimension mismatch ({q.size} vs {quantized.dimension()})")
    mixed = F(load().innr_mixed_dot_u8_f32(_vp(q), _vp(quantized.data()), q.size))
    return float(F(F(F(params.alpha) / F(255.0)) * mixed) + F(F(params.offset) * F(ctx.query_sum)))


def asymmetric_dot_u8(query, quantized: QuantizedU8, params: QuantizationParams) -> float:
    """scalar.rs:261-278."""
    q = np.asarray(query, dtype=np.float32).reshape(-1)
    if q.size != quantized.dimension():
        raise InnrPanic(f"asymmetric_dot_u8: dimension mismatch ({q.size} vs {quantized.dimension()})")
    return asymmetric_dot_u8_precomputed(q, quantized, params, query_context(q))


class QuantizedCorpus:
    """Device-resident packed code array of a &[QuantizedU8] (addition; see module docstring)."""

    def __init__(self, handle, n: int, dim: int, params: QuantizationParams, ctx: _lib.Context):
        self._h, self._n, self._d, self.params, self._ctx = handle, int(n), int(dim), params, ctx
        ctx._children.add(self)

    @classmethod
    def from_quantized(cls, corpus: Sequence[QuantizedU8], params: QuantizationParams,
                       ctx: Optional[_lib.Context] = None) -> "QuantizedCorpus":
        ctx = ctx or default_context()
        n = len(corpus)
        dim = corpus[0].dimension() if n else 0
        # documents of a different dimension make the reference panic inside its scoring loop (scalar.rs:290)
        for qd in corpus:
            if qd.dimension() != dim:
                raise InnrPanic(f"asymmetric_dot_u8_precomputed: dimension mismatch ({dim} vs {qd.dimension()})")
        codes = np.ascontiguousarray(np.stack([qd.data() for qd in corpus])) if n and dim else np.empty(0, np.uint8)
        return cls.from_codes(codes, n, dim, params, ctx)

    @classmethod
    def from_codes(cls, codes, n: int, dim: int, params: QuantizationParams,
                   ctx: Optional[_lib.Context] = None) -> "QuantizedCorpus":
        ctx = ctx or default_context()
        codes = np.ascontiguousarray(codes, dtype=np.uint8).reshape(-1)
        if codes.size != n * dim:
            raise InnrPanic("codes.len() != n * dim")
        h = C.c_void_p()
        check(load().innr_batch_upload_u8(ctx.handle, _vp(codes) if codes.size else None, n, dim,
                                          C.c_float(params.alpha), C.c_float(params.offset), C.byref(h)))
        return cls(h, n, dim, params, ctx)

    @classmethod
    def generate(cls, n: int, dim: int, params: QuantizationParams, seed: int = 0, row0: int = 0,
                 ctx: Optional[_lib.Context] = None) -> "QuantizedCorpus":
        """quantize_u8(uniform row (row0+i) of stream `seed`, params), generated on the device."""
        ctx = ctx or default_context()
        h = C.c_void_p()
        check(load().innr_batch_generate_u8(ctx.handle, n, dim, C.c_uint64(seed), C.c_uint64(row0),
                                            C.c_float(params.alpha), C.c_float(params.offset), C.byref(h)))
        return cls(h, n, dim, params, ctx)

    @classmethod
    def from_batch(cls, batch, params: QuantizationParams) -> "QuantizedCorpus":
        """quantize_u8 of every value of a device-resident f32 VerticalBatch, on the device (corpus ingest without a
        host round trip); inherits the batch's index base."""
        h = C.c_void_p()
        check(load().innr_batch_quantize_u8(batch._h, C.c_float(params.alpha), C.c_float(params.offset), C.byref(h)))
        return cls(h, batch.num_vectors(), batch.dimension(), params, batch._ctx)

    def __len__(self) -> int: return self._n
    def dimension(self) -> int: return self._d

    def set_index_base(self, base: int) -> None:
        """Global index of this shard's first document (multi-GPU range partition)."""
        check(load().innr_batch_set_index_base(self._h, C.c_uint64(int(base))))

    def codes(self) -> np.ndarray:
        """Codes back on the host as (n, dim) rows."""
        out = np.empty((self._d, self._n), dtype=np.uint8)
        check(load().innr_batch_download_u8(self._h, _vp(out) if out.size else None))
        return np.ascontiguousarray(out.T)

    def scores(self, query) -> np.ndarray:
        """asymmetric_dot_u8_precomputed(query, doc_i) for every document."""
        q = np.ascontiguousarray(query, dtype=np.float32).reshape(-1)
        out = np.empty(self._n, dtype=np.float32)
        check(load().innr_batch_scores_u8(self._h, _vp(q) if q.size else None, q.size, _vp(out) if out.size else None))
        return out

    def knn_multi(self, queries, k: int, engine: int = KNN_AUTO, stats: Optional[KnnStats] = None):
        q = np.ascontiguousarray(queries, dtype=np.float32)
        if q.ndim == 1:
            q = q.reshape(1, -1)
        nq, d = q.shape
        kk = max(min(int(k), self._n), 1)
        idx = np.empty((nq, kk), dtype=np.uint64)
        sc = np.empty((nq, kk), dtype=np.float32)
        out_k = C.c_size_t(0)
        st = stats if stats is not None else KnnStats()
        check(load().innr_batch_knn_u8(self._h, _vp(q) if q.size else None, nq, d, int(k), engine, _vp(idx), _vp(sc),
                                       C.byref(out_k), C.byref(st)))
        r = int(out_k.value)
        return idx[:, :r].reshape(nq, r), sc[:, :r].reshape(nq, r)

    # ---- persistence: the codes in their device order data[d*N + i] behind a 32-byte header (cf. VerticalBatch.save) -----
    _MAGIC = b"INNRU8C1"  # magic, u64 n, u64 dim, f32 alpha, f32 offset, then dim x n bytes

    def save(self, path: str) -> None:
        """Write the code corpus and its QuantizationParams; the device copy goes straight into the memory-mapped file."""
        with open(path, "wb") as f:
            f.write(self._MAGIC + np.array([self._n, self._d], dtype="<u8").tobytes() +
                    np.array([self.params.alpha, self.params.offset], dtype="<f4").tobytes())
            f.truncate(32 + self._n * self._d)
        if self._n * self._d:
            mm = np.memmap(path, dtype=np.uint8, mode="r+", offset=32, shape=(self._d * self._n,))
            check(load().innr_batch_download_u8(self._h, _vp(mm)))
            mm.flush()
            del mm

    @classmethod
    def load(cls, path: str, ctx: Optional[_lib.Context] = None) -> "QuantizedCorpus":
        """Inverse of save(): the file is memory-mapped and uploaded as it is (already the device's layout order)."""
        import os
        with open(path, "rb") as f:
            head = f.read(32)
        if len(head) != 32 or head[:8] != cls._MAGIC:
            raise InnrPanic(f"{path}: not an innr u8 corpus file")
        n, d = (int(x) for x in np.frombuffer(head[8:24], dtype="<u8"))
        alpha, offset = (float(x) for x in np.frombuffer(head[24:32], dtype="<f4"))
        if os.path.getsize(path) != 32 + n * d:
            raise InnrPanic(f"{path}: size does not match its header ({n} x {d})")
        ctx = ctx or default_context()
        h = C.c_void_p()
        mm = np.memmap(path, dtype=np.uint8, mode="r", offset=32, shape=(d * n,)) if n * d else np.empty(0, np.uint8)
        check(load().innr_batch_upload_u8_colmajor(ctx.handle, _vp(mm) if n * d else None, n, d, C.c_float(alpha), C.c_float(offset),
                                                   C.byref(h)))
        return cls(h, n, d, QuantizationParams(alpha, offset), ctx)

    def prefix(self, prefix_dims: int) -> "QuantizedCorpus":
        """The corpus restricted to the first min(prefix_dims, dim) code dimensions (same params): a view of the leading
        rows of the dimension-major code matrix, cf. VerticalBatch.prefix. Closing the parent closes its views."""
        h = C.c_void_p()
        check(load().innr_batch_prefix_view(self._h, int(prefix_dims), C.byref(h)))
        v = QuantizedCorpus(h, self._n, min(int(prefix_dims), self._d), self.params, self._ctx)
        v._parent = self
        if not hasattr(self, "_views"):
            self._views = weakref.WeakSet()
        self._views.add(v)
        return v

    def close(self) -> None:
        for v in list(getattr(self, "_views", ())):
            v.close()
        if getattr(self, "_h", None):
            if getattr(self._ctx, "handle", None):
                load().innr_batch_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def fit_batch(batch) -> QuantizationParams:
    """QuantizationParams::fit (scalar.rs:68-87) over every value of a device-resident f32 VerticalBatch."""
    if batch.num_vectors() * batch.dimension() == 0:  # scalar.rs:69-74: no values at all
        return QuantizationParams(1.0, 0.0)
    mn, mx, any_ = C.c_float(0.0), C.c_float(0.0), C.c_int(0)
    check(load().innr_batch_minmax(batch._h, C.byref(mn), C.byref(mx), C.byref(any_)))
    # The reference's scan starts from (f32::MAX, f32::MIN) and only ever takes `v < min` / `v > max` (scalar.rs:76-85), and
    # `fit` -- unlike fit_vectors (:156-161) -- has no min > max guard: a non-empty all-NaN corpus therefore yields
    # from_range(f32::MAX, f32::MIN) = {alpha 1.0, offset 3.4028235e38}, and a corpus whose only non-NaN values are +inf
    # keeps min = f32::MAX (inf < MAX is false). The device reports the true extrema of the non-NaN values: clamp them the same way.
    lo, hi = F(3.4028235e38), F(-3.4028235e38)
    if any_.value:
        lo, hi = min(lo, F(mn.value)), max(hi, F(mx.value))
    return QuantizationParams.from_range(float(lo), float(hi))


def fit_quantile_batch(batch, quantile: float) -> QuantizationParams:
    """QuantizationParams::fit_quantile (scalar.rs:104-139) over every value of a device-resident f32 VerticalBatch: the two
    rank values come from a radix select on the device (no sort, no download of the corpus)."""
    if not (0.0 < quantile <= 1.0):
        raise InnrPanic("quantile must be in (0.0, 1.0]")
    lo, hi, any_ = C.c_float(0.0), C.c_float(0.0), C.c_int(0)
    check(load().innr_batch_quantile_range(batch._h, C.c_float(quantile), C.byref(lo), C.byref(hi), C.byref(any_)))
    if not any_.value:
        return QuantizationParams(1.0, 0.0)
    return QuantizationParams.from_range(float(lo.value), float(hi.value))


def two_stage_knn(queries, coarse: "QuantizedCorpus", fine, k: int, k_coarse: int, metric: Optional[int] = None):
    """The two-stage retrieval the reference describes for this module (scalar.rs:366-368): batch_knn_u8 over the
    quantised corpus for k_coarse candidates, then an exact re-rank on the full-precision batch `fine` (both resident
    on the GPU: 50M x 768 is 38 GB of codes + 154 GB of f32). Returns (indices [Q, k'], exact scores [Q, k'])."""
    from . import batch as B
    idx, _ = coarse.knn_multi(queries, k_coarse)
    return B.batch_rerank(queries, fine, idx, k, B.METRIC_DOT if metric is None else metric)


def batch_knn_u8(query, corpus, params: QuantizationParams, k: int, engine: int = KNN_AUTO) -> List[Tuple[int, float]]:
    """scalar.rs:370-393: [(index, score)] of the k best documents by asymmetric dot, best first.
    `corpus` is a sequence of QuantizedU8 (uploaded for this call) or a device-resident QuantizedCorpus."""
    if isinstance(corpus, QuantizedCorpus):
        qc, own = corpus, False
    else:
        if len(corpus) == 0 or k == 0:  # scalar.rs:376-378
            return []
        qc, own = QuantizedCorpus.from_quantized(corpus, params), True
    try:
        if len(qc) == 0 or k == 0:
            return []
        idx, sc = qc.knn_multi(np.asarray(query, dtype=np.float32).reshape(1, -1), k, engine)
        return [(int(i), float(s)) for i, s in zip(idx[0], sc[0])]
    finally:
        if own:
            qc.close()
