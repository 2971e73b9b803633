"""Host-side mirror of innr's `batch` module (reference: src/batch.rs) over the HIP C ABI.

Same function names, argument meaning and error behaviour as the reference:
  VerticalBatch (batch.rs:88-220), batch_dot/_into (:270,:284), batch_l2_squared/_into (:236,:250),
  batch_norms/_into (:663,:672), batch_cosine/_into (:690,:705), batch_knn (:385), batch_knn_dot (:742),
  batch_knn_cosine (:777), BatchKnnResult (:368-377).
A reference panic (assert_eq! on a dimension mismatch) is raised as InnrPanic (an AssertionError).
Vec<f32> <-> numpy float32 arrays; `_into` variants refill a caller-owned Python list in place.

Additions (the reference has no multi-query API; parity = "the reference called in a loop"):
  batch_knn_dot_multi / batch_knn_cosine_multi / batch_knn_multi.

Every computation runs on the GPU through include/innr_hip.h; nothing here computes scores on the CPU.
"""
from __future__ import annotations

import ctypes as C
import os
import weakref
from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence

import numpy as np

from . import _lib
from ._lib import (KNN_AUTO, METRIC_COSINE, METRIC_DOT, METRIC_L2SQ, InnrPanic, KnnStats, check, default_context,
                   load)


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _vp(a: np.ndarray) -> C.c_void_p:
    return C.c_void_p(a.ctypes.data)


class VerticalBatch:
    """Vertical (columnar, PDX) storage: data[d * num_vectors + i] (batch.rs:88-95), resident on the GPU."""

    def __init__(self, handle: C.c_void_p, num_vectors: int, dimension: int, ctx: _lib.Context):
        self._h = handle
        self._n = int(num_vectors)
        self._d = int(dimension)
        self._ctx = ctx
        ctx._children.add(self)
        self._host: Optional[np.ndarray] = None  # lazily downloaded dimension-major copy

    # ---- constructors --------------------------------------------------------------------------
    @classmethod
    def _upload(cls, fn_name: str, arr: np.ndarray, n: int, d: int, ctx: Optional[_lib.Context]):
        ctx = ctx or default_context()
        h = C.c_void_p()
        check(getattr(load(), fn_name)(ctx.handle, _vp(arr) if arr.size else None, n, d, C.byref(h)))
        return cls(h, n, d, ctx)

    @classmethod
    def from_rows(cls, vectors: Sequence[Sequence[float]], ctx: Optional[_lib.Context] = None) -> "VerticalBatch":
        """batch.rs:103-131. Panics on inconsistent vector dimension (batch.rs:120)."""
        rows = [np.asarray(v, dtype=np.float32).reshape(-1) for v in vectors]
        if len(rows) == 0:
            return cls._upload("innr_batch_upload_rowmajor", np.empty(0, np.float32), 0, 0, ctx)
        d = rows[0].size
        for r in rows:
            if r.size != d:
                raise InnrPanic("Inconsistent vector dimension")
        flat = np.ascontiguousarray(np.stack(rows)) if d else np.empty(0, np.float32)
        return cls._upload("innr_batch_upload_rowmajor", flat, len(rows), d, ctx)

    from_slices = from_rows  # batch.rs:138-164: same contract for borrowed slices

    @classmethod
    def from_flat(cls, data, num_vectors: int, dimension: int, ctx: Optional[_lib.Context] = None) -> "VerticalBatch":
        """batch.rs:167-183: flat row-major data; asserts len == num_vectors * dimension."""
        flat = _f32(data).reshape(-1)
        if flat.size != num_vectors * dimension:
            raise InnrPanic(f"assertion failed: data.len() == num_vectors * dimension ({flat.size} vs "
                            f"{num_vectors * dimension})")
        return cls._upload("innr_batch_upload_rowmajor", flat, num_vectors, dimension, ctx)

    @classmethod
    def from_data(cls, data, num_vectors: int, dimension: int, ctx: Optional[_lib.Context] = None) -> "VerticalBatch":
        """Inverse of data(): adopt an already dimension-major buffer (VerticalBatch::data(), batch.rs:212)."""
        col = _f32(data).reshape(-1)
        if col.size != num_vectors * dimension:
            raise InnrPanic("data.len() != num_vectors * dimension")
        return cls._upload("innr_batch_upload_colmajor", col, num_vectors, dimension, ctx)

    @classmethod
    def generate(cls, num_vectors: int, dimension: int, seed: int = 0, generator: int = _lib.GEN_UNIFORM,
                 row0: int = 0, ctx: Optional[_lib.Context] = None) -> "VerticalBatch":
        """Synthetic corpus made on the device; row i = row (row0 + i) of the chosen stream.
        GEN_EXAMPLE_LCG: generate_embedding(dimension, seed + row) (examples/batch_demo.rs:167, 233-242);
        GEN_UNIFORM: i.i.d. uniform[-1,1) (distribution of benches/batch.rs:11-21)."""
        ctx = ctx or default_context()
        h = C.c_void_p()
        check(load().innr_batch_generate(ctx.handle, num_vectors, dimension, int(generator), C.c_uint64(seed),
                                         C.c_uint64(row0), C.byref(h)))
        return cls(h, num_vectors, dimension, ctx)

    # ---- persistence: a minimal file around VerticalBatch::data() (batch.rs:208-214) -----------------------------
    _MAGIC = b"INNRPDX1"  # 8-byte magic, u64 num_vectors, u64 dimension, then dimension x num_vectors f32 (data() order)

    def save(self, path: str) -> None:
        """Write the batch in its own dimension-major order. The device copy goes straight into a memory-mapped file:
        no second host buffer, so a 30 GB shard does not need 30 GB of RAM."""
        with open(path, "wb") as f:
            f.write(self._MAGIC + np.array([self._n, self._d], dtype="<u8").tobytes())
            f.truncate(24 + 4 * self._n * self._d)
        if self._n * self._d:
            mm = np.memmap(path, dtype="<f4", mode="r+", offset=24, shape=(self._d, self._n))
            check(load().innr_batch_download_colmajor(self._h, _vp(mm)))
            mm.flush()
            del mm

    @classmethod
    def load(cls, path: str, ctx: Optional[_lib.Context] = None) -> "VerticalBatch":
        """Inverse of save(): the file is memory-mapped and uploaded as it is (already the device's layout order)."""
        with open(path, "rb") as f:
            head = f.read(24)
        if len(head) != 24 or head[:8] != cls._MAGIC:
            raise InnrPanic(f"{path}: not an innr PDX file")
        n, d = (int(x) for x in np.frombuffer(head[8:], dtype="<u8"))
        if os.path.getsize(path) != 24 + 4 * n * d:
            raise InnrPanic(f"{path}: size does not match its header ({n} x {d})")
        if n * d == 0:
            return cls._upload("innr_batch_upload_colmajor", np.empty(0, np.float32), n, d, ctx)
        mm = np.memmap(path, dtype="<f4", mode="r", offset=24, shape=(d * n,))
        return cls._upload("innr_batch_upload_colmajor", mm, n, d, ctx)

    # ---- accessors (batch.rs:187-219) --------------------------------------------------------------
    def num_vectors(self) -> int:
        return self._n

    def dimension(self) -> int:
        return self._d

    def data(self) -> np.ndarray:
        """Raw data in dimension-major order, shape (dimension, num_vectors) viewable as flat [d*N+i]."""
        if self._host is None:
            out = np.empty((self._d, self._n), dtype=np.float32)
            check(load().innr_batch_download_colmajor(self._h, _vp(out) if out.size else None))
            self._host = out
        return self._host

    def get(self, dim: int, vec_idx: int) -> float:
        return float(self.data()[dim, vec_idx])

    def dimension_slice(self, dim: int) -> np.ndarray:
        return self.data()[dim]

    def extract_vector(self, vec_idx: int) -> np.ndarray:
        return self.data()[:, vec_idx].copy()

    def set_index_base(self, base: int) -> None:
        """Range-partitioned corpus: indices reported by kNN become base + local index."""
        check(load().innr_batch_set_index_base(self._h, C.c_uint64(base)))

    def prefix(self, prefix_dims: int) -> "VerticalBatch":
        """The batch restricted to its first min(prefix_dims, dimension) dimensions -- the batch-level form of
        matryoshka_dot / matryoshka_cosine (dense.rs:436-462). Dimension-major storage makes this a view of the leading
        rows on the device: nothing is copied. Closing the parent closes its views."""
        h = C.c_void_p()
        check(load().innr_batch_prefix_view(self._h, int(prefix_dims), C.byref(h)))
        v = VerticalBatch(h, self._n, min(int(prefix_dims), self._d), self._ctx)
        v._parent = self
        if not hasattr(self, "_views"):
            self._views = weakref.WeakSet()
        self._views.add(v)
        return v

    def close(self) -> None:
        for v in list(getattr(self, "_views", ())):
            v.close()
        if getattr(self, "_h", None):
            if getattr(self._ctx, "handle", None):  # a closed ctx has already freed its batches
                load().innr_batch_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


@dataclass
class BatchKnnResult:
    """batch.rs:368-377: indices and scores, best first."""
    indices: List[int] = field(default_factory=list)
    scores: List[float] = field(default_factory=list)

    def __eq__(self, other):
        return (isinstance(other, BatchKnnResult) and list(self.indices) == list(other.indices)
                and np.array_equal(np.asarray(self.scores, np.float32), np.asarray(other.scores, np.float32)))


def _assert_dim(query: np.ndarray, batch: VerticalBatch) -> None:
    if query.size != batch.dimension():  # assert_eq!(query.len(), batch.dimension)
        raise InnrPanic(f"assertion `left == right` failed\n  left: {query.size}\n right: {batch.dimension()}")


def _scores(metric: int, query, batch: VerticalBatch, norms=None) -> np.ndarray:
    q = _f32(query).reshape(-1)
    _assert_dim(q, batch)
    out = np.empty(batch.num_vectors(), dtype=np.float32)
    nrm = None
    if norms is not None:
        nrm = _f32(norms).reshape(-1)
        if nrm.size != batch.num_vectors():  # assert_eq!(norms.len(), batch.num_vectors) batch.rs:711
            raise InnrPanic(f"assertion `left == right` failed\n  left: {nrm.size}\n right: {batch.num_vectors()}")
    check(load().innr_batch_scores(batch._h, metric, _vp(q) if q.size else None, q.size,
                                   _vp(nrm) if nrm is not None and nrm.size else None,
                                   _vp(out) if out.size else None))
    return out


def _refill(dst: list, values: np.ndarray) -> None:
    dst.clear()            # Vec::clear + resize: the caller's allocation is reused
    dst.extend(values.tolist())


def batch_l2_squared(query, batch: VerticalBatch) -> np.ndarray:
    """batch.rs:236-240."""
    return _scores(METRIC_L2SQ, query, batch)


def batch_l2_squared_into(query, batch: VerticalBatch, distances: list) -> None:
    """batch.rs:250-266."""
    _refill(distances, _scores(METRIC_L2SQ, query, batch))


def batch_dot(query, batch: VerticalBatch) -> np.ndarray:
    """batch.rs:270-274."""
    return _scores(METRIC_DOT, query, batch)


def batch_dot_into(query, batch: VerticalBatch, products: list) -> None:
    """batch.rs:284-297."""
    _refill(products, _scores(METRIC_DOT, query, batch))


def batch_norms(batch: VerticalBatch) -> np.ndarray:
    """batch.rs:663-667."""
    out = np.empty(batch.num_vectors(), dtype=np.float32)
    check(load().innr_batch_norms(batch._h, _vp(out) if out.size else None))
    return out


def batch_norms_into(batch: VerticalBatch, norms: list) -> None:
    """batch.rs:672-686."""
    _refill(norms, batch_norms(batch))


def batch_cosine(query, batch: VerticalBatch, norms) -> np.ndarray:
    """batch.rs:690-694."""
    if len(norms) != batch.num_vectors():
        raise InnrPanic(f"assertion `left == right` failed\n  left: {len(norms)}\n right: {batch.num_vectors()}")
    return _scores(METRIC_COSINE, query, batch, norms if batch.num_vectors() else None)


def batch_cosine_into(query, batch: VerticalBatch, norms, cosines: list) -> None:
    """batch.rs:705-728."""
    _refill(cosines, batch_cosine(query, batch, norms))


# ---- kNN ---------------------------------------------------------------------------------------------
def knn_multi(metric: int, queries, batch: VerticalBatch, k: int, engine: int = KNN_AUTO,
              stats: Optional[KnnStats] = None):
    """Q queries at once. Returns (indices uint64 [Q, k'], scores float32 [Q, k'])."""
    q = _f32(queries)
    if q.ndim == 1:
        q = q.reshape(1, -1)
    nq, d = q.shape
    if d != batch.dimension():
        raise InnrPanic(f"assertion `left == right` failed\n  left: {d}\n right: {batch.dimension()}")
    kk = min(int(k), batch.num_vectors())
    idx = np.empty((nq, max(kk, 1)), dtype=np.uint64)
    sc = np.empty((nq, max(kk, 1)), dtype=np.float32)
    out_k = C.c_size_t(0)
    st = stats if stats is not None else KnnStats()
    check(load().innr_batch_knn(batch._h, metric, _vp(q) if q.size else None, nq, d, int(k), engine, _vp(idx),
                                _vp(sc), C.byref(out_k), C.byref(st)))
    r = int(out_k.value)
    return idx[:, :r].reshape(nq, r), sc[:, :r].reshape(nq, r)


def batch_rerank(queries, batch: VerticalBatch, candidates, k: int, metric: int = METRIC_DOT):
    """Second stage of the two-stage pipeline (scalar.rs:366-368): exact scores of `candidates` [Q, kc] (global indices
    inside `batch`, no duplicates per query; any kc -- beyond 256 every query's candidates are sorted on the device) in the
    reference's arithmetic order, best min(k, kc) per query in the kNN functions' order. Returns (indices uint64 [Q, k'], scores float32 [Q, k'])."""
    q = _f32(queries)
    if q.ndim == 1:
        q = q.reshape(1, -1)
    nq, d = q.shape
    if d != batch.dimension():
        raise InnrPanic(f"assertion `left == right` failed\n  left: {d}\n right: {batch.dimension()}")
    cand = np.ascontiguousarray(candidates, dtype=np.uint64)
    if cand.ndim == 1:
        cand = cand.reshape(1, -1)
    if cand.shape[0] != nq:
        raise InnrPanic("one candidate row per query")
    kc = cand.shape[1]
    kk = max(min(int(k), kc), 1)
    idx = np.empty((nq, kk), dtype=np.uint64)
    sc = np.empty((nq, kk), dtype=np.float32)
    out_k = C.c_size_t(0)
    check(load().innr_batch_rerank(batch._h, metric, _vp(q) if q.size else None, nq, d, _vp(cand) if cand.size else None,
                                   kc, int(k), _vp(idx), _vp(sc), C.byref(out_k)))
    r = int(out_k.value)
    return idx[:, :r].reshape(nq, r), sc[:, :r].reshape(nq, r)


def matryoshka_knn(queries, batch: VerticalBatch, prefix_dims: int, k_coarse: int, k: int, metric: int = METRIC_COSINE,
                   engine: int = KNN_AUTO, coarse: Optional[VerticalBatch] = None):
    """Matryoshka progressive search (examples/matryoshka_search.rs:49-73) for a batch of queries: coarse top-k_coarse on
    the first prefix_dims dimensions (matryoshka_cosine / matryoshka_dot, dense.rs:436-462), then the exact score at
    full dimension for those candidates, best k. `coarse` may hold batch.prefix(prefix_dims) across calls (its norms are
    cached on the device). Returns (indices uint64 [Q, k'], scores float32 [Q, k'])."""
    q = _f32(queries)
    if q.ndim == 1:
        q = q.reshape(1, -1)
    if q.shape[1] != batch.dimension():
        raise InnrPanic(f"assertion `left == right` failed\n  left: {q.shape[1]}\n right: {batch.dimension()}")
    view = coarse if coarse is not None else batch.prefix(prefix_dims)
    try:
        idx, _ = knn_multi(metric, np.ascontiguousarray(q[:, :view.dimension()]), view, k_coarse, engine)
    finally:
        if coarse is None:
            view.close()
    return batch_rerank(q, batch, idx, k, metric)


def _knn_single(metric: int, query, batch: VerticalBatch, k: int, engine: int) -> BatchKnnResult:
    q = _f32(query).reshape(-1)
    _assert_dim(q, batch)
    idx, sc = knn_multi(metric, q.reshape(1, -1), batch, k, engine)
    return BatchKnnResult(indices=[int(i) for i in idx[0]], scores=[float(s) for s in sc[0]])


def batch_knn(query, batch: VerticalBatch, k: int, engine: int = KNN_AUTO) -> BatchKnnResult:
    """batch.rs:385-411: k nearest by squared L2, ascending."""
    return _knn_single(METRIC_L2SQ, query, batch, k, engine)


def batch_knn_dot(query, batch: VerticalBatch, k: int, engine: int = KNN_AUTO) -> BatchKnnResult:
    """batch.rs:742-764: k highest dot products, descending."""
    return _knn_single(METRIC_DOT, query, batch, k, engine)


def batch_knn_cosine(query, batch: VerticalBatch, k: int, engine: int = KNN_AUTO) -> BatchKnnResult:
    """batch.rs:777-800: k highest cosine similarities, descending."""
    return _knn_single(METRIC_COSINE, query, batch, k, engine)


def batch_knn_dot_multi(queries, batch: VerticalBatch, k: int, engine: int = KNN_AUTO, stats=None):
    return knn_multi(METRIC_DOT, queries, batch, k, engine, stats)


def batch_knn_cosine_multi(queries, batch: VerticalBatch, k: int, engine: int = KNN_AUTO, stats=None):
    return knn_multi(METRIC_COSINE, queries, batch, k, engine, stats)


def batch_knn_multi(queries, batch: VerticalBatch, k: int, engine: int = KNN_AUTO, stats=None):
    return knn_multi(METRIC_L2SQ, queries, batch, k, engine, stats)


def batch_dimension_variance(batch: VerticalBatch) -> np.ndarray:
    """batch.rs:572-592: per-dimension variance across all vectors (sequential sums, reference order)."""
    out = np.empty(batch.dimension(), dtype=np.float32)
    check(load().innr_batch_dimension_variance(batch._h, _vp(out) if out.size else None))
    return out


def _l2_variant(fn_name: str, query, batch: VerticalBatch, k: int, *extra) -> BatchKnnResult:
    q = _f32(query).reshape(-1)
    _assert_dim(q, batch)
    kk = max(min(int(k), batch.num_vectors()), 1)
    idx = np.empty(kk, dtype=np.uint64)
    sc = np.empty(kk, dtype=np.float32)
    out_k = C.c_size_t(0)
    check(getattr(load(), fn_name)(batch._h, _vp(q) if q.size else None, q.size, int(k), *extra, _vp(idx), _vp(sc),
                                   C.byref(out_k)))
    r = int(out_k.value)
    return BatchKnnResult(indices=[int(i) for i in idx[:r]], scores=[float(s) for s in sc[:r]])


def batch_knn_filtered(query, batch: VerticalBatch, k: int, predicate: Callable[[int], bool]) -> BatchKnnResult:
    """batch.rs:820-882: kNN (squared L2) over the vectors where predicate(index) is true. Like the reference,
    the predicate is evaluated once per index up front (batch.rs:839) into a mask; the scan runs on the GPU."""
    mask = np.fromiter((1 if predicate(i) else 0 for i in range(batch.num_vectors())), dtype=np.uint8,
                       count=batch.num_vectors())
    return _l2_variant("innr_batch_knn_filtered", query, batch, k, _vp(mask) if mask.size else None)


def batch_knn_reordered(query, batch: VerticalBatch, k: int) -> BatchKnnResult:
    """batch.rs:621-659: exact kNN with distances accumulated in decreasing-variance dimension order."""
    return _l2_variant("innr_batch_knn_reordered", query, batch, k)


def batch_l2_squared_pruning(query, batch: VerticalBatch, threshold: float):
    """batch.rs:320-365: [(index, squared_distance)] of the vectors within `threshold`, in index order."""
    q = _f32(query).reshape(-1)
    _assert_dim(q, batch)
    n = batch.num_vectors()
    idx = np.empty(max(n, 1), dtype=np.uint64)
    ds = np.empty(max(n, 1), dtype=np.float32)
    out_n = C.c_size_t(0)
    check(load().innr_batch_l2_squared_pruning(batch._h, _vp(q) if q.size else None, q.size, C.c_float(threshold),
                                               _vp(idx), _vp(ds), n, C.byref(out_n)))
    r = int(out_n.value)
    return [(int(idx[i]), float(ds[i])) for i in range(r)]
