"""Host-side mirror of innr's `maxsim` module (reference: src/maxsim.rs:96-194).

maxsim(query_tokens, doc_tokens) = sum_i max_j q_i . d_j for ONE (query, document) pair -- the reference's
signature; evaluated by libinnr_hip.so's host function in the portable arithmetic order (a single pair cannot
amortise a kernel launch). The many-documents scan (ColBERT-style top-k over a corpus) is the device path.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import InnrPanic, check, load


def _tokens(t, what: str) -> np.ndarray:
    rows = [np.asarray(r, dtype=np.float32).reshape(-1) for r in t]
    if not rows:
        return np.empty((0, 0), np.float32)
    dim = rows[0].size
    if any(r.size != dim for r in rows):
        raise InnrPanic(f"dimension mismatch ({what})")  # maxsim.rs:103-110
    return np.ascontiguousarray(np.stack(rows))


def _maxsim(query_tokens, doc_tokens, cosine: int) -> float:
    q = _tokens(query_tokens, "query")
    d = _tokens(doc_tokens, "doc")
    if q.shape[0] == 0 or d.shape[0] == 0:
        return 0.0  # maxsim.rs:97-99
    if q.shape[1] != d.shape[1]:
        raise InnrPanic("dimension mismatch (doc)")
    out = C.c_float(0.0)
    check(load().innr_maxsim_pair(C.c_void_p(q.ctypes.data), q.shape[0], C.c_void_p(d.ctypes.data), d.shape[0], q.shape[1],
                                  cosine, C.byref(out)))
    return float(out.value)


def maxsim(query_tokens, doc_tokens) -> float:
    """maxsim.rs:96-137"""
    return _maxsim(query_tokens, doc_tokens, 0)


def maxsim_cosine(query_tokens, doc_tokens) -> float:
    """maxsim.rs:168-194"""
    return _maxsim(query_tokens, doc_tokens, 1)


# ---- many documents: the device path -------------------------------------------------------------------------
from typing import Optional, Sequence  # noqa: E402

from . import _lib  # noqa: E402
from ._lib import KnnStats, default_context  # noqa: E402


class DocumentCorpus:
    """Token embeddings of many documents resident on the GPU (addition: the reference scores one (query, doc)
    pair per call and the caller loops and sorts, examples/maxsim_colbert.rs:171-187). `scores` returns
    maxsim(query, doc_i) for every document, bit-identical to calling the reference in a loop; `topk` the k best
    (score descending, ties -> lower document index)."""

    def __init__(self, handle, ndocs: int, T: int, dim: int, ctx: _lib.Context):
        self._h, self._n, self._T, self._dim, self._ctx = handle, int(ndocs), int(T), int(dim), ctx
        ctx._children.add(self)

    @classmethod
    def from_tokens(cls, tokens, doc_len: Optional[Sequence[int]] = None, ctx: Optional[_lib.Context] = None):
        """tokens: array [docs, T, dim]; doc_len[i] <= T = number of valid tokens of document i (default: T)."""
        ctx = ctx or default_context()
        t = np.ascontiguousarray(tokens, dtype=np.float32)
        if t.ndim != 3:
            raise InnrPanic("tokens must be [docs, T, dim]")
        n, T, dim = t.shape
        dl = None if doc_len is None else np.ascontiguousarray(doc_len, dtype=np.uint32)
        if dl is not None and dl.size != n:
            raise InnrPanic("doc_len.len() != docs")
        h = C.c_void_p()
        check(load().innr_maxsim_upload(ctx.handle, C.c_void_p(t.ctypes.data) if t.size else None,
                                        C.c_void_p(dl.ctypes.data) if dl is not None and dl.size else None, n, T, dim,
                                        C.byref(h)))
        return cls(h, n, T, dim, ctx)

    @classmethod
    def generate(cls, ndocs: int, T: int, dim: int, seed: int = 0, row0: int = 0, ctx: Optional[_lib.Context] = None):
        """Synthetic corpus made on the device: token (doc, t) = normalised uniform row (row0 + doc*T + t)."""
        ctx = ctx or default_context()
        h = C.c_void_p()
        check(load().innr_maxsim_generate(ctx.handle, ndocs, T, dim, C.c_uint64(seed), C.c_uint64(row0), C.byref(h)))
        return cls(h, ndocs, T, dim, ctx)

    # ---- persistence: tokens in their device order [doc][token][dim] behind a 40-byte header (cf. VerticalBatch.save) ----
    _MAGIC = b"INNRDOC1"  # magic, u64 docs, u64 T, u64 dim, u64 has_doc_len, then docs*T*dim f32, then [docs u32 lengths]

    def save(self, path: str) -> None:
        has = C.c_int(0)
        check(load().innr_docs_shape(self._h, None, None, None, C.byref(has)))
        ntok = self._n * self._T * self._dim
        with open(path, "wb") as f:
            f.write(self._MAGIC + np.array([self._n, self._T, self._dim, has.value], dtype="<u8").tobytes())
            f.truncate(40 + 4 * ntok + (4 * self._n if has.value else 0))
        if ntok:
            tok = np.memmap(path, dtype="<f4", mode="r+", offset=40, shape=(ntok,))
            dl = np.empty(self._n, np.uint32) if has.value else None
            check(load().innr_docs_download(self._h, C.c_void_p(tok.ctypes.data), C.c_void_p(dl.ctypes.data) if dl is not None else None))
            tok.flush()
            del tok
            if dl is not None:
                with open(path, "r+b") as f:
                    f.seek(40 + 4 * ntok)
                    f.write(dl.astype("<u4").tobytes())

    @classmethod
    def load(cls, path: str, ctx: Optional[_lib.Context] = None) -> "DocumentCorpus":
        import os
        with open(path, "rb") as f:
            head = f.read(40)
        if len(head) != 40 or head[:8] != cls._MAGIC:
            raise InnrPanic(f"{path}: not an innr document corpus file")
        n, T, dim, has = (int(x) for x in np.frombuffer(head[8:], dtype="<u8"))
        ntok = n * T * dim
        if os.path.getsize(path) != 40 + 4 * ntok + (4 * n if has else 0):
            raise InnrPanic(f"{path}: size does not match its header ({n} x {T} x {dim})")
        tok = np.memmap(path, dtype="<f4", mode="r", offset=40, shape=(n, T, dim)) if ntok else np.empty((n, T, dim), np.float32)
        dl = np.fromfile(path, dtype="<u4", count=n, offset=40 + 4 * ntok) if has else None
        return cls.from_tokens(tok, dl, ctx)

    def set_index_base(self, base: int) -> None:
        """Global index of this shard's first document (multi-GPU range partition)."""
        check(load().innr_docs_set_index_base(self._h, C.c_uint64(int(base))))

    def __len__(self) -> int:
        return self._n

    def _q(self, query_tokens) -> np.ndarray:
        q = _tokens(query_tokens, "query")
        if q.shape[0] and self._T and self._n and q.shape[1] != self._dim:
            raise InnrPanic("dimension mismatch (doc)")
        return q

    def scores(self, query_tokens, cosine: bool = False) -> np.ndarray:
        q = self._q(query_tokens)
        out = np.empty(self._n, dtype=np.float32)
        check(load().innr_maxsim_scores(self._h, 1 if cosine else 0, C.c_void_p(q.ctypes.data) if q.size else None,
                                        q.shape[0], self._dim if q.size == 0 else q.shape[1],
                                        C.c_void_p(out.ctypes.data) if out.size else None))
        return out

    def topk(self, query_tokens, k: int, cosine: bool = False, stats: Optional[KnnStats] = None,
             engine: int = _lib.KNN_AUTO):
        """k best documents (index, exact maxsim score), best first; engine KNN_EXACT / KNN_MFMA / KNN_AUTO."""
        q = self._q(query_tokens)
        kk = max(min(int(k), self._n), 1)
        idx = np.empty(kk, dtype=np.uint64)
        sc = np.empty(kk, dtype=np.float32)
        out_k = C.c_size_t(0)
        st = stats if stats is not None else KnnStats()
        check(load().innr_maxsim_topk(self._h, 1 if cosine else 0, C.c_void_p(q.ctypes.data) if q.size else None,
                                      q.shape[0], self._dim if q.size == 0 else q.shape[1], int(k), int(engine),
                                      C.c_void_p(idx.ctypes.data), C.c_void_p(sc.ctypes.data), C.byref(out_k), C.byref(st)))
        r = int(out_k.value)
        return idx[:r].copy(), sc[:r].copy()

    def topk_multi(self, queries, k: int, cosine: bool = False, stats: Optional[KnnStats] = None,
                   engine: int = _lib.KNN_AUTO):
        """Several queries (a list of [Tq_i, dim] token matrices) in one call; on the MFMA engine they share corpus
        passes four at a time. Returns (indices [Q, k'], scores [Q, k']), row i = topk(queries[i], k)."""
        qs = [self._q(q) for q in queries]
        nq = len(qs)
        kk = max(min(int(k), self._n), 1)
        idx = np.empty((max(nq, 1), kk), dtype=np.uint64)
        sc = np.empty((max(nq, 1), kk), dtype=np.float32)
        if nq == 0:
            return idx[:0], sc[:0]
        stride = max(max(q.shape[0] for q in qs), 1)
        packed = np.zeros((nq, stride, self._dim), dtype=np.float32)
        tq = np.zeros(nq, dtype=np.uint32)
        for i, q in enumerate(qs):
            if q.size and q.shape[1] != self._dim:
                raise InnrPanic(f"dimension mismatch (doc): query dim {q.shape[1]}, document dim {self._dim}")
            packed[i, :q.shape[0]] = q
            tq[i] = q.shape[0]
        out_k = C.c_size_t(0)
        st = stats if stats is not None else KnnStats()
        check(load().innr_maxsim_topk_multi(self._h, 1 if cosine else 0, C.c_void_p(packed.ctypes.data), nq,
                                            C.c_void_p(tq.ctypes.data), stride, self._dim, int(k), int(engine),
                                            C.c_void_p(idx.ctypes.data), C.c_void_p(sc.ctypes.data), C.byref(out_k),
                                            C.byref(st)))
        r = int(out_k.value)
        return idx[:, :r].copy(), sc[:, :r].copy()

    def close(self) -> None:
        if getattr(self, "_h", None):
            if getattr(self._ctx, "handle", None):
                load().innr_docs_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
