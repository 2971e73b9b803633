"""Two implementations of the same innr API surface, so one KAT body checks both.

* OracleBackend -- oracle/ (CPU restatement; the checker).
* HipBackend    -- innr_amd (the product: ctypes -> C ABI -> HIP kernels). GPU only.

Both expose the reference's function names (batch.rs / scalar.rs / maxsim.rs) with numpy in/out:
knn functions return (indices uint64[k'], scores float32[k']).
"""
from __future__ import annotations

import numpy as np


class OracleBackend:
    name = "oracle"

    def __init__(self):
        import oracle
        self.o = oracle

    # batch handles are the dimension-major (dim, n) array itself
    def from_rows(self, rows):
        rows = np.asarray(rows, dtype=np.float32)
        if rows.size == 0 and rows.ndim < 2:
            return np.empty((0, 0), np.float32)
        return self.o.from_rows(rows)

    def num_vectors(self, b): return b.shape[1]
    def dimension(self, b): return b.shape[0]
    def batch_dot(self, q, b): return self.o.batch_dot(q, b)
    def batch_l2_squared(self, q, b): return self.o.batch_l2_squared(q, b)
    def batch_norms(self, b): return self.o.batch_norms(b)
    def batch_cosine(self, q, b, norms): return self.o.batch_cosine(q, b, norms)
    def batch_knn(self, q, b, k): return self.o.batch_knn(q, b, k)
    def batch_knn_dot(self, q, b, k): return self.o.batch_knn_dot(q, b, k)
    def batch_knn_cosine(self, q, b, k): return self.o.batch_knn_cosine(q, b, k)
    def batch_knn_filtered(self, q, b, k, pred):
        mask = np.array([1 if pred(i) else 0 for i in range(b.shape[1])], dtype=np.uint8)
        return self.o.batch_knn_filtered(q, b, k, mask)
    def batch_knn_reordered(self, q, b, k): return self.o.batch_knn_reordered(q, b, k)
    def batch_l2_squared_pruning(self, q, b, t): return self.o.batch_l2_squared_pruning(q, b, t)
    def maxsim(self, q, d): return self.o.maxsim(q, d)
    def maxsim_cosine(self, q, d): return self.o.maxsim_cosine(q, d)
    def batch_knn_u8(self, q, codes, alpha, offset, k):
        p = self.o.QParams(alpha, offset)
        return self.o.batch_knn_u8(q, codes, p, k)


class HipBackend:
    name = "hip"

    def __init__(self, engine=None):
        import innr_amd
        from innr_amd import batch
        self.m = innr_amd
        self.batch = batch
        self.engine = innr_amd.KNN_AUTO if engine is None else engine

    @property
    def ms(self):
        from innr_amd import maxsim
        return maxsim

    @property
    def scalar(self):
        from innr_amd import scalar
        return scalar

    def from_rows(self, rows):
        rows = [np.asarray(r, dtype=np.float32) for r in rows]
        return self.batch.VerticalBatch.from_rows(rows)

    def num_vectors(self, b): return b.num_vectors()
    def dimension(self, b): return b.dimension()
    def batch_dot(self, q, b): return self.batch.batch_dot(q, b)
    def batch_l2_squared(self, q, b): return self.batch.batch_l2_squared(q, b)
    def batch_norms(self, b): return self.batch.batch_norms(b)
    def batch_cosine(self, q, b, norms): return self.batch.batch_cosine(q, b, norms)

    @staticmethod
    def _r(res):
        return np.asarray(res.indices, dtype=np.uint64), np.asarray(res.scores, dtype=np.float32)

    def batch_knn(self, q, b, k): return self._r(self.batch.batch_knn(q, b, k))
    def batch_knn_dot(self, q, b, k): return self._r(self.batch.batch_knn_dot(q, b, k, engine=self.engine))
    def batch_knn_cosine(self, q, b, k): return self._r(self.batch.batch_knn_cosine(q, b, k, engine=self.engine))
    def batch_knn_filtered(self, q, b, k, pred): return self._r(self.batch.batch_knn_filtered(q, b, k, pred))
    def batch_knn_reordered(self, q, b, k): return self._r(self.batch.batch_knn_reordered(q, b, k))
    def batch_l2_squared_pruning(self, q, b, t):
        pairs = self.batch.batch_l2_squared_pruning(q, b, t)
        return (np.array([p[0] for p in pairs], dtype=np.uint64), np.array([p[1] for p in pairs], dtype=np.float32))
    def maxsim(self, q, d): return self.ms.maxsim(q, d)
    def maxsim_cosine(self, q, d): return self.ms.maxsim_cosine(q, d)
    def batch_knn_u8(self, q, codes, alpha, offset, k):
        params = self.scalar.QuantizationParams(alpha, offset)
        corpus = [self.scalar.QuantizedU8(np.asarray(c, dtype=np.uint8), len(c)) for c in codes]
        res = self.scalar.batch_knn_u8(q, corpus, params, k)
        return (np.array([r[0] for r in res], dtype=np.uint64), np.array([r[1] for r in res], dtype=np.float32))
