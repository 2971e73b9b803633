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
