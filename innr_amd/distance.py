"""Host-side mirror of innr's `distance` module (reference: src/distance.rs:66-143).

`Distance<T>::eval(a, b) -> f32`, smaller = more similar. These are PAIRWISE metrics that an external index
(hnsw_rs via anndists, distance.rs:148-193) calls hundreds of times per query on single pairs, so they stay on
the host: plain functions of libinnr_hip.so in the reference's portable arithmetic order.
The batched scans are what runs on the GPU (innr_amd.batch).
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from ._lib import InnrPanic, load


def _pair(fn_name: str, what: str, a, b) -> float:
    a = np.ascontiguousarray(a, dtype=np.float32).reshape(-1)
    b = np.ascontiguousarray(b, dtype=np.float32).reshape(-1)
    if a.size != b.size:  # assert_eq!(a.len(), b.len(), "innr::dot: slice length mismatch ...") dense.rs:57-63
        raise InnrPanic(f"innr::{what}: slice length mismatch ({a.size} vs {b.size})")
    return float(getattr(load(), fn_name)(C.c_void_p(a.ctypes.data), C.c_void_p(b.ctypes.data), a.size))


def dot(a, b) -> float: return _pair("innr_dot_f32", "dot", a, b)
def cosine(a, b) -> float: return _pair("innr_cosine_f32", "cosine", a, b)
def l2_distance_squared(a, b) -> float: return _pair("innr_l2sq_f32", "l2_distance_squared", a, b)
def l2_distance(a, b) -> float: return float(np.sqrt(np.float32(l2_distance_squared(a, b))))
def l1_distance(a, b) -> float: return _pair("innr_l1_f32", "l1_distance", a, b)


def _prefix(a, b, prefix_len: int):
    a = np.asarray(a, dtype=np.float32).reshape(-1)
    b = np.asarray(b, dtype=np.float32).reshape(-1)
    end = min(int(prefix_len), a.size, b.size)  # prefix_len.min(a.len()).min(b.len()), dense.rs:437,459
    return a[:end], b[:end]


def matryoshka_dot(a, b, prefix_len: int) -> float:
    """dense.rs:436-440: dot of the first prefix_len dimensions (clamped to the shorter slice: never a length panic)."""
    return dot(*_prefix(a, b, prefix_len))


def matryoshka_cosine(a, b, prefix_len: int) -> float:
    """dense.rs:458-462: cosine of the first prefix_len dimensions."""
    return cosine(*_prefix(a, b, prefix_len))


class Distance:
    """distance.rs:66-69"""
    def eval(self, a, b) -> float:  # pragma: no cover - interface
        raise NotImplementedError


class DistCosine(Distance):
    """distance.rs:73-80: 1 - cosine_similarity, range [0, 2]."""
    def eval(self, a, b) -> float: return float(np.float32(1.0) - np.float32(cosine(a, b)))


class DistDot(Distance):
    """distance.rs:85-92: negated dot product."""
    def eval(self, a, b) -> float: return -dot(a, b)


class DistL2(Distance):
    """distance.rs:96-103: Euclidean distance."""
    def eval(self, a, b) -> float: return l2_distance(a, b)


class DistL1(Distance):
    """distance.rs:107-114: Manhattan distance."""
    def eval(self, a, b) -> float: return l1_distance(a, b)


def hamming_distance(a, b) -> int:
    """quant.rs:220-258: differing bits between two byte-packed binary vectors; panics on a length mismatch."""
    x = np.ascontiguousarray(a, dtype=np.uint8).reshape(-1)
    y = np.ascontiguousarray(b, dtype=np.uint8).reshape(-1)
    if x.size != y.size:
        raise InnrPanic(f"innr::hamming_distance: slice length mismatch ({x.size} vs {y.size})")
    return int(load().innr_hamming_u8(C.c_void_p(x.ctypes.data), C.c_void_p(y.ctypes.data), x.size))


def jaccard_distance(a, b) -> float:
    """slot.rs:392-405: fraction of differing u32 slots (0.0 for empty inputs); panics on a length mismatch."""
    x = np.ascontiguousarray(a, dtype=np.uint32).reshape(-1)
    y = np.ascontiguousarray(b, dtype=np.uint32).reshape(-1)
    if x.size != y.size:
        raise InnrPanic(f"innr::jaccard_distance: slice length mismatch ({x.size} vs {y.size})")
    return float(load().innr_slot_distance_u32(C.c_void_p(x.ctypes.data), C.c_void_p(y.ctypes.data), x.size))


class DistHamming(Distance):
    """distance.rs:116-126"""
    def eval(self, a, b) -> float: return float(hamming_distance(a, b))


class DistSlotU32(Distance):
    """distance.rs:128-143"""
    def eval(self, a, b) -> float: return jaccard_distance(a, b)
