"""Backend introspection, the counterpart of innr's `backend` module (src/backend.rs:20-67): which code path a call
will take. The reference reports its SIMD family per input length; here the choice is between the two device engines
(per batch and query count) and the host functions of the pairwise surface."""
from __future__ import annotations

import enum

from . import _lib


class Backend(enum.Enum):
    HIP_EXACT = "hip-exact"      # bit-exact scan engine (HBM-bound)
    HIP_MFMA = "hip-mfma"        # f32 MFMA GEMM + fused top-k + exact re-score
    HOST_PORTABLE = "portable"   # per-pair host functions in the reference's portable order

    def __str__(self) -> str:
        return self.value


def batch_backend(batch, num_queries: int = 1) -> Backend:
    """Engine a `batch_knn*` call with KNN_AUTO takes for `num_queries` queries on this batch."""
    e = _lib.load().innr_batch_auto_engine(batch._h, int(num_queries))
    return Backend.HIP_MFMA if e == _lib.KNN_MFMA else Backend.HIP_EXACT


def dense_backend(length: int = 0) -> Backend:
    """Pairwise dot/cosine/l2/l1 (`distance`, one-pair `maxsim`): always the host's portable path."""
    return Backend.HOST_PORTABLE


def version() -> str:
    return _lib.load().innr_version().decode()
