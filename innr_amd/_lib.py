"""ctypes binding of the C ABI (include/innr_hip.h) -> innr_amd/lib/libinnr_hip.so.

There is no CPU fallback anywhere in this package: if the shared library is missing, or no GPU is
visible, every operation raises. (The CPU oracle under oracle/ is test infrastructure and is never
imported from here.)
"""
from __future__ import annotations

import atexit
import contextlib
import ctypes as C
import os
import sys
import threading
import weakref
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("INNR_HIP_LIB_PATH") or os.path.join(_HERE, "lib", "libinnr_hip.so")  # (the override: A/B builds of tools/)

OK = 0
E_DIM_MISMATCH = -1
E_BAD_ARG = -2
E_OOM = -3
E_HIP = -4
E_RCCL = -5
E_UNSUPPORTED = -6

METRIC_DOT = 0
METRIC_L2SQ = 1
METRIC_COSINE = 2

KNN_AUTO = 0
KNN_EXACT = 1
KNN_MFMA = 2
KNN_MFMA_BF16 = 3  # bf16 filter + exact f32 re-score (dot, k <= 48); same results
KNN_MFMA_I8 = 4  # u8 code corpora: int8-MFMA filter (two int8 limbs per query value) + exact f32 re-score; same results

MAX_K = 240

GEN_EXAMPLE_LCG = 0
GEN_UNIFORM = 1


class InnrError(RuntimeError):
    """A failure reported by the HIP library (status < 0 other than a dimension mismatch)."""

    def __init__(self, status: int, msg: str):
        super().__init__(f"innr_hip status {status}: {msg}")
        self.status = status


class InnrPanic(AssertionError):
    """What the reference reports by panicking (assert_eq!/assert!): same condition, Python exception."""


class KnnStats(C.Structure):
    _fields_ = [("engine", C.c_int), ("queries_fallback", C.c_uint32), ("candidates_kept", C.c_uint32),
                ("gemm_ms", C.c_float), ("total_ms", C.c_float)]


_f32p = C.POINTER(C.c_float)
_u64p = C.POINTER(C.c_uint64)
_szp = C.POINTER(C.c_size_t)
_vp = C.c_void_p
_sz = C.c_size_t

# name -> (restype, argtypes): every symbol include/innr_hip.h declares (tests/test_abi.py checks both ways)
SIGNATURES = {
    "innr_ctx_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "innr_ctx_destroy": (None, [_vp]),
    "innr_ctx_set_stream": (C.c_int, [_vp, _vp]),
    "innr_ctx_synchronize": (C.c_int, [_vp]),
    "innr_ctx_set_option": (C.c_int, [_vp, C.c_char_p, C.c_long]),
    "innr_ctx_get_option": (C.c_int, [_vp, C.c_char_p, C.POINTER(C.c_long)]),
    "innr_last_error": (C.c_char_p, []),
    "innr_version": (C.c_char_p, []),
    "innr_batch_upload_colmajor": (C.c_int, [_vp, _vp, _sz, _sz, C.POINTER(_vp)]),
    "innr_batch_upload_rowmajor": (C.c_int, [_vp, _vp, _sz, _sz, C.POINTER(_vp)]),
    "innr_batch_generate": (C.c_int, [_vp, _sz, _sz, C.c_int, C.c_uint64, C.c_uint64, C.POINTER(_vp)]),
    "innr_batch_free": (None, [_vp]),
    "innr_batch_auto_engine": (C.c_int, [_vp, _sz]),
    "innr_batch_num_vectors": (_sz, [_vp]),
    "innr_batch_dimension": (_sz, [_vp]),
    "innr_batch_download_colmajor": (C.c_int, [_vp, _vp]),
    "innr_batch_set_index_base": (C.c_int, [_vp, C.c_uint64]),
    "innr_batch_scores": (C.c_int, [_vp, C.c_int, _vp, _sz, _vp, _vp]),
    "innr_batch_norms": (C.c_int, [_vp, _vp]),
    "innr_batch_knn": (C.c_int, [_vp, C.c_int, _vp, _sz, _sz, _sz, C.c_int, _vp, _vp, _szp, C.POINTER(KnnStats)]),
    "innr_batch_knn_dev": (C.c_int, [_vp, C.c_int, _vp, _sz, _sz, _sz, C.c_int, _vp, _vp, _szp, C.POINTER(KnnStats)]),
    "innr_batch_upload_u8": (C.c_int, [_vp, _vp, _sz, _sz, C.c_float, C.c_float, C.POINTER(_vp)]),
    "innr_batch_generate_u8": (C.c_int, [_vp, _sz, _sz, C.c_uint64, C.c_uint64, C.c_float, C.c_float, C.POINTER(_vp)]),
    "innr_batch_download_u8": (C.c_int, [_vp, _vp]),
    "innr_batch_scores_u8": (C.c_int, [_vp, _vp, _sz, _vp]),
    "innr_batch_knn_u8": (C.c_int, [_vp, _vp, _sz, _sz, _sz, C.c_int, _vp, _vp, _szp, C.POINTER(KnnStats)]),
    "innr_batch_knn_u8_dev": (C.c_int, [_vp, _vp, _sz, _sz, _sz, C.c_int, _vp, _vp, _szp, C.POINTER(KnnStats)]),
    "innr_quantize_u8": (None, [_vp, _sz, C.c_float, C.c_float, _vp]),
    "innr_batch_quantize_u8": (C.c_int, [_vp, C.c_float, C.c_float, C.POINTER(_vp)]),
    "innr_batch_prefix_view": (C.c_int, [_vp, _sz, C.POINTER(_vp)]),
    "innr_batch_minmax": (C.c_int, [_vp, _f32p, _f32p, C.POINTER(C.c_int)]),
    "innr_batch_rerank": (C.c_int, [_vp, C.c_int, _vp, _sz, _sz, _vp, _sz, _sz, _vp, _vp, _szp]),
    "innr_batch_rerank_dev": (C.c_int, [_vp, C.c_int, _vp, _sz, _sz, _vp, _sz, _sz, _vp, _vp, _szp]),
    "innr_mixed_dot_u8_f32": (C.c_float, [_vp, _vp, _sz]),
    "innr_maxsim_upload": (C.c_int, [_vp, _vp, _vp, _sz, _sz, _sz, C.POINTER(_vp)]),
    "innr_maxsim_generate": (C.c_int, [_vp, _sz, _sz, _sz, C.c_uint64, C.c_uint64, C.POINTER(_vp)]),
    "innr_docs_free": (None, [_vp]),
    "innr_docs_count": (_sz, [_vp]),
    "innr_docs_set_index_base": (C.c_int, [_vp, C.c_uint64]),
    "innr_maxsim_scores": (C.c_int, [_vp, C.c_int, _vp, _sz, _sz, _vp]),
    "innr_maxsim_topk": (C.c_int, [_vp, C.c_int, _vp, _sz, _sz, _sz, C.c_int, _vp, _vp, _szp, C.POINTER(KnnStats)]),
    "innr_maxsim_topk_multi": (C.c_int, [_vp, C.c_int, _vp, _sz, _vp, _sz, _sz, _sz, C.c_int, _vp, _vp, _szp,
                                        C.POINTER(KnnStats)]),
    "innr_batch_dimension_variance": (C.c_int, [_vp, _vp]),
    "innr_batch_knn_filtered": (C.c_int, [_vp, _vp, _sz, _sz, _vp, _vp, _vp, _szp]),
    "innr_batch_knn_reordered": (C.c_int, [_vp, _vp, _sz, _sz, _vp, _vp, _szp]),
    "innr_batch_l2_squared_pruning": (C.c_int, [_vp, _vp, _sz, C.c_float, _vp, _vp, _sz, _szp]),
    "innr_merge_topk_dev": (C.c_int, [_vp, C.c_int, _vp, _vp, _sz, _sz, _sz, _sz, _vp, _vp]),
    "innr_dot_f32": (C.c_float, [_vp, _vp, _sz]),
    "innr_cosine_f32": (C.c_float, [_vp, _vp, _sz]),
    "innr_l2sq_f32": (C.c_float, [_vp, _vp, _sz]),
    "innr_l1_f32": (C.c_float, [_vp, _vp, _sz]),
    "innr_hamming_u8": (C.c_uint32, [_vp, _vp, _sz]),
    "innr_slot_distance_u32": (C.c_float, [_vp, _vp, _sz]),
    "innr_maxsim_pair": (C.c_int, [_vp, _sz, _vp, _sz, _sz, C.c_int, _f32p]),
    "innr_batch_quantile_range": (C.c_int, [_vp, C.c_float, _f32p, _f32p, C.POINTER(C.c_int)]),
    "innr_batch_upload_u8_colmajor": (C.c_int, [_vp, _vp, _sz, _sz, C.c_float, C.c_float, C.POINTER(_vp)]),
    "innr_docs_shape": (C.c_int, [_vp, _szp, _szp, _szp, C.POINTER(C.c_int)]),
    "innr_docs_download": (C.c_int, [_vp, _vp, _vp]),
    "innr_comm_unique_id": (C.c_int, [_vp]),
    "innr_comm_create": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.POINTER(_vp)]),
    "innr_comm_attach": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.POINTER(_vp)]),
    "innr_comm_destroy": (None, [_vp]),
    "innr_comm_rank": (C.c_int, [_vp]),
    "innr_comm_world": (C.c_int, [_vp]),
    "innr_topk_block_words": (_sz, [_sz, _sz]),
    "innr_topk_pack_dev": (C.c_int, [_vp, _vp, _vp, C.c_uint64, C.c_uint64, _sz, _sz, _sz, _vp]),
    "innr_allgather_topk_dev": (C.c_int, [_vp, _vp, _sz, _sz, _vp]),
    "innr_merge_blocks_dev": (C.c_int, [_vp, C.c_int, _vp, _sz, _sz, _sz, _vp, _vp, _szp]),
    "innr_sharded_knn_dev": (C.c_int, [_vp, _vp, C.c_int, _vp, _sz, _sz, _sz, C.c_int, _vp, _vp, _szp, C.POINTER(KnnStats)]),
    "innr_sharded_knn": (C.c_int, [_vp, _vp, C.c_int, _vp, _sz, _sz, _sz, C.c_int, _vp, _vp, _szp, C.POINTER(KnnStats)]),
    "innr_sharded_maxsim": (C.c_int, [_vp, _vp, C.c_int, _vp, _sz, _sz, _sz, C.c_int, _vp, _vp, _szp, C.POINTER(KnnStats)]),
}
COMM_ID_BYTES = 128

_lib: Optional[C.CDLL] = None
_lock = threading.Lock()


def _preload_torch_rccl() -> None:
    """The library binds librccl at run time and prefers the copy already mapped into the process. PyTorch bundles its
    own (torch/lib/librccl.so, loaded when torch.distributed first needs it): map that copy now, so the library's
    communicator and torch's share one RCCL, whichever is used first. No torch installed: the system copy is used."""
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "librccl.so")
        if os.path.exists(path) and "INNR_RCCL_LIB" not in os.environ:
            os.environ["INNR_RCCL_LIB"] = path  # load_rccl() in api.hip opens this path first
    except Exception:
        pass


def _preload_torch_hip_runtime() -> None:
    """One HIP runtime per process. The PyTorch-ROCm wheel bundles its own libamdhip64.so (SONAME
    libamdhip64.so.7) and loads it by the name 'libamdhip64.so'; if /opt/rocm's copy is already mapped under the
    SONAME, the loader maps torch's copy as a SECOND runtime and torch then reports "No HIP GPUs are available".
    Mapping torch's copy first (no `import torch` needed) makes libinnr_hip.so's NEEDED libamdhip64.so.7 resolve
    to it, whichever of the two is used first. Without torch installed this is a no-op (system ROCm is used)."""
    if os.environ.get("INNR_HIP_SYSTEM_RUNTIME") == "1" or "torch" in sys.modules:
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(path):
            C.CDLL(path, mode=C.RTLD_GLOBAL)
    except Exception:
        pass


def load() -> C.CDLL:
    """Load libinnr_hip.so (built by __graft_entry__.build() / innr_amd/csrc/Makefile). Raises if absent."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). innr_amd has no CPU fallback.")
        _preload_torch_hip_runtime()
        _preload_torch_rccl()
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError here = ABI drift, fail loudly
            fn.restype = res
            fn.argtypes = args
        _lib = L
        return L


def last_error() -> str:
    return load().innr_last_error().decode("utf-8", "replace")


def check(status: int) -> None:
    if status == OK:
        return
    msg = last_error()
    if status == E_DIM_MISMATCH:
        raise InnrPanic(msg)
    raise InnrError(status, msg)


class Context:
    """One GPU (innr_ctx). One per process in the multi-GPU layout (one process per GPU)."""

    def __init__(self, device: int = 0):
        L = load()
        h = _vp()
        check(L.innr_ctx_create(int(device), C.byref(h)))
        self.handle = h
        self.device = int(device)
        self.stream = None  # the ctx's private stream until set_stream()
        self._children = weakref.WeakSet()  # device objects that hold a raw pointer to this ctx

    def set_stream(self, stream_ptr: int | None) -> None:
        check(load().innr_ctx_set_stream(self.handle, _vp(stream_ptr or 0)))
        self.stream = int(stream_ptr or 0)  # what the library's kernels are ordered with (None until the first call)

    def bind_torch_stream(self) -> None:
        """Run the library's kernels on torch's CURRENT stream of this device, so that they are ordered with the caller's
        torch kernels and collectives (queries produced by torch, results consumed by torch) without extra events. A
        no-op when already bound to it. Called by every device-pointer path (innr_amd.dist)."""
        import torch
        cur = int(torch.cuda.current_stream(self.device).cuda_stream)
        if getattr(self, "stream", None) != cur:
            self.set_stream(cur)

    def synchronize(self) -> None:
        check(load().innr_ctx_synchronize(self.handle))

    def set_option(self, name: str, value: int) -> None:
        """A tuning / experiment switch of this context (include/innr_hip.h: innr_ctx_set_option). Never changes a result."""
        check(load().innr_ctx_set_option(self.handle, name.encode(), int(value)))

    def get_option(self, name: str) -> int:
        v = C.c_long(0)
        check(load().innr_ctx_get_option(self.handle, name.encode(), C.byref(v)))
        return int(v.value)

    @contextlib.contextmanager
    def option(self, name: str, value: int):
        """`with ctx.option("i8_two_limb", 1): ...` -- set for the block, restored afterwards."""
        old = self.get_option(name)
        self.set_option(name, value)
        try:
            yield
        finally:
            self.set_option(name, old)

    def close(self) -> None:
        """Free every batch created on this context, then the context (a batch must not outlive its ctx)."""
        if getattr(self, "handle", None):
            for child in list(self._children):
                child.close()
            load().innr_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx: Optional[Context] = None


def default_context() -> Context:
    """Process-wide context on LOCAL_RANK's GPU (or device 0)."""
    global _default_ctx
    if _default_ctx is None:
        dev = int(os.environ.get("INNR_HIP_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        _default_ctx = Context(dev)
        atexit.register(_default_ctx.close)  # before interpreter teardown randomises __del__ order
    return _default_ctx
