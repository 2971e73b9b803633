import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Load (and, if its source is newer, re-make) the CPU oracle now, before any test can have initialised the GPU: a
    # `make` child started later would be an exec from a HIP-initialised process, which the GPU boxes forbid.
    import oracle
    oracle.lib()


def hooks_lib():
    """innr_amd/lib/libinnr_hip_testhooks.so: the product sources built with -DINNR_TEST_HOOKS (make hooks) -- the innrdbg_*
    layout hooks live only there. It takes handles made by the product library (same structs, same HIP runtime)."""
    import ctypes as C
    global _HOOKS
    if _HOOKS is None:
        path = os.path.join(ROOT, "innr_amd", "lib", "libinnr_hip_testhooks.so")
        if not os.path.exists(path):
            pytest.skip("libinnr_hip_testhooks.so not built (make -C innr_amd/csrc hooks)")
        _HOOKS = C.CDLL(path)
    return _HOOKS


_HOOKS = None


@pytest.fixture
def ctx_option():
    """set a tuning option of the default context for one test (innr_ctx_set_option), restored afterwards"""
    from innr_amd import _lib
    ctx = _lib.default_context()
    saved = []

    def _set(name, value):
        saved.append((name, ctx.get_option(name)))
        ctx.set_option(name, value)

    yield _set
    for name, old in reversed(saved):
        ctx.set_option(name, old)


def _have_gpu() -> bool:
    # /dev/kfd is what the HIP runtime needs; avoids initialising anything in CPU-only runs.
    return os.path.exists("/dev/kfd")


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container (/dev/kfd absent)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
