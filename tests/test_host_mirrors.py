"""CPU tests of the host-side surface of the product: the pairwise metrics / one-pair maxsim that stay on the host
by design (distance.rs:66-143, maxsim.rs:96-194) and the TopK tracker (topk.rs). They live in libinnr_hip.so /
innr_amd (NOT in oracle/) and are compared bit-for-bit with the oracle and with the reference's KATs."""
from __future__ import annotations

import math

import numpy as np
import pytest

import oracle
import kat_cases as K


@pytest.fixture(scope="module")
def mods():
    import __graft_entry__ as g
    g.build()
    from innr_amd import distance, maxsim, topk
    return distance, maxsim, topk


class _HostBackend:
    def __init__(self, ms):
        self.ms = ms

    def maxsim(self, q, d): return self.ms.maxsim(q, d)
    def maxsim_cosine(self, q, d): return self.ms.maxsim_cosine(q, d)


@pytest.mark.parametrize("kat", K.MAXSIM_KATS, ids=lambda f: f.__name__)
def test_maxsim_reference_kats(mods, kat):  # src/maxsim.rs:196-382, examples/maxsim_colbert.rs:64,104
    kat(_HostBackend(mods[1]))


def test_pairwise_bit_exact_vs_oracle(mods):
    D, M, _ = mods
    rng = np.random.default_rng(0)
    for n in (0, 1, 3, 4, 5, 16, 17, 63, 64, 65, 128, 769):
        a = rng.normal(size=n).astype(np.float32)
        b = rng.normal(size=n).astype(np.float32)
        assert D.dot(a, b) == oracle.dot_portable(a, b)
        assert D.cosine(a, b) == oracle.cosine_portable(a, b)
        assert D.l2_distance_squared(a, b) == oracle.l2_distance_squared_portable(a, b)
        assert D.l1_distance(a, b) == oracle.l1_distance_portable(a, b)
        assert D.DistCosine().eval(a, b) == oracle.dist_cosine(a, b)
        assert D.DistDot().eval(a, b) == oracle.dist_dot(a, b)
        assert D.DistL2().eval(a, b) == oracle.dist_l2(a, b)
        assert D.DistL1().eval(a, b) == oracle.dist_l1(a, b)
        ua, ub = rng.integers(0, 256, n, dtype=np.uint8), rng.integers(0, 256, n, dtype=np.uint8)
        assert D.DistHamming().eval(ua, ub) == oracle.dist_hamming(ua, ub) == float(np.unpackbits(ua ^ ub).sum())
        sa, sb = rng.integers(0, 3, n, dtype=np.uint32), rng.integers(0, 3, n, dtype=np.uint32)
        assert D.DistSlotU32().eval(sa, sb) == oracle.dist_slot_u32(sa, sb)
    for nq, nd, dim in ((1, 1, 1), (5, 9, 33), (32, 64, 128), (3, 2, 7)):
        q = rng.normal(size=(nq, dim)).astype(np.float32)
        d = rng.normal(size=(nd, dim)).astype(np.float32)
        assert M.maxsim(q, d) == oracle.maxsim(q, d)
        assert M.maxsim_cosine(q, d) == oracle.maxsim_cosine(q, d)


def test_pairwise_panics_and_guards(mods):
    D, M, _ = mods
    from innr_amd import InnrPanic
    with pytest.raises(InnrPanic):
        D.hamming_distance([1, 2], [1])          # quant.rs:221-227
    with pytest.raises(InnrPanic):
        D.jaccard_distance([1, 2, 3], [1, 2])    # slot.rs:393-399
    assert D.jaccard_distance([], []) == 0.0     # slot.rs:401-403
    with pytest.raises(InnrPanic):
        D.dot([1.0, 2.0], [1.0])  # dense.rs:57-63
    with pytest.raises(InnrPanic):
        M.maxsim([[1.0, 2.0]], [[1.0]])  # maxsim.rs:107-110 "dimension mismatch (doc)"
    with pytest.raises(InnrPanic):
        M.maxsim([[1.0, 2.0], [1.0]], [[1.0, 1.0]])  # maxsim.rs:103-106 "dimension mismatch (query)"
    assert D.cosine([1.0, 0.0], [0.0, 0.0]) == 0.0  # zero-vector guard, dense.rs:341-345
    assert D.DistDot().eval([1, 2, 3], [4, 5, 6]) == -32.0
    assert abs(D.DistL2().eval([0, 0], [3, 4]) - 5.0) < 1e-6


def test_topk_reference_kats(mods):  # src/topk.rs:191-346
    T = mods[2].TopK
    t = T(3)
    for i, d in [(0, 1.5), (1, 0.3), (2, 2.0), (3, 0.8), (4, 5.0)]:
        t.insert(i, d)
    assert len(t) == 3 and t.into_sorted() == [(1, np.float32(0.3)), (3, np.float32(0.8)), (0, 1.5)]
    t = T(3)
    assert t.threshold() == math.inf
    for i, d, th in [(0, 1.0, math.inf), (1, 2.0, math.inf), (2, 3.0, 3.0), (3, 1.5, 2.0), (4, 0.5, 1.5), (5, 10.0, 1.5)]:
        t.insert(i, d)
        assert t.threshold() == th
    t = T(3)
    for i in range(4):
        t.insert(i, 1.0)
    r = t.into_sorted()
    assert len(r) == 3 and all(d == 1.0 for _, d in r) and 3 not in [i for i, _ in r]
    t = T(10)
    for i in range(10_000):
        t.insert(i, float(i))
    assert t.into_sorted() == [(i, float(i)) for i in range(10)]
    t = T(2)  # NaN does not poison the gate (topk.rs:191-209)
    t.insert(0, float("nan")); t.insert(1, 1.0); t.insert(2, 0.5)
    assert sorted(i for i, _ in t.into_sorted()) == [1, 2]
    from innr_amd import InnrPanic
    with pytest.raises(InnrPanic):
        T(0)


def test_topk_matches_oracle_on_random_streams(mods):
    T = mods[2].TopK
    rng = np.random.default_rng(3)
    for k in (1, 2, 7, 32):
        vals = rng.integers(0, 40, size=500).astype(np.float32) / np.float32(4.0)  # many exact ties
        a, b = T(k), oracle.TopK(k)
        for i, v in enumerate(vals):
            a.insert(i, float(v)); b.insert(i, float(v))
            assert a.threshold() == b.threshold()
        assert a.into_sorted() == b.into_sorted()


def test_matryoshka_reference_kats(mods):  # src/dense.rs:681-703, 928-1003
    D, _, _ = mods
    for impl_dot, impl_cos, full_dot, full_cos in (
            (D.matryoshka_dot, D.matryoshka_cosine, D.dot, D.cosine),
            (oracle.matryoshka_dot, oracle.matryoshka_cosine, oracle.dot_portable, oracle.cosine_portable)):
        a, b = [1.0, 2.0, 3.0, 4.0, 5.0], [5.0, 4.0, 3.0, 2.0, 1.0]
        for prefix, want in zip((1, 2, 3, 4, 5), (5.0, 13.0, 22.0, 30.0, 35.0)):  # test_matryoshka_dot_equals_prefix_dot
            assert impl_dot(a, b, prefix) == want == full_dot(a[:prefix], b[:prefix])
        assert impl_dot([1.0, 0.0, -1.0], [2.0, 3.0, 4.0], 3) == -2.0          # ..._full_prefix_equals_dot
        assert impl_dot([1.0, 2.0], [3.0, 4.0], 100) == 11.0                    # ..._prefix_longer_than_vec_clips
        a, b = [1.0, 2.0, 3.0, 4.0], [4.0, 3.0, 2.0, 1.0]
        for prefix in (1, 2, 3, 4):                                             # test_matryoshka_cosine_equals_prefix_cosine
            assert impl_cos(a, b, prefix) == full_cos(a[:prefix], b[:prefix])
        assert abs(impl_cos([1.0, 0.0], [0.0, 1.0], 2)) < 1e-6                  # ..._full_prefix_equals_cosine
        assert abs(impl_cos([3.0, -99.0, -99.0], [5.0, 1.0, 1.0], 1) - 1.0) < 1e-5  # ..._prefix_one
        q, d1, d2, d3 = [1.0, 0.5, 0.2, 0.1], [0.9, 0.4, 0.1, 0.05], [0.1, 0.1, 0.1, 0.1], [-0.5, -0.2, 0.0, 0.0]
        assert full_cos(q, d1) > full_cos(q, d2) > full_cos(q, d3)              # test_matryoshka_ranking_preservation
        assert impl_cos(q, d1, 2) > impl_cos(q, d2, 2) > impl_cos(q, d3, 2)
    rng = np.random.default_rng(5)
    for n, prefix in ((0, 4), (7, 0), (33, 16), (128, 128), (100, 64), (9, 200)):
        a = rng.normal(size=n).astype(np.float32)
        b = rng.normal(size=n + 3).astype(np.float32)  # unequal lengths clip, never panic (dense.rs:437)
        assert D.matryoshka_dot(a, b, prefix) == oracle.matryoshka_dot(a, b, prefix)
        assert D.matryoshka_cosine(a, b, prefix) == oracle.matryoshka_cosine(a, b, prefix)
