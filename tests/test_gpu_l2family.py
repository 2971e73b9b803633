"""GPU parity tests, part 3: the L2 variants of src/batch.rs on the device -- batch_knn_filtered (:820-882),
batch_knn_reordered (:621-659) with batch_dimension_variance (:572-592), batch_l2_squared_pruning (:320-365)."""
from __future__ import annotations

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle
import kat_cases as K
from backends import HipBackend
from test_gpu_exact import _corpus, _queries, bits_equal, same_knn


@pytest.fixture(scope="module")
def B():
    from innr_amd import batch
    return batch


@pytest.mark.parametrize("kat", K.L2_FAMILY_KATS, ids=lambda f: f.__name__)
def test_reference_kat_l2_family(kat):
    kat(HipBackend())


@pytest.mark.parametrize("n,dim", [(1, 3), (2, 1), (300, 17), (5000, 64), (20_001, 128)])
def test_dimension_variance_bit_exact(B, n, dim):
    rows, data = _corpus(n, dim, 4, uniform=True)
    vb = B.VerticalBatch.from_rows(rows)
    assert bits_equal(B.batch_dimension_variance(vb), oracle.batch_dimension_variance(data))


@pytest.mark.parametrize("n,dim,k", [(50, 16, 5), (200, 64, 10), (3000, 33, 40), (10_000, 128, 10)])
def test_reordered_equals_oracle(B, n, dim, k):
    rows, data = _corpus(n, dim, 21, uniform=True)
    vb = B.VerticalBatch.from_rows(rows)
    for q in _queries(3, dim, 5, uniform=True):
        r = B.batch_knn_reordered(q, vb, k)
        oi, os_ = oracle.batch_knn_reordered(q, data, k)
        assert r.indices == oi.tolist() and bits_equal(np.float32(r.scores), os_)
        # the reference's claim (batch.rs:607): same neighbours as the plain exact kNN on well-separated data
        assert set(r.indices) == set(B.batch_knn(q, vb, k).indices)


@pytest.mark.parametrize("n,dim,k,mod", [(100, 2, 5, 2), (5000, 32, 10, 7), (5000, 32, 100, 3), (4099, 16, 240, 17)])
def test_filtered_equals_oracle(B, n, dim, k, mod):
    rows, data = _corpus(n, dim, 8, uniform=True)
    vb = B.VerticalBatch.from_rows(rows)
    pred = lambda i: i % mod == 0  # noqa: E731
    mask = np.array([1 if pred(i) else 0 for i in range(n)], dtype=np.uint8)
    for q in _queries(3, dim, 6, uniform=True):
        r = B.batch_knn_filtered(q, vb, k, pred)
        oi, os_ = oracle.batch_knn_filtered(q, data, k, mask)
        assert r.indices == oi.tolist() and bits_equal(np.float32(r.scores), os_)
    # fewer passing than k -> min(k, passing) results (batch.rs:849)
    r = B.batch_knn_filtered(rows[0], vb, 50, lambda i: i < 3)
    assert sorted(r.indices) == [0, 1, 2] and r.indices[0] == 0


@pytest.mark.parametrize("n,dim", [(10, 2), (1000, 16), (70_000, 24)])
def test_pruning_equals_oracle(B, n, dim):
    rows, data = _corpus(n, dim, 9, uniform=True)
    vb = B.VerticalBatch.from_rows(rows)
    q = _queries(1, dim, 3, uniform=True)[0]
    d = oracle.batch_l2_squared(q, data)
    for thr in (0.0, float(np.percentile(d, 1)), float(np.median(d)), 1e9):
        got = B.batch_l2_squared_pruning(q, vb, thr)
        oi, od = oracle.batch_l2_squared_pruning(q, data, thr)
        assert [g[0] for g in got] == oi.tolist()
        assert bits_equal(np.float32([g[1] for g in got]), od)
