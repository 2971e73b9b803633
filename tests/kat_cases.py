"""Known-answer tests lifted from the reference's OWN tests (inputs + expected values only).

Each function takes a backend `be` (tests/backends.py) and mirrors one reference test, cited by
file:line (relative to /root/reference). They run against the CPU oracle in test_oracle_kat.py
(pinning the oracle) and against the HIP path in test_gpu_kat.py (parity on the same vectors).
"""
from __future__ import annotations

import math

import numpy as np

F = np.float32


def _sin_rows(n, dim):
    # ((i*7 + d*3) as f32).sin()  -- closed-form generator used by batch.rs:1221 / batch_tests.rs:414
    i = np.arange(n, dtype=np.float32)[:, None]
    d = np.arange(dim, dtype=np.float32)[None, :]
    return np.sin((i * F(7) + d * F(3)).astype(np.float32)).astype(np.float32)


# ---------------------------------------------------------------------------- batch.rs unit tests
def kat_batch_l2_squared(be):  # src/batch.rs:903-917
    b = be.from_rows([[0, 0, 0], [1, 0, 0], [0, 1, 0]])
    d = be.batch_l2_squared([1, 1, 0], b)
    assert abs(d[0] - 2.0) < 1e-6 and abs(d[1] - 1.0) < 1e-6 and abs(d[2] - 1.0) < 1e-6


def kat_batch_l2_known_value(be):  # tests/batch_tests.rs:126-138
    b = be.from_rows([[0, 0, 0]])
    d = be.batch_l2_squared([3, 4, 0], b)
    assert abs(d[0] - 25.0) < 1e-6


def kat_batch_dot(be):  # src/batch.rs:930-939
    b = be.from_rows([[1, 0], [0, 1], [1, 1]])
    d = be.batch_dot([1, 2], b)
    assert abs(d[0] - 1.0) < 1e-6 and abs(d[1] - 2.0) < 1e-6 and abs(d[2] - 3.0) < 1e-6


def kat_batch_dot_orthogonal(be):  # tests/batch_tests.rs:141-156
    b = be.from_rows([[1, 0, 0], [0, 1, 0], [0, 0, 1]])
    d = be.batch_dot([1, 0, 0], b)
    assert abs(d[0] - 1.0) < 1e-6 and abs(d[1]) < 1e-6 and abs(d[2]) < 1e-6


def kat_batch_dot_zero_query(be):  # src/batch.rs:1138-1146
    b = be.from_rows([[1, 2], [3, 4]])
    d = be.batch_dot([0, 0], b)
    assert list(d) == [0.0, 0.0]


def kat_batch_norms(be):  # src/batch.rs:1101-1108
    b = be.from_rows([[3, 4], [0, 0], [1, 0]])
    n = be.batch_norms(b)
    assert abs(n[0] - 5.0) < 1e-6 and abs(n[1]) < 1e-6 and abs(n[2] - 1.0) < 1e-6


def kat_batch_cosine(be):  # tests/batch_tests.rs:159-186 (+ src/batch.rs:1000-1010)
    b = be.from_rows([[1, 0], [0, 1], [1, 1], [-1, 0]])
    c = be.batch_cosine([1, 0], b, be.batch_norms(b))
    assert abs(c[0] - 1.0) < 1e-6 and abs(c[1]) < 1e-6
    assert abs(c[2] - 1.0 / math.sqrt(2.0)) < 1e-5 and abs(c[3] + 1.0) < 1e-6


def kat_batch_cosine_zero_query(be):  # src/batch.rs:1158-1167, tests/batch_tests.rs:476-488
    b = be.from_rows([[1, 0], [0, 1]])
    c = be.batch_cosine([0, 0], b, be.batch_norms(b))
    assert list(c) == [0.0, 0.0]


def kat_batch_cosine_zero_norm_vector(be):  # src/batch.rs:1170-1179
    b = be.from_rows([[1, 0], [0, 0]])
    c = be.batch_cosine([1, 0], b, be.batch_norms(b))
    assert abs(c[0] - 1.0) < 1e-6 and c[1] == 0.0


def kat_batch_knn(be):  # src/batch.rs:952-968
    b = be.from_rows([[0, 0], [1, 0], [2, 0], [3, 0]])
    idx, sc = be.batch_knn([0.5, 0.0], b, 2)
    assert len(idx) == 2 and set(idx.tolist()) == {0, 1}


def kat_batch_knn_k_zero_empty_klarge(be):  # src/batch.rs:1422-1448
    b = be.from_rows([[1, 0], [0, 1]])
    idx, sc = be.batch_knn([1, 0], b, 0)
    assert len(idx) == 0 and len(sc) == 0
    e = be.from_rows([])
    idx, _ = be.batch_knn([], e, 5)
    assert len(idx) == 0
    b = be.from_rows([[1.0], [2.0]])
    idx, _ = be.batch_knn([1.5], b, 10)
    assert len(idx) == 2


def kat_batch_knn_sorted(be):  # src/batch.rs:1451-1468
    b = be.from_rows([[10, 0], [1, 0], [5, 0], [0, 0]])
    idx, sc = be.batch_knn([0, 0], b, 4)
    assert all(sc[i] <= sc[i + 1] for i in range(3)) and idx[0] == 3


def kat_knn_returns_k(be):  # tests/batch_tests.rs:220-230
    b = be.from_rows([[float(i), 0.0] for i in range(100)])
    for k in (1, 5, 10, 50, 100):
        idx, sc = be.batch_knn([50.0, 0.0], b, k)
        assert len(idx) == k and len(sc) == k


def kat_knn_exact_match(be):  # tests/batch_tests.rs:251-266
    b = be.from_rows([[0, 0], [1, 0], [0, 1], [1, 1]])
    idx, sc = be.batch_knn([0, 1], b, 1)
    assert idx[0] == 2 and sc[0] < 1e-6


def kat_batch_knn_dot_basic(be):  # src/batch.rs:1190-1202
    b = be.from_rows([[1, 0], [0, 1], [-1, 0]])
    idx, sc = be.batch_knn_dot([1, 0], b, 2)
    assert idx[0] == 0 and abs(sc[0] - 1.0) < 1e-6


def kat_batch_knn_dot_sorted(be):  # src/batch.rs:1204-1214
    b = be.from_rows([[0.5, 0.5], [1, 0], [0, 1]])
    _, sc = be.batch_knn_dot([1, 0], b, 3)
    assert all(sc[i] >= sc[i + 1] for i in range(2))


def kat_batch_knn_cosine_basic(be):  # src/batch.rs:1300-1315
    b = be.from_rows([[1, 0], [0, 1], [-1, 0]])
    idx, sc = be.batch_knn_cosine([1, 0], b, 2)
    assert list(idx) == [0, 1] and abs(sc[0] - 1.0) < 1e-5 and abs(sc[1]) < 1e-5


def kat_batch_knn_cosine_sorted(be):  # src/batch.rs:1325-1346
    b = be.from_rows([[0.1, 1.0], [1.0, 0.0], [0.5, 0.5]])
    idx, sc = be.batch_knn_cosine([1, 0], b, 3)
    assert all(sc[i] >= sc[i + 1] for i in range(2)) and idx[0] == 1


def kat_batch_knn_cosine_empty(be):  # src/batch.rs:1317-1322
    idx, _ = be.batch_knn_cosine([], be.from_rows([]), 5)
    assert len(idx) == 0


def kat_reordered_matches_exact(be):  # src/batch.rs:1221-1239 and tests/batch_tests.rs:414-426
    for n, dim, k in ((50, 16, 5), (200, 64, 10)):
        b = be.from_rows(_sin_rows(n, dim))
        q = np.cos(np.arange(dim, dtype=np.float32) * F(0.1)).astype(np.float32)
        ei, es = be.batch_knn(q, b, k)
        ri, rs = be.batch_knn_reordered(q, b, k)
        assert list(ei) == list(ri)
        assert np.all(np.abs(es - rs) < 1e-4)


def kat_cosine_knn_matches_dot_knn_normalized(be):  # tests/batch_tests.rs:429-458
    raw = _sin_rows(50, 8)
    vec = (raw / np.sqrt((raw * raw).sum(axis=1, dtype=np.float32))[:, None]).astype(np.float32)
    q = np.cos(np.arange(8, dtype=np.float32) * F(0.3)).astype(np.float32)
    q = (q / np.sqrt((q * q).sum(dtype=np.float32))).astype(np.float32)
    b = be.from_rows(vec)
    ci, _ = be.batch_knn_cosine(q, b, 5)
    di, _ = be.batch_knn_dot(q, b, 5)
    assert list(ci) == list(di)


def kat_filtered(be):  # src/batch.rs:1353-1419, tests/batch_tests.rs:461-474
    b = be.from_rows([[0, 0], [1, 0], [0.1, 0], [10, 0]])
    idx, _ = be.batch_knn_filtered([0, 0], b, 2, lambda i: i % 2 == 0)
    assert list(idx) == [0, 2]
    b = be.from_rows([[1, 0], [2, 0]])
    idx, _ = be.batch_knn_filtered([0, 0], b, 2, lambda i: False)
    assert len(idx) == 0
    b = be.from_rows([[0, 0], [1, 0], [2, 0]])
    fi, _ = be.batch_knn_filtered([0, 0], b, 2, lambda i: True)
    ui, _ = be.batch_knn([0, 0], b, 2)
    assert list(fi) == list(ui)
    b = be.from_rows([[1.0], [2.0], [3.0]])
    idx, _ = be.batch_knn_filtered([0.0], b, 10, lambda i: i == 0)
    assert list(idx) == [0]
    b = be.from_rows([[100.0], [100.0], [0.1], [100.0], [0.2]])
    idx, _ = be.batch_knn_filtered([0.0], b, 2, lambda i: i in (2, 4))
    assert list(idx) == [2, 4]
    b = be.from_rows([[float(i), 0.0] for i in range(100)])
    idx, _ = be.batch_knn_filtered([50.0, 0.0], b, 5, lambda i: i % 2 == 0)
    assert len(idx) == 5 and idx[0] == 50 and all(int(i) % 2 == 0 for i in idx)


def kat_pruning(be):  # src/batch.rs:971-988, 1475-1506
    b = be.from_rows([[0, 0], [1, 0], [10, 0]])
    idx, _ = be.batch_l2_squared_pruning([0, 0], b, 2.0)
    assert set(idx.tolist()) == {0, 1}
    b = be.from_rows([[0, 0], [1, 0], [0, 1]])
    idx, ds = be.batch_l2_squared_pruning([0, 0], b, 0.0)
    assert list(idx) == [0] and abs(ds[0]) < 1e-9
    b = be.from_rows([[0.1, 0], [0, 0.1]])
    idx, _ = be.batch_l2_squared_pruning([0, 0], b, 100.0)
    assert len(idx) == 2
    b = be.from_rows([[10, 0], [0, 10]])
    idx, _ = be.batch_l2_squared_pruning([0, 0], b, 0.5)
    assert len(idx) == 0


# ---------------------------------------------------------------------------- maxsim.rs / example
def kat_maxsim(be):  # src/maxsim.rs:201-381, examples/maxsim_colbert.rs:64,104-105
    assert abs(be.maxsim([[1, 0], [0, 1]], [[0.9, 0.1], [0.1, 0.9]]) - 1.8) < 1e-6
    assert be.maxsim([[1, 0]], np.empty((0, 2), np.float32)) == 0.0
    assert be.maxsim(np.empty((0, 2), np.float32), [[1, 0]]) == 0.0
    q, d = [[1, 0]], [[0.5, 0.5], [0.5, 0.5]]
    assert abs(be.maxsim(q, d) - 0.5) < 1e-6 and abs(be.maxsim(d, q) - 1.0) < 1e-6
    assert abs(be.maxsim([[1, 2, 3]], [[4, 5, 6]]) - 32.0) < 1e-6
    assert abs(be.maxsim([[1, 0, 0], [0, 1, 0], [0, 0, 1]], [[0.5, 0.3, 0.0], [0.0, 0.7, 0.9]]) - 2.1) < 1e-6
    v = [1, 0, 0]
    assert abs(be.maxsim([v, v, v], [v, v]) - 3.0) < 1e-6
    assert abs(be.maxsim([[1, 0, 0, 0], [0, 1, 0, 0]], [[0, 0, 1, 0], [0, 0, 0, 1]])) < 1e-6
    assert abs(be.maxsim([[1, 0, 0, 0, 0, 0, 0, 0]], [[0, 0, 0, 0, 0, 0, 0, 1], [0.5, 0.5, 0, 0, 0, 0, 0, 0]]) - 0.5) < 1e-6
    # example: 2 query x 3 doc tokens, dim 4 -> 1.7 ; non-commutative 0.8 / 1.6
    q = [[1, 0, 0, 0]]
    d = [[0.5, 0.5, 0, 0], [0.3, 0.7, 0, 0], [0.8, 0.2, 0, 0]]
    assert abs(be.maxsim(q, d) - 0.8) < 1e-5 and abs(be.maxsim(d, q) - 1.6) < 1e-5


def kat_maxsim_cosine(be):  # src/maxsim.rs:247-275, 344-367
    assert abs(be.maxsim([[1, 0]], [[1, 0]]) - be.maxsim_cosine([[1, 0]], [[1, 0]])) < 1e-6
    assert abs(be.maxsim_cosine([[2, 0]], [[3, 0]]) - 1.0) < 1e-6
    assert abs(be.maxsim_cosine([[1, 0]], [[0, 1]])) < 1e-6
    assert abs(be.maxsim_cosine([[3, 4], [3, 4]], [[3, 4]]) - 2.0) < 1e-6
    assert be.maxsim_cosine([[1, 0]], np.empty((0, 2), np.float32)) == 0.0


# ---------------------------------------------------------------------------- scalar.rs
def kat_batch_knn_u8(be, quantize):  # src/scalar.rs:582-606 ; quantize(values, alpha, offset) -> u8 codes
    alpha, offset = 2.0, -1.0  # from_range(-1, 1)
    rows = [[1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [-1.0, 0.0, 0.0], [0.7, 0.7, 0.0]]
    codes = np.stack([quantize(r, alpha, offset) for r in rows])
    idx, sc = be.batch_knn_u8([1.0, 0.0, 0.0], codes, alpha, offset, 2)
    assert len(idx) == 2 and int(idx[0]) in (0, 3) and sc[0] >= sc[1]
    idx, _ = be.batch_knn_u8([1.0], np.empty((0, 1), np.uint8), 1.0, 0.0, 5)
    assert len(idx) == 0


BATCH_KATS = [
    kat_batch_l2_squared, kat_batch_l2_known_value, kat_batch_dot, kat_batch_dot_orthogonal,
    kat_batch_dot_zero_query, kat_batch_norms, kat_batch_cosine, kat_batch_cosine_zero_query,
    kat_batch_cosine_zero_norm_vector, kat_batch_knn, kat_batch_knn_k_zero_empty_klarge,
    kat_batch_knn_sorted, kat_knn_returns_k, kat_knn_exact_match, kat_batch_knn_dot_basic,
    kat_batch_knn_dot_sorted, kat_batch_knn_cosine_basic, kat_batch_knn_cosine_sorted,
    kat_batch_knn_cosine_empty, kat_cosine_knn_matches_dot_knn_normalized,
]
L2_FAMILY_KATS = [kat_reordered_matches_exact, kat_filtered, kat_pruning]
MAXSIM_KATS = [kat_maxsim, kat_maxsim_cosine]
