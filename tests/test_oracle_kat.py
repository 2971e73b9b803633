"""Pins the CPU oracle (oracle/innr_oracle.c) against every known-answer test the reference holds
for the hot path, plus an independent numpy f32 re-derivation of the
sequential arithmetic. CPU only."""
from __future__ import annotations

import math

import numpy as np
import pytest

import oracle
from backends import OracleBackend
import kat_cases as K

F = np.float32
be = OracleBackend()


@pytest.mark.parametrize("kat", K.BATCH_KATS + K.L2_FAMILY_KATS + K.MAXSIM_KATS, ids=lambda f: f.__name__)
def test_reference_kat(kat):
    kat(be)


def test_batch_knn_u8_kat():  # src/scalar.rs:582-606
    K.kat_batch_knn_u8(be, lambda v, a, o: oracle.quantize_u8(v, oracle.QParams(a, o)))


# ------------------------------------------------------------------ VerticalBatch layout KATs
def test_layout_dimension_major():  # src/batch.rs:889-901, 1086-1094
    b = oracle.from_rows([[1, 2, 3], [4, 5, 6]])
    assert b.shape == (3, 2)
    assert b[0, 0] == 1 and b[0, 1] == 4 and b[1, 0] == 2 and b[2, 1] == 6
    assert b.reshape(-1).tolist() == [1, 4, 2, 5, 3, 6]  # data[d*N+i]
    b = oracle.from_rows([[1, 2], [3, 4], [5, 6]])
    assert b[0].tolist() == [1, 3, 5] and b[1].tolist() == [2, 4, 6]


def test_layout_from_flat_matches_from_rows():  # src/batch.rs:1050-1070
    rows = np.array([[1, 2, 3], [4, 5, 6], [7, 8, 9]], dtype=np.float32)
    assert np.array_equal(oracle.from_rows(rows), oracle.from_flat(rows.reshape(-1), 3, 3))
    b = oracle.from_flat([10.0, 20.0], 1, 2)
    assert b.shape == (2, 1) and b[:, 0].tolist() == [10.0, 20.0]


def test_layout_empty_and_roundtrip():  # src/batch.rs:1026-1044, 1600-1615
    assert oracle.from_rows([]).shape == (0, 0)
    rows = np.array([[1, 2, 3], [4, 5, 6], [7, 8, 9]], dtype=np.float32)
    b = oracle.from_rows(rows)
    assert np.array_equal(b.T, rows)


# ------------------------------------------------------------------ total order / stable sort
def test_total_cmp_key_order():
    nan = float("nan")
    vals = [-nan, float("-inf"), -1.0, -0.0, 0.0, 1e-45, 1.0, float("inf"), nan]
    # -NaN: flip the sign bit of NaN
    neg_nan = np.frombuffer(np.uint32(0xFFC00000).tobytes(), dtype=np.float32)[0]
    keys = [oracle.total_key(neg_nan)] + [oracle.total_key(v) for v in vals[1:]]
    assert keys == sorted(keys) and len(set(keys)) == len(keys)


def test_knn_dot_ties_resolve_to_lower_index():
    # stable sort_by (batch.rs:757): equal scores keep index order
    b = oracle.from_rows([[1, 0], [1, 0], [2, 0], [1, 0], [2, 0]])
    idx, sc = oracle.batch_knn_dot([1, 0], b, 4)
    assert idx.tolist() == [2, 4, 0, 1] and sc.tolist() == [2, 2, 1, 1]


def test_knn_dot_nan_sorts_first():
    # total_cmp: +NaN is greater than +inf, so a descending sort puts it first
    b = oracle.from_rows([[1.0], [float("nan")], [3.0]])
    idx, sc = oracle.batch_knn_dot([1.0], b, 3)
    assert idx.tolist() == [1, 2, 0] and math.isnan(sc[0])


# ------------------------------------------------------------------ TopK KATs (src/topk.rs:191-346)
def test_topk_basic_top3():
    t = oracle.TopK(3)
    for i, d in [(0, 1.5), (1, 0.3), (2, 2.0), (3, 0.8), (4, 5.0)]:
        t.insert(i, d)
    assert len(t) == 3
    r = t.into_sorted()
    assert r == [(1, F(0.3)), (3, F(0.8)), (0, F(1.5))]


def test_topk_threshold_tracking():
    t = oracle.TopK(3)
    assert t.threshold() == math.inf
    t.insert(0, 1.0); assert t.threshold() == math.inf
    t.insert(1, 2.0); assert t.threshold() == math.inf
    t.insert(2, 3.0); assert t.threshold() == 3.0
    t.insert(3, 1.5); assert t.threshold() == 2.0
    t.insert(4, 0.5); assert t.threshold() == 1.5
    t.insert(5, 10.0); assert t.threshold() == 1.5


def test_topk_duplicates_k1_large_sorted():
    t = oracle.TopK(3)
    for i in range(4):
        t.insert(i, 1.0)
    r = t.into_sorted()
    assert len(r) == 3 and all(d == 1.0 for _, d in r)
    assert 3 not in [i for i, _ in r]  # equal-to-worst is rejected (strict less, topk.rs:101)
    t = oracle.TopK(1)
    for i, d, th in [(0, 5.0, 5.0), (1, 3.0, 3.0), (2, 10.0, 3.0), (3, 1.0, 1.0)]:
        t.insert(i, d); assert t.threshold() == th
    assert t.into_sorted() == [(3, 1.0)]
    t = oracle.TopK(10)
    for i in range(10_000):
        t.insert(i, float(i))
    assert t.into_sorted() == [(i, float(i)) for i in range(10)]
    t = oracle.TopK(5)
    for i in reversed(range(5)):
        t.insert(i, float(i))
    r = t.into_sorted()
    assert all(r[i][1] <= r[i + 1][1] for i in range(4))


def test_topk_len_and_sorted_insert_and_nan():
    t = oracle.TopK(4)
    assert t.is_empty() and len(t) == 0
    for i in range(4):
        t.insert(i, float(i + 1))
    t.insert(4, 5.0)
    assert len(t) == 4
    t = oracle.TopK(4)
    for i in range(4):
        t.insert(i, float(i + 1))
    t.insert(4, 0.5)
    r = t.into_sorted()
    assert r[0] == (4, 0.5) and r[3] == (2, 3.0)
    t = oracle.TopK(2)  # nan_candidate_does_not_poison_topk, topk.rs:191-209
    t.insert(0, float("nan")); t.insert(1, 1.0); t.insert(2, 0.5)
    ids = [i for i, _ in t.into_sorted()]
    assert 2 in ids and 1 in ids
    with pytest.raises(AssertionError):
        oracle.TopK(0)


# ------------------------------------------------------------------ scalar.rs KATs
def test_quantize_range_and_roundtrip():  # src/scalar.rs:399-427
    p = oracle.qparams_fit([-1.0, 0.0, 1.0])
    q = oracle.quantize_u8([-1.0, 0.0, 1.0], p)
    assert q[0] == 0 and q[2] == 255 and abs(int(q[1]) - 128) <= 1
    vals = np.array([0.0, 0.5, 1.0, -1.0, 0.25], dtype=np.float32)
    p = oracle.qparams_fit(vals)
    q = oracle.quantize_u8(vals, p)
    deq = p.alpha * (q.astype(np.float32) / F(255.0)) + p.offset
    assert np.all(np.abs(vals - deq) < p.alpha / 255.0 + 1e-6)


def test_qparams():  # src/scalar.rs:496-511, 478-493, 557-580
    p = oracle.qparams_from_range(-1.0, 1.0)
    assert abs(p.alpha - 2.0) < 1e-6 and abs(p.offset + 1.0) < 1e-6
    p = oracle.qparams_fit(np.empty(0, np.float32))
    assert p.alpha == 1.0 and p.offset == 0.0
    p = oracle.qparams_fit([5.0] * 10)  # constant -> alpha falls back to 1.0 (scalar.rs:57)
    assert p.alpha == 1.0 and p.offset == 5.0
    assert np.all(oracle.quantize_u8([5.0] * 10, p) == 0)
    vals = [(i / 49.0) - 1.0 for i in range(98)] + [100.0, -100.0]
    full = oracle.qparams_fit(vals)
    clip = oracle.qparams_fit_quantile(vals, 0.95)
    assert clip.alpha < full.alpha and clip.alpha < 10.0
    with pytest.raises(AssertionError):
        oracle.qparams_fit_quantile(vals, 0.0)


def test_mixed_dot_exact_small():  # src/scalar.rs:466-476
    q = np.array([0.5, -2.0, 3.0, 4.5], dtype=np.float32)
    c = np.array([2, 7, 11, 13], dtype=np.uint8)
    exp = F(-0.0)
    for a, b in zip(q, c):
        exp = F(exp + F(a * F(b)))
    assert oracle.mixed_dot_u8_f32(q, c) == exp
    with pytest.raises(AssertionError):
        oracle.mixed_dot_u8_f32([1.0, 2.0], [1])


def test_mixed_dot_bit_exact_integer_sweep():  # tests/simd_correctness.rs:365-388
    for dim in (8, 16, 31, 32, 33, 64, 65, 128):
        for seed in range(5):
            corpus = np.array([(i * 31 + seed * 7) % 256 for i in range(dim)], dtype=np.uint8)
            query = np.array([(i * 13 + seed * 3) % 8 for i in range(dim)], dtype=np.float32)
            expect = int(sum(int(a) * int(b) for a, b in zip(query, corpus)))  # exact in f32
            assert expect < 2 ** 24
            assert oracle.mixed_dot_u8_f32(query, corpus) == float(expect)


def test_asymmetric_dot():  # src/scalar.rs:429-463, 513-537
    doc = np.array([1, 2, 3, 4], dtype=np.float32)
    q = np.full(4, 0.5, dtype=np.float32)
    p = oracle.qparams_fit(doc)
    approx = oracle.asymmetric_dot_u8(q, oracle.quantize_u8(doc, p), p)
    assert abs(float((doc * q).sum()) - approx) < p.alpha / 255.0 * 4
    dim = 128
    doc = np.sin(np.arange(dim, dtype=np.float32) * F(0.1)).astype(np.float32)
    q = np.cos(np.arange(dim, dtype=np.float32) * F(0.3)).astype(np.float32)
    p = oracle.qparams_fit(doc)
    approx = oracle.asymmetric_dot_u8(q, oracle.quantize_u8(doc, p), p)
    assert abs(float((doc.astype(np.float64) * q).sum()) - approx) < p.alpha / 255.0 * math.sqrt(dim) + 0.1
    with pytest.raises(AssertionError):
        oracle.asymmetric_dot_u8([1.0, 2.0, 3.0], oracle.quantize_u8([0.5, 0.5], oracle.qparams_from_range(0, 1)),
                                 oracle.qparams_from_range(0, 1))


# ------------------------------------------------------------------ adaptive (CPU-only function)
def test_adaptive():  # src/batch.rs:1512-1570, tests/batch_tests.rs:268-290
    assert len(oracle.batch_knn_adaptive([], oracle.from_rows([]), 5, 2)[0]) == 0
    assert len(oracle.batch_knn_adaptive([1, 2], oracle.from_rows([[1, 2]]), 0, 1)[0]) == 0
    b = oracle.from_rows([[0, 0, 0, 0], [100, 100, 100, 100], [0.1, 0.1, 0.1, 0.1]])
    assert oracle.batch_knn([0, 0, 0, 0], b, 1)[0][0] == 0
    assert oracle.batch_knn_adaptive([0, 0, 0, 0], b, 1, 2)[0][0] == 0
    b = oracle.from_rows([[0.0, 1.0]])
    ai, asc = oracle.batch_knn_adaptive([0, 0], b, 1, 1)
    ei, esc = oracle.batch_knn([0, 0], b, 1)
    assert ai.tolist() == ei.tolist() and asc.tolist() == esc.tolist()
    b = np.empty((0, 3), np.float32)  # three zero-dimensional vectors
    i, s = oracle.batch_knn_adaptive([], b, 2, 1)
    assert i.tolist() == [0, 1] and s.tolist() == [0.0, 0.0]
    rows = [[float(i), math.sin(i * 0.1), math.cos(i * 0.1)] for i in range(100)]
    b = oracle.from_rows(rows)
    basic = oracle.batch_knn([50, 0, 1], b, 10)[0]
    adapt = oracle.batch_knn_adaptive([50, 0, 1], b, 10, 1)[0]
    assert set(basic.tolist()) <= set(adapt.tolist())


def test_dimension_variance():  # src/batch.rs:1248-1262
    v = oracle.batch_dimension_variance(oracle.from_rows([[1, 0], [1, 5], [1, 10]]))
    assert abs(v[0]) < 1e-6 and v[1] > 10.0 and abs(v[1] - 50.0 / 3.0) < 1e-4


# ------------------------------------------------------------------ distance.rs
def test_distance_trait_metrics():  # src/distance.rs:195-264 shapes
    a, b = [1.0, 0.0], [0.0, 1.0]
    assert abs(oracle.dist_cosine(a, b) - 1.0) < 1e-6
    assert abs(oracle.dist_cosine(a, a)) < 1e-6
    assert oracle.dist_dot([1, 2, 3], [4, 5, 6]) == -32.0
    assert abs(oracle.dist_l2([0, 0], [3, 4]) - 5.0) < 1e-6
    assert oracle.dist_l1([0, 0], [3, -4]) == 7.0
    assert oracle.cosine_portable([1, 0], [0, 0]) == 0.0  # zero vector guard (dense.rs:341-345)
    # integer metrics (distance.rs:230-243, quant.rs doc example, slot.rs:385-388)
    assert oracle.dist_hamming([0b11110000], [0b10101010]) == 4.0
    assert oracle.dist_hamming([0b11110000, 0xFF], [0b10101010, 0x00]) == 12.0
    assert oracle.dist_slot_u32([1, 2, 3, 4], [1, 0, 3, 9]) == 0.5
    assert oracle.dist_slot_u32([1, 2, 3, 4], [1, 2, 3, 9]) == 0.25
    assert oracle.dist_slot_u32([], []) == 0.0
    sketches = [[1, 2, 3, 4], [1, 2, 3, 9], [9, 9, 9, 9]]  # generic_index_over_metric, distance.rs:260-262
    assert min(range(3), key=lambda i: oracle.dist_slot_u32([1, 2, 3, 4], sketches[i])) == 0


# ------------------------------------------------------------------ example-level checks
def test_example_batch_demo_knn_matches_naive():  # examples/batch_demo.rs:77-122
    dim, n, k = 8, 20, 3
    corpus = oracle.generate_corpus(n, dim, 0)
    q = oracle.generate_embedding(dim, 999)
    b = oracle.from_rows(corpus)
    idx, _ = oracle.batch_knn(q, b, k)
    naive = sorted(range(n), key=lambda i: (oracle.l2_distance_squared_portable(q, corpus[i]), i))[:k]
    assert idx.tolist() == naive


def test_example_generator_properties():  # examples/batch_demo.rs:233-242
    v = oracle.generate_embedding(128, 12345)
    assert v.dtype == np.float32 and np.all(v >= -1.0) and np.all(v < 1.0)
    assert oracle.generate_embedding(4, 0)[0] == -1.0  # seed 0, i 0 -> x = 0
    # independent integer re-derivation
    seed, dim = 50_017, 16
    M = (1 << 64) - 1
    exp = []
    for i in range(dim):
        x = (seed * 6364136223846793005 + i * 1442695040888963407) & M
        exp.append(F(F(F(np.float32(x >> 33)) / F(2147483648.0)) * F(2.0)) - F(1.0))
    assert oracle.generate_embedding(dim, seed).tolist() == [float(e) for e in exp]
    u = oracle.generate_normalized(128, 5000)
    assert abs(float(np.sqrt((u.astype(np.float64) ** 2).sum())) - 1.0) < 1e-5


def test_example_timing_checksum_c1_shape():  # examples/batch_demo.rs:159-227 at reduced query count
    dim, n, nq = 128, 10_000, 8
    corpus = oracle.generate_corpus(n, dim, 0)
    b = oracle.from_rows(corpus)
    tot_b = 0.0; tot_n = 0.0
    for j in range(nq):
        q = oracle.generate_embedding(dim, j + 50_000)
        tot_b += float(oracle.batch_l2_squared(q, b).sum(dtype=np.float64))
        diff = corpus.astype(np.float64) - q.astype(np.float64)
        tot_n += float((diff * diff).sum())
    assert abs(tot_b - tot_n) / max(abs(tot_n), 1.0) < 1e-3


# ------------------------------------------------------------------ independent numpy restatement
def _np_seq_dot(q, data):
    acc = np.zeros(data.shape[1], dtype=np.float32)
    for d in range(data.shape[0]):
        acc = (acc + (q[d] * data[d]).astype(np.float32)).astype(np.float32)
    return acc


def _np_seq_l2(q, data):
    acc = np.zeros(data.shape[1], dtype=np.float32)
    for d in range(data.shape[0]):
        diff = (q[d] - data[d]).astype(np.float32)
        acc = (acc + (diff * diff).astype(np.float32)).astype(np.float32)
    return acc


@pytest.mark.parametrize("n,dim", [(1, 1), (7, 3), (1000, 128), (4097, 768)])
def test_scans_bit_exact_vs_numpy_sequential(n, dim):
    corpus = oracle.generate_corpus(n, dim, 7)
    data = oracle.from_rows(corpus)
    q = oracle.generate_embedding(dim, 123_456)
    assert np.array_equal(oracle.batch_dot(q, data), _np_seq_dot(q, data))
    assert np.array_equal(oracle.batch_l2_squared(q, data), _np_seq_l2(q, data))
    nsq = np.zeros(n, np.float32)
    for d in range(dim):
        nsq = (nsq + (data[d] * data[d]).astype(np.float32)).astype(np.float32)
    norms = np.sqrt(nsq).astype(np.float32)
    assert np.array_equal(oracle.batch_norms(data), norms)
    qs = F(-0.0)
    for x in q:
        qs = F(qs + F(x * x))
    qn = F(np.sqrt(qs))
    cos = np.where(norms > F(1e-9), _np_seq_dot(q, data) / (qn * norms).astype(np.float32), F(0.0)).astype(np.float32)
    assert np.array_equal(oracle.batch_cosine(q, data, norms), cos)


def test_knn_dot_equals_stable_argsort_of_numpy_scores():
    n, dim, k = 5000, 64, 25
    data = oracle.from_rows(oracle.generate_corpus(n, dim, 11))
    q = oracle.generate_embedding(dim, 99)
    sc = _np_seq_dot(q, data)
    order = np.argsort(-sc.astype(np.float64), kind="stable")[:k]
    idx, s = oracle.batch_knn_dot(q, data, k)
    assert idx.tolist() == order.tolist() and np.array_equal(s, sc[order])
    # L2 kNN (TopK path) == ascending stable order on distinct distances
    d2 = _np_seq_l2(q, data)
    order = np.argsort(d2.astype(np.float64), kind="stable")[:k]
    idx, s = oracle.batch_knn(q, data, k)
    assert idx.tolist() == order.tolist() and np.array_equal(s, d2[order])


def test_batch_dot_differential_vs_pairwise_dot():  # tests/property_tests.rs:398-415 tolerance
    rng = np.random.default_rng(0)
    for n, dim in [(3, 5), (17, 33), (64, 128)]:
        rows = rng.uniform(-10, 10, size=(n, dim)).astype(np.float32)
        q = rng.uniform(-10, 10, size=dim).astype(np.float32)
        data = oracle.from_rows(rows)
        bd = oracle.batch_dot(q, data)
        bl = oracle.batch_l2_squared(q, data)
        for i in range(n):
            pd = oracle.dot_portable(q, rows[i])
            tol = 1e-4 * float(np.abs(q * rows[i]).sum()) + 1e-4
            assert abs(bd[i] - pd) <= tol
            pl = oracle.l2_distance_squared_portable(q, rows[i])
            assert abs(bl[i] - pl) <= 1e-4 * abs(pl) + 1e-5  # property_tests.rs:385


def test_maxsim_properties():  # tests/maxsim_tests.rs:75-132 (additivity, single query == max dot)
    rng = np.random.default_rng(1)
    q = rng.normal(size=(5, 32)).astype(np.float32)
    d = rng.normal(size=(9, 32)).astype(np.float32)
    total = oracle.maxsim(q, d)
    parts = sum(oracle.maxsim(q[i:i + 1], d) for i in range(5))
    assert abs(total - parts) < 1e-4
    single = oracle.maxsim(q[:1], d)
    assert single == max(oracle.dot_portable(q[0], d[j]) for j in range(9))
