"""GPU parity tests: matryoshka prefix views (dense.rs:436-462 at batch level) and the progressive search of
examples/matryoshka_search.rs, through the C ABI, against the CPU oracle run on the first prefix rows of the same corpus.
Bar: bit-exact scores, identical index lists (same_knn of test_gpu_exact.py)."""
from __future__ import annotations

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle
from test_gpu_exact import B, _check_knn, _corpus, _queries, bits_equal, innr, same_knn  # noqa: F401  (fixtures)


@pytest.mark.parametrize("n,dim,prefix", [(1000, 64, 1), (1000, 64, 20), (777, 100, 32), (3000, 96, 96), (500, 40, 1000)])
def test_prefix_view_scans_and_norms(B, n, dim, prefix):
    rows, data = _corpus(n, dim, 3, uniform=True)
    vb = B.VerticalBatch.from_rows(rows)
    v = vb.prefix(prefix)
    p = min(prefix, dim)
    assert (v.num_vectors(), v.dimension()) == (n, p)
    pd = np.ascontiguousarray(data[:p])
    q = _queries(1, dim, uniform=True)[0][:p]
    assert bits_equal(B.batch_dot(q, v), oracle.batch_dot(q, pd))
    assert bits_equal(B.batch_l2_squared(q, v), oracle.batch_l2_squared(q, pd))
    norms = B.batch_norms(v)
    assert bits_equal(norms, oracle.batch_norms(pd))
    assert bits_equal(B.batch_cosine(q, v, norms), oracle.batch_cosine(q, pd, oracle.batch_norms(pd)))
    # the parent is untouched: its own norms are the full-dimension ones
    assert bits_equal(B.batch_norms(vb), oracle.batch_norms(data))
    with pytest.raises(Exception):
        B.batch_dot(np.zeros(p + 1, np.float32), v)
    with pytest.raises(Exception):
        vb.prefix(0)


@pytest.mark.parametrize("prefix,engine", [(32, "mfma"), (64, "mfma"), (20, "mfma"), (33, "exact"), (128, "auto")])
def test_prefix_view_knn_all_metrics(B, innr, prefix, engine):
    # prefix 20 / 33: rows prefix..Dpad of the view hold the parent's NEXT dimensions, so the GEMM engine (which multiplies
    # all Dpad rows) must be refused and the exact engine used -- the results say so
    rows, data = _corpus(70_000, 128, 11, uniform=True)
    vb = B.VerticalBatch.from_rows(rows)
    v = vb.prefix(prefix)
    pd = np.ascontiguousarray(data[:prefix])
    eng = {"mfma": innr.KNN_MFMA, "exact": innr.KNN_EXACT, "auto": innr.KNN_AUTO}[engine]
    qs = np.ascontiguousarray(_queries(40, 128, 99, uniform=True)[:, :prefix])
    for metric in ("dot", "cos", "l2"):
        _check_knn(B, innr, metric, v, pd, qs, 10, eng)
    v.close()
    _check_knn(B, innr, "dot", vb, data, _queries(3, 128, 5, uniform=True), 5, innr.KNN_EXACT)  # parent still alive


def _xorshift_vec(dim, seed):
    """examples/matryoshka_search.rs:178-188 (generate_vec)."""
    state = np.uint64(seed) ^ np.uint64(0x517CC1B727220A95)
    out = np.empty(dim, np.float32)
    m = (1 << 64) - 1
    s = int(state)
    for i in range(dim):
        s ^= (s << 13) & m
        s ^= s >> 7
        s ^= (s << 17) & m
        out[i] = np.float32(np.float32(np.float32(s) / np.float32(m)) * np.float32(2.0)) - np.float32(1.0)
    return out


def _normalize(v):
    """examples/matryoshka_search.rs:191-197 with innr::norm = sqrt(dot(v, v)) (dense.rs norm)."""
    nrm = np.float32(np.sqrt(np.float32(oracle.dot_portable(v, v))))
    return v if nrm < 1e-9 else (v / nrm).astype(np.float32)


def _oracle_two_stage(q, data, prefix, k_coarse, k, metric):
    pd = np.ascontiguousarray(data[:prefix])
    knn = {"cos": oracle.batch_knn_cosine, "dot": oracle.batch_knn_dot}[metric]
    ci, _ = knn(q[:prefix], pd, k_coarse)
    full = oracle.batch_cosine(q, data, oracle.batch_norms(data)) if metric == "cos" else oracle.batch_dot(q, data)
    order = sorted((int(i) for i in ci), key=lambda i: (-oracle.total_key(full[i]), i))[:k]
    return np.array(order, np.uint64), full[order]


@pytest.mark.parametrize("metric", ["cos", "dot"])
def test_matryoshka_search_example_shape(B, innr, metric):
    # examples/matryoshka_search.rs:14-18: 10K x 768, prefix 128, coarse top-100, final top-10, normalised xorshift data
    n, dim, prefix, kc, k = 10_000, 768, 128, 100, 10
    rng_rows = np.stack([_normalize(_xorshift_vec(dim, i)) for i in range(0, n, 40)])  # every 40th row by the example's generator
    rows = oracle.generate_uniform(n, dim, 21)
    rows /= np.sqrt((rows.astype(np.float64) ** 2).sum(1, keepdims=True)).astype(np.float32)
    rows[::40] = rng_rows
    data = oracle.from_rows(rows)
    vb = B.VerticalBatch.from_rows(rows)
    qs = np.stack([_normalize(_xorshift_vec(dim, 0xDEAD + j)) for j in range(6)])
    m = {"cos": innr.METRIC_COSINE, "dot": innr.METRIC_DOT}[metric]
    idx, sc = B.matryoshka_knn(qs, vb, prefix, kc, k, m)
    coarse = vb.prefix(prefix)  # a cached view gives the same answer
    idx2, sc2 = B.matryoshka_knn(qs, vb, prefix, kc, k, m, coarse=coarse)
    assert np.array_equal(idx, idx2) and bits_equal(sc, sc2)
    for j, q in enumerate(qs):
        oi, os_ = _oracle_two_stage(q, data, prefix, kc, k, metric)
        assert same_knn(metric, idx[j], sc[j], oi, os_), (j, idx[j], oi, sc[j], os_)


def test_prefix_view_of_u8_codes(innr):
    from innr_amd import scalar as S
    n, dim, prefix = 5000, 96, 32
    rows = oracle.generate_uniform(n, dim, 8)
    op = oracle.qparams_from_range(-1.0, 1.0)
    codes = oracle.quantize_u8(rows, op)
    qc = S.QuantizedCorpus.from_codes(codes, n, dim, S.QuantizationParams(op.alpha, op.offset))
    v = qc.prefix(prefix)
    qs = np.ascontiguousarray(oracle.generate_uniform(5, dim, 77)[:, :prefix])
    idx, sc = v.knn_multi(qs, 7)
    for j, q in enumerate(qs):
        oi, os_ = oracle.batch_knn_u8(q, np.ascontiguousarray(codes[:, :prefix]), op, 7)
        assert [int(i) for i in idx[j]] == [int(i) for i in oi] and bits_equal(sc[j], os_)
    qc.close()
    assert v._h is None
