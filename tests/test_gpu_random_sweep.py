"""GPU parity, randomized: shapes, k, metric and engine drawn from a fixed-seed generator -- the breadth the reference
gets from its property tests (tests/property_tests.rs), here as HIP path == oracle, bit for bit, on every draw.
Data deliberately contains duplicates (exact ties), zero vectors and a zero query now and then."""
from __future__ import annotations

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle
from test_gpu_exact import bits_equal, same_knn


def _data(rng, n, dim):
    rows = rng.uniform(-1, 1, size=(n, dim)).astype(np.float32)
    if n > 4 and rng.random() < 0.5:
        rows[rng.integers(0, n, size=max(1, n // 10))] = rows[rng.integers(0, n)]  # exact duplicates -> ties
    if rng.random() < 0.3:
        rows[rng.integers(0, n)] = 0.0  # zero-norm vector (cosine guard)
    if rng.random() < 0.3:
        rows = (rows * np.float32(rng.choice([1e-3, 7.0, 300.0]))).astype(np.float32)
    return rows


@pytest.mark.parametrize("seed", range(24))
def test_random_knn_all_metrics_and_engines(seed):
    import innr_amd
    from innr_amd import batch as B
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(1, 4000))
    dim = int(rng.integers(1, 200))
    nq = int(rng.integers(1, 40))
    k = int(rng.integers(1, min(240, n + 5) + 1))
    rows = _data(rng, n, dim)
    qs = rng.uniform(-1, 1, size=(nq, dim)).astype(np.float32)
    if rng.random() < 0.3:
        qs[0] = 0.0
    if rng.random() < 0.3:
        qs[-1] = rows[rng.integers(0, n)]  # a query that IS a corpus vector
    data = oracle.from_rows(rows)
    vb = B.VerticalBatch.from_rows(rows)
    for metric in ("dot", "cos", "l2"):
        fn = {"dot": B.batch_knn_dot_multi, "cos": B.batch_knn_cosine_multi, "l2": B.batch_knn_multi}[metric]
        ofn = {"dot": oracle.batch_knn_dot, "cos": oracle.batch_knn_cosine, "l2": oracle.batch_knn}[metric]
        # KNN_MFMA_BF16: bf16 filter for dot with k <= 48, served by the f32 engine otherwise -- same answers either way
        for engine in (innr_amd.KNN_EXACT, innr_amd.KNN_MFMA, innr_amd.KNN_MFMA_BF16):
            idx, sc = fn(qs, vb, k, engine=engine)
            assert idx.shape == (nq, min(k, n))
            for j in range(nq):
                oi, os_ = ofn(qs[j], data, k)
                if metric != "l2":
                    assert same_knn(metric, idx[j], sc[j], oi, os_), (seed, metric, engine, j)
                else:  # ties at the cut: membership is core::binary_search's business
                    assert bits_equal(sc[j], os_), (seed, engine, j)
                    inner = os_ != os_[-1]
                    assert sorted(idx[j][inner].tolist()) == sorted(oi[inner].tolist()), (seed, engine, j)
    q0 = qs[0]
    assert bits_equal(B.batch_dot(q0, vb), oracle.batch_dot(q0, data))
    assert bits_equal(B.batch_l2_squared(q0, vb), oracle.batch_l2_squared(q0, data))
    norms = B.batch_norms(vb)
    assert bits_equal(norms, oracle.batch_norms(data))
    assert bits_equal(B.batch_cosine(q0, vb, norms), oracle.batch_cosine(q0, data, oracle.batch_norms(data)))


@pytest.mark.parametrize("seed", range(10))
def test_random_u8_and_maxsim(seed):
    import innr_amd
    from innr_amd import maxsim as M
    from innr_amd import scalar as S
    rng = np.random.default_rng(2000 + seed)
    # scalar::batch_knn_u8
    n, dim, nq = int(rng.integers(1, 3000)), int(rng.integers(1, 150)), int(rng.integers(1, 30))
    k = int(rng.integers(1, min(200, n + 3) + 1))
    lo, hi = sorted(rng.uniform(-2, 2, size=2).tolist())
    p = S.QuantizationParams.from_range(lo, hi)
    rows = _data(rng, n, dim)
    op = oracle.QParams(p.alpha, p.offset)
    codes = oracle.quantize_u8(rows, op)
    qc = S.QuantizedCorpus.from_codes(codes, n, dim, p)
    qs = rng.uniform(-1, 1, size=(nq, dim)).astype(np.float32)
    for engine in (innr_amd.KNN_EXACT, innr_amd.KNN_MFMA):
        idx, sc = qc.knn_multi(qs, k, engine=engine)
        for j in range(nq):
            oi, os_ = oracle.batch_knn_u8(qs[j], codes, op, k)
            assert same_knn("dot", idx[j], sc[j], oi, os_), (seed, engine, j)
    # maxsim over a ragged corpus
    ndocs, T, dim = int(rng.integers(1, 400)), int(rng.integers(1, 90)), int(rng.choice([8, 24, 32, 40, 64, 96, 128, 33]))
    Tq, k = int(rng.integers(1, 45)), int(rng.integers(1, 30))
    toks = rng.uniform(-1, 1, size=(ndocs, T, dim)).astype(np.float32)
    lens = rng.integers(0, T + 1, size=ndocs).astype(np.uint32)
    q = rng.uniform(-1, 1, size=(Tq, dim)).astype(np.float32)
    dc = M.DocumentCorpus.from_tokens(toks, lens)
    for cosine in (False, True):
        want = np.array([oracle.maxsim(q, toks[i][:lens[i]], cosine=cosine) if lens[i] else 0.0 for i in range(ndocs)],
                        dtype=np.float32)
        assert bits_equal(dc.scores(q, cosine=cosine), want), (seed, cosine)
        order = np.argsort(-want.astype(np.float64), kind="stable")[:k]
        for engine in (innr_amd.KNN_EXACT, innr_amd.KNN_AUTO):
            idx, sc = dc.topk(q, k, cosine=cosine, engine=engine)
            assert idx.tolist() == order.tolist() and bits_equal(sc, want[order]), (seed, cosine, engine)


@pytest.mark.parametrize("seed", range(8))
def test_random_large_mfma_equals_exact_engine(seed):
    """Larger draws (seeded thresholds, 8-wave tiles, many slices): the GEMM engine against the bit-exact engine, which
    the smaller draws above pin to the oracle."""
    import innr_amd
    from innr_amd import batch as B
    rng = np.random.default_rng(3000 + seed)
    n = int(rng.integers(70_000, 400_000))
    dim = int(rng.choice([24, 64, 100, 128, 200, 256]))
    nq = int(rng.integers(200, 700))
    k = int(rng.choice([1, 5, 10, 16, 17, 40, 100]))
    vb = B.VerticalBatch.generate(n, dim, seed=int(rng.integers(0, 1 << 30)))
    qs = rng.uniform(-1, 1, size=(nq, dim)).astype(np.float32)
    if seed % 2:
        qs[: nq // 8] = np.ascontiguousarray(vb.data()[:, rng.integers(0, n, size=nq // 8)].T)  # queries that are corpus rows
    sub = rng.choice(nq, size=12, replace=False)
    for fn in (B.batch_knn_dot_multi, B.batch_knn_cosine_multi, B.batch_knn_multi):
        st = innr_amd.KnnStats()
        i1, s1 = fn(qs, vb, k, engine=innr_amd.KNN_MFMA, stats=st)
        i2, s2 = fn(qs[sub], vb, k, engine=innr_amd.KNN_EXACT)
        assert bits_equal(s1[sub], s2), (seed, fn.__name__)
        if fn is not B.batch_knn_multi:
            assert np.array_equal(i1[sub], i2), (seed, fn.__name__)
        assert st.queries_fallback <= nq // 4
