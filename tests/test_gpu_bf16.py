"""GPU parity tests of the bf16 FILTER engine (INNR_KNN_MFMA_BF16, kernels_gemm_bf16.h): approximate scores on the bf16
matrix pipe, candidates re-scored in the reference's f32 order, answers proven against the bf16 error bound. The bar is
the same as for every engine: scores bit-identical to the oracle's, index lists identical."""
from __future__ import annotations

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle
from test_gpu_exact import B, _check_knn, _corpus, _queries, bits_equal, innr, same_knn  # noqa: F401  (fixtures)


@pytest.mark.parametrize("n,dim,nq,k", [(10_000, 128, 100, 10), (20_000, 100, 513, 10), (70_000, 64, 600, 10), (3000, 20, 300, 48),
                                        (257, 33, 5, 16), (1000, 64, 9, 33), (5, 3, 2, 10), (200_000, 32, 40, 1)])
@pytest.mark.parametrize("filt", ["bf16", "int8"])
def test_bf16_filter_dot_matches_oracle(B, innr, n, dim, nq, k, filt):
    rows, data = _corpus(n, dim, 77, uniform=True)
    rows = (rows * (1.0 + 0.5 * np.sin(np.arange(n, dtype=np.float32)))[:, None]).astype(np.float32)  # unequal norms: cosine != dot order
    data = oracle.from_rows(rows)
    vb = B.VerticalBatch.from_rows(rows)
    qs = _queries(nq, dim, 4242, uniform=True)
    engine = innr.KNN_MFMA_BF16 if filt == "bf16" else innr.KNN_MFMA_I8  # int8: the scalar-quantised corpus as the filter
    for metric, fn, ofn in (("dot", B.batch_knn_dot_multi, oracle.batch_knn_dot), ("cos", B.batch_knn_cosine_multi, oracle.batch_knn_cosine),
                            ("l2", B.batch_knn_multi, oracle.batch_knn)):
        st = innr.KnnStats()
        idx, sc = fn(qs, vb, k, engine=engine, stats=st)
        # (squared L2 on the int8 filter: |v|^2 as two 8-bit limbs in R + 1 more dimensions of its own corpus copy)
        assert st.engine == engine
        for j, q in enumerate(qs):
            oi, os_ = ofn(q, data, k)
            assert same_knn(metric, idx[j], sc[j], oi, os_), (metric, j, idx[j], oi, sc[j], os_)
        if n >= 10_000:  # well-separated uniform data: (almost) every answer is proven, few queries redone exactly
            assert st.queries_fallback <= max(2, nq // 20), (metric, st.queries_fallback)


def test_bf16_cosine_zero_norms_and_auto(B, innr):
    """cosine on the normalised bf16 copies with zero-norm rows and a zero query (the reference's epsilon guards,
    batch.rs:716-727), and INNR_KNN_AUTO choosing the bf16 filter for a large batch with small k (same answers)"""
    rows, data = _corpus(70_000, 48, 21, uniform=True)
    rows[5] = 0.0
    rows[777] = 1e-12
    data = oracle.from_rows(rows)
    qs = _queries(130, 48, 5, uniform=True)
    qs[3] = 0.0
    vb = _check_knn(B, innr, "cos", rows, data, qs, 10, innr.KNN_MFMA_BF16)
    st = innr.KnnStats()
    B.batch_knn_cosine_multi(qs, vb, 10, engine=innr.KNN_AUTO, stats=st)
    assert st.engine == innr.KNN_MFMA_I8  # there is room for the int8 filter copy: AUTO takes the fastest filter that applies
    _check_knn(B, innr, "cos", vb, data, qs, 10, innr.KNN_AUTO)
    from innr_amd import _lib
    with _lib.default_context().option("no_auto_i8", 1):
        B.batch_knn_cosine_multi(qs, vb, 10, engine=innr.KNN_AUTO, stats=st)
        assert st.engine == innr.KNN_MFMA_BF16  # ... the bf16 one when the int8 one is ruled out (its copy exists already)
    _check_knn(B, innr, "dot", vb, data, qs, 10, innr.KNN_AUTO)
    B.batch_knn_multi(qs, vb, 10, engine=innr.KNN_AUTO, stats=st)
    assert st.engine == innr.KNN_MFMA_I8  # squared L2 has its own int8 copy (D + R + 1 dimensions)
    _check_knn(B, innr, "l2", vb, data, qs, 10, innr.KNN_AUTO)
    with _lib.default_context().option("no_auto_i8", 1):
        B.batch_knn_multi(qs, vb, 10, engine=innr.KNN_AUTO, stats=st)
        assert st.engine == innr.KNN_MFMA_BF16
    B.batch_knn_dot_multi(qs, vb, 100, engine=innr.KNN_AUTO, stats=st)
    # k > 48: the int8 filter with lists of k + 16 and its completion pass (collect mode) behind them; same answers
    assert st.engine == innr.KNN_MFMA_I8
    _check_knn(B, innr, "dot", vb, data, qs, 100, innr.KNN_AUTO)
    _check_knn(B, innr, "cos", vb, data, qs, 200, innr.KNN_MFMA_I8)  # lists of 256: the two-limb kernel, then the one-limb collect pass


def test_bf16_filter_near_ties_are_redone_exactly(B, innr):
    # the reference example's LCG data: a one-parameter family with seas of near-ties; proofs fail, results must not
    rows, data = _corpus(10_000, 128, 0)
    _check_knn(B, innr, "dot", rows, data, _queries(40, 128), 10, innr.KNN_MFMA_BF16)


def test_bf16_engine_l2_and_what_the_f32_engine_still_serves(B, innr):
    rows, data = _corpus(70_000, 64, 5, uniform=True)
    vb = B.VerticalBatch.from_rows(rows)
    qs = _queries(20, 64, 9, uniform=True)
    st = innr.KnnStats()
    for engine in (innr.KNN_MFMA_BF16, innr.KNN_MFMA_I8):
        B.batch_knn_multi(qs, vb, 10, engine=engine, stats=st)
        assert st.engine == engine and st.queries_fallback <= 2
        _check_knn(B, innr, "l2", vb, data, qs, 10, engine)
    # D + 6 columns no longer fit the last K-step pair (D = 64 -> 128 columns), a query far outside the corpus, a zero query,
    # a zero row, a corpus far from the origin (|v|^2 dwarfs the differences: proofs fail, answers must not)
    qs2 = qs.copy()
    qs2[0] *= 1e3
    qs2[1] = 0.0
    rows2 = rows.copy()
    rows2[11] = 0.0
    _check_knn(B, innr, "l2", rows2, oracle.from_rows(rows2), qs2, 7, innr.KNN_MFMA_BF16)
    far = (rows + np.float32(300.0)).astype(np.float32)
    _check_knn(B, innr, "l2", far, oracle.from_rows(far), (qs + np.float32(300.0)).astype(np.float32), 5, innr.KNN_MFMA_BF16)
    huge = (rows * np.float32(1e17)).astype(np.float32)  # |v|^2 would overflow: the f32 engine's direct differences serve it
    B.batch_knn_multi(qs, B.VerticalBatch.from_rows(huge), 5, engine=innr.KNN_MFMA_BF16, stats=st)
    assert st.engine == innr.KNN_MFMA
    st = innr.KnnStats()
    B.batch_knn_dot_multi(qs, vb, 100, engine=innr.KNN_MFMA_BF16, stats=st)  # k > 48: candidate lists would not fit
    assert st.engine == innr.KNN_MFMA
    _check_knn(B, innr, "dot", vb, data, qs, 100, innr.KNN_MFMA_BF16)


def test_bf16_filter_tiny_and_special_values(B, innr):
    # scores near the denormal range (bf16 products may flush): the engine must hand those to the exact path
    rows, data = _corpus(70_000, 64, 3, uniform=True)
    tiny = (rows * np.float32(1e-20)).astype(np.float32)
    qs = _queries(6, 64, 1, uniform=True)
    _check_knn(B, innr, "dot", tiny, oracle.from_rows(tiny), (qs * np.float32(1e-18)).astype(np.float32), 10, innr.KNN_MFMA_BF16)
    _check_knn(B, innr, "l2", tiny, oracle.from_rows(tiny), (qs * np.float32(1e-18)).astype(np.float32), 10, innr.KNN_MFMA_BF16)
    rows[17, 3] = np.inf  # a non-finite norm: nothing can be proven
    _check_knn(B, innr, "dot", rows, oracle.from_rows(rows), qs, 5, innr.KNN_MFMA_BF16)
    _check_knn(B, innr, "l2", rows, oracle.from_rows(rows), qs, 5, innr.KNN_MFMA_BF16)


def test_bf16_filter_on_a_prefix_view(B, innr):
    rows, data = _corpus(70_000, 128, 11, uniform=True)
    vb = B.VerticalBatch.from_rows(rows)
    v = vb.prefix(64)
    for metric in ("dot", "l2"):
        _check_knn(B, innr, metric, v, np.ascontiguousarray(data[:64]), np.ascontiguousarray(_queries(30, 128, 99, uniform=True)[:, :64]), 10,
                   innr.KNN_MFMA_BF16)


def test_int8_filter_of_f32_corpus_special_cases(B, innr):
    """INNR_KNN_MFMA_I8 on an f32 batch: a corpus with outliers (the global range is wide, the bound large: proofs fail,
    answers must not), a constant corpus (nothing to quantise against: the f32 engine serves it), non-finite values, squared L2,
    k > 48 (lists of k + 16 + the completion pass), near-tie data, zero-norm rows / queries under cosine."""
    rows, data = _corpus(70_000, 64, 13, uniform=True)
    qs = _queries(40, 64, 3, uniform=True)
    out = rows.copy()
    out[5, 3] = 1e4
    out[9, 0] = -3e3
    _check_knn(B, innr, "dot", out, oracle.from_rows(out), qs, 10, innr.KNN_MFMA_I8)
    _check_knn(B, innr, "cos", out, oracle.from_rows(out), qs, 10, innr.KNN_MFMA_I8)
    const = np.full((70_000, 16), 0.5, np.float32)
    st = innr.KnnStats()
    vb = B.VerticalBatch.from_rows(const)
    B.batch_knn_dot_multi(_queries(20, 16, 1, uniform=True), vb, 5, engine=innr.KNN_MFMA_I8, stats=st)
    assert st.engine == innr.KNN_MFMA
    _check_knn(B, innr, "dot", vb, oracle.from_rows(const), _queries(20, 16, 1, uniform=True), 5, innr.KNN_MFMA_I8)
    bad = rows.copy()
    bad[17, 3] = np.inf
    _check_knn(B, innr, "dot", bad, oracle.from_rows(bad), qs, 5, innr.KNN_MFMA_I8)
    vb = B.VerticalBatch.from_rows(rows)
    B.batch_knn_multi(qs, vb, 10, engine=innr.KNN_MFMA_I8, stats=st)
    assert st.engine == innr.KNN_MFMA_I8  # squared L2: the augmented copy
    _check_knn(B, innr, "l2", out, oracle.from_rows(out), qs, 10, innr.KNN_MFMA_I8)  # outliers: |v|^2 up to 1e8 beside ~21
    _check_knn(B, innr, "l2", bad, oracle.from_rows(bad), qs, 5, innr.KNN_MFMA_I8)
    B.batch_knn_dot_multi(qs, vb, 100, engine=innr.KNN_MFMA_I8, stats=st)
    assert st.engine == innr.KNN_MFMA_I8  # k > 48: lists of k + 16 and the filter's completion pass
    _check_knn(B, innr, "dot", vb, data, qs, 100, innr.KNN_MFMA_I8)
    B.batch_knn_dot_multi(qs, vb, 241, engine=innr.KNN_MFMA_I8, stats=st)
    assert st.engine == innr.KNN_EXACT  # beyond INNR_MAX_K: all scores + a sort
    _check_knn(B, innr, "l2", vb, data, qs, 10, innr.KNN_MFMA_I8)
    z = rows.copy()
    z[5] = 0.0
    qz = qs.copy()
    qz[3] = 0.0
    _check_knn(B, innr, "cos", z, oracle.from_rows(z), qz, 10, innr.KNN_MFMA_I8)
    lrows, ldata = _corpus(70_000, 128, 0)  # the reference example's LCG data: proofs fail, results must not
    _check_knn(B, innr, "dot", lrows, ldata, _queries(24, 128), 10, innr.KNN_MFMA_I8)


def test_int8_filter_off_centre_range_and_extreme_thresholds(B, innr):
    """(1) a corpus whose range is far from zero (values in [0, 1]: ReLU embeddings, tf-idf) with equal-magnitude
    queries -- the query SUM that multiplies the range centre is accumulated in f32, and its error belongs in the proof's bound;
    (2) thresholds in the top band of the integer domain (D = 768, a constant negative query against near-minimum rows with
    distinct scores): the integer threshold must saturate only at the real limit of V. Answers: the exact engine's."""
    rng = np.random.default_rng(77)
    rows = rng.uniform(0.0, 1.0, size=(60_000, 768)).astype(np.float32)
    vb = B.VerticalBatch.from_rows(rows)
    qs = np.where(rng.uniform(size=(160, 768)) < 0.5, np.float32(-1.0), np.float32(1.0)).astype(np.float32)
    qs[1] = 1.0  # the largest possible |sum|
    qs[2] = -1.0  # a constant negative query: the best rows are the near-minimum ones
    rows2 = rows.copy()
    rows2[:300] = (np.arange(300, dtype=np.float32)[:, None] * np.float32(1e-6)) + np.float32(1e-4)  # > 256 near-minimum rows, distinct scores
    vb2 = B.VerticalBatch.from_rows(rows2)
    for batch, k in ((vb, 10), (vb2, 32), (vb2, 10)):
        for fn in (B.batch_knn_dot_multi, B.batch_knn_cosine_multi):
            st = innr.KnnStats()
            i1, s1 = fn(qs, batch, k, engine=innr.KNN_MFMA_I8, stats=st)
            assert st.engine == innr.KNN_MFMA_I8
            i0, s0 = fn(qs[:24], batch, k, engine=innr.KNN_EXACT)
            assert np.array_equal(i1[:24], i0) and bits_equal(s1[:24], s0)


@pytest.mark.parametrize("n,dim,nq,k", [(70_000, 64, 130, 10), (70_000, 100, 40, 100), (66_000, 7, 33, 5), (100_000, 200, 3, 10),
                                        (70_000, 64, 1, 48)])
def test_int8_filter_squared_l2(B, innr, n, dim, nq, k):
    """squared L2 on the int8 filter (pack_corpus_f32_i8_kernel: the corpus copy carries |v|^2 as two 8-bit limbs in R + 1 more
    dimensions, the query is [2q, -w1 x R, -w2]): centred rows with unequal norms, an off-centre range (ReLU-like: the offset
    terms of K0 and B_j matter), rows far from the origin (C_j - distance loses bits: the f32 terms of the bound), queries far
    outside the corpus; k beyond the direct lists goes through the collect pass. Same distances bit for bit, same index lists."""
    rows, _ = _corpus(n, dim, 31, uniform=True)
    rows = (rows * (1.0 + 0.7 * np.sin(np.arange(n, dtype=np.float32)))[:, None]).astype(np.float32)
    qs = _queries(nq, dim, 777, uniform=True)
    st = innr.KnnStats()
    vb = _check_knn(B, innr, "l2", rows, oracle.from_rows(rows), qs, k, innr.KNN_MFMA_I8)
    B.batch_knn_multi(qs, vb, k, engine=innr.KNN_MFMA_I8, stats=st)
    assert st.engine == innr.KNN_MFMA_I8
    if nq >= 16 and k <= 48:
        assert st.queries_fallback <= max(2, nq // 10), st.queries_fallback
    relu = np.maximum(rows, 0).astype(np.float32)
    _check_knn(B, innr, "l2", relu, oracle.from_rows(relu), np.abs(qs), k, innr.KNN_MFMA_I8)
    far = (rows + np.float32(50.0)).astype(np.float32)
    _check_knn(B, innr, "l2", far, oracle.from_rows(far), (qs + np.float32(50.0)).astype(np.float32), k, innr.KNN_MFMA_I8)
    _check_knn(B, innr, "l2", vb, oracle.from_rows(rows), (qs * np.float32(40.0)).astype(np.float32), k, innr.KNN_MFMA_I8)
    _check_knn(B, innr, "dot", vb, oracle.from_rows(rows), qs, min(k, 48), innr.KNN_MFMA_I8)  # (the dot copy beside it, same range)


@pytest.mark.parametrize("dim,nq", [(64, 64), (200, 5), (300, 1), (500, 3), (600, 2), (768, 4), (1000, 2),
                                    (500, 65), (600, 100), (768, 128), (1000, 70),  # (65 .. 128 queries: four column tiles per wave)
                                    (600, 200), (768, 256)])  # (two query groups: a block per group and corpus slice)
def test_int8_small_batch_kernel_every_k_step_count(B, innr, dim, nq, ctx_option):
    """gemm_i8s_filter_kernel (at most 128 queries on a corpus large enough for seeded bounds: every wave streams quarter tiles of its
    own, the queries' high limbs in LDS): one instantiation per K-step count 2, 4, ... 16 -- dot / cosine / squared L2 (whose copy
    has D + R + 1 dimensions: the next count up for some of these) against the oracle, and against the 512-query-tile kernel
    (option i8_no_small) bit for bit."""
    n = 140_000  # >= 32 x 4096 rows: seeded
    rows, _ = _corpus(n, dim, 5, uniform=True)
    rows = (rows * (1.0 + 0.5 * np.sin(np.arange(n, dtype=np.float32)))[:, None]).astype(np.float32)
    data = oracle.from_rows(rows)
    qs = _queries(nq, dim, 99, uniform=True)
    vb = B.VerticalBatch.from_rows(rows)
    for metric, fn in (("dot", B.batch_knn_dot_multi), ("cos", B.batch_knn_cosine_multi), ("l2", B.batch_knn_multi)):
        _check_knn(B, innr, metric, vb, data, qs, 10, innr.KNN_MFMA_I8)
        st = innr.KnnStats()
        i1, s1 = fn(qs, vb, 10, engine=innr.KNN_MFMA_I8, stats=st)
        assert st.engine == innr.KNN_MFMA_I8 and st.queries_fallback <= 1
        ctx_option("i8_no_small", 1)
        i2, s2 = fn(qs, vb, 10, engine=innr.KNN_MFMA_I8)
        ctx_option("i8_no_small", 0)
        assert np.array_equal(i1, i2) and bits_equal(s1, s2)


@pytest.mark.parametrize("k,nq", [(20, 9), (48, 70), (33, 1)])
def test_int8_small_batch_k_17_to_48(B, innr, k, nq):
    """k = 17 .. 48 in a small batch: lists of 128 on the small-batch kernel plus its collect pass for the proofs that fail (direct
    lists would be 256 long: the two-limb kernel). Same answers as the oracle."""
    n, dim = 140_000, 128
    rows, _ = _corpus(n, dim, 8, uniform=True)
    data = oracle.from_rows(rows)
    qs = _queries(nq, dim, 17, uniform=True)
    vb = None
    for metric in ("dot", "cos", "l2"):
        vb = _check_knn(B, innr, metric, vb if vb is not None else rows, data, qs, k, innr.KNN_MFMA_I8)
