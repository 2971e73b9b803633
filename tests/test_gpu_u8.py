"""GPU parity tests, part 4: scalar-quantised corpus (src/scalar.rs) -- device code layout, bit-exact asymmetric
scores, batch_knn_u8 on the exact engine and on the GEMM engine (u8 codes widened to f32 for the f32 MFMA)."""
from __future__ import annotations

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle
import kat_cases as K
from backends import HipBackend
from test_gpu_exact import bits_equal, same_knn


@pytest.fixture(scope="module")
def S():
    from innr_amd import scalar
    return scalar


@pytest.fixture(scope="module")
def innr():
    import innr_amd
    return innr_amd


def _codes(n, dim, seed, alpha=2.0, offset=-1.0):
    rows = oracle.generate_uniform(n, dim, seed)
    return oracle.quantize_u8(rows, oracle.QParams(alpha, offset))  # (n, dim) uint8


def _oracle_knn(q, codes, alpha, offset, k):
    return oracle.batch_knn_u8(q, codes, oracle.QParams(alpha, offset), k)


def test_reference_kat_batch_knn_u8(S):  # src/scalar.rs:582-606
    K.kat_batch_knn_u8(HipBackend(), lambda v, a, o: S.quantize_u8(v, S.QuantizationParams(a, o)).data())


@pytest.mark.parametrize("n,dim", [(1, 1), (5, 3), (1023, 16), (1024, 33), (1025, 64), (5000, 768)])
def test_u8_layout_generator_and_scores_bit_exact(S, n, dim):
    p = S.QuantizationParams.from_range(-1.0, 1.0)
    codes = _codes(n, dim, 11)
    qc = S.QuantizedCorpus.from_codes(codes, n, dim, p)
    assert np.array_equal(qc.codes(), codes)
    gen = S.QuantizedCorpus.generate(n, dim, p, seed=11)
    assert np.array_equal(gen.codes(), codes)  # device quantize_u8(generator) == host quantize_u8(generator)
    op = oracle.QParams(p.alpha, p.offset)
    for q in oracle.generate_uniform(3, dim, 5):
        exp = np.array([oracle.asymmetric_dot_u8(q, codes[i], op) for i in range(min(n, 300))], dtype=np.float32)
        got = qc.scores(q)
        assert got.shape == (n,) and bits_equal(got[:len(exp)], exp)


@pytest.mark.parametrize("n,dim,nq,k", [(4, 3, 1, 2), (300, 16, 3, 10), (5000, 64, 9, 33), (20_000, 128, 5, 100),
                                        (3000, 20, 2, 240), (3000, 20, 2, 241), (3000, 20, 2, 3000), (700, 33, 1, 10**6)])
def test_batch_knn_u8_exact_engine(S, innr, n, dim, nq, k):
    alpha, offset = 2.0, -1.0
    codes = _codes(n, dim, 3)
    qc = S.QuantizedCorpus.from_codes(codes, n, dim, S.QuantizationParams(alpha, offset))
    qs = oracle.generate_uniform(nq, dim, 77)
    idx, sc = qc.knn_multi(qs, k, engine=innr.KNN_EXACT)
    for j in range(nq):
        oi, os_ = _oracle_knn(qs[j], codes, alpha, offset, k)
        assert same_knn("dot", idx[j], sc[j], oi, os_), (j, idx[j], oi)
    # reference-shaped call: list of QuantizedU8, returns [(index, score)]
    corpus = [S.QuantizedU8(codes[i], dim) for i in range(min(n, 500))]
    res = S.batch_knn_u8(qs[0], corpus, S.QuantizationParams(alpha, offset), 7)
    oi, os_ = _oracle_knn(qs[0], codes[:len(corpus)], alpha, offset, 7)
    assert [r[0] for r in res] == oi.tolist() and bits_equal(np.float32([r[1] for r in res]), os_)


def _i8_dense(qc, queries):
    """test hook (outside include/innr_hip.h): the int8 engine's dense approximate scores + per-query constants"""
    import ctypes as C
    from innr_amd import _lib
    from conftest import hooks_lib
    fn = hooks_lib().innrdbg_i8_scores
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t)]
    q = np.ascontiguousarray(queries, np.float32)
    nq = q.shape[0]
    out = np.empty((nq, len(qc)), np.float32)
    qcst = np.empty((5, (nq + 511) // 512 * 512), np.float32)  # room for either tile width
    got_pad = C.c_size_t(0)
    _lib.check(fn(qc._h, q.ctypes.data, nq, q.shape[1], out.ctypes.data, qcst.ctypes.data, C.byref(got_pad)))
    qpad = got_pad.value
    assert qpad in ((nq + 255) // 256 * 256, (nq + 511) // 512 * 512)
    return out, qcst.reshape(-1)[:5 * qpad].reshape(5, qpad)[:, :nq]


@pytest.mark.parametrize("two_limb", ["0", "1"])
@pytest.mark.parametrize("n,dim,nq", [(128, 64, 1), (300, 33, 5), (1000, 128, 70), (1025, 200, 300), (3000, 768, 33)])
def test_i8_engine_dense_scores_are_the_limb_arithmetic_exactly(S, n, dim, nq, two_limb, ctx_option):
    """The int8 MFMA operand layout and the limb arithmetic, checked EXACTLY: with s = max|q| / T and t = round(q / s)
    (T = (R1 << S) + 2^(S-1) - 1, R1 from the dimension; S = 6: one limb on the matrix pipe + the exact low limb of
    v_dot4_i32_i8, S = 8: both limbs on the pipe), the kernel's V must equal sum_d (c_d - 128) t_d as integers -- asymmetric
    corpus and queries catch a transposed accumulator map or a k-order mismatch between the operands -- and its approximate
    score A V + B must stay within the engine's own bound E of the reference's asymmetric dot."""
    ctx_option("i8_two_limb", int(two_limb))
    shift = 8 if two_limb == "1" else 6
    alpha, offset = 2.0, -1.0
    codes = _codes(n, dim, 21, alpha, offset)
    qc = S.QuantizedCorpus.from_codes(codes, n, dim, S.QuantizationParams(alpha, offset))
    qs = oracle.generate_uniform(nq, dim, 22)
    qs[0, :] *= np.float32(1e-3)  # a small-magnitude query: its own scale
    got, (A, Bc, invA, E, lob) = _i8_dense(qc, qs)
    tmax = (2 ** 31 - 1) // (dim * 128)
    lo_max = 2 ** (shift - 1) - 1
    r1 = min(127, (tmax - lo_max) >> shift)
    T = np.float32((r1 << shift) + lo_max)
    a255 = np.float32(alpha) / np.float32(255.0)
    cp = codes.astype(np.int64) - 128
    op = oracle.QParams(alpha, offset)
    for j in range(nq):
        mx = np.abs(qs[j]).max()
        s = np.float32(mx) / T
        t = np.clip(np.rint(qs[j] * (np.float32(1.0) / s)), -T, T).astype(np.int64)
        V = cp @ t  # exact
        assert np.abs(V).max() < 2 ** 31
        # the low limb's reach the one-limb kernel bounds its fast reject with: LOB >= |sum c' r2| for every row
        r1v = (t + 2 ** (shift - 1)) >> shift
        r2v = t - (r1v << shift)
        assert int(lob[j].view(np.int32)) == 128 * int(np.abs(r2v).sum()) and np.abs(cp @ r2v).max() <= int(lob[j].view(np.int32))
        assert np.float32(a255 * s).view(np.uint32) == A[j].view(np.uint32)
        want = (np.float64(A[j]) * V.astype(np.float64) + np.float64(Bc[j])).astype(np.float32)  # one rounding, like the fma
        ulp = np.spacing(np.abs(want).astype(np.float32))
        assert np.all(np.abs(got[j].astype(np.float64) - want.astype(np.float64)) <= ulp), j
        exact = np.array([oracle.asymmetric_dot_u8(qs[j], codes[i], op) for i in range(0, n, max(1, n // 64))], np.float32)
        sub = got[j][::max(1, n // 64)]
        assert np.all(np.abs(sub.astype(np.float64) - exact.astype(np.float64)) <= E[j] + 1e-6 * np.abs(exact)), (j, E[j])
        assert E[j] < 0.02 * (np.abs(exact).max() + 1.0)  # the bound is tiny against the scores (14- / 16-bit query values)


@pytest.mark.parametrize("engine_name", ["f32-mfma", "int8-mfma", "int8-mfma-two-limbs"])
@pytest.mark.parametrize("n,dim,nq,k,alpha,offset", [(300, 16, 20, 10, 2.0, -1.0), (10_000, 128, 100, 10, 2.0, -1.0),
                                                     (20_000, 96, 300, 100, 3.5, -0.25), (1030, 768, 17, 16, 2.0, -1.0),
                                                     (70_000, 40, 513, 5, 1.0, 0.0), (5000, 130, 9, 240, 2.0, -1.0)])
def test_batch_knn_u8_gemm_engine(S, innr, n, dim, nq, k, alpha, offset, engine_name, ctx_option):
    ctx_option("i8_two_limb", 1 if engine_name == "int8-mfma-two-limbs" else 0)
    engine = {"f32-mfma": innr.KNN_MFMA, "int8-mfma": innr.KNN_MFMA_I8, "int8-mfma-two-limbs": innr.KNN_MFMA_I8}[engine_name]
    codes = _codes(n, dim, 4, alpha, offset)
    qc = S.QuantizedCorpus.from_codes(codes, n, dim, S.QuantizationParams(alpha, offset))
    qs = oracle.generate_uniform(nq, dim, 78)
    st = innr.KnnStats()
    idx, sc = qc.knn_multi(qs, k, engine=engine, stats=st)
    assert st.engine == engine
    for j in range(nq):
        oi, os_ = _oracle_knn(qs[j], codes, alpha, offset, k)
        assert same_knn("dot", idx[j], sc[j], oi, os_), (j, idx[j], oi)
    # u8 codes collide often (coarse grid), so some margin proofs legitimately fail; most must hold
    assert st.queries_fallback <= max(2, nq // 4), st.queries_fallback


@pytest.mark.parametrize("nq,dim", [(1, 128), (9, 96), (64, 200), (100, 512), (128, 768)])
def test_batch_knn_u8_small_batch_kernel(S, innr, nq, dim, ctx_option):
    """at most 128 queries, lists of 128 (k = 100) and a corpus large enough for seeded bounds: the int8 engine runs
    gemm_i8s_filter_kernel (one wave = one slice, queries in LDS); same answers as the oracle and as the 512-query-tile kernel"""
    n, k, alpha, offset = 140_000, 100, 2.0, -1.0
    codes = _codes(n, dim, 11, alpha, offset)
    qc = S.QuantizedCorpus.from_codes(codes, n, dim, S.QuantizationParams(alpha, offset))
    qs = oracle.generate_uniform(nq, dim, 321)
    st = innr.KnnStats()
    idx, sc = qc.knn_multi(qs, k, engine=innr.KNN_MFMA_I8, stats=st)
    assert st.engine == innr.KNN_MFMA_I8
    for j in range(nq):
        oi, os_ = _oracle_knn(qs[j], codes, alpha, offset, k)
        assert same_knn("dot", idx[j], sc[j], oi, os_), (j, idx[j], oi)
    ctx_option("i8_no_small", 1)
    idx2, sc2 = qc.knn_multi(qs, k, engine=innr.KNN_MFMA_I8)
    assert np.array_equal(idx, idx2) and np.array_equal(sc.view(np.uint32), sc2.view(np.uint32))


def test_u8_edge_cases(S, innr):
    p = S.QuantizationParams.from_range(0.0, 1.0)
    assert S.batch_knn_u8([1.0], [], p, 5) == []  # scalar.rs:601-605
    qc = S.QuantizedCorpus.from_codes(np.zeros((3, 4), np.uint8), 3, 4, p)
    assert S.batch_knn_u8([1.0, 2.0, 3.0, 4.0], qc, p, 0) == []
    with pytest.raises(innr.InnrPanic):
        qc.knn_multi(np.ones((1, 5), np.float32), 2)  # "dimension mismatch" scalar.rs:290
    r = S.batch_knn_u8([1.0, 1.0, 1.0, 1.0], qc, p, 10)  # all-equal scores, k > N: indices 0,1,2
    assert [i for i, _ in r] == [0, 1, 2]
    with pytest.raises(innr.InnrError):
        from innr_amd import batch as B
        B.batch_dot([1.0, 2.0, 3.0, 4.0], type("X", (), {"_h": qc._h, "dimension": lambda s: 4, "num_vectors": lambda s: 3})())


def test_u8_engines_agree_large(S, innr):
    p = S.QuantizationParams.from_range(-1.0, 1.0)
    qc = S.QuantizedCorpus.generate(2_000_000, 128, p, seed=1)
    qs = oracle.generate_uniform(256, 128, 9)
    i2, s2 = qc.knn_multi(qs[:32], 10, engine=innr.KNN_EXACT)
    for engine in (innr.KNN_MFMA, innr.KNN_MFMA_I8, innr.KNN_AUTO):
        st = innr.KnnStats()
        i1, s1 = qc.knn_multi(qs, 10, engine=engine, stats=st)
        assert np.array_equal(i1[:32], i2) and bits_equal(s1[:32], s2)
        assert st.engine == (innr.KNN_MFMA_I8 if engine == innr.KNN_AUTO else engine)  # AUTO: the int8 filter
        print(f"u8 2Mx128 256q engine {st.engine}: gemm {st.gemm_ms:.2f} ms total {st.total_ms:.2f} ms fallback {st.queries_fallback}")
    st = innr.KnnStats()
    i1, s1 = qc.knn_multi(qs[:13], 10, engine=innr.KNN_AUTO, stats=st)  # the copy exists: the int8 engine's small-batch kernel
    assert st.engine == innr.KNN_MFMA_I8 and np.array_equal(i1, i2[:13]) and bits_equal(s1, s2[:13])
    i1, s1 = qc.knn_multi(qs[:1], 10, engine=innr.KNN_AUTO, stats=st)  # one query: one pass of the exact scan streams the same bytes
    assert st.engine == innr.KNN_EXACT and np.array_equal(i1, i2[:1]) and bits_equal(s1, s2[:1])
    qc2 = S.QuantizedCorpus.generate(200_000, 128, p, seed=2)
    qc2.knn_multi(qs[:3], 10, engine=innr.KNN_AUTO, stats=st)  # no copy yet, three queries: not worth building
    assert st.engine == innr.KNN_EXACT
    qc2.knn_multi(qs[:4], 10, engine=innr.KNN_AUTO, stats=st)
    assert st.engine == innr.KNN_MFMA_I8


@pytest.mark.parametrize("n,dim,nq,k", [(300_000, 320, 600, 10), (200_000, 1024, 130, 100), (150_000, 256, 1030, 33)])
def test_i8_engine_longer_rows_and_many_tiles(S, innr, n, dim, nq, k):
    """4 to 16 K-steps of 64 dimensions per tile, several tiles per slice, lists of 64 / 128 / 256 (two limbs): the answers are
    the exact engine's, bit for bit"""
    p = S.QuantizationParams.from_range(-1.0, 1.0)
    qc = S.QuantizedCorpus.generate(n, dim, p, seed=5)
    qs = oracle.generate_uniform(nq, dim, 77)
    st = innr.KnnStats()
    i1, s1 = qc.knn_multi(qs, k, engine=innr.KNN_MFMA_I8, stats=st)
    assert st.engine == innr.KNN_MFMA_I8
    i2, s2 = qc.knn_multi(qs[:64], k, engine=innr.KNN_EXACT)
    assert np.array_equal(i1[:64], i2) and bits_equal(s1[:64], s2)
    i3, s3 = qc.knn_multi(qs[-40:], k, engine=innr.KNN_EXACT)
    assert np.array_equal(i1[-40:], i3) and bits_equal(s1[-40:], s3)


def test_no_environment_variable_reaches_the_call_path(S, innr):
    """The timing probes of the int8 kernel are compile-time switches of separate builds (tools/i8h_probe.py) and the tuning
    options are read once, at innr_ctx_create: an environment variable set afterwards changes nothing -- in particular
    INNR_I8H_PROBE, which in round 2 made the product library skip its visits, is inert."""
    import os
    p = S.QuantizationParams.from_range(-1.0, 1.0)
    qc = S.QuantizedCorpus.generate(200_000, 256, p, seed=2)
    qs = oracle.generate_uniform(600, 256, 3)
    ref_i, ref_s = qc.knn_multi(qs, 10, engine=innr.KNN_MFMA_I8)
    e_i, e_s = qc.knn_multi(qs[:24], 10, engine=innr.KNN_EXACT)
    assert np.array_equal(ref_i[:24], e_i) and bits_equal(ref_s[:24], e_s)
    os.environ["INNR_I8H_PROBE"] = "1"
    os.environ["INNR_I8_TWO_LIMB"] = "1"
    try:
        st = innr.KnnStats()
        i2, s2 = qc.knn_multi(qs, 10, engine=innr.KNN_MFMA_I8, stats=st)
    finally:
        del os.environ["INNR_I8H_PROBE"]
        del os.environ["INNR_I8_TWO_LIMB"]
    assert np.array_equal(i2, ref_i) and bits_equal(s2, ref_s) and st.engine == innr.KNN_MFMA_I8


def test_i8_engine_special_queries_and_params(S, innr):
    """zero / tiny / huge / non-finite queries and parameter sets the int8 limbs cannot represent: the answer is the
    oracle's in every case (unprovable queries take the exact engine; alpha <= 0 is served by the f32 GEMM engine)"""
    n, dim, k = 70_000, 48, 10
    codes = _codes(n, dim, 31)
    qs = oracle.generate_uniform(40, dim, 32)
    qs[1] = 0.0
    qs[2] *= np.float32(1e-30)
    qs[3] *= np.float32(1e30)
    qs[4, 7] = np.inf
    qs[5, 0] = np.nan
    qs[6] = np.float32(0.5)            # constant query: ties everywhere in the quantised domain too
    qs[7, 1:] = 0.0                    # one non-zero dimension: massive exact ties among codes
    for alpha, offset in ((2.0, -1.0), (0.5, 3.0), (-2.0, 1.0)):
        qc = S.QuantizedCorpus.from_codes(codes, n, dim, S.QuantizationParams(alpha, offset))
        st = innr.KnnStats()
        idx, sc = qc.knn_multi(qs, k, engine=innr.KNN_MFMA_I8, stats=st)
        assert st.engine == (innr.KNN_MFMA_I8 if alpha > 0 else innr.KNN_MFMA)
        for j in range(len(qs)):
            oi, os_ = _oracle_knn(qs[j], codes, alpha, offset, k)
            assert same_knn("dot", idx[j], sc[j], oi, os_), (alpha, j, idx[j], oi)


# ---------------------------------------------------------------- corpus ingest on the device, two-stage pipeline
def test_device_quantize_and_fit_match_host(S, innr):
    from innr_amd import batch as B
    n, dim = 3001, 37
    rows = (oracle.generate_uniform(n, dim, 8) * np.float32(2.5) + np.float32(0.3)).astype(np.float32)
    rows[5, 7] = np.nan   # ignored by fit (scalar.rs:76-83: both comparisons false), quantises to 0 (`as u8`)
    rows[9, 1] = 1e30     # saturates at 255
    vb = B.VerticalBatch.from_rows(rows)
    p = S.fit_batch(vb)
    hp = oracle.qparams_fit(rows.reshape(-1))
    assert np.float32(p.alpha).view(np.uint32) == np.float32(hp.alpha).view(np.uint32)
    assert np.float32(p.offset).view(np.uint32) == np.float32(hp.offset).view(np.uint32)
    p2 = S.QuantizationParams.from_range(-2.0, 3.0)
    qc = S.QuantizedCorpus.from_batch(vb, p2)
    assert np.array_equal(qc.codes(), oracle.quantize_u8(rows, oracle.QParams(p2.alpha, p2.offset)))
    empty = B.VerticalBatch.from_rows(np.empty((0, 0), np.float32))
    e = S.fit_batch(empty)
    assert (e.alpha, e.offset) == (1.0, 0.0)
    # degenerate corpora: `fit` scans from (f32::MAX, f32::MIN) and has no min > max guard (scalar.rs:68-87) -- a non-empty
    # all-NaN corpus gives {1.0, f32::MAX}; values beyond +-f32::MAX (the infinities) never replace the starting values
    for vals in (np.full((4, 3), np.nan, np.float32), np.full((2, 5), np.inf, np.float32), np.full((3, 2), -np.inf, np.float32),
                 np.array([[np.nan, np.inf], [-np.inf, np.nan]], np.float32), np.array([[np.nan, 2.0], [np.nan, np.nan]], np.float32)):
        dv = B.VerticalBatch.from_rows(vals)
        got, want, host = S.fit_batch(dv), oracle.qparams_fit(vals.reshape(-1)), S.QuantizationParams.fit(vals.reshape(-1))
        for x in (got, host):
            assert np.float32(x.alpha).view(np.uint32) == np.float32(want.alpha).view(np.uint32), (vals, x, want)
            assert np.float32(x.offset).view(np.uint32) == np.float32(want.offset).view(np.uint32), (vals, x, want)
        dv.close()
    nanp = S.fit_batch(B.VerticalBatch.from_rows(np.full((4, 3), np.nan, np.float32)))
    assert (nanp.alpha, np.float32(nanp.offset)) == (1.0, np.float32(3.4028235e38))


@pytest.mark.parametrize("metric", ["dot", "cos", "l2"])
@pytest.mark.parametrize("n,dim,nq,kc,k", [(2000, 48, 7, 40, 10), (500, 16, 3, 256, 256), (9000, 128, 20, 100, 100),
                                           (3000, 24, 5, 257, 20), (3000, 24, 4, 2500, 2500), (700, 8, 3, 700, 10**6)])
def test_batch_rerank_matches_oracle(innr, metric, n, dim, nq, kc, k):
    from innr_amd import batch as B
    rows = oracle.generate_uniform(n, dim, 12)
    data = oracle.from_rows(rows)
    vb = B.VerticalBatch.from_rows(rows)
    qs = oracle.generate_uniform(nq, dim, 13)
    rng = np.random.default_rng(5)
    cand = np.stack([rng.choice(n, size=kc, replace=False) for _ in range(nq)]).astype(np.uint64)
    met = {"dot": innr.METRIC_DOT, "cos": innr.METRIC_COSINE, "l2": innr.METRIC_L2SQ}[metric]
    idx, sc = B.batch_rerank(qs, vb, cand, k, met)
    norms = oracle.batch_norms(data)
    for j in range(nq):
        full = {"dot": lambda: oracle.batch_dot(qs[j], data), "cos": lambda: oracle.batch_cosine(qs[j], data, norms),
                "l2": lambda: oracle.batch_l2_squared(qs[j], data)}[metric]()
        c = np.sort(cand[j].astype(np.int64))  # index asc, then a stable sort by score: the kNN functions' order
        s = full[c]
        order = np.argsort(s.astype(np.float64) if metric == "l2" else -s.astype(np.float64), kind="stable")[:k]
        assert idx[j].tolist() == c[order].tolist() and bits_equal(sc[j], s[order]), (metric, j)
    with pytest.raises(innr.InnrError):
        B.batch_rerank(qs, vb, np.full((nq, 4), n + 5, np.uint64), 2, met)  # outside the batch
    if kc > 256:  # the sorted path flags out-of-range candidates too
        bad = cand.copy()
        bad[0, 5] = n + 1
        with pytest.raises(innr.InnrError):
            B.batch_rerank(qs, vb, bad, 2, met)


def test_two_stage_pipeline_u8_then_exact(S, innr):
    from innr_amd import batch as B
    n, dim, nq, k, kc = 200_000, 96, 64, 10, 100
    vb = B.VerticalBatch.generate(n, dim, seed=3)
    p = S.fit_batch(vb)
    qc = S.QuantizedCorpus.from_batch(vb, p)
    gen = S.QuantizedCorpus.generate(n, dim, p, seed=3)
    assert np.array_equal(qc.codes()[:2000], gen.codes()[:2000])  # device quantise(resident f32) == quantise(generator)
    qs = oracle.generate_uniform(nq, dim, 21)
    idx, sc = S.two_stage_knn(qs, qc, vb, k, kc)
    ei, es = B.batch_knn_dot_multi(qs, vb, k, engine=innr.KNN_EXACT)
    # exact scores, and with 10x over-fetch the 8-bit first pass loses nothing of the true top-10 on this data
    recall = np.mean([len(set(idx[j].tolist()) & set(ei[j].tolist())) / k for j in range(nq)])
    assert recall >= 0.99, recall
    for j in range(nq):
        common = [i for i in idx[j].tolist() if i in set(ei[j].tolist())]
        pos = {int(i): t for t, i in enumerate(ei[j].tolist())}
        got = {int(i): t for t, i in enumerate(idx[j].tolist())}
        assert all(sc[j][got[i]].view(np.uint32) == es[j][pos[i]].view(np.uint32) for i in common)


@pytest.mark.parametrize("engine_name", ["int8-mfma", "f32-mfma"])
def test_full_size_properties_c3(S, innr, engine_name):
    # BASELINE.json configs[2]: 50M x 768 u8 codes, 1024 queries, k = 100 ("int8 MFMA") -- through size-independent
    # properties: the filter engine equals the bit-exact engine on a query subset; results sorted, unique, in range; few redone.
    engine = {"f32-mfma": innr.KNN_MFMA, "int8-mfma": innr.KNN_MFMA_I8}[engine_name]
    n, dim, nq, k = 50_000_000, 768, 1024, 100
    p = S.QuantizationParams.from_range(-1.0, 1.0)
    qc = S.QuantizedCorpus.generate(n, dim, p, seed=0)
    qs = oracle.generate_uniform(nq, dim, 0xBE7C)
    st = innr.KnnStats()
    idx, sc = qc.knn_multi(qs, k, engine=engine, stats=st)
    assert idx.shape == (nq, k) and st.engine == engine and st.queries_fallback <= 8, st.queries_fallback
    assert np.all(sc[:, :-1] >= sc[:, 1:]) and int(idx.max()) < n
    assert all(len(set(r.tolist())) == k for r in idx[::37])
    i2, s2 = qc.knn_multi(qs[:8], k, engine=innr.KNN_EXACT)
    assert np.array_equal(idx[:8], i2) and bits_equal(sc[:8], s2)
    # spot check of the scores themselves: the oracle's asymmetric dot on the winners' regenerated codes
    op = oracle.QParams(p.alpha, p.offset)
    for j in (0, 511):
        for r in (0, k - 1):
            row = oracle.generate_uniform(1, dim, 0, row0=int(idx[j, r]))
            code = oracle.quantize_u8(row, op)[0]
            assert np.float32(oracle.asymmetric_dot_u8(qs[j], code, op)).view(np.uint32) == sc[j, r].view(np.uint32)
    print(f"C3 u8 ({engine_name}): gemm {st.gemm_ms:.1f} ms, total {st.total_ms:.1f} ms, fallback {st.queries_fallback}")
    qc.close()


def test_rerank_and_ingest_edge_cases(S, innr):
    from innr_amd import batch as B
    rows = oracle.generate_uniform(50, 8, 2)
    vb = B.VerticalBatch.from_rows(rows)
    q = oracle.generate_uniform(2, 8, 3)
    # k larger than the candidate list: everything comes back, ordered; k = 0 / no candidates: empty
    idx, sc = B.batch_rerank(q, vb, np.array([[7, 3, 9], [1, 2, 4]], np.uint64), 10)
    assert idx.shape == (2, 3) and np.all(sc[:, :-1] >= sc[:, 1:])
    assert sorted(idx[0].tolist()) == [3, 7, 9]
    idx, sc = B.batch_rerank(q, vb, np.array([[7, 3, 9], [1, 2, 4]], np.uint64), 0)
    assert idx.shape[1] == 0
    idx, sc = B.batch_rerank(q, vb, np.empty((2, 0), np.uint64), 5)
    assert idx.shape[1] == 0
    with pytest.raises(innr.InnrPanic):
        B.batch_rerank(np.ones((2, 9), np.float32), vb, np.array([[1], [2]], np.uint64), 1)  # dimension mismatch
    idx, sc = B.batch_rerank(q, vb, np.zeros((2, 300), np.uint64), 1)  # more than 256 candidates (here: all the same one)
    assert idx.tolist() == [[0], [0]]
    # index base: candidates are GLOBAL indices
    vb.set_index_base(1000)
    idx, sc = B.batch_rerank(q[:1], vb, np.array([[1007, 1003]], np.uint64), 2)
    assert sorted(idx[0].tolist()) == [1003, 1007]
    with pytest.raises(innr.InnrError):
        B.batch_rerank(q[:1], vb, np.array([[7]], np.uint64), 1)  # below the base
    vb.set_index_base(0)
    # quantising an all-NaN / constant corpus: fit gives alpha 1 (scalar.rs:57); NaN quantises to 0 (`as u8`)
    nanb = B.VerticalBatch.from_rows(np.full((4, 3), np.nan, np.float32))
    p = S.fit_batch(nanb)
    assert (p.alpha, np.float32(p.offset)) == (1.0, np.float32(3.4028235e38))  # scalar.rs:68-87 has no min > max guard
    assert np.all(S.QuantizedCorpus.from_batch(nanb, p).codes() == 0)
    const = B.VerticalBatch.from_rows(np.full((4, 3), 2.5, np.float32))
    p = S.fit_batch(const)
    assert (p.alpha, p.offset) == (1.0, 2.5)
    with pytest.raises(innr.InnrError):
        S.QuantizedCorpus.from_batch(S.QuantizedCorpus.from_codes(np.zeros((2, 2), np.uint8), 2, 2, p), p)  # not an f32 batch


@pytest.mark.parametrize("n,dim", [(1, 1), (257, 3), (3001, 37), (20_000, 64)])
@pytest.mark.parametrize("quantile", [0.5, 0.9, 0.99, 0.999, 1.0])
def test_device_fit_quantile_matches_reference_ranks(S, innr, n, dim, quantile):
    """QuantizationParams::fit_quantile (scalar.rs:104-139) on a resident batch: the two rank values of the FINITE values in
    total_cmp order, found by a radix select on the device -- bit-equal alpha / offset to the oracle's sort-based version,
    with NaN / +-inf / +-0.0 / duplicates / outliers in the data."""
    from innr_amd import batch as B
    rows = (oracle.generate_uniform(n, dim, 8) * np.float32(2.5) + np.float32(0.3)).astype(np.float32)
    flat = rows.reshape(-1)
    if flat.size > 50:
        flat[5] = np.nan
        flat[9] = 1e30
        flat[11] = -np.inf
        flat[12] = np.inf
        flat[13] = -0.0
        flat[14] = 0.0
        flat[20:40] = flat[19]  # duplicates
    vb = B.VerticalBatch.from_rows(rows)
    p = S.fit_quantile_batch(vb, quantile)
    hp = oracle.qparams_fit_quantile(rows.reshape(-1), quantile)
    assert np.float32(p.alpha).view(np.uint32) == np.float32(hp.alpha).view(np.uint32), (p, hp.alpha, hp.offset)
    assert np.float32(p.offset).view(np.uint32) == np.float32(hp.offset).view(np.uint32), (p, hp.alpha, hp.offset)


def test_device_fit_quantile_edge_cases(S, innr):
    from innr_amd import batch as B
    vb = B.VerticalBatch.from_rows(np.full((4, 3), np.nan, np.float32))
    p = S.fit_quantile_batch(vb, 0.9)
    assert (p.alpha, p.offset) == (1.0, 0.0)  # no finite value (scalar.rs:124-129)
    with pytest.raises(innr.InnrPanic):
        S.fit_quantile_batch(vb, 0.0)  # "quantile must be in (0.0, 1.0]"
    with pytest.raises(innr.InnrPanic):
        S.fit_quantile_batch(vb, 1.5)
    const = B.VerticalBatch.from_rows(np.full((10, 2), 0.25, np.float32))
    p = S.fit_quantile_batch(const, 0.5)
    assert (p.alpha, p.offset) == (1.0, 0.25)  # zero range -> alpha 1 (scalar.rs:57)


def test_quantized_corpus_save_load_roundtrip(S, innr, tmp_path):
    n, dim = 2049, 37
    p = S.QuantizationParams(3.5, -0.25)
    codes = _codes(n, dim, 13, p.alpha, p.offset)
    qc = S.QuantizedCorpus.from_codes(codes, n, dim, p)
    path = str(tmp_path / "corpus.u8")
    qc.save(path)
    back = S.QuantizedCorpus.load(path)
    assert len(back) == n and back.dimension() == dim and back.params == p
    assert np.array_equal(back.codes(), codes)
    q = oracle.generate_uniform(3, dim, 2)
    i1, s1 = qc.knn_multi(q, 7)
    i2, s2 = back.knn_multi(q, 7)
    assert np.array_equal(i1, i2) and bits_equal(s1, s2)
    with open(path, "r+b") as f:
        f.write(b"XXXXXXXX")
    with pytest.raises(innr.InnrPanic):
        S.QuantizedCorpus.load(path)
    empty = S.QuantizedCorpus.from_codes(np.empty((0, 5), np.uint8), 0, 5, p)
    empty.save(path)
    assert len(S.QuantizedCorpus.load(path)) == 0
