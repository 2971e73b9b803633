"""GPU parity tests, part 2: the f32-MFMA GEMM engine (fused top-k filter + exact re-score + margin proof).
Bar (BASELINE.json north_star): top-k index lists identical to the portable CPU path, scores within 1e-4
relative -- the engine actually delivers bit-identical scores because every reported score is re-computed in
the reference's arithmetic order."""
from __future__ import annotations

import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle
from test_gpu_exact import _check_knn, _corpus, _queries, bits_equal, same_knn


@pytest.fixture(scope="module")
def B():
    from innr_amd import batch
    return batch


@pytest.fixture(scope="module")
def innr():
    import innr_amd
    return innr_amd


def _dense_scores(vb, metric, queries):
    from innr_amd import _lib
    from conftest import hooks_lib
    fn = hooks_lib().innrdbg_gemm_scores  # test hook: outside include/innr_hip.h and outside the product library
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]
    q = np.ascontiguousarray(queries, np.float32)
    out = np.empty((q.shape[0], vb.num_vectors()), np.float32)
    _lib.check(fn(vb._h, metric, q.ctypes.data, q.shape[0], q.shape[1], out.ctypes.data))
    return out


def _mfma_knn_checked(B, innr, metric, rows, data, qs, k, max_fallback):
    """MFMA engine == oracle, AND the GEMM path itself did the work (few/no exact-engine fallbacks)."""
    vb = _check_knn(B, innr, metric, rows, data, qs, k, innr.KNN_MFMA)
    st = innr.KnnStats()
    fn = {"dot": B.batch_knn_dot_multi, "cos": B.batch_knn_cosine_multi, "l2": B.batch_knn_multi}[metric]
    fn(qs, vb, k, engine=innr.KNN_MFMA, stats=st)
    assert st.engine == innr.KNN_MFMA and st.queries_fallback <= max_fallback, st.queries_fallback
    return vb


# ------------------------------------------------------------------------------- operand / accumulator layout
@pytest.mark.parametrize("n,dim,nq", [(128, 16, 1), (300, 33, 5), (1000, 128, 70), (257, 2, 300), (5000, 768, 33)])
def test_gemm_dense_scores_match_oracle(B, innr, n, dim, nq):
    # tolerance: the reference's own SIMD-vs-scalar convention, 1e-4 * sum|a*b| + 1e-4 (tests/property_tests.rs:408)
    rows, data = _corpus(n, dim, 5, uniform=True)
    vb = B.VerticalBatch.from_rows(rows)
    qs = _queries(nq, dim, 777, uniform=True)
    got = _dense_scores(vb, innr.METRIC_DOT, qs)
    norms = oracle.batch_norms(data)
    gotc = _dense_scores(vb, innr.METRIC_COSINE, qs)
    for j in range(nq):
        exp = oracle.batch_dot(qs[j], data)
        tol = 1e-4 * (np.abs(qs[j])[:, None] * np.abs(data)).sum(axis=0) + 1e-4
        assert np.all(np.abs(got[j] - exp) <= tol), (j, np.abs(got[j] - exp).max())
        expc = oracle.batch_cosine(qs[j], data, norms)
        assert np.all(np.abs(gotc[j] - expc) <= 2e-4), (j, np.abs(gotc[j] - expc).max())


def test_gemm_layout_asymmetric_integers(B, innr):
    # exact small integers: every product and partial sum is exact in f32, so the MFMA result must EQUAL the
    # oracle's; an asymmetric corpus catches a transposed accumulator map (guide: "A=I with asymmetric B")
    n, dim, nq = 384, 48, 260
    rows = ((np.arange(n)[:, None] * 7 + np.arange(dim)[None, :] * 3) % 11 - 5).astype(np.float32)
    qs = ((np.arange(nq)[:, None] * 5 + np.arange(dim)[None, :]) % 7 - 3).astype(np.float32)
    vb = B.VerticalBatch.from_rows(rows)
    got = _dense_scores(vb, innr.METRIC_DOT, qs)
    exp = (qs.astype(np.int64) @ rows.astype(np.int64).T).astype(np.float32)
    assert np.array_equal(got, exp)


# ------------------------------------------------------------------------------- kNN parity
@pytest.mark.parametrize("metric", ["dot", "cos", "l2"])
def test_knn_mfma_c1_shape(B, innr, metric):
    # BASELINE.json configs[0] shape (10K x 128, 100 queries, k = 10) on the GEMM engine, uniform data
    rows, data = _corpus(10_000, 128, 0, uniform=True)
    _mfma_knn_checked(B, innr, metric, rows, data, _queries(100, 128, uniform=True), 10, max_fallback=1)


@pytest.mark.parametrize("metric", ["dot", "cos", "l2"])
def test_knn_mfma_c1_example_generator(B, innr, metric):
    # the reference example's own LCG data (examples/batch_demo.rs:167-170): a one-parameter family with seas of
    # near-ties, so most margin proofs fail and those queries are redone on the exact engine. Results must
    # still be identical.
    rows, data = _corpus(10_000, 128, 0)
    _check_knn(B, innr, metric, rows, data, _queries(100, 128), 10, innr.KNN_MFMA)


@pytest.mark.parametrize("n,dim,nq,k", [(1, 4, 1, 1), (5, 3, 2, 10), (255, 16, 3, 7), (257, 33, 5, 16), (1000, 64, 9, 33),
                                        (3000, 20, 300, 100), (2049, 8, 4, 240), (20_000, 100, 513, 10)])
def test_knn_mfma_ragged(B, innr, n, dim, nq, k):
    rows, data = _corpus(n, dim, 77, uniform=True)
    vb = None
    for metric in ("dot", "cos", "l2"):
        vb = _check_knn(B, innr, metric, vb if vb is not None else rows, data, _queries(nq, dim, 4242, uniform=True), k,
                        innr.KNN_MFMA)


def test_knn_mfma_many_tiles_and_compactions(B, innr):
    # 200K x 32 on few slices: candidate lists fill and compact repeatedly inside the GEMM kernel
    rows, data = _corpus(200_000, 32, 11, uniform=True)
    qs = _queries(40, 32, 99, uniform=True)
    _mfma_knn_checked(B, innr, "dot", rows, data, qs, 10, max_fallback=2)
    _mfma_knn_checked(B, innr, "cos", rows, data, qs, 100, max_fallback=2)
    _mfma_knn_checked(B, innr, "l2", rows, data, qs, 10, max_fallback=2)


def test_knn_mfma_ties_zero_query_and_stats(B, innr):
    base = oracle.generate_uniform(700, 24, 9)
    rows = np.concatenate([base, base, base[::-1]])  # every vector three times: exact ties at every rank
    data = oracle.from_rows(rows)
    qs = np.concatenate([_queries(20, 24, 31, uniform=True), np.zeros((1, 24), np.float32)])
    for metric in ("dot", "cos", "l2"):
        _check_knn(B, innr, metric, rows, data, qs, 12, innr.KNN_MFMA)
    vb = B.VerticalBatch.from_rows(rows)
    st = innr.KnnStats()
    B.batch_knn_dot_multi(qs, vb, 12, engine=innr.KNN_MFMA, stats=st)
    assert st.engine == innr.KNN_MFMA and st.candidates_kept == 32 and st.gemm_ms > 0.0
    # 12 = 4 triples: the cut falls between groups, the proof holds for random queries; the all-zero query ties
    # every vector at 0.0 and can only be settled by the exact engine
    assert 1 <= st.queries_fallback <= 4


def test_knn_mfma_nonfinite_falls_back_to_exact(B, innr):
    rows = oracle.generate_uniform(4000, 32, 1)
    rows[17, 3] = np.nan
    rows[300, 0] = np.inf
    data = oracle.from_rows(rows)
    qs = _queries(6, 32, 5, uniform=True)
    vb = B.VerticalBatch.from_rows(rows)
    st = innr.KnnStats()
    idx, sc = B.batch_knn_dot_multi(qs, vb, 5, engine=innr.KNN_MFMA, stats=st)
    assert st.queries_fallback == len(qs)  # the error bound is not finite: nothing can be proven, all redone
    for j, q in enumerate(qs):
        oi, os_ = oracle.batch_knn_dot(q, data, 5)
        assert same_knn("dot", idx[j], sc[j], oi, os_)
    idx, sc = B.batch_knn_multi(qs, vb, 5, engine=innr.KNN_MFMA, stats=st)
    assert st.queries_fallback == len(qs)
    for j, q in enumerate(qs):
        oi, os_ = oracle.batch_knn(q, data, 5)
        assert same_knn("l2", idx[j], sc[j], oi, os_)
    # cosine: its bound does not carry the corpus norm, yet a row with a NaN / inf component scores NaN * 0 on the matrix pipe
    # where the reference says 0.0 (norm NaN -> `vn > eps` false, batch.rs:722) -- every query must take the exact engine. The
    # corpus here is the NEGATED queries' neighbourhood: all finite cosines are negative, so the 0.0 of the broken rows is top-1.
    neg = (-np.abs(oracle.generate_uniform(4000, 32, 2))).astype(np.float32)
    neg[17, 3] = np.nan
    neg[300, 0] = np.inf
    qpos = np.abs(_queries(6, 32, 6, uniform=True)).astype(np.float32)
    dneg = oracle.from_rows(neg)
    for engine in (innr.KNN_MFMA, innr.KNN_MFMA_BF16):
        vbn = B.VerticalBatch.from_rows(neg)
        idx, sc = B.batch_knn_cosine_multi(qpos, vbn, 5, engine=engine, stats=st)
        assert st.queries_fallback == len(qpos)
        for j, q in enumerate(qpos):
            oi, os_ = oracle.batch_knn_cosine(q, dneg, 5)
            assert same_knn("cos", idx[j], sc[j], oi, os_), (engine, j, idx[j], oi, sc[j], os_)
        oi, os_ = oracle.batch_knn_cosine(qpos[0], dneg, 5)
        assert 17 in oi[:2].tolist() and 0.0 in os_[:2].tolist()  # the NaN-norm row scores 0.0 and leads the finite (negative) cosines


def test_knn_auto_builds_the_int8_copy_on_the_fourth_small_call(B, innr):
    """a caller that keeps sending one query (the reference's own signature in a loop) gets the int8 copy with its fourth AUTO call;
    answers unchanged"""
    vb = B.VerticalBatch.generate(150_000, 64, 3)
    qs = _queries(6, 64, uniform=True)
    st = innr.KnnStats()
    engines = []
    for j in range(6):
        i1, s1 = B.batch_knn_dot_multi(qs[j:j + 1], vb, 5, stats=st)
        engines.append(st.engine)
        e1, es1 = B.batch_knn_dot_multi(qs[j:j + 1], vb, 5, engine=innr.KNN_EXACT)
        assert np.array_equal(i1, e1) and bits_equal(s1, es1)
    assert engines == [innr.KNN_EXACT] * 3 + [innr.KNN_MFMA_I8] * 3, engines


def test_knn_auto_engine_selection(B, innr):
    """INNR_KNN_AUTO (api.hip, innr_batch_knn_dev): up to 3 queries the exact engine (one HBM-bound corpus pass, no extra memory);
    from 4 queries on the int8 filter for dot / cosine when its corpus copy fits -- and for EVERY batch size once it exists --,
    the bf16 filter for squared L2 from 9 queries on; a corpus of a few tiles per slice stays on the exact engine."""
    vb = B.VerticalBatch.generate(100_000, 64, 0)
    st = innr.KnnStats()
    B.batch_knn_dot_multi(_queries(3, 64, uniform=True), vb, 5, stats=st)
    assert st.engine == innr.KNN_EXACT
    i4, s4 = B.batch_knn_dot_multi(_queries(4, 64, uniform=True), vb, 5, stats=st)
    assert st.engine == innr.KNN_MFMA_I8
    e4, es4 = B.batch_knn_dot_multi(_queries(4, 64, uniform=True), vb, 5, engine=innr.KNN_EXACT)
    assert np.array_equal(i4, e4) and bits_equal(s4, es4)
    i1, s1 = B.batch_knn_dot_multi(_queries(1, 64, uniform=True), vb, 5, stats=st)
    assert st.engine == innr.KNN_MFMA_I8  # the copy exists now: one query too
    e1, es1 = B.batch_knn_dot_multi(_queries(1, 64, uniform=True), vb, 5, engine=innr.KNN_EXACT)
    assert np.array_equal(i1, e1) and bits_equal(s1, es1)
    B.batch_knn_dot_multi(_queries(64, 64, uniform=True), vb, 5, stats=st)
    assert st.engine == innr.KNN_MFMA_I8
    B.batch_knn_multi(_queries(3, 64, uniform=True), vb, 5, stats=st)
    assert st.engine == innr.KNN_EXACT  # squared L2, 3 queries, no copy for it yet: one exact pass
    B.batch_knn_multi(_queries(64, 64, uniform=True), vb, 5, stats=st)
    assert st.engine == innr.KNN_MFMA_I8 and st.queries_fallback <= 1  # (squared L2 has its own int8 copy)
    from innr_amd import _lib
    with _lib.default_context().option("no_auto_i8", 1):
        B.batch_knn_multi(_queries(64, 64, uniform=True), vb, 5, stats=st)
        assert st.engine == innr.KNN_MFMA_BF16 and st.queries_fallback <= 1
    with _lib.default_context().option("no_auto_bf16", 1):  # no low-precision filter at all: the f32 GEMM engine
        B.batch_knn_dot_multi(_queries(64, 64, uniform=True), vb, 5, stats=st)
        assert st.engine == innr.KNN_MFMA
    small = B.VerticalBatch.generate(3000, 64, 0)  # a few tiles per slice: the exact engine, all query groups in one launch
    B.batch_knn_dot_multi(_queries(64, 64, uniform=True), small, 5, stats=st)
    assert st.engine == innr.KNN_EXACT


# ------------------------------------------------------------------------------- larger sizes: engine agreement
def test_engines_agree_1m(B, innr):
    # 1M x 128 (the oracle would need minutes for 256 queries): the exact engine is pinned to the oracle above,
    # so agreement of the two engines carries parity to this size.
    vb = B.VerticalBatch.generate(1_000_000, 128, 0)
    qs = _queries(256, 128, 10_000_000, uniform=True)
    for fn in (B.batch_knn_dot_multi, B.batch_knn_cosine_multi, B.batch_knn_multi):
        st = innr.KnnStats()
        i1, s1 = fn(qs, vb, 10, engine=innr.KNN_MFMA, stats=st)
        i2, s2 = fn(qs[:64], vb, 10, engine=innr.KNN_EXACT)
        assert np.array_equal(i1[:64], i2) and bits_equal(s1[:64], s2)
        assert st.queries_fallback <= 2, st.queries_fallback
    # spot-check 3 queries of the same corpus against the CPU oracle
    data = oracle.from_rows(oracle.generate_uniform(1_000_000, 128, 0))
    i1, s1 = B.batch_knn_dot_multi(qs[:3], vb, 10, engine=innr.KNN_MFMA)
    for j in range(3):
        oi, os_ = oracle.batch_knn_dot(qs[j], data, 10)
        assert same_knn("dot", i1[j], s1[j], oi, os_)


def test_full_size_properties_c2(B, innr):
    # BASELINE.json configs[1]: 10M x 768 f32, 1024 queries, k = 10 -- through size-independent properties:
    #  (1) self-match: a query equal to corpus row r must rank r first under cosine with score ~ 1;
    #  (2) results sorted best-first, indices unique and in range;
    #  (3) the GEMM engine equals the bit-exact engine on a subset of the same queries.
    n, dim, nq, k = 10_000_000, 768, 1024, 10
    vb = B.VerticalBatch.generate(n, dim, 0)
    picks = (np.arange(nq, dtype=np.int64) * 9_765 + 123) % n
    qs = np.concatenate([oracle.generate_uniform(1, dim, 0, row0=int(r)) for r in picks])  # query j = corpus row picks[j]
    st = innr.KnnStats()
    idx, sc = B.batch_knn_cosine_multi(qs, vb, k, engine=innr.KNN_MFMA, stats=st)
    assert idx.shape == (nq, k)
    assert np.array_equal(idx[:, 0].astype(np.int64), picks)
    assert np.all(np.abs(sc[:, 0] - 1.0) < 1e-5)
    assert np.all(sc[:, :-1] >= sc[:, 1:]) and idx.max() < n
    assert all(len(set(r.tolist())) == k for r in idx)
    assert st.queries_fallback <= 4, st.queries_fallback
    i2, s2 = B.batch_knn_cosine_multi(qs[:16], vb, k, engine=innr.KNN_EXACT)
    assert np.array_equal(idx[:16], i2) and bits_equal(sc[:16], s2)
    idx, sc = B.batch_knn_dot_multi(qs, vb, k, engine=innr.KNN_MFMA, stats=st)
    i2, s2 = B.batch_knn_dot_multi(qs[:16], vb, k, engine=innr.KNN_EXACT)
    assert np.array_equal(idx[:16], i2) and bits_equal(sc[:16], s2)
    assert np.all(sc[:, :-1] >= sc[:, 1:]) and st.queries_fallback <= 4
    print(f"C2 dot: gemm {st.gemm_ms:.1f} ms, total {st.total_ms:.1f} ms, fallback {st.queries_fallback}")
    # batch_knn (squared L2) on the GEMM engine: row r is its own nearest neighbour at distance 0
    idx, sc = B.batch_knn_multi(qs, vb, k, engine=innr.KNN_MFMA, stats=st)
    assert np.array_equal(idx[:, 0].astype(np.int64), picks) and np.all(sc[:, 0] == 0.0)
    assert np.all(sc[:, :-1] <= sc[:, 1:]) and st.queries_fallback <= 4, st.queries_fallback
    i2, s2 = B.batch_knn_multi(qs[:16], vb, k, engine=innr.KNN_EXACT)
    assert np.array_equal(idx[:16], i2) and bits_equal(sc[:16], s2)
    print(f"C2 l2: gemm {st.gemm_ms:.1f} ms, total {st.total_ms:.1f} ms, fallback {st.queries_fallback}")


@pytest.mark.parametrize("cluster", [20, 200])
def test_knn_mfma_near_ties_around_the_cut(B, innr, cluster):
    """Adversarial for the margin proof: `cluster` corpus vectors are copies of the query direction perturbed in the
    last bits, so their exact scores differ by a few ulps and the MFMA (fma-chain) order among them is arbitrary.
    cluster < KP: all of them are candidates, exact re-scoring settles the order, the proof holds. cluster > KP: the
    cut runs through the cluster, nothing can be proven and the exact engine must take over. Either way the answer is
    the oracle's, bit for bit."""
    n, dim, k = 60_000, 64, 10
    rows = oracle.generate_uniform(n, dim, 17) * np.float32(0.25)
    qs = oracle.generate_uniform(24, dim, 18)
    rng = np.random.default_rng(3)
    for j in range(len(qs)):  # each query gets its own cluster, scattered over the corpus
        pos = rng.choice(n, size=cluster, replace=False)
        noise = (1.0 + rng.integers(-3, 4, size=(cluster, dim)) * 2.0 ** -23).astype(np.float32)
        rows[pos] = (qs[j] * np.float32(1.5)) * noise
    data = oracle.from_rows(rows)
    vb = B.VerticalBatch.from_rows(rows)
    for metric in ("dot", "cos", "l2"):
        fn = {"dot": B.batch_knn_dot_multi, "cos": B.batch_knn_cosine_multi, "l2": B.batch_knn_multi}[metric]
        ofn = {"dot": oracle.batch_knn_dot, "cos": oracle.batch_knn_cosine, "l2": oracle.batch_knn}[metric]
        st = innr.KnnStats()
        idx, sc = fn(qs, vb, k, engine=innr.KNN_MFMA, stats=st)
        for j, q in enumerate(qs):
            oi, os_ = ofn(q, data, k)
            if metric != "l2":
                assert same_knn(metric, idx[j], sc[j], oi, os_), (metric, cluster, j)
                continue
            # batch_knn keeps k of the vectors tied at the k-th distance; WHICH of them is a property of core's
            # binary_search inside TopK::insert (topk.rs:177-185), so at the cut only the distances are comparable:
            # identical score lists, identical indices above the last distance value, and every index reported at the
            # last value really lies at that distance
            assert bits_equal(sc[j], os_), (cluster, j)
            last = os_[-1]
            inner = os_ != last
            assert sorted(idx[j][inner].tolist()) == sorted(oi[inner].tolist()), (cluster, j)
            full = oracle.batch_l2_squared(q, data)
            assert all(full[int(i)] == last for i in idx[j][~inner]) and len(set(idx[j].tolist())) == k
        if cluster > 32 and metric != "l2":
            assert st.queries_fallback >= len(qs) // 2, (metric, st.queries_fallback)  # the cut is inside the cluster
        if cluster < 32 and metric == "dot":
            assert st.queries_fallback <= 2, st.queries_fallback


def test_knn_mfma_minority_of_unproven_queries_retries_with_longer_lists(B, innr):
    """20 of 96 queries get a cluster of 100 near-identical top vectors: more than the 32 candidates the first pass keeps
    (proof fails), fewer than 256 -- the engine re-runs those 20 as one batch with 256-entry lists instead of an exact
    corpus scan per group. Results identical to the oracle either way; the other 76 queries stay proven."""
    n, dim, k = 80_000, 64, 10
    rows = oracle.generate_uniform(n, dim, 41) * np.float32(0.25)
    qs = oracle.generate_uniform(96, dim, 42)
    rng = np.random.default_rng(9)
    for j in range(0, 96, 5):
        pos = rng.choice(n, size=100, replace=False)
        noise = (1.0 + rng.integers(-3, 4, size=(100, dim)) * 2.0 ** -23).astype(np.float32)
        # scaled so that the cluster tops ITS query's ranking (0.2 |q|^2 ~ 4.3 against ~2.5 for the best ordinary row)
        # and stays out of every other query's top-k (0.2 q.q' < 1.7): only the 20 cluster queries are near-tied
        rows[pos] = (qs[j] * np.float32(0.2)) * noise
    data = oracle.from_rows(rows)
    vb = B.VerticalBatch.from_rows(rows)
    for metric in ("dot", "cos"):
        fn = {"dot": B.batch_knn_dot_multi, "cos": B.batch_knn_cosine_multi}[metric]
        ofn = {"dot": oracle.batch_knn_dot, "cos": oracle.batch_knn_cosine}[metric]
        st = innr.KnnStats()
        idx, sc = fn(qs, vb, k, engine=innr.KNN_MFMA, stats=st)
        for j, q in enumerate(qs):
            oi, os_ = ofn(q, data, k)
            assert same_knn(metric, idx[j], sc[j], oi, os_), (metric, j)
        assert 16 <= st.queries_fallback <= 24, st.queries_fallback


def test_config5_cosine_4096_queries_two_logical_shards(B, innr):
    """BASELINE.json configs[4] per-GPU workload: batch_knn_cosine f32, a 10M x 768 shard, a 4096-query batch, k = 10 --
    and the exchange contract on G = 2 logical shards of 10M rows each (index bases 0 and 10M): per-shard top-k through
    the sharded path's device-resident local search (innr_batch_knn_dev), merge by innr_merge_topk_dev.
    Size-independent checks: self-match (query j IS corpus row picks[j], somewhere in the 20M rows -> global top-1 with
    cosine ~ 1), sortedness, uniqueness, range; the GEMM engine against the bit-exact engine on a query subset of each
    shard; the device merge against a numpy (score desc, index asc) merge of the two shard results for all 4096 queries."""
    import torch
    from innr_amd.dist import _gpu_local_search, _gpu_merge
    n, dim, nq, k, g = 10_000_000, 768, 4096, 10, 2
    ctx = innr.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    dev = torch.device("cuda", 0)
    picks = (np.arange(nq, dtype=np.int64) * 4_877 + 321) % (g * n)
    qs = np.concatenate([oracle.generate_uniform(1, dim, 0, row0=int(r)) for r in picks])
    q_dev = torch.from_numpy(qs).to(dev)
    all_i = torch.empty((g, nq, k), dtype=torch.int64, device=dev)
    all_s = torch.empty((g, nq, k), dtype=torch.float32, device=dev)
    for r in range(g):
        vb = B.VerticalBatch.generate(n, dim, seed=0, row0=r * n, ctx=ctx)
        vb.set_index_base(r * n)
        st = innr.KnnStats()
        idx, sc = _gpu_local_search(vb, innr.METRIC_COSINE, innr.KNN_MFMA)(q_dev, k, st)
        torch.cuda.synchronize()
        assert st.engine == innr.KNN_MFMA and st.queries_fallback <= 8, st.queries_fallback
        print(f"C5 shard {r}: cosine 4096 q x 10M x 768: gemm {st.gemm_ms:.1f} ms, total {st.total_ms:.1f} ms, "
              f"fallback {st.queries_fallback}")
        all_i[r], all_s[r] = idx, sc
        hi, hs = idx.cpu().numpy(), sc.cpu().numpy()
        assert hi.min() >= r * n and hi.max() < (r + 1) * n
        assert np.all(hs[:, :-1] >= hs[:, 1:]) and all(len(set(row.tolist())) == k for row in hi)
        mine = (picks // n) == r  # queries that are rows of THIS shard: self-match first, cosine ~ 1
        assert np.array_equal(hi[mine, 0], picks[mine]) and np.all(np.abs(hs[mine, 0] - 1.0) < 1e-5)
        sub = np.arange(0, nq, 293)[:14]  # 14 queries spread over the batch (two 8-query passes of the exact engine)
        ei, es = _gpu_local_search(vb, innr.METRIC_COSINE, innr.KNN_EXACT)(q_dev[torch.from_numpy(sub).to(dev)].contiguous(), k)
        assert np.array_equal(hi[sub], ei.cpu().numpy()) and bits_equal(hs[sub], es.cpu().numpy())
        # the bf16 filter engine (what INNR_KNN_AUTO picks at this shape): normalised bf16 copies, same answers for all 4096
        st2 = innr.KnnStats()
        bi, bs = _gpu_local_search(vb, innr.METRIC_COSINE, innr.KNN_MFMA_BF16)(q_dev, k, st2)
        torch.cuda.synchronize()
        assert st2.engine == innr.KNN_MFMA_BF16 and torch.equal(bi, idx) and torch.equal(bs.view(torch.int32), sc.view(torch.int32))
        print(f"C5 shard {r} on the bf16 filter: gemm {st2.gemm_ms:.1f} ms, total {st2.total_ms:.1f} ms, fallback {st2.queries_fallback}")
        ii, i_s = _gpu_local_search(vb, innr.METRIC_COSINE, innr.KNN_MFMA_I8)(q_dev, k, st2)  # the int8 filter (scalar-quantised copy)
        torch.cuda.synchronize()
        assert st2.engine == innr.KNN_MFMA_I8 and torch.equal(ii, idx) and torch.equal(i_s.view(torch.int32), sc.view(torch.int32))
        print(f"C5 shard {r} on the int8 filter: gemm {st2.gemm_ms:.1f} ms, total {st2.total_ms:.1f} ms, fallback {st2.queries_fallback}")
        vb.close()
    out_i, out_s = _gpu_merge(ctx, innr.METRIC_COSINE)(all_i, all_s, k)
    torch.cuda.synchronize()
    out_i, out_s = out_i.cpu().numpy(), out_s.cpu().numpy()
    ci = all_i.cpu().numpy().transpose(1, 0, 2).reshape(nq, g * k)
    cs = all_s.cpu().numpy().transpose(1, 0, 2).reshape(nq, g * k)
    order = np.lexsort((ci, -cs.astype(np.float64)), axis=1)[:, :k]  # score descending, then global index ascending
    assert np.array_equal(out_i, np.take_along_axis(ci, order, 1)) and bits_equal(out_s, np.take_along_axis(cs, order, 1))
    assert np.array_equal(out_i[:, 0], picks) and np.all(np.abs(out_s[:, 0] - 1.0) < 1e-5)
    ctx.close()


@pytest.mark.parametrize("waves", ["1", "2", "4", "8"])
def test_knn_mfma_both_block_shapes_every_metric(B, innr, waves):
    # plan_gemm picks 8-wave (512-query) tiles for dot with many queries, 4-wave tiles for cosine / L2, 1- and 2-wave
    # (64- / 128-query) tiles for small batches; the context option gemm_waves forces any of them, so every (kind, block shape)
    # instantiation stays under test -- here with 600 queries, i.e. ten 64-query tiles down to two 512-query ones
    from innr_amd import _lib
    rows, data = _corpus(70_000, 64, 31, uniform=True)
    vb = None
    with _lib.default_context().option("gemm_waves", int(waves)):
        for metric in ("dot", "cos", "l2"):
            vb = _check_knn(B, innr, metric, vb if vb is not None else rows, data, _queries(600, 64, 777, uniform=True), 10,
                            innr.KNN_MFMA)


@pytest.mark.parametrize("metric", ["dot", "cos", "l2"])
def test_completion_pass_every_path(B, innr, metric, ctx_option):
    """Queries whose margin proof fails (here: every query -- the corpus holds each vector 40 times) are settled by ONE completion pass --
    the GEMM kernel in collect mode with fixed thresholds, exact re-score of everything collected, radix select -- with the
    answer the oracle gives; from the row-major copy, by column gathers (what a full HBM falls back to), and with the
    completion pass switched off (exact engine, 8 queries per corpus pass)."""
    base, _ = _corpus(1500, 96, 5, uniform=True)
    rows = np.repeat(base, 40, axis=0)  # every vector 40 times: the cut at k = 10 lies inside a group of exactly equal scores
    rows[::7] *= np.float32(1.0 + 2.0 ** -20)  # ... and near-equal ones around it
    data = oracle.from_rows(rows)
    qs = _queries(40, 96, 9, uniform=True)
    st = innr.KnnStats()
    fn = {"dot": B.batch_knn_dot_multi, "cos": B.batch_knn_cosine_multi, "l2": B.batch_knn_multi}[metric]
    vb = _check_knn(B, innr, metric, rows, data, qs, 10, innr.KNN_MFMA)
    fn(qs, vb, 10, engine=innr.KNN_MFMA, stats=st)
    assert st.queries_fallback > 8  # the completion pass ran (up to 8 unproven queries take the exact engine directly)
    ctx_option("no_rows_copy", 1)
    _check_knn(B, innr, metric, vb, data, qs, 10, innr.KNN_MFMA)
    ctx_option("no_completion", 1)
    _check_knn(B, innr, metric, vb, data, qs, 10, innr.KNN_MFMA)
    if metric != "l2":  # (which members of a tie group that straddles the cut survive is core::slice::binary_search's choice
        _check_knn(B, innr, metric, vb, data, qs, 100, innr.KNN_MFMA)  # in TopK: DESIGN.md 2(a); k = 10 keeps the first ten either way)


def test_k100_on_the_int8_filter_and_its_completion_pass(B, innr):
    """k beyond the direct lists (4k + 64 > 256): lists of k + 16, most proofs fail by design, the int8 filter's own completion
    pass (collect mode) settles them -- the reference benches k = 100 (benches/batch.rs:127,146)."""
    rows, data = _corpus(150_000, 64, 3, uniform=True)
    qs = _queries(200, 64, 11, uniform=True)
    vb = None
    for metric in ("dot", "cos"):
        for k in (49, 100, 240):
            vb = _check_knn(B, innr, metric, vb if vb is not None else rows, data, qs, k, innr.KNN_MFMA_I8)
    st = innr.KnnStats()
    B.batch_knn_dot_multi(qs, vb, 100, engine=innr.KNN_AUTO, stats=st)
    assert st.engine == innr.KNN_MFMA_I8
