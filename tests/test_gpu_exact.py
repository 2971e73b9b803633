"""GPU parity tests, part 1: layout, generator, bit-exact scans and the EXACT kNN engine, all through the
C ABI (ctypes -> libinnr_hip.so), compared with the CPU oracle on the same seeded inputs.
Bar: bit-exact scores, identical index lists."""
from __future__ import annotations

import math

import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle
import kat_cases as K
from backends import HipBackend


@pytest.fixture(scope="module")
def B():
    from innr_amd import batch
    return batch


@pytest.fixture(scope="module")
def innr():
    import innr_amd
    return innr_amd


def _corpus(n, dim, seed0=0, uniform=False):
    """LCG rows (the reference example's generator, examples/batch_demo.rs:167) or i.i.d. uniform[-1,1) rows
    (the distribution of the reference's criterion benches, benches/batch.rs:11-21)."""
    rows = oracle.generate_uniform(n, dim, seed0) if uniform else oracle.generate_corpus(n, dim, seed0)
    return rows, oracle.from_rows(rows)


def _queries(nq, dim, seed0=50_000, uniform=False):
    if uniform:
        return oracle.generate_uniform(nq, dim, seed0 + 0x5EED)
    return np.stack([oracle.generate_embedding(dim, seed0 + j) for j in range(nq)])


# ------------------------------------------------------------------------------- reference KATs on the GPU
@pytest.mark.parametrize("kat", K.BATCH_KATS, ids=lambda f: f.__name__)
def test_reference_kat_exact_engine(kat, innr):
    kat(HipBackend(engine=innr.KNN_EXACT))


def test_native_library_is_loaded():
    from innr_amd import _lib
    _lib.load()
    maps = open("/proc/self/maps").read()
    assert "libinnr_hip.so" in maps


# ------------------------------------------------------------------------------- layout / ingest
@pytest.mark.parametrize("n,dim", [(1, 1), (2, 3), (255, 7), (256, 32), (257, 33), (1000, 128), (4099, 100)])
def test_layout_roundtrip_all_constructors(B, n, dim):
    rows, data = _corpus(n, dim, 3)
    for vb in (B.VerticalBatch.from_rows(rows), B.VerticalBatch.from_flat(rows.reshape(-1), n, dim),
               B.VerticalBatch.from_data(data.reshape(-1), n, dim)):
        assert vb.num_vectors() == n and vb.dimension() == dim
        assert np.array_equal(vb.data(), data)
        assert np.array_equal(vb.extract_vector(n - 1), rows[n - 1])
        assert vb.get(dim - 1, 0) == rows[0, dim - 1]


def test_layout_errors_and_empty(B, innr):
    with pytest.raises(innr.InnrPanic):
        B.VerticalBatch.from_rows([[1.0, 2.0], [1.0]])  # batch.rs:120 "Inconsistent vector dimension"
    with pytest.raises(innr.InnrPanic):
        B.VerticalBatch.from_flat([1.0, 2.0, 3.0], 2, 2)  # batch.rs:168
    e = B.VerticalBatch.from_rows([])
    assert e.num_vectors() == 0 and e.dimension() == 0 and e.data().shape == (0, 0)
    assert len(B.batch_dot([], e)) == 0 and len(B.batch_norms(e)) == 0
    vb = B.VerticalBatch.from_rows([[1.0, 2.0]])
    for fn in (B.batch_dot, B.batch_l2_squared, lambda q, b: B.batch_knn(q, b, 1), lambda q, b: B.batch_knn_dot(q, b, 1),
               lambda q, b: B.batch_knn_cosine(q, b, 1)):
        with pytest.raises(innr.InnrPanic):
            fn([1.0, 2.0, 3.0], vb)  # assert_eq!(query.len(), batch.dimension)
    with pytest.raises(innr.InnrPanic):
        B.batch_cosine([1.0, 2.0], vb, [1.0, 2.0])  # norms.len() != num_vectors, batch.rs:711


@pytest.mark.parametrize("n,dim,seed0", [(1000, 128, 0), (257, 33, 1 << 40), (5000, 768, 12345)])
def test_device_generator_bit_exact(B, innr, n, dim, seed0):
    vb = B.VerticalBatch.generate(n, dim, seed0, generator=innr.GEN_EXAMPLE_LCG)
    assert np.array_equal(vb.data(), oracle.from_rows(oracle.generate_corpus(n, dim, seed0)))
    vb = B.VerticalBatch.generate(n, dim, seed0, generator=innr.GEN_EXAMPLE_LCG, row0=17)
    assert np.array_equal(vb.data(), oracle.from_rows(oracle.generate_corpus(n, dim, seed0 + 17)))
    vb = B.VerticalBatch.generate(n, dim, seed0)  # default: uniform stream
    assert np.array_equal(vb.data(), oracle.from_rows(oracle.generate_uniform(n, dim, seed0)))
    vb = B.VerticalBatch.generate(n, dim, seed0, row0=12_345_678_901)
    assert np.array_equal(vb.data(), oracle.from_rows(oracle.generate_uniform(n, dim, seed0, row0=12_345_678_901)))


# ------------------------------------------------------------------------------- bit-exact scans
@pytest.mark.parametrize("n,dim", [(1, 1), (3, 2), (255, 16), (257, 31), (1000, 128), (10_000, 128), (4100, 768)])
def test_scans_bit_exact(B, n, dim):
    rows, data = _corpus(n, dim, 7)
    vb = B.VerticalBatch.from_rows(rows)
    for q in _queries(3, dim, 999):
        assert np.array_equal(B.batch_dot(q, vb), oracle.batch_dot(q, data))
        assert np.array_equal(B.batch_l2_squared(q, vb), oracle.batch_l2_squared(q, data))
        norms = B.batch_norms(vb)
        assert np.array_equal(norms, oracle.batch_norms(data))
        assert np.array_equal(B.batch_cosine(q, vb, norms), oracle.batch_cosine(q, data, norms))
    out = [123.0]
    B.batch_dot_into(rows[0], vb, out)
    assert len(out) == n and np.array_equal(np.float32(out), oracle.batch_dot(rows[0], data))


def test_scans_special_values(B):
    rows = np.array([[1.0, 0.0, -2.0], [0.0, 0.0, 0.0], [np.inf, 1.0, 1.0], [np.nan, 1.0, 1.0], [1e-30, 1e-30, 0.0],
                     [-0.0, -0.0, -0.0], [3e38, 3e38, 3e38]], dtype=np.float32)
    data = oracle.from_rows(rows)
    vb = B.VerticalBatch.from_rows(rows)
    for q in ([1.0, 2.0, 3.0], [0.0, 0.0, 0.0], [1e-10, 0.0, 0.0], [-1.0, np.inf, 0.5]):
        q = np.float32(q)
        for got, exp in ((B.batch_dot(q, vb), oracle.batch_dot(q, data)),
                         (B.batch_l2_squared(q, vb), oracle.batch_l2_squared(q, data)),
                         (B.batch_cosine(q, vb, B.batch_norms(vb)), oracle.batch_cosine(q, data, oracle.batch_norms(data)))):
            # A NaN *generated* by an invalid operation (inf*0, inf/inf) has an ISA-defined sign (x86: negative,
            # CDNA/ARM: positive), so the reference itself differs across hosts there: compare NaN-ness for
            # those, bits (incl. -0.0) for everything else.
            gn, en = np.isnan(got), np.isnan(exp)
            assert np.array_equal(gn, en), (q, got, exp)
            assert bits_equal(got[~gn], exp[~en]), (q, got, exp)
    # caller-supplied norms are honoured (reference signature takes them, batch.rs:690)
    fake = np.float32([1, 2, 3, 4, 5, 6, 7])
    q = np.float32([1, 2, 3])
    got, exp = B.batch_cosine(q, vb, fake), oracle.batch_cosine(q, data, fake)
    assert np.array_equal(np.isnan(got), np.isnan(exp)) and bits_equal(got[~np.isnan(got)], exp[~np.isnan(exp)])


# ------------------------------------------------------------------------------- exact kNN engine
def bits_equal(a, b) -> bool:
    """Bitwise equality of two f32 arrays (so -0.0 != +0.0 and NaN == NaN of the same sign/payload)."""
    a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def same_knn(metric, idx, sc, oi, os_) -> bool:
    """Scores bit-identical; index lists identical. For L2 (`batch_knn`, TopK path) the reference leaves the
    order AMONG EXACTLY EQUAL distances to core's binary_search (topk.rs:177-185, SURVEY.md hard part 2), so
    equal-distance groups are compared as sets; everything else is compared position by position."""
    if not bits_equal(sc, os_):
        return False
    idx = [int(i) for i in idx]; oi = [int(i) for i in oi]
    if metric != "l2":
        return idx == oi
    keys = np.asarray(os_, np.float32).view(np.uint32)
    for kbits in set(keys.tolist()):
        sel = [p for p in range(len(oi)) if keys[p] == kbits]
        if sorted(idx[p] for p in sel) != sorted(oi[p] for p in sel):
            return False
    return True


def _check_knn(B, innr, metric, rows, data, queries, k, engine):
    fn = {"dot": B.batch_knn_dot_multi, "cos": B.batch_knn_cosine_multi, "l2": B.batch_knn_multi}[metric]
    ofn = {"dot": oracle.batch_knn_dot, "cos": oracle.batch_knn_cosine, "l2": oracle.batch_knn}[metric]
    vb = rows if hasattr(rows, "num_vectors") else B.VerticalBatch.from_rows(rows)
    idx, sc = fn(queries, vb, k, engine=engine)
    assert idx.shape == (len(queries), min(k, data.shape[1]))
    for j, q in enumerate(queries):
        oi, os_ = ofn(q, data, k)
        assert same_knn(metric, idx[j], sc[j], oi, os_), (metric, j, idx[j], oi, sc[j], os_)
    return vb


@pytest.mark.parametrize("metric", ["dot", "cos", "l2"])
def test_knn_exact_c1_shape(B, innr, metric):
    # BASELINE.json configs[0]: 10K x 128, 100 queries, k = 10 (examples/batch_demo.rs:159-170)
    rows, data = _corpus(10_000, 128, 0)
    _check_knn(B, innr, metric, rows, data, _queries(100, 128), 10, innr.KNN_EXACT)


@pytest.mark.parametrize("n,dim,nq,k", [(1, 4, 1, 1), (5, 3, 2, 10), (255, 16, 3, 7), (257, 33, 5, 32), (1000, 64, 9, 33),
                                        (3000, 20, 17, 100), (2049, 8, 4, 240)])
def test_knn_exact_ragged(B, innr, n, dim, nq, k):
    rows, data = _corpus(n, dim, 77)
    vb = None
    for metric in ("dot", "cos", "l2"):
        vb = _check_knn(B, innr, metric, vb if vb is not None else rows, data, _queries(nq, dim, 4242), k, innr.KNN_EXACT)


def test_knn_exact_single_query_api_matches_reference_shape(B, innr):
    rows, data = _corpus(2000, 48, 5)
    vb = B.VerticalBatch.from_rows(rows)
    q = _queries(1, 48)[0]
    for fn, ofn in ((B.batch_knn_dot, oracle.batch_knn_dot), (B.batch_knn_cosine, oracle.batch_knn_cosine), (B.batch_knn, oracle.batch_knn)):
        r = fn(q, vb, 10)
        oi, os_ = ofn(q, data, 10)
        assert isinstance(r, B.BatchKnnResult) and r.indices == oi.tolist() and np.array_equal(np.float32(r.scores), os_)
    assert B.batch_knn_dot(q, vb, 0).indices == [] and B.batch_knn(q, vb, 0).scores == []


@pytest.mark.parametrize("metric", ["dot", "cos", "l2"])
def test_knn_exact_uniform_data(B, innr, metric):
    rows, data = _corpus(20_000, 96, 3, uniform=True)
    _check_knn(B, innr, metric, rows, data, _queries(24, 96, 8, uniform=True), 20, innr.KNN_EXACT)


def test_no_k_limit_anywhere(B, innr):
    # every kNN entry point takes any k <= N, like the reference (batch.rs:395, 621-659, 820-882): beyond INNR_MAX_K the
    # candidate-list engines hand over to "all scores + device sort"; INNR_E_UNSUPPORTED is not reachable through k
    n, dim = 3000, 24
    rows, data = _corpus(n, dim, 4, uniform=True)
    vb = B.VerticalBatch.from_rows(rows)
    q = _queries(1, dim, 12, uniform=True)[0]
    assert len(B.batch_knn_dot(np.zeros(dim, np.float32), vb, 241).indices) == 241
    for k in (241, 1000, n, n + 5):
        r = B.batch_knn_reordered(q, vb, k)
        oi, os_ = oracle.batch_knn_reordered(q, data, k)
        assert r.indices == oi.tolist() and bits_equal(np.float32(r.scores), os_), k
        pred = lambda i: i % 3 != 1  # noqa: E731
        mask = np.array([1 if pred(i) else 0 for i in range(n)], dtype=np.uint8)
        r = B.batch_knn_filtered(q, vb, k, pred)
        oi, os_ = oracle.batch_knn_filtered(q, data, k, mask)
        assert r.indices == oi.tolist() and bits_equal(np.float32(r.scores), os_), k
    # fewer vectors pass than k (> INNR_MAX_K): k = min(k, passing), batch.rs:849
    r = B.batch_knn_filtered(q, vb, 500, lambda i: i < 300)
    oi, os_ = oracle.batch_knn_filtered(q, data, 500, (np.arange(n) < 300).astype(np.uint8))
    assert len(r.indices) == 300 and r.indices == oi.tolist() and bits_equal(np.float32(r.scores), os_)


@pytest.mark.parametrize("nq", [2, 3, 5, 6, 7, 9, 11])
def test_knn_exact_ragged_query_tails(B, innr, nq):
    # 2-3 queries share ONE 4-query corpus pass, 5-7 one 8-query pass (zero rows as padding): results per query unchanged
    rows, data = _corpus(9000, 40, 23, uniform=True)
    vb = None
    for metric in ("dot", "cos", "l2"):
        vb = _check_knn(B, innr, metric, vb if vb is not None else rows, data, _queries(nq, 40, 99, uniform=True), 7, innr.KNN_EXACT)


def test_knn_ties_resolve_to_lower_index(B, innr):
    # duplicated vectors: the reference's stable sort keeps the lower index first (batch.rs:757)
    base = oracle.generate_corpus(50, 16, 9)
    rows = np.concatenate([base, base, base[::-1]])  # every vector appears three times
    data = oracle.from_rows(rows)
    for metric in ("dot", "cos"):
        _check_knn(B, innr, metric, rows, data, _queries(6, 16, 31), 12, innr.KNN_EXACT)
    # L2 ties: same SET and same (distance, index) order as the documented device rule
    vb = B.VerticalBatch.from_rows(rows)
    q = _queries(1, 16, 31)[0]
    r = B.batch_knn(q, vb, 12)
    oi, os_ = oracle.batch_knn(q, data, 12)
    assert sorted(r.indices) == sorted(oi.tolist()) and np.array_equal(np.float32(r.scores), os_)
    assert r.indices == [i for _, i in sorted(zip(r.scores, r.indices))]
    # all-equal scores: zero query -> every dot is 0.0 -> first k indices
    r = B.batch_knn_dot(np.zeros(16, np.float32), vb, 7)
    assert r.indices == list(range(7)) and r.scores == [0.0] * 7
    r = B.batch_knn_cosine(np.zeros(16, np.float32), vb, 7)  # zero-norm query: all cosines 0 (batch.rs:716)
    assert r.indices == list(range(7)) and r.scores == [0.0] * 7


def test_knn_nan_inf_ordering(B, innr):
    # A NaN *input* propagates with its sign on every ISA: +NaN sorts above +inf under total_cmp, so it comes
    # FIRST in a descending sort (batch.rs:757) and LAST in the ascending TopK (topk.rs:101).
    rows = oracle.generate_corpus(600, 8, 1)
    rows[17, 3] = np.nan
    rows[300, 0] = np.inf
    rows[301, 0] = -np.inf
    data = oracle.from_rows(rows)
    q = np.float32([1, 1, 1, 1, 1, 1, 1, 1])
    for metric in ("dot", "l2"):  # cosine would divide inf/inf: generated NaN, sign is ISA-defined (see above)
        _check_knn(B, innr, metric, rows, data, q.reshape(1, -1), 5, innr.KNN_EXACT)
    r = B.batch_knn_dot(q, B.VerticalBatch.from_rows(rows), 3)
    assert r.indices[:2] == [17, 300] and math.isnan(r.scores[0]) and r.scores[1] == math.inf
    rows[300, 0] = 5.0; rows[301, 0] = -5.0
    _check_knn(B, innr, "cos", rows, oracle.from_rows(rows), q.reshape(1, -1), 5, innr.KNN_EXACT)


def test_generated_nan_sign_and_rank_are_pinned(B, innr):
    """A NaN that an invalid operation GENERATES (inf * 0 under dot, inf / inf under cosine) has an ISA-defined sign. Measured on
    gfx950 (tools/nan_probe.py): 0xFFC00000, the default NaN with the sign bit SET -- the same pattern the reference's usual x86
    hosts produce -- which total_cmp ranks below -inf: LAST in batch_knn_dot / batch_knn_cosine's descending order. include/innr_hip.h states
    this as the ABI's guarantee; pinned here on every engine, together with equality to the oracle (an x86 host)."""
    rows = oracle.generate_uniform(4000, 16, 3)
    rows[1234, 5] = np.inf
    q = oracle.generate_uniform(1, 16, 8)[0]
    q[5] = 0.0  # inf * 0
    q2 = oracle.generate_uniform(1, 16, 9)[0]  # cosine: the row's norm is +inf and its dot product +-inf: inf / inf
    data = oracle.from_rows(rows)
    vb = B.VerticalBatch.from_rows(rows)
    assert B.batch_dot(q, vb)[1234:1235].view(np.uint32)[0] == 0xFFC00000
    assert B.batch_cosine(q2, vb, B.batch_norms(vb))[1234:1235].view(np.uint32)[0] == 0xFFC00000
    for engine in (innr.KNN_EXACT, innr.KNN_MFMA):
        idx, sc = B.batch_knn_dot_multi(q.reshape(1, -1), vb, 4000, engine=engine)
        assert idx[0][-1] == 1234 and sc[0][-1:].view(np.uint32)[0] == 0xFFC00000, (engine, idx[0][-3:], sc[0][-3:].view(np.uint32))
        idx, sc = B.batch_knn_cosine_multi(q2.reshape(1, -1), vb, 4000, engine=engine)
        assert idx[0][-1] == 1234 and sc[0][-1:].view(np.uint32)[0] == 0xFFC00000, (engine, idx[0][-3:], sc[0][-3:].view(np.uint32))
        _check_knn(B, innr, "dot", vb, data, q.reshape(1, -1), 6, engine)      # the top of the list: untouched by it
        _check_knn(B, innr, "cos", vb, data, q2.reshape(1, -1), 6, engine)
    oi, os_ = oracle.batch_knn_dot(q, data, 4000)  # the oracle on this (x86) host: the same bits at the same place
    assert int(oi[-1]) == 1234 and os_[-1:].view(np.uint32)[0] == 0xFFC00000


def test_knn_monotone_scores_force_many_compactions(B, innr):
    # scores increase with the index: every vector beats the running threshold, so candidate lists fill and
    # compact over and over (the worst case of the threshold filter).
    n = 200_000
    rows = np.zeros((n, 2), dtype=np.float32)
    rows[:, 0] = np.arange(n, dtype=np.float32) * np.float32(0.001)
    rows[:, 1] = 1.0
    data = oracle.from_rows(rows)
    q = np.float32([[1.0, 0.5], [-1.0, 0.25], [0.0, 1.0]])
    for metric, k in (("dot", 10), ("l2", 10), ("dot", 100), ("cos", 33)):
        _check_knn(B, innr, metric, rows, data, q, k, innr.KNN_EXACT)


def test_index_base_offsets_reported_indices(B, innr):
    rows, data = _corpus(700, 12, 2)
    vb = B.VerticalBatch.from_rows(rows)
    vb.set_index_base(1 << 33)
    q = _queries(1, 12)[0]
    r = B.batch_knn_dot(q, vb, 4)
    oi, _ = oracle.batch_knn_dot(q, data, 4)
    assert r.indices == [int(i) + (1 << 33) for i in oi]


def test_backend_introspection(B, innr):
    from innr_amd import backend as BK
    small, big = B.VerticalBatch.generate(3000, 16, 0), B.VerticalBatch.generate(70_000, 16, 0)
    assert BK.batch_backend(small, 100) == BK.Backend.HIP_EXACT and BK.batch_backend(big, 1) == BK.Backend.HIP_EXACT
    assert BK.batch_backend(big, 100) == BK.Backend.HIP_MFMA and str(BK.dense_backend(768)) == "portable"
    st = innr.KnnStats()
    B.batch_knn_dot_multi(oracle.generate_uniform(100, 16, 1), big, 3, stats=st)
    # a matrix-pipe engine: the int8 filter when its corpus copy fits (what AUTO prefers at every batch size from 4 queries on)
    assert st.engine in (innr.KNN_MFMA, innr.KNN_MFMA_I8, innr.KNN_MFMA_BF16) and "gfx950" in BK.version()


def test_save_load_roundtrip(B, tmp_path):
    rows = oracle.generate_uniform(1234, 37, 4)
    vb = B.VerticalBatch.from_rows(rows)
    path = str(tmp_path / "corpus.pdx")
    vb.save(path)
    assert os.path.getsize(path) == 24 + 4 * 1234 * 37
    vb2 = B.VerticalBatch.load(path)
    assert vb2.num_vectors() == 1234 and vb2.dimension() == 37 and np.array_equal(vb2.data(), vb.data())
    q = oracle.generate_uniform(1, 37, 9)[0]
    assert B.batch_knn_dot(q, vb2, 5) == B.batch_knn_dot(q, vb, 5)
    empty = B.VerticalBatch.from_rows([])
    empty.save(path)
    e2 = B.VerticalBatch.load(path)
    assert e2.num_vectors() == 0 and e2.dimension() == 0
    with open(path, "wb") as f:
        f.write(b"not a pdx file, but long enough....")
    import innr_amd
    with pytest.raises(innr_amd.InnrPanic):
        B.VerticalBatch.load(path)


# ------------------------------------------------------------------------------- k beyond the candidate lists
@pytest.mark.parametrize("n,dim,nq,k", [(1000, 16, 3, 241), (5000, 33, 2, 5000), (5000, 33, 2, 10**9), (20_000, 64, 2, 1500)])
def test_knn_large_k_full_sort(B, innr, n, dim, nq, k):
    # k > INNR_MAX_K (240): all scores + a full device sort, the reference's own algorithm (batch.rs:754-763); k = N is
    # the complete ranking, k > N clamps to N (batch.rs:752). Whatever engine the caller names, the result is exact.
    rows, data = _corpus(n, dim, 19)
    vb = None
    for metric in ("dot", "cos", "l2"):
        for engine in (innr.KNN_AUTO, innr.KNN_MFMA):
            vb = _check_knn(B, innr, metric, vb if vb is not None else rows, data, _queries(nq, dim, 4242), k, engine)


def test_knn_large_k_ties_keep_index_order(B, innr):
    # duplicates everywhere: the full ranking must list equal scores by ascending index (stable sort, batch.rs:757)
    base = oracle.generate_uniform(50, 24, 3)
    rows = np.concatenate([base] * 20)
    data = oracle.from_rows(rows)
    _check_knn(B, innr, "dot", rows, data, _queries(2, 24, 7, uniform=True), 1000, innr.KNN_AUTO)
    _check_knn(B, innr, "cos", rows, data, _queries(2, 24, 7, uniform=True), 600, innr.KNN_AUTO)


def test_knn_large_k_nan_inf_zero_ordering(B, innr):
    # the complete ranking (k = N > INNR_MAX_K) with +NaN, +-inf and +-0.0 scores in it: total_cmp order end to end
    rows = oracle.generate_corpus(600, 8, 1)
    rows[17, 3] = np.nan
    rows[300, 0] = np.inf
    rows[301, 0] = -np.inf
    rows[5] = 0.0
    rows[6] = -0.0
    data = oracle.from_rows(rows)
    q = np.float32([1, 1, 1, 1, 1, 1, 1, 1])
    for metric in ("dot", "l2"):
        _check_knn(B, innr, metric, rows, data, q.reshape(1, -1), 600, innr.KNN_AUTO)
    r = B.batch_knn_dot(q, B.VerticalBatch.from_rows(rows), 600)
    assert r.indices[:2] == [17, 300] and r.indices[-1] == 301 and len(set(r.indices)) == 600
