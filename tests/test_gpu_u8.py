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
                                        (3000, 20, 2, 240)])
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


@pytest.mark.parametrize("n,dim,nq,k,alpha,offset", [(300, 16, 20, 10, 2.0, -1.0), (10_000, 128, 100, 10, 2.0, -1.0),
                                                     (20_000, 96, 300, 100, 3.5, -0.25), (1030, 768, 17, 16, 2.0, -1.0)])
def test_batch_knn_u8_gemm_engine(S, innr, n, dim, nq, k, alpha, offset):
    codes = _codes(n, dim, 4, alpha, offset)
    qc = S.QuantizedCorpus.from_codes(codes, n, dim, S.QuantizationParams(alpha, offset))
    qs = oracle.generate_uniform(nq, dim, 78)
    st = innr.KnnStats()
    idx, sc = qc.knn_multi(qs, k, engine=innr.KNN_MFMA, stats=st)
    assert st.engine == innr.KNN_MFMA
    for j in range(nq):
        oi, os_ = _oracle_knn(qs[j], codes, alpha, offset, k)
        assert same_knn("dot", idx[j], sc[j], oi, os_), (j, idx[j], oi)
    # u8 codes collide often (coarse grid), so some margin proofs legitimately fail; most must hold
    assert st.queries_fallback <= max(2, nq // 4), st.queries_fallback


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
    st = innr.KnnStats()
    i1, s1 = qc.knn_multi(qs, 10, engine=innr.KNN_MFMA, stats=st)
    i2, s2 = qc.knn_multi(qs[:32], 10, engine=innr.KNN_EXACT)
    assert np.array_equal(i1[:32], i2) and bits_equal(s1[:32], s2)
    print(f"u8 2Mx128 256q: gemm {st.gemm_ms:.2f} ms total {st.total_ms:.2f} ms fallback {st.queries_fallback}")
