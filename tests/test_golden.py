"""Golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py): the oracle reproduces them bit for bit
(CPU), and so does the HIP path through the C ABI (GPU). See tests/golden_cases.py for the inputs."""
from __future__ import annotations

import importlib.util
import os

import numpy as np
import pytest

import golden_cases as G
import oracle


def _load(name):
    return dict(np.load(G.path(name)))


def _maker():
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(G.GOLDEN_DIR, "make_golden.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("name", list(G.CASES))
def test_oracle_reproduces_golden(name):
    want, got = _load(name), _maker().expected(name)
    assert sorted(want) == sorted(got)
    for key in want:
        assert np.array_equal(want[key], got[key]), (name, key)


def test_golden_knn_independent_numpy_check():
    """the stored top-k of one case re-derived without the oracle: sequential f32 accumulation in numpy + stable sort"""
    p = G.CASES["uniform_777x33_k50"]
    rows, qs, want = G.corpus_rows(oracle, p), G.query_rows(oracle, p), _load("uniform_777x33_k50")
    for j, q in enumerate(qs):
        acc = np.zeros(p["n"], np.float32)
        for d in range(p["dim"]):
            acc = (acc + (q[d] * rows[:, d]).astype(np.float32)).astype(np.float32)
        order = np.argsort(-acc.astype(np.float64), kind="stable")[: p["k"]]
        assert order.tolist() == want["dot_idx"][j].tolist()
        assert np.array_equal(acc[order].view(np.uint32), want["dot_bits"][j])


# ---------------------------------------------------------------- GPU: the product path against the same files
gpu = pytest.mark.gpu


def _same_l2(idx, sc_bits, widx, wbits):
    """order among exactly equal L2 distances is implementation-defined in the reference (topk.rs:177-185)"""
    if not np.array_equal(np.sort(sc_bits), np.sort(wbits)):
        return False
    for b in np.unique(wbits):
        if set(idx[sc_bits == b].tolist()) != set(widx[wbits == b].tolist()):
            return False
    return True


@gpu
@pytest.mark.parametrize("name", [n for n, p in G.CASES.items() if p["kind"] == "knn"])
@pytest.mark.parametrize("engine", ["exact", "mfma"])
def test_hip_knn_matches_golden(name, engine):
    import innr_amd
    from innr_amd import batch as B
    p, want = G.CASES[name], _load(name)
    vb = B.VerticalBatch.from_rows(G.corpus_rows(oracle, p))
    qs = G.query_rows(oracle, p)
    eng = innr_amd.KNN_EXACT if engine == "exact" else innr_amd.KNN_MFMA
    idx, sc = B.batch_knn_dot_multi(qs, vb, p["k"], engine=eng)
    assert np.array_equal(idx, want["dot_idx"]) and np.array_equal(G.bits(sc), want["dot_bits"])
    idx, sc = B.batch_knn_cosine_multi(qs, vb, p["k"], engine=eng)
    assert np.array_equal(idx, want["cos_idx"]) and np.array_equal(G.bits(sc), want["cos_bits"])
    idx, sc = B.batch_knn_multi(qs, vb, p["k"], engine=innr_amd.KNN_EXACT)
    for j in range(p["nq"]):
        assert _same_l2(idx[j], G.bits(sc[j]), want["l2_idx"][j], want["l2_bits"][j])


@gpu
def test_hip_scores_match_golden():
    from innr_amd import batch as B
    p, want = G.CASES["scores_1000x33"], _load("scores_1000x33")
    vb = B.VerticalBatch.from_rows(G.corpus_rows(oracle, p))
    norms = B.batch_norms(vb)
    assert np.array_equal(G.bits(norms), want["norms_bits"])
    assert np.array_equal(G.bits(B.batch_dimension_variance(vb)), want["var_bits"])
    for j, q in enumerate(G.query_rows(oracle, p)):
        assert np.array_equal(G.bits(B.batch_dot(q, vb)), want["dot_bits"][j])
        assert np.array_equal(G.bits(B.batch_l2_squared(q, vb)), want["l2_bits"][j])
        assert np.array_equal(G.bits(B.batch_cosine(q, vb, norms)), want["cos_bits"][j])


@gpu
def test_hip_l2family_matches_golden():
    from innr_amd import batch as B
    p, want = G.CASES["l2family_3000x40"], _load("l2family_3000x40")
    vb = B.VerticalBatch.from_rows(G.corpus_rows(oracle, p))
    for j, q in enumerate(G.query_rows(oracle, p)):
        r = B.batch_knn_filtered(q, vb, p["k"], lambda i: i % p["mod"] == 0)
        assert _same_l2(np.asarray(r.indices, np.uint64), G.bits(r.scores), want[f"filt_idx{j}"], want[f"filt_bits{j}"])
        r = B.batch_knn_reordered(q, vb, p["k"])
        assert _same_l2(np.asarray(r.indices, np.uint64), G.bits(r.scores), want[f"reord_idx{j}"], want[f"reord_bits{j}"])
        pairs = B.batch_l2_squared_pruning(q, vb, p["thr"])
        assert [a for a, _ in pairs] == want[f"prune_idx{j}"].tolist()
        assert np.array_equal(G.bits([b for _, b in pairs]), want[f"prune_bits{j}"])


@gpu
@pytest.mark.parametrize("engine", ["exact", "mfma"])
def test_hip_u8_matches_golden(engine):
    import innr_amd
    from innr_amd import scalar as S
    p, want = G.CASES["u8_4000x64"], _load("u8_4000x64")
    params = S.QuantizationParams.from_range(p["mn"], p["mx"])
    qc = S.QuantizedCorpus.generate(p["n"], p["dim"], params, seed=0)  # device quantize_u8 of the same stream
    codes = qc.codes().astype(np.uint64)
    crc = [int(codes.sum()), int((codes * (np.arange(codes.size, dtype=np.uint64).reshape(codes.shape) % 251)).sum())]
    assert crc == want["codes_crc"].tolist()
    qs = oracle.generate_uniform(p["nq"], p["dim"], p["qseed"])
    idx, sc = qc.knn_multi(qs, p["k"], engine=innr_amd.KNN_EXACT if engine == "exact" else innr_amd.KNN_MFMA)
    assert np.array_equal(idx, want["idx"]) and np.array_equal(G.bits(sc), want["bits"])


@gpu
def test_hip_maxsim_matches_golden():
    from innr_amd import maxsim as M
    p, want = G.CASES["maxsim_300x16x48"], _load("maxsim_300x16x48")
    tok, q = G.maxsim_inputs(oracle, p)
    dc = M.DocumentCorpus.from_tokens(tok)
    assert np.array_equal(G.bits(dc.scores(q)), want["dot_bits"])
    assert np.array_equal(G.bits(dc.scores(q, cosine=True)), want["cos_bits"])
    assert G.bits([M.maxsim(q, tok[7])])[0] == want["dot_bits"][7]  # host pair function, same value
