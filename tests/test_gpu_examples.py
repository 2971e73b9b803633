"""The example walkthroughs (examples/*.py, the counterparts of the reference's examples) run end to end."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "examples", name + ".py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_batch_demo_runs(capsys):
    m = _load("batch_demo")
    m.demo_layout()
    m.demo_knn()
    m.demo_batch_dot()
    m.demo_timing(n=2000, dim=32, num_queries=10, k=5)
    assert "match: indices and distances agree" in capsys.readouterr().out


def test_maxsim_colbert_runs(capsys):
    _load("maxsim_colbert").main(n_docs=300, n_doc_tokens=24, n_query_tokens=8, dim=64)
    assert "ranking equals a stable sort" in capsys.readouterr().out


def test_matryoshka_search_runs(capsys):
    _load("matryoshka_search").main(corpus_size=3000, full_dim=96, prefix_dim=32, coarse_k=50, final_k=5, num_queries=3)
    assert "re-ranked scores are the exact full-dimension cosines" in capsys.readouterr().out
