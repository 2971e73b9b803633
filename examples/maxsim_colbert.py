#!/usr/bin/env python3
"""The reference's late-interaction walkthrough (examples/maxsim_colbert.rs) on the MI355X library: one (query, document)
pair through the host function, then the example's batch step -- score one query against many documents and rank them
-- as ONE device call instead of the caller's loop + sort (examples/maxsim_colbert.rs:159-193), checked against that loop.

    python examples/maxsim_colbert.py [n_docs]          (needs a GPU)
"""
from __future__ import annotations

import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from innr_amd import KnnStats, distance
from innr_amd import maxsim as M


def naive_maxsim(query, doc) -> float:
    """sum over query tokens of the best dot against any document token, with the pairwise dot of the library"""
    total = np.float32(-0.0)
    for q in query:
        total = np.float32(total + np.float32(max(distance.dot(q, d) for d in doc)))
    return float(total)


def main(n_docs: int = 1000, n_doc_tokens: int = 64, n_query_tokens: int = 32, dim: int = 128) -> None:
    print("MaxSim (ColBERT-style late interaction) on the GPU\n")
    corpus = M.DocumentCorpus.generate(n_docs, n_doc_tokens, dim, seed=5000)  # unit-norm token embeddings, made on the device
    rng = np.random.default_rng(7)
    q = rng.normal(size=(n_query_tokens, dim)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)

    scores = corpus.scores(q)  # maxsim(query, doc_i) for every document, bit-identical to the per-pair function
    st = KnnStats()
    t0 = time.perf_counter()
    top, top_scores = corpus.topk(q, 5, stats=st)
    dt = time.perf_counter() - t0
    print(f"   {n_docs} documents x {n_doc_tokens} tokens x {dim} dims, {n_query_tokens}-token query")
    print(f"   top-5 in one call: {dt * 1e3:.2f} ms ({dt / n_docs * 1e6:.2f} us/doc), engine {'MFMA' if st.engine == 2 else 'exact'}")
    for rank, (i, s) in enumerate(zip(top, top_scores)):
        print(f"     #{rank + 1}: doc {int(i)} = {s:.4f}")
    # the caller's loop of the reference example, on the first few documents
    order = np.argsort(-scores.astype(np.float64), kind="stable")[:5]
    assert top.tolist() == order.tolist()
    print("   ranking equals a stable sort of the per-document scores")
    print()


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 1000)
