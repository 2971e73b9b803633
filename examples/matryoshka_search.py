#!/usr/bin/env python3
"""The reference's matryoshka walkthrough (examples/matryoshka_search.rs) on the MI355X library: exact brute force at
full dimension, then the two-stage search -- coarse top-100 on the 128-dimension prefix, exact re-rank at 768 -- with
recall of the two-stage result against the exact one. On the dimension-major device layout the prefix is a VIEW of the
first 128 rows of the corpus: the coarse stage reads a sixth of the bytes and nothing is copied.

    python examples/matryoshka_search.py [corpus_size]          (needs a GPU)
"""
from __future__ import annotations

import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from innr_amd import METRIC_COSINE
from innr_amd import batch as B


def recall(retrieved, ground_truth) -> float:
    """examples/matryoshka_search.rs:169-175"""
    return sum(1 for g in ground_truth if g in set(retrieved)) / len(ground_truth)


def main(corpus_size: int = 10_000, full_dim: int = 768, prefix_dim: int = 128, coarse_k: int = 100, final_k: int = 10,
         num_queries: int = 8) -> None:
    print("Matryoshka Progressive Search")
    print("=============================\n")
    print(f"Corpus: {corpus_size} vectors, {full_dim}d (prefix {prefix_dim}d)")
    print(f"Pipeline: coarse top-{coarse_k} at {prefix_dim}d -> fine top-{final_k} at {full_dim}d\n")
    # MRL-like synthetic data: the leading dimensions carry most of the energy (scale decays with the dimension index),
    # so a prefix preserves most of the ranking -- i.i.d. dimensions would not (the reference's example data has none
    # either and reports the recall it gets). Queries are noisy copies of corpus rows.
    rng = np.random.default_rng(0xDEAD)
    scale = (1.0 / np.sqrt(1.0 + np.arange(full_dim) / 16.0)).astype(np.float32)
    rows = rng.uniform(-1, 1, (corpus_size, full_dim)).astype(np.float32) * scale
    vb = B.VerticalBatch.from_flat(rows.reshape(-1), corpus_size, full_dim)
    queries = (rows[rng.integers(0, corpus_size, num_queries)] + 0.5 * scale * rng.uniform(-1, 1, (num_queries, full_dim))).astype(np.float32)
    coarse = vb.prefix(prefix_dim)

    B.batch_knn_cosine_multi(queries, vb, final_k)                     # warm both paths (norm caches, workspaces)
    B.matryoshka_knn(queries, vb, prefix_dim, coarse_k, final_k, METRIC_COSINE, coarse=coarse)
    t0 = time.perf_counter()
    exact_idx, exact_sc = B.batch_knn_cosine_multi(queries, vb, final_k)
    t1 = time.perf_counter()
    coarse_idx, _ = B.batch_knn_cosine_multi(np.ascontiguousarray(queries[:, :prefix_dim]), coarse, coarse_k)
    t2 = time.perf_counter()
    fine_idx, fine_sc = B.matryoshka_knn(queries, vb, prefix_dim, coarse_k, final_k, METRIC_COSINE, coarse=coarse)
    t3 = time.perf_counter()

    print("Timing (all queries)")
    print("------")
    print(f"  Exact brute-force ({full_dim}d):          {1e3 * (t1 - t0):.3f} ms")
    print(f"  Coarse pass ({prefix_dim}d):                {1e3 * (t2 - t1):.3f} ms")
    print(f"  Two-stage total (coarse + re-rank): {1e3 * (t3 - t2):.3f} ms\n")
    cr = np.mean([recall(coarse_idx[j].tolist(), exact_idx[j].tolist()) for j in range(num_queries)])
    fr = np.mean([recall(fine_idx[j].tolist(), exact_idx[j].tolist()) for j in range(num_queries)])
    print("Recall")
    print("------")
    print(f"  Coarse recall@{final_k} (top-{coarse_k} at {prefix_dim}d): {100 * cr:.1f}%")
    print(f"  Final recall@{final_k}:                  {100 * fr:.1f}%\n")
    print(f"Top-{final_k} of query 0 (exact vs two-stage)")
    for r in range(final_k):
        tag = " " if exact_idx[0][r] == fine_idx[0][r] else "*"
        print(f"  #{r + 1:>2} exact: idx={exact_idx[0][r]:>7} sim={exact_sc[0][r]:.6f}  | two-stage: idx={fine_idx[0][r]:>7} "
              f"sim={fine_sc[0][r]:.6f} {tag}")
    # every two-stage score is the exact full-dimension cosine of its vector: wherever both lists hold a vector, bits agree
    for j in range(num_queries):
        ex = dict(zip(exact_idx[j].tolist(), exact_sc[j].tolist()))
        assert all(ex[i] == s for i, s in zip(fine_idx[j].tolist(), fine_sc[j].tolist()) if i in ex)
    print("\nre-ranked scores are the exact full-dimension cosines (bitwise, wherever the lists overlap)")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 10_000)
