#!/usr/bin/env python3
"""The reference's batch walkthrough (examples/batch_demo.rs) on the MI355X library: the same four stops -- layout,
batch kNN against a naive per-vector loop, batch dot, timing at the example's scale (10 000 x 128, 100 queries) -- with
innr_amd in place of innr::batch and the host pairwise functions in place of innr::{dot, l2_distance_squared}.

    python examples/batch_demo.py            (needs a GPU; corpus from the example's own generator, on the device)
"""
from __future__ import annotations

import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from innr_amd import GEN_EXAMPLE_LCG, distance
from innr_amd import batch as B


def generate_embedding(dim: int, seed: int) -> np.ndarray:
    """examples/batch_demo.rs:233-242, via the library's device generator (row `seed` of the example stream)."""
    return B.VerticalBatch.generate(1, dim, seed=seed, generator=GEN_EXAMPLE_LCG).extract_vector(0)


def demo_layout() -> None:
    print("1. Row-major vs column-major layout")
    vectors = [[1.0, 2.0, 3.0], [4.0, 5.0, 6.0], [7.0, 8.0, 9.0]]
    batch = B.VerticalBatch.from_rows(vectors)
    for d in range(batch.dimension()):
        print(f"     dim {d}: {batch.dimension_slice(d).tolist()}")
    for i, v in enumerate(vectors):
        assert batch.extract_vector(i).tolist() == v, f"round-trip failed for vector {i}"
    print("   round trip verified\n")


def demo_knn(n: int = 20, dim: int = 8, k: int = 3) -> None:
    print("2. Batch kNN vs a naive brute-force loop")
    batch = B.VerticalBatch.generate(n, dim, seed=0, generator=GEN_EXAMPLE_LCG)  # row i = generate_embedding(dim, i)
    corpus = [batch.extract_vector(i) for i in range(n)]
    query = generate_embedding(dim, 999)
    res = B.batch_knn(query, batch, k)
    naive = sorted(((distance.l2_distance_squared(query, v), i) for i, v in enumerate(corpus)))[:k]
    for rank, (idx, dist) in enumerate(zip(res.indices, res.scores)):
        print(f"     #{rank + 1}: index={idx}, dist_sq={dist:.6f}   (naive: index={naive[rank][1]}, dist_sq={naive[rank][0]:.6f})")
        assert idx == naive[rank][1] and abs(dist - naive[rank][0]) < 1e-5
    print("   match: indices and distances agree\n")


def demo_batch_dot() -> None:
    print("3. Batch dot product: one query against many documents")
    docs = [[1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.7, 0.7, 0.0], [-1.0, 0.0, 0.0]]
    batch = B.VerticalBatch.from_rows(docs)
    query = [1.0, 0.0, 0.0]
    for i, (p, d) in enumerate(zip(B.batch_dot(query, batch), docs)):
        assert abs(p - distance.dot(query, d)) < 1e-6
        print(f"     doc {i}: dot={p:.4f}")
    print()


def demo_timing(n: int = 10_000, dim: int = 128, num_queries: int = 100, k: int = 10) -> None:
    print("4. Timing at the example's scale")
    batch = B.VerticalBatch.generate(n, dim, seed=0, generator=GEN_EXAMPLE_LCG)
    queries = B.VerticalBatch.generate(num_queries, dim, seed=50_000, generator=GEN_EXAMPLE_LCG)
    qs = np.ascontiguousarray(queries.data().T)
    B.batch_l2_squared(qs[0], batch)  # warm up
    t0 = time.perf_counter()
    checksum = 0.0
    for q in qs:
        checksum += float(B.batch_l2_squared(q, batch).sum(dtype=np.float64))
    t_single = time.perf_counter() - t0
    t0 = time.perf_counter()
    idx, sc = B.batch_knn_multi(qs, batch, k)
    t_multi = time.perf_counter() - t0
    print(f"   corpus {n} x {dim}, {num_queries} queries")
    print(f"   batch_l2_squared, one query per call: {t_single * 1e3:8.2f} ms total ({t_single / num_queries * 1e6:.0f} us per query), checksum {checksum:.3f}")
    print(f"   batch_knn, all queries in one call:   {t_multi * 1e3:8.2f} ms total, first result {idx[0, :3].tolist()}")
    one = B.batch_knn(qs[0], batch, k)
    assert one.indices == [int(i) for i in idx[0]]
    print()


if __name__ == "__main__":
    print("Batch operations with the PDX-style columnar layout, on the GPU\n")
    demo_layout()
    demo_knn()
    demo_batch_dot()
    demo_timing()
    print("Done!")
