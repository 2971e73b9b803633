#!/usr/bin/env python3
"""Create tests/golden/*.npz with the CPU oracle (run from the repo root: python tests/golden/make_golden.py).
The oracle itself is pinned by the reference's known-answer tests (tests/test_oracle_kat.py); these files freeze
its outputs on larger, generator-defined inputs so that neither the oracle nor the HIP path can drift unnoticed."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import numpy as np

import golden_cases as G
import oracle


def expected(name: str) -> dict:
    p = G.CASES[name]
    out = {}
    if p["kind"] in ("knn", "scores", "l2family"):
        data = oracle.from_rows(G.corpus_rows(oracle, p))
        qs = G.query_rows(oracle, p)
    if p["kind"] == "knn":
        for metric, fn in (("dot", oracle.batch_knn_dot), ("cos", oracle.batch_knn_cosine), ("l2", oracle.batch_knn)):
            res = [fn(q, data, p["k"]) for q in qs]
            out[f"{metric}_idx"] = np.stack([r[0] for r in res]).astype(np.uint64)
            out[f"{metric}_bits"] = np.stack([G.bits(r[1]) for r in res])
    elif p["kind"] == "scores":
        norms = oracle.batch_norms(data)
        out["norms_bits"] = G.bits(norms)
        out["dot_bits"] = np.stack([G.bits(oracle.batch_dot(q, data)) for q in qs])
        out["l2_bits"] = np.stack([G.bits(oracle.batch_l2_squared(q, data)) for q in qs])
        out["cos_bits"] = np.stack([G.bits(oracle.batch_cosine(q, data, norms)) for q in qs])
        out["var_bits"] = G.bits(oracle.batch_dimension_variance(data))
    elif p["kind"] == "l2family":
        mask = (np.arange(p["n"]) % p["mod"] == 0).astype(np.uint8)
        for j, q in enumerate(qs):
            i, s = oracle.batch_knn_filtered(q, data, p["k"], mask)
            out[f"filt_idx{j}"], out[f"filt_bits{j}"] = i.astype(np.uint64), G.bits(s)
            i, s = oracle.batch_knn_reordered(q, data, p["k"])
            out[f"reord_idx{j}"], out[f"reord_bits{j}"] = i.astype(np.uint64), G.bits(s)
            i, s = oracle.batch_l2_squared_pruning(q, data, p["thr"])
            out[f"prune_idx{j}"], out[f"prune_bits{j}"] = i.astype(np.uint64), G.bits(s)
    elif p["kind"] == "u8":
        qp = oracle.qparams_from_range(p["mn"], p["mx"])
        codes = oracle.quantize_u8(oracle.generate_uniform(p["n"], p["dim"], 0), qp)
        qs = oracle.generate_uniform(p["nq"], p["dim"], p["qseed"])
        res = [oracle.batch_knn_u8(q, codes, qp, p["k"]) for q in qs]
        out["idx"] = np.stack([r[0] for r in res]).astype(np.uint64)
        out["bits"] = np.stack([G.bits(r[1]) for r in res])
        out["codes_crc"] = np.array([int(codes.astype(np.uint64).sum()), int((codes.astype(np.uint64) * (np.arange(codes.size, dtype=np.uint64).reshape(codes.shape) % 251)).sum())], dtype=np.uint64)
    elif p["kind"] == "maxsim":
        tok, q = G.maxsim_inputs(oracle, p)
        out["dot_bits"] = G.bits([oracle.maxsim(q, d) for d in tok])
        out["cos_bits"] = G.bits([oracle.maxsim_cosine(q, d) for d in tok])
    return out


if __name__ == "__main__":
    for name in G.CASES:
        np.savez_compressed(G.path(name), **expected(name))
        print(name, os.path.getsize(G.path(name)), "bytes")
