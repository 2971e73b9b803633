"""Golden-vector cases (tests/golden/*.npz): inputs are regenerated from integer-exact generators
(oracle.generate_uniform / the reference example's LCG, examples/batch_demo.rs:233-242), expected outputs are stored.
`python tests/golden/make_golden.py` (re)creates the files with the CPU oracle; tests/test_golden.py checks that the
oracle still reproduces them (CPU) and that the HIP path reproduces them (GPU). Scores are stored as raw f32 bits."""
from __future__ import annotations

import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# name -> parameters
CASES = {
    # the reference example's CPU-runnable configuration (BASELINE.json configs[0]): corpus row i = seed i,
    # query j = seed j + 50_000 (examples/batch_demo.rs:167-170), first 16 of the 100 queries
    "c1_lcg_10000x128": dict(kind="knn", gen="lcg", n=10_000, dim=128, nq=16, qseed0=50_000, k=10),
    "uniform_5000x96": dict(kind="knn", gen="uniform", n=5_000, dim=96, nq=8, qseed=77, k=10),
    "uniform_777x33_k50": dict(kind="knn", gen="uniform", n=777, dim=33, nq=4, qseed=5, k=50),
    "scores_1000x33": dict(kind="scores", gen="uniform", n=1_000, dim=33, nq=2, qseed=9),
    "l2family_3000x40": dict(kind="l2family", gen="uniform", n=3_000, dim=40, nq=3, qseed=21, k=12, mod=3, thr=11.5),
    "u8_4000x64": dict(kind="u8", n=4_000, dim=64, nq=6, qseed=31, k=20, mn=-1.0, mx=1.0),
    "maxsim_300x16x48": dict(kind="maxsim", ndocs=300, T=16, dim=48, Tq=8, qseed=41),
}


def corpus_rows(oracle, p):
    if p["gen"] == "lcg":
        return oracle.generate_corpus(p["n"], p["dim"], 0)
    return oracle.generate_uniform(p["n"], p["dim"], 0)


def query_rows(oracle, p):
    if p.get("gen") == "lcg":
        return oracle.generate_corpus(p["nq"], p["dim"], p["qseed0"])
    return oracle.generate_uniform(p["nq"], p["dim"], p["qseed"])


def bits(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32)).view(np.uint32)


def maxsim_inputs(oracle, p):
    """documents: rows of the uniform stream, L2-normalised per token like examples/maxsim_colbert.rs:212-228 (the
    normalisation is done in float64 then rounded once: any implementation gets the same f32 inputs)"""
    raw = oracle.generate_uniform(p["ndocs"] * p["T"], p["dim"], 3).astype(np.float64)
    tok = (raw / np.sqrt((raw * raw).sum(axis=1, keepdims=True))).astype(np.float32)
    rq = oracle.generate_uniform(p["Tq"], p["dim"], p["qseed"]).astype(np.float64)
    q = (rq / np.sqrt((rq * rq).sum(axis=1, keepdims=True))).astype(np.float32)
    return tok.reshape(p["ndocs"], p["T"], p["dim"]), q


def path(name: str) -> str:
    return os.path.join(GOLDEN_DIR, name + ".npz")
