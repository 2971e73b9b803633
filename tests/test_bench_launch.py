"""`python bench.py --gpus N` from a bare shell (no torchrun around it, WORLD_SIZE unset) must start its N ranks itself, relay rank
0's one JSON line and exit non-zero when a rank fails. Runs on the CPU box through --rehearse-cpu: launcher, gloo rendezvous on
127.0.0.1, range partition, ONE all_gather of the exchange blocks, merge, barrier / MAX timing and the relay are the real code
paths of bench.py / innr_amd.dist; the per-shard search is a torch stand-in (nothing is measured, value = null)."""
from __future__ import annotations

import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *extra], capture_output=True, text=True, env=env,
                          timeout=600, cwd=ROOT)


def test_bench_self_launches_two_ranks():
    r = _run("--gpus", "2", "--steps", "2", "--warmup", "1", "--rehearse-cpu", "--n-per-gpu", "3000", "--dim", "24", "--queries", "17",
             "--k", "5")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["value"] is None and "rehearsal" in out and out["sharded_result_equals_whole_corpus"] is True


def test_bench_self_launch_three_ranks_uneven_k():
    r = _run("--gpus", "3", "--steps", "1", "--warmup", "0", "--rehearse-cpu", "--n-per-gpu", "7", "--dim", "8", "--queries", "5", "--k", "10")
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 3


def test_bench_self_launch_reports_a_failing_rank():
    """Without --rehearse-cpu the ranks need a GPU: on the CPU box every rank fails, and the launcher's exit code must say so
    (no JSON line on stdout)."""
    if os.path.exists("/dev/kfd"):
        import pytest
        pytest.skip("needs a box without a GPU")
    r = _run("--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--n-per-gpu", "1000")
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.lstrip().startswith("{")]
