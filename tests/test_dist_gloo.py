"""N > 1 path on CPU: world_size-2 (and 3) `gloo` runs of innr_amd.dist.ShardedKnn -- range partition, index
bases, the all-gather layout and the merge contract. The per-shard search and the merge are injected
stand-ins here (the oracle, test infrastructure): the product's own defaults are the HIP kernels, which the
GPU test `test_gpu_shards.py` covers with G logical shards on one device."""
from __future__ import annotations

import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _ord(x: torch.Tensor) -> torch.Tensor:
    """f32::total_cmp key as int64 (monotone)."""
    b = x.contiguous().view(torch.int32).to(torch.int64)
    return torch.where(b < 0, -(b & 0x7FFFFFFF) - 1, b)


def torch_merge(metric_desc: bool):
    """Reference merge used by the gloo test: sort G*k candidates by (score order, index ascending)."""
    def run(all_idx, all_sc, kout):
        g, nq, kin = all_idx.shape
        idx = all_idx.permute(1, 0, 2).reshape(nq, g * kin)
        sc = all_sc.permute(1, 0, 2).reshape(nq, g * kin)
        key = _ord(sc)
        key = torch.where(idx < 0, torch.full_like(key, -(1 << 40)), key if metric_desc else -key)
        out_i = torch.empty((nq, kout), dtype=torch.int64)
        out_s = torch.empty((nq, kout), dtype=torch.float32)
        for q in range(nq):
            order = sorted(range(g * kin), key=lambda c: (-int(key[q, c]), int(idx[q, c]) if idx[q, c] >= 0 else 1 << 62))
            out_i[q] = idx[q, order[:kout]]
            out_s[q] = sc[q, order[:kout]]
        return out_i, out_s
    return run


def _worker(rank, world, port, n_total, dim, nq, k, metric, retq):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from innr_amd.dist import ShardedKnn, shard_range

    start, count = shard_range(n_total, world, rank)
    rows = oracle.generate_uniform(count, dim, 7, row0=start)  # this rank's range of the global stream
    data = oracle.from_rows(rows) if count else np.empty((dim, 0), np.float32)
    ofn = {"dot": oracle.batch_knn_dot, "cos": oracle.batch_knn_cosine, "l2": oracle.batch_knn}[metric]

    def local_search(queries, kk):  # stand-in for the GPU shard search: local top-k with GLOBAL indices
        qs = queries.numpy()
        kloc = min(kk, count)
        idx = torch.empty((len(qs), kloc), dtype=torch.int64)
        sc = torch.empty((len(qs), kloc), dtype=torch.float32)
        for j, q in enumerate(qs):
            i, s = ofn(q, data, kk) if count else (np.empty(0, np.uint64), np.empty(0, np.float32))
            idx[j] = torch.from_numpy((i.astype(np.int64) + start))
            sc[j] = torch.from_numpy(s)
        return idx, sc

    sk = ShardedKnn(n_total, local_search=local_search, merge=torch_merge(metric != "l2"))
    assert (sk.start, sk.count) == (start, count)
    queries = torch.from_numpy(oracle.generate_uniform(nq, dim, 99))
    idx, sc = sk.search(queries, k)
    retq.put((rank, idx.numpy(), sc.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_total,k,metric", [(2, 1001, 10, "dot"), (2, 1001, 10, "cos"), (2, 600, 25, "l2"),
                                                    (3, 50, 40, "dot"), (2, 3, 5, "dot")])
def test_sharded_knn_equals_single_corpus(world, n_total, k, metric):
    import oracle
    dim, nq = 24, 5
    ctx = mp.get_context("spawn")
    retq = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, dim, nq, k, metric, retq)) for r in range(world)]
    for p in procs:
        p.start()
    results = [retq.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    data = oracle.from_rows(oracle.generate_uniform(n_total, dim, 7))
    queries = oracle.generate_uniform(nq, dim, 99)
    ofn = {"dot": oracle.batch_knn_dot, "cos": oracle.batch_knn_cosine, "l2": oracle.batch_knn}[metric]
    for rank, idx, sc in results:
        assert idx.shape == (nq, min(k, n_total))
        for j in range(nq):
            oi, os_ = ofn(queries[j], data, k)
            assert idx[j].tolist() == oi.astype(np.int64).tolist(), (rank, j)
            assert np.array_equal(sc[j].view(np.uint32), os_.view(np.uint32))


def test_shard_range_partitions_exactly():
    from innr_amd.dist import shard_range
    for n in (0, 1, 7, 8, 80_000_000, 10**10 + 3):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == n
            assert all(spans[r][0] + spans[r][1] == spans[r + 1][0] for r in range(w - 1))
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1


def _failing_worker(rank, world, port, bad_rank, retq):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from innr_amd import _lib
    from innr_amd.dist import ShardedKnn, shard_range
    n_total, dim, nq, k = 90, 8, 4, 5
    start, count = shard_range(n_total, world, rank)
    rows = torch.arange(n_total * dim, dtype=torch.float32).reshape(n_total, dim)[start:start + count].sin()

    def local_search(queries, kk):
        if rank == bad_rank:
            raise _lib.InnrError(_lib.E_OOM, "hipMalloc for the filter copy failed on this rank only")
        sc, idx = torch.topk(queries @ rows.T, min(kk, count), dim=1)
        return idx.to(torch.int64) + start, sc

    sk = ShardedKnn(n_total, local_search=local_search, merge=torch_merge(True))
    queries = torch.ones((nq, dim))
    try:
        sk.search(queries, k)
        outcome = ("ok", 0, "")
    except _lib.InnrError as exc:
        outcome = ("error", exc.status, str(exc))
    dist.barrier()  # every rank is out of the exchange: nobody is left waiting in the all_gather
    retq.put((rank, outcome))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,bad_rank", [(2, 1), (3, 0)])
def test_failing_local_search_makes_every_rank_return_an_error(world, bad_rank):
    """The exchange is symmetric: a rank whose local search fails still gathers (an error block), raises its own error, and
    every other rank raises INNR_E_RCCL naming it -- nobody hangs in the collective (include/innr_hip.h, innr_sharded_*)."""
    from innr_amd import _lib
    ctx = mp.get_context("spawn")
    retq = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_failing_worker, args=(r, world, port, bad_rank, retq)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(retq.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, (kind, status, msg) in results.items():
        assert kind == "error", (rank, kind)
        if rank == bad_rank:
            assert status == _lib.E_OOM and "this rank only" in msg
        else:
            assert status == _lib.E_RCCL and f"rank {bad_rank}" in msg and f"status {_lib.E_OOM}" in msg


def _uid_worker(rank, world, port, retq):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from innr_amd import _lib
    from innr_amd.dist import Comm

    def no_rccl():
        raise _lib.InnrError(_lib.E_RCCL, "RCCL is not available: librccl.so[.1] could not be loaded")

    Comm.unique_id = staticmethod(no_rccl)  # what a box without librccl does on rank 0
    try:
        Comm.from_torch_group(ctx=None)
        outcome = "ok"
    except _lib.InnrError as exc:
        outcome = f"error {exc.status}: {exc}"
    # the fallback path of bench.py: every rank must arrive at the SAME next collective
    ok = torch.tensor([0 if outcome.startswith("error") else 1], dtype=torch.int32)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    retq.put((rank, outcome, int(ok.item())))
    dist.barrier()
    dist.destroy_process_group()


def test_comm_bootstrap_failure_on_rank0_reaches_every_rank():
    """Comm.from_torch_group: rank 0 failing to draw the communicator id must not leave the other ranks in the broadcast
    (round 2: rank 0 raised first and went on to bench.py's all_reduce while rank 1 still sat in broadcast_object_list)."""
    world = 2
    ctx = mp.get_context("spawn")
    retq = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_uid_worker, args=(r, world, port, retq)) for r in range(world)]
    for p in procs:
        p.start()
    results = [retq.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, outcome, agreed in results:
        assert outcome.startswith("error -5") and "librccl" in outcome and agreed == 0, (rank, outcome)
