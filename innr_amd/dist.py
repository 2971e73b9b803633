"""Range-partitioned corpus across the GPUs of one node (one process per GPU, torch.distributed / RCCL).

SURVEY.md 8(e): every corpus vector's score is independent and top-k is a selection, so the path shards with
ONE exchange step: rank g owns the contiguous index range [start_g, start_g + n_g) as its own PDX batch
(index_base = start_g), searches it locally, all-gathers the per-shard (score, global index) top-k -- Q*k*12
bytes per rank, latency-bound over xGMI -- and every rank merges G*k -> k per query by
(score order, index ascending). Contiguous ranges + the index tie-break reproduce the reference's stable-sort
tie rule (batch.rs:757) globally. No all-reduce, no data-path collective besides the gather.

`local_search` and `merge` are injectable so the orchestration (ranges, bases, gather layout, merge contract)
is covered by world_size-2 gloo tests on CPU; the defaults are the HIP kernels (no CPU fallback).
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Optional, Tuple

import numpy as np

from . import _lib
from ._lib import KNN_AUTO, METRIC_COSINE, METRIC_DOT, METRIC_L2SQ, KnnStats, check, load

INVALID_INDEX = -1  # int64 view of UINT64_MAX: "this shard had fewer than k vectors"


def shard_range(n_total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous range of rank `rank`: (start, count); the first n_total % world ranks get one extra."""
    base, rem = divmod(int(n_total), int(world))
    count = base + (1 if rank < rem else 0)
    start = rank * base + min(rank, rem)
    return start, count


def _gpu_local_search(batch, metric: int, engine: int):
    import torch

    def run(queries: "torch.Tensor", k: int, stats: Optional[KnnStats] = None):
        assert queries.is_cuda and queries.dtype == torch.float32 and queries.is_contiguous()
        nq, d = queries.shape
        kk = min(int(k), batch.num_vectors())
        # empty, not filled: a fill kernel on torch's stream could land after the library's writes
        idx = torch.empty((nq, max(kk, 1)), dtype=torch.int64, device=queries.device)
        sc = torch.empty((nq, max(kk, 1)), dtype=torch.float32, device=queries.device)
        out_k = C.c_size_t(0)
        st = stats if stats is not None else KnnStats()
        check(load().innr_batch_knn_dev(batch._h, metric, C.c_void_p(queries.data_ptr()), nq, d, int(k), engine,
                                        C.c_void_p(idx.data_ptr()), C.c_void_p(sc.data_ptr()), C.byref(out_k),
                                        C.byref(st)))
        return idx[:, :out_k.value], sc[:, :out_k.value]

    return run


def _gpu_local_search_u8(qcorpus, engine: int):
    """Shard-local batch_knn_u8 (scalar.rs:370-393) with device-resident queries and results."""
    import torch

    def run(queries: "torch.Tensor", k: int, stats: Optional[KnnStats] = None):
        assert queries.is_cuda and queries.dtype == torch.float32 and queries.is_contiguous()
        nq, d = queries.shape
        kk = min(int(k), len(qcorpus))
        idx = torch.empty((nq, max(kk, 1)), dtype=torch.int64, device=queries.device)
        sc = torch.empty((nq, max(kk, 1)), dtype=torch.float32, device=queries.device)
        out_k = C.c_size_t(0)
        st = stats if stats is not None else KnnStats()
        check(load().innr_batch_knn_u8_dev(qcorpus._h, C.c_void_p(queries.data_ptr()), nq, d, int(k), engine,
                                           C.c_void_p(idx.data_ptr()), C.c_void_p(sc.data_ptr()), C.byref(out_k),
                                           C.byref(st)))
        return idx[:, :out_k.value], sc[:, :out_k.value]

    return run


def _gpu_local_search_docs(corpus, cosine: bool, engine: int):
    """Shard-local maxsim top-k: `queries` is ONE query's token matrix [Tq, dim]; returns [1, k'] blocks."""
    import torch

    def run(query_tokens: "torch.Tensor", k: int, stats: Optional[KnnStats] = None):
        q = query_tokens.detach().cpu().numpy()  # Tq x dim floats: the entry point takes the query from the host
        idx, sc = corpus.topk(q, k, cosine=cosine, stats=stats, engine=engine)
        dev = query_tokens.device
        return (torch.from_numpy(idx.astype(np.int64)).reshape(1, -1).to(dev),
                torch.from_numpy(sc).reshape(1, -1).to(dev))

    return run


def _gpu_merge(ctx: _lib.Context, metric: int):
    import torch

    def run(all_idx: "torch.Tensor", all_sc: "torch.Tensor", kout: int):
        g, nq, kin = all_idx.shape
        out_i = torch.empty((nq, kout), dtype=torch.int64, device=all_idx.device)
        out_s = torch.empty((nq, kout), dtype=torch.float32, device=all_idx.device)
        check(load().innr_merge_topk_dev(ctx.handle, metric, C.c_void_p(all_idx.data_ptr()),
                                         C.c_void_p(all_sc.data_ptr()), g, nq, kin, kout,
                                         C.c_void_p(out_i.data_ptr()), C.c_void_p(out_s.data_ptr())))
        return out_i, out_s

    return run


class ShardedKnn:
    """One rank's view of a corpus range-partitioned over the ranks of `group`."""

    def __init__(self, n_total: int, k_pad_to: Optional[int] = None, group=None, rank: Optional[int] = None,
                 world: Optional[int] = None, local_search: Optional[Callable] = None,
                 merge: Optional[Callable] = None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        self.n_total = int(n_total)
        self.start, self.count = shard_range(n_total, self.world, self.rank)
        self.local_search = local_search
        self.merge = merge

    def attach_gpu_batch(self, batch, metric: int, engine: int = KNN_AUTO) -> None:
        """Use a device-resident VerticalBatch holding rows [start, start+count) as this rank's shard."""
        assert batch.num_vectors() == self.count
        batch.set_index_base(self.start)
        self.local_search = _gpu_local_search(batch, metric, engine)
        self.merge = _gpu_merge(batch._ctx, metric)

    def attach_gpu_u8(self, qcorpus, engine: int = KNN_AUTO) -> None:
        """Shard = a device-resident QuantizedCorpus holding documents [start, start+count) (scalar::batch_knn_u8)."""
        assert len(qcorpus) == self.count
        qcorpus.set_index_base(self.start)
        self.local_search = _gpu_local_search_u8(qcorpus, engine)
        self.merge = _gpu_merge(qcorpus._ctx, METRIC_DOT)

    def attach_gpu_docs(self, corpus, cosine: bool = False, engine: int = KNN_AUTO) -> None:
        """Shard = a device-resident maxsim DocumentCorpus holding documents [start, start+count); search() then takes
        ONE query's token matrix and returns [1, k] blocks."""
        assert len(corpus) == self.count
        corpus.set_index_base(self.start)
        self.local_search = _gpu_local_search_docs(corpus, cosine, engine)
        self.merge = _gpu_merge(corpus._ctx, METRIC_DOT)
        self._one_query = True

    def search(self, queries, k: int, stats: Optional[KnnStats] = None):
        """queries: [Q, D] tensor on this rank's device (identical on every rank). Returns the global top-k:
        (indices int64 [Q, k'], scores float32 [Q, k']) with k' = min(k, n_total), identical on every rank."""
        import torch
        kout = min(int(k), self.n_total)
        idx, sc = self.local_search(queries, k, stats) if stats is not None else self.local_search(queries, k)
        nq = 1 if getattr(self, "_one_query", False) else queries.shape[0]
        # pad to a fixed [Q, kin] block so every rank gathers the same shape (a shard may hold < k vectors)
        kin = min(int(k), max(shard_range(self.n_total, self.world, r)[1] for r in range(self.world)))
        kin = max(kin, 1)
        pad_i = torch.full((nq, kin), INVALID_INDEX, dtype=torch.int64, device=idx.device)
        pad_s = torch.zeros((nq, kin), dtype=torch.float32, device=idx.device)
        pad_i[:, :idx.shape[1]] = idx
        pad_s[:, :sc.shape[1]] = sc
        # one exchange step: all-gather of the per-shard candidates (RCCL over xGMI on GPUs, gloo in CPU tests). A gloo
        # group with device tensors (a rehearsal of the N > 1 path on a box whose ranks share one GPU) gathers through the
        # host: Q*k*12 bytes per rank.
        via_host = idx.is_cuda and self.dist.get_backend(self.group) == "gloo"
        gdev = torch.device("cpu") if via_host else idx.device
        all_i = torch.empty((self.world, nq, kin), dtype=torch.int64, device=gdev)
        all_s = torch.empty((self.world, nq, kin), dtype=torch.float32, device=gdev)
        self.dist.all_gather(list(all_i.unbind(0)), pad_i.contiguous().to(gdev), group=self.group)
        self.dist.all_gather(list(all_s.unbind(0)), pad_s.contiguous().to(gdev), group=self.group)
        if via_host:
            all_i, all_s = all_i.to(idx.device), all_s.to(idx.device)
        return self.merge(all_i, all_s, kout)
