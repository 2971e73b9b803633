"""Range-partitioned corpus across the GPUs of one node (one process per GPU).

SURVEY.md 8(e): every corpus vector's score is independent and top-k is a selection, so the path shards with
ONE exchange step: rank g owns the contiguous index range [start_g, start_g + n_g) as its own PDX batch
(index_base = start_g), searches it locally, all-gathers the per-shard top-k -- one block of 2 + Q*k uint64 per rank:
{index base, vector count, then 8 bytes per candidate = u32 index local to the shard + f32 score}, latency-bound over
xGMI -- and every rank merges G*k -> k per query by (score order, global index ascending). Contiguous ranges + the index
tie-break reproduce the reference's stable-sort tie rule (batch.rs:757) globally. No all-reduce, no data-path
collective besides the gather.

The exchange lives BEHIND the C ABI (include/innr_hip.h: innr_comm_*, innr_sharded_knn_dev): `Comm` wraps an RCCL
communicator owned by the library, and ShardedKnn.search() with a Comm attached is one library call (local search + pack
+ ncclAllGather + merge on the ctx stream). torch.distributed is used only to hand the communicator id to the other
ranks (and by bench.py for its barrier / timing reduction).

Without a Comm (`local_search` / `merge` injected: the world_size-2 gloo tests on CPU, or a one-GPU rehearsal where RCCL
refuses two ranks per device) the same block format travels through ONE torch.distributed.all_gather.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Optional, Tuple

import numpy as np

from . import _lib
from ._lib import KNN_AUTO, METRIC_COSINE, METRIC_DOT, METRIC_L2SQ, KnnStats, check, load

INVALID_INDEX = -1  # int64 view of UINT64_MAX: "this shard had fewer than k vectors"
_NO_CANDIDATE = 0xFFFFFFFF


def shard_range(n_total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous range of rank `rank`: (start, count); the first n_total % world ranks get one extra."""
    base, rem = divmod(int(n_total), int(world))
    count = base + (1 if rank < rem else 0)
    start = rank * base + min(rank, rem)
    return start, count


class Comm:
    """innr_comm: this rank's context + an RCCL communicator over all ranks, owned by the library."""

    def __init__(self, ctx: _lib.Context, rank: int, world: int, uid: bytes):
        assert len(uid) == _lib.COMM_ID_BYTES
        h = C.c_void_p()
        buf = (C.c_char * _lib.COMM_ID_BYTES).from_buffer_copy(uid)
        check(load().innr_comm_create(ctx.handle, buf, int(rank), int(world), C.byref(h)))  # collective
        self._h, self.ctx, self.rank, self.world = h, ctx, int(rank), int(world)
        ctx._children.add(self)

    @staticmethod
    def unique_id() -> bytes:
        buf = (C.c_char * _lib.COMM_ID_BYTES)()
        check(load().innr_comm_unique_id(buf))
        return bytes(buf.raw)

    @classmethod
    def from_torch_group(cls, ctx: _lib.Context, group=None) -> "Comm":
        """Rank 0 draws the id, torch.distributed carries it to the other ranks (bootstrap only), every rank joins."""
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        # Rank 0 ALWAYS takes part in the broadcast -- with the id, or with the reason it has none (no librccl, a missing
        # symbol): the other ranks are already waiting in it, and a rank 0 that raised first would leave them there while it
        # moves on to its caller's next collective. After the broadcast every rank raises the same error.
        uid, err = None, None
        if rank == 0:
            try:
                uid = cls.unique_id()
            except Exception as exc:
                err = f"{type(exc).__name__}: {exc}"
        box = [(uid, err)]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        uid, err = box[0]
        if uid is None:
            raise _lib.InnrError(_lib.E_RCCL, f"rank 0 could not draw a communicator id: {err}")
        return cls(ctx, rank, world, uid)

    def close(self) -> None:
        if getattr(self, "_h", None):
            load().innr_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _gpu_local_search(batch, metric: int, engine: int):
    import torch

    def run(queries: "torch.Tensor", k: int, stats: Optional[KnnStats] = None):
        assert queries.is_cuda and queries.dtype == torch.float32 and queries.is_contiguous()
        batch._ctx.bind_torch_stream()  # the queries were produced on torch's stream, the results are consumed there
        nq, d = queries.shape
        kk = min(int(k), batch.num_vectors())
        # empty, not filled: a fill kernel on another stream could land after the library's writes
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
        qcorpus._ctx.bind_torch_stream()
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
    """[G, Q, kin] (global index, score) arrays -> best kout per query (innr_merge_topk_dev)."""
    import torch

    def run(all_idx: "torch.Tensor", all_sc: "torch.Tensor", kout: int):
        ctx.bind_torch_stream()
        g, nq, kin = all_idx.shape
        out_i = torch.empty((nq, kout), dtype=torch.int64, device=all_idx.device)
        out_s = torch.empty((nq, kout), dtype=torch.float32, device=all_idx.device)
        check(load().innr_merge_topk_dev(ctx.handle, metric, C.c_void_p(all_idx.data_ptr()),
                                         C.c_void_p(all_sc.data_ptr()), g, nq, kin, kout,
                                         C.c_void_p(out_i.data_ptr()), C.c_void_p(out_s.data_ptr())))
        return out_i, out_s

    return run


def gpu_pack_block(ctx: _lib.Context, idx: "torch.Tensor", sc: "torch.Tensor", index_base: int, shard_vectors: int, k: int):
    """A shard's device-resident kNN result -> its exchange block (innr_topk_pack_dev): int64 tensor [2 + Q*k]."""
    import torch
    ctx.bind_torch_stream()
    nq, kin = idx.shape
    block = torch.empty((2 + nq * int(k),), dtype=torch.int64, device=idx.device)
    idx, sc = idx.contiguous(), sc.contiguous()
    check(load().innr_topk_pack_dev(ctx.handle, C.c_void_p(idx.data_ptr()), C.c_void_p(sc.data_ptr()), int(index_base),
                                    int(shard_vectors), nq, kin, int(k), C.c_void_p(block.data_ptr())))
    return block


def gpu_merge_blocks(ctx: _lib.Context, metric: int, blocks: "torch.Tensor", nq: int, k: int):
    """[G, 2 + Q*k] gathered blocks -> (indices int64 [Q, k'], scores [Q, k']) (innr_merge_blocks_dev)."""
    import torch
    ctx.bind_torch_stream()
    blocks = blocks.contiguous()
    g = blocks.shape[0]
    out_i = torch.empty((nq, max(int(k), 1)), dtype=torch.int64, device=blocks.device)
    out_s = torch.empty((nq, max(int(k), 1)), dtype=torch.float32, device=blocks.device)
    out_k = C.c_size_t(0)
    check(load().innr_merge_blocks_dev(ctx.handle, metric, C.c_void_p(blocks.data_ptr()), g, nq, int(k),
                                       C.c_void_p(out_i.data_ptr()), C.c_void_p(out_s.data_ptr()), C.byref(out_k)))
    r = int(out_k.value)
    # the kernel writes rows of k' entries: re-view the flat storage with that stride
    return (out_i.reshape(-1)[:nq * r].reshape(nq, r), out_s.reshape(-1)[:nq * r].reshape(nq, r))


def pack_block_host(idx: "torch.Tensor", sc: "torch.Tensor", index_base: int, shard_vectors: int, k: int) -> "torch.Tensor":
    """The same block built with torch ops (CPU stand-in path of the gloo tests): int64 [2 + Q*k]."""
    import torch
    nq, kin = idx.shape
    local = torch.full((nq, int(k)), _NO_CANDIDATE, dtype=torch.int64)
    bits = torch.zeros((nq, int(k)), dtype=torch.int64)
    if kin:
        local[:, :kin] = idx.cpu().to(torch.int64) - int(index_base)
        bits[:, :kin] = sc.cpu().contiguous().view(torch.int32).to(torch.int64) & 0xFFFFFFFF
    hi = torch.where(bits >= (1 << 31), bits - (1 << 32), bits)  # the score bits as the int64's (signed) upper half
    entries = (hi << 32) | local
    head = torch.tensor([int(index_base), int(shard_vectors)], dtype=torch.int64)
    return torch.cat([head, entries.reshape(-1)])


def unpack_blocks_host(blocks: "torch.Tensor", nq: int, k: int):
    """[G, 2 + Q*k] blocks -> ([G, Q, k] global indices (INVALID_INDEX where a shard had no candidate), [G, Q, k] scores,
    total vectors)."""
    import torch
    blocks = blocks.cpu()
    g = blocks.shape[0]
    base = blocks[:, 0]
    total = int(blocks[:, 1].sum())
    e = blocks[:, 2:].reshape(g, nq, int(k))
    local = e & 0xFFFFFFFF
    sc = (e >> 32).to(torch.int32).view(torch.float32)
    idx = torch.where(local == _NO_CANDIDATE, torch.full_like(local, INVALID_INDEX), local + base.view(g, 1, 1))
    return idx, sc, total


class ShardedKnn:
    """One rank's view of a corpus range-partitioned over the ranks of `group`."""

    def __init__(self, n_total: int, k_pad_to: Optional[int] = None, group=None, rank: Optional[int] = None,
                 world: Optional[int] = None, local_search: Optional[Callable] = None,
                 merge: Optional[Callable] = None, comm: Optional[Comm] = None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        self.n_total = int(n_total)
        self.start, self.count = shard_range(n_total, self.world, self.rank)
        self.local_search = local_search
        self.merge = merge
        self.comm = comm
        self._shard = None      # (handle, metric, engine) of a batch the library searches itself (comm path)
        self._ctx = None
        self._metric = METRIC_DOT

    def _attach(self, obj, ctx, metric):
        self._ctx, self._metric = ctx, metric
        ctx.bind_torch_stream()  # torch's pad / gather kernels and the library's kernels: one stream, one order

    def attach_gpu_batch(self, batch, metric: int, engine: int = KNN_AUTO) -> None:
        """Use a device-resident VerticalBatch holding rows [start, start+count) as this rank's shard."""
        assert batch.num_vectors() == self.count
        batch.set_index_base(self.start)
        self.local_search = _gpu_local_search(batch, metric, engine)
        self.merge = _gpu_merge(batch._ctx, metric)
        self._attach(batch, batch._ctx, metric)
        self._shard = (batch, metric, engine)

    def attach_gpu_u8(self, qcorpus, engine: int = KNN_AUTO) -> None:
        """Shard = a device-resident QuantizedCorpus holding documents [start, start+count) (scalar::batch_knn_u8)."""
        assert len(qcorpus) == self.count
        qcorpus.set_index_base(self.start)
        self.local_search = _gpu_local_search_u8(qcorpus, engine)
        self.merge = _gpu_merge(qcorpus._ctx, METRIC_DOT)
        self._attach(qcorpus, qcorpus._ctx, METRIC_DOT)
        self._shard = (qcorpus, METRIC_DOT, engine)

    def attach_gpu_docs(self, corpus, cosine: bool = False, engine: int = KNN_AUTO) -> None:
        """Shard = a device-resident maxsim DocumentCorpus holding documents [start, start+count); search() then takes
        ONE query's token matrix and returns [1, k] blocks."""
        assert len(corpus) == self.count
        corpus.set_index_base(self.start)
        self.local_search = _gpu_local_search_docs(corpus, cosine, engine)
        self.merge = _gpu_merge(corpus._ctx, METRIC_DOT)
        self._attach(corpus, corpus._ctx, METRIC_DOT)
        self._one_query = True
        self._docs = (corpus, bool(cosine), engine)

    def search(self, queries, k: int, stats: Optional[KnnStats] = None):
        """queries: [Q, D] tensor on this rank's device (identical on every rank). Returns the global top-k:
        (indices int64 [Q, k'], scores float32 [Q, k']) with k' = min(k, n_total), identical on every rank."""
        import torch
        if self.comm is not None and getattr(self, "_docs", None) is not None:
            # maxsim through the library's exchange (innr_sharded_maxsim): the query and the result are host arrays, like
            # innr_maxsim_topk's; the local top-k, the blocks and the merge stay on the device
            corpus, cosine, engine = self._docs
            q = np.ascontiguousarray(queries.detach().cpu().numpy() if hasattr(queries, "detach") else queries, np.float32)
            tq, dim = q.shape
            kk = max(int(k), 1)
            out_i = np.empty(kk, np.uint64)
            out_s = np.empty(kk, np.float32)
            out_k = C.c_size_t(0)
            st = stats if stats is not None else KnnStats()
            check(load().innr_sharded_maxsim(self.comm._h, corpus._h, int(cosine), q.ctypes.data, tq, dim, int(k), engine,
                                             out_i.ctypes.data, out_s.ctypes.data, C.byref(out_k), C.byref(st)))
            r = int(out_k.value)
            dev = queries.device if hasattr(queries, "device") else "cpu"
            return (torch.from_numpy(out_i[:r].astype(np.int64)).reshape(1, r).to(dev), torch.from_numpy(out_s[:r].copy()).reshape(1, r).to(dev))
        if self.comm is not None and self._shard is not None:
            # the whole sharded call behind the boundary: local search + pack + ONE ncclAllGather + merge, ctx stream
            obj, metric, engine = self._shard
            self._ctx.bind_torch_stream()
            assert queries.is_cuda and queries.dtype == torch.float32 and queries.is_contiguous()
            nq, d = queries.shape
            kout = max(min(int(k), self.n_total), 1)
            out_i = torch.empty((nq, kout), dtype=torch.int64, device=queries.device)
            out_s = torch.empty((nq, kout), dtype=torch.float32, device=queries.device)
            out_k = C.c_size_t(0)
            st = stats if stats is not None else KnnStats()
            check(load().innr_sharded_knn_dev(self.comm._h, obj._h, metric, C.c_void_p(queries.data_ptr()), nq, d, int(k), engine,
                                              C.c_void_p(out_i.data_ptr()), C.c_void_p(out_s.data_ptr()), C.byref(out_k),
                                              C.byref(st)))
            r = int(out_k.value)
            return out_i.reshape(-1)[:nq * r].reshape(nq, r), out_s.reshape(-1)[:nq * r].reshape(nq, r)
        nq = 1 if getattr(self, "_one_query", False) else queries.shape[0]
        kk = max(int(k), 1)
        # The exchange is SYMMETRIC (include/innr_hip.h, failure semantics of innr_sharded_*): a rank whose local search fails
        # still gathers -- a block whose header says so -- and every rank raises, instead of one rank leaving the others in
        # the collective.
        failure = None
        try:
            idx, sc = self.local_search(queries, k, stats) if stats is not None else self.local_search(queries, k)
        except Exception as exc:
            failure = exc
        # ONE exchange step: all-gather of every rank's block (gloo in the CPU tests; through the host when ranks of a
        # rehearsal share one GPU, where RCCL cannot run): (2 + Q*k) * 8 bytes per rank
        if failure is not None:
            block = torch.full((2 + nq * kk,), _NO_CANDIDATE, dtype=torch.int64)
            block[0], block[1] = int(getattr(failure, "status", _lib.E_HIP)) * -1, -1  # word 1 all ones: "this rank failed"
            if getattr(queries, "is_cuda", False):
                block = block.to(queries.device)
        elif idx.is_cuda:
            block = gpu_pack_block(self._ctx, idx, sc, self.start, self.count, kk)
        else:
            block = pack_block_host(idx, sc, self.start, self.count, kk)
        via_host = block.is_cuda and self.dist.get_backend(self.group) == "gloo"
        send = block.cpu() if via_host else block
        blocks = torch.empty((self.world, send.numel()), dtype=torch.int64, device=send.device)
        self.dist.all_gather(list(blocks.unbind(0)), send, group=self.group)
        if failure is not None:
            raise failure
        failed = [g for g in range(self.world) if int(blocks[g, 1]) == -1]
        if failed:
            raise _lib.InnrError(_lib.E_RCCL, f"rank {failed[0]} of the sharded call failed its local search "
                                              f"(status {-int(blocks[failed[0], 0])}): no rank has a result")
        if idx.is_cuda:
            return gpu_merge_blocks(self._ctx, self._metric, blocks.to(idx.device), nq, kk)
        all_i, all_s, total = unpack_blocks_host(blocks, nq, kk)
        return self.merge(all_i, all_s, min(int(k), total))
