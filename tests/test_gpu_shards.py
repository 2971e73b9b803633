"""GPU: the multi-GPU data path with G LOGICAL shards on one device ("emulation"):
range-partitioned batches with index bases, per-shard top-k on the GEMM engine with device-resident queries
(innr_batch_knn_dev), and the HIP merge kernel (innr_merge_topk_dev) -- against the oracle on the whole corpus."""
from __future__ import annotations

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle


@pytest.mark.parametrize("metric_name,g,n_total,k", [("dot", 4, 40_000, 10), ("cos", 3, 10_001, 33), ("l2", 2, 5_000, 7),
                                                      ("dot", 8, 20, 5),
                                                      # 800 candidates per query: the merge kernel's LDS staging at the size the
                                                      # reference's k = 100 bench gives 8 ranks; 4000: beyond it (global re-reads)
                                                      ("dot", 8, 30_000, 100), ("l2", 8, 16_000, 500)])
def test_logical_shards_merge_equals_whole_corpus(metric_name, g, n_total, k):
    import torch
    import innr_amd
    from innr_amd import batch as B
    from innr_amd.dist import INVALID_INDEX, _gpu_local_search, _gpu_merge, shard_range

    dim, nq = 48, 70
    metric = {"dot": innr_amd.METRIC_DOT, "cos": innr_amd.METRIC_COSINE, "l2": innr_amd.METRIC_L2SQ}[metric_name]
    ctx = innr_amd.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    dev = torch.device("cuda", 0)
    queries = oracle.generate_uniform(nq, dim, 99)
    q_dev = torch.from_numpy(queries).to(dev)
    kin = min(k, max(shard_range(n_total, g, r)[1] for r in range(g)))
    all_i = torch.full((g, nq, kin), INVALID_INDEX, dtype=torch.int64, device=dev)
    all_s = torch.zeros((g, nq, kin), dtype=torch.float32, device=dev)
    shards = []
    for r in range(g):
        start, count = shard_range(n_total, g, r)
        vb = B.VerticalBatch.generate(count, dim, seed=7, row0=start, ctx=ctx)
        vb.set_index_base(start)
        shards.append(vb)
        idx, sc = _gpu_local_search(vb, metric, innr_amd.KNN_AUTO)(q_dev, k)
        all_i[r, :, :idx.shape[1]] = idx
        all_s[r, :, :sc.shape[1]] = sc
    kout = min(k, n_total)
    out_i, out_s = _gpu_merge(ctx, metric)(all_i, all_s, kout)
    # the exchange format of the library's own sharded call (innr_topk_pack_dev -> [all-gather] -> innr_merge_blocks_dev):
    # one block of 2 + Q*k words per shard, 8 bytes per candidate; here the "gather" is a torch.stack
    from innr_amd.dist import gpu_merge_blocks, gpu_pack_block
    blocks = []
    for r in range(g):
        start, count = shard_range(n_total, g, r)
        kk = min(k, count)
        blocks.append(gpu_pack_block(ctx, all_i[r, :, :kk].contiguous(), all_s[r, :, :kk].contiguous(), start, count, k))
    b_i, b_s = gpu_merge_blocks(ctx, metric, torch.stack(blocks), nq, k)
    torch.cuda.synchronize()
    assert b_i.shape == (nq, kout) and torch.equal(b_i, out_i) and torch.equal(b_s.view(torch.int32), out_s.view(torch.int32))
    # a rank that failed its local search gathers a block whose header says so (word 1 all ones, word 0 = -status): the merge
    # then returns an error naming it on every rank instead of a result (include/innr_hip.h, failure semantics)
    bad = torch.stack(blocks).clone()
    bad[g - 1, 0], bad[g - 1, 1] = 3, -1
    with pytest.raises(innr_amd.InnrError) as ei:
        gpu_merge_blocks(ctx, metric, bad, nq, k)
    assert ei.value.status == innr_amd._lib.E_RCCL and f"rank {g - 1}" in str(ei.value) and "status -3" in str(ei.value)
    out_i, out_s = out_i.cpu().numpy(), out_s.cpu().numpy()
    data = oracle.from_rows(oracle.generate_uniform(n_total, dim, 7))
    ofn = {"dot": oracle.batch_knn_dot, "cos": oracle.batch_knn_cosine, "l2": oracle.batch_knn}[metric_name]
    for j in range(nq):
        oi, os_ = ofn(queries[j], data, k)
        assert out_i[j].tolist() == oi.astype(np.int64).tolist(), (j, out_i[j], oi)
        assert np.array_equal(out_s[j].view(np.uint32), os_.view(np.uint32))
    for vb in shards:
        vb.close()
    ctx.close()


def test_logical_shards_u8_and_maxsim():
    """the same merge contract for the other two sharded paths: scalar::batch_knn_u8 and maxsim top-k"""
    import torch
    import innr_amd
    from innr_amd import maxsim as M
    from innr_amd import scalar as S
    from innr_amd.dist import (INVALID_INDEX, _gpu_local_search_docs, _gpu_local_search_u8, _gpu_merge, shard_range)

    ctx = innr_amd.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    dev = torch.device("cuda", 0)
    merge = _gpu_merge(ctx, innr_amd.METRIC_DOT)

    # u8: 3 shards of a 30K x 64 code corpus
    g, n_total, dim, nq, k = 3, 30_000, 64, 40, 20
    p = S.QuantizationParams.from_range(-1.0, 1.0)
    queries = oracle.generate_uniform(nq, dim, 5)
    q_dev = torch.from_numpy(queries).to(dev)
    all_i = torch.full((g, nq, k), INVALID_INDEX, dtype=torch.int64, device=dev)
    all_s = torch.zeros((g, nq, k), dtype=torch.float32, device=dev)
    keep = []
    for r in range(g):
        start, count = shard_range(n_total, g, r)
        qc = S.QuantizedCorpus.generate(count, dim, p, seed=4, row0=start, ctx=ctx)
        qc.set_index_base(start)
        keep.append(qc)
        idx, sc = _gpu_local_search_u8(qc, innr_amd.KNN_AUTO)(q_dev, k)
        all_i[r, :, :idx.shape[1]] = idx
        all_s[r, :, :sc.shape[1]] = sc
    out_i, out_s = merge(all_i, all_s, k)
    torch.cuda.synchronize()
    codes = oracle.quantize_u8(oracle.generate_uniform(n_total, dim, 4), oracle.QParams(p.alpha, p.offset))
    for j in range(nq):
        oi, os_ = oracle.batch_knn_u8(queries[j], codes, oracle.QParams(p.alpha, p.offset), k)
        assert out_i[j].cpu().numpy().tolist() == oi.astype(np.int64).tolist()
        assert np.array_equal(out_s[j].cpu().numpy().view(np.uint32), os_.view(np.uint32))

    # maxsim: 4 shards of 6000 documents x 32 tokens x 64 dims, one 8-token query
    g, ndocs, T, dim, Tq, k = 4, 6000, 32, 64, 8, 15
    whole = M.DocumentCorpus.generate(ndocs, T, dim, seed=9, ctx=ctx)
    rq = oracle.generate_uniform(Tq, dim, 77)
    q = (rq / np.sqrt((rq.astype(np.float64) ** 2).sum(axis=1, keepdims=True))).astype(np.float32)
    want_i, want_s = whole.topk(q, k, engine=innr_amd.KNN_EXACT)
    q_dev = torch.from_numpy(q).to(dev)
    all_i = torch.full((g, 1, k), INVALID_INDEX, dtype=torch.int64, device=dev)
    all_s = torch.zeros((g, 1, k), dtype=torch.float32, device=dev)
    for r in range(g):
        start, count = shard_range(ndocs, g, r)
        dc = M.DocumentCorpus.generate(count, T, dim, seed=9, row0=start * T, ctx=ctx)
        dc.set_index_base(start)
        keep.append(dc)
        idx, sc = _gpu_local_search_docs(dc, False, innr_amd.KNN_AUTO)(q_dev, k)
        all_i[r, :, :idx.shape[1]] = idx
        all_s[r, :, :sc.shape[1]] = sc
    out_i, out_s = merge(all_i, all_s, k)
    torch.cuda.synchronize()
    assert out_i[0].cpu().numpy().tolist() == want_i.astype(np.int64).tolist()
    assert np.array_equal(out_s[0].cpu().numpy().view(np.uint32), want_s.view(np.uint32))
    for o in keep:
        o.close()
    whole.close()
    ctx.close()


def test_rccl_exchange_behind_the_abi_world_1():
    """The library's own exchange step on a real RCCL communicator: innr_comm_unique_id / innr_comm_create (world = 1:
    the one-GPU box; RCCL refuses two ranks per device) and innr_sharded_knn_dev = local search + pack + ncclAllGather +
    merge -- for an f32 shard and a u8 code shard, against the plain local search and the oracle."""
    import torch
    import innr_amd
    from innr_amd import batch as B
    from innr_amd import scalar as S
    from innr_amd.dist import Comm, ShardedKnn, _gpu_local_search

    ctx = innr_amd.Context(0)
    dev = torch.device("cuda", 0)
    comm = Comm(ctx, 0, 1, Comm.unique_id())
    n, dim, nq, k = 30_000, 40, 37, 12
    queries = oracle.generate_uniform(nq, dim, 99)
    q_dev = torch.from_numpy(queries).to(dev)
    vb = B.VerticalBatch.generate(n, dim, seed=7, row0=0, ctx=ctx)
    data = oracle.from_rows(oracle.generate_uniform(n, dim, 7))
    for metric, ofn in ((innr_amd.METRIC_DOT, oracle.batch_knn_dot), (innr_amd.METRIC_COSINE, oracle.batch_knn_cosine),
                        (innr_amd.METRIC_L2SQ, oracle.batch_knn)):
        sk = ShardedKnn(n, rank=0, world=1, comm=comm)
        sk.attach_gpu_batch(vb, metric)
        st = innr_amd.KnnStats()
        idx, sc = sk.search(q_dev, k, st)
        torch.cuda.synchronize()
        assert idx.shape == (nq, k) and st.total_ms > 0
        for j in range(nq):
            oi, os_ = ofn(queries[j], data, k)
            assert idx[j].cpu().numpy().tolist() == oi.astype(np.int64).tolist()
            assert np.array_equal(sc[j].cpu().numpy().view(np.uint32), os_.view(np.uint32))
    # k larger than the shard: k' = min(k, total)
    small = B.VerticalBatch.generate(5, dim, seed=7, row0=0, ctx=ctx)
    sk = ShardedKnn(5, rank=0, world=1, comm=comm)
    sk.attach_gpu_batch(small, innr_amd.METRIC_DOT)
    idx, sc = sk.search(q_dev, 9)
    assert idx.shape == (nq, 5) and sorted(idx[0].cpu().tolist()) == [0, 1, 2, 3, 4]
    # u8 codes
    p = S.QuantizationParams.from_range(-1.0, 1.0)
    qc = S.QuantizedCorpus.generate(n, dim, p, seed=4, row0=0, ctx=ctx)
    sk = ShardedKnn(n, rank=0, world=1, comm=comm)
    sk.attach_gpu_u8(qc)
    idx, sc = sk.search(q_dev, k)
    torch.cuda.synchronize()
    codes = oracle.quantize_u8(oracle.generate_uniform(n, dim, 4), oracle.QParams(p.alpha, p.offset))
    for j in range(0, nq, 5):
        oi, os_ = oracle.batch_knn_u8(queries[j], codes, oracle.QParams(p.alpha, p.offset), k)
        assert idx[j].cpu().numpy().tolist() == oi.astype(np.int64).tolist()
        assert np.array_equal(sc[j].cpu().numpy().view(np.uint32), os_.view(np.uint32))
    # a dimension mismatch is reported like the local call's (the rank still gathers: an error block)
    with pytest.raises(innr_amd.InnrPanic):
        sk.search(torch.zeros((2, dim + 1), device=dev), k)
    # a rank-local failure of the local search (here on request): this rank returns its own status -- after taking part in the
    # exchange -- and the communicator stays usable
    with ctx.option("fail_local_search", 1):
        with pytest.raises(innr_amd.InnrError) as ei:
            sk.search(q_dev, k)
        assert ei.value.status == innr_amd._lib.E_HIP and "on request" in str(ei.value)
    idx2, sc2 = sk.search(q_dev, k)
    assert torch.equal(idx2, idx) and torch.equal(sc2.view(torch.int32), sc.view(torch.int32))
    # maxsim through the same exchange (innr_sharded_maxsim): a document shard, one query
    from innr_amd import maxsim as M
    ndocs, T, mdim, Tq, mk = 5000, 32, 64, 8, 15
    dc = M.DocumentCorpus.generate(ndocs, T, mdim, seed=9, ctx=ctx)
    rq = oracle.generate_uniform(Tq, mdim, 77)
    mq = (rq / np.sqrt((rq.astype(np.float64) ** 2).sum(axis=1, keepdims=True))).astype(np.float32)
    for cosine in (False, True):
        want_i, want_s = dc.topk(mq, mk, cosine=cosine, engine=innr_amd.KNN_EXACT)
        skd = ShardedKnn(ndocs, rank=0, world=1, comm=comm)
        skd.attach_gpu_docs(dc, cosine=cosine)
        mi, ms = skd.search(torch.from_numpy(mq).to(dev), mk)
        assert mi.shape == (1, mk) and mi[0].cpu().numpy().tolist() == want_i.astype(np.int64).tolist()
        assert np.array_equal(ms[0].cpu().numpy().view(np.uint32), want_s.view(np.uint32))
    mi, ms = skd.search(torch.from_numpy(mq).to(dev), ndocs + 7)  # k beyond the corpus: k' = ndocs
    assert mi.shape == (1, ndocs)
    comm.close()
    for o in (vb, small, qc, dc):
        o.close()
    ctx.close()
