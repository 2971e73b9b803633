"""GPU: the multi-GPU data path with G LOGICAL shards on one device ("emulation"):
range-partitioned batches with index bases, per-shard top-k on the GEMM engine with device-resident queries
(innr_batch_knn_dev), and the HIP merge kernel (innr_merge_topk_dev) -- against the oracle on the whole corpus."""
from __future__ import annotations

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle


@pytest.mark.parametrize("metric_name,g,n_total,k", [("dot", 4, 40_000, 10), ("cos", 3, 10_001, 33), ("l2", 2, 5_000, 7),
                                                      ("dot", 8, 20, 5)])
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
    torch.cuda.synchronize()
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
