#!/bin/bash
# Rehearsal of the N > 1 bench path on a ONE-GPU box: two ranks, both on device 0, gloo instead of RCCL (RCCL refuses two
# ranks on one GPU). Everything but the collective's transport is the real thing: torchrun launch, per-rank shard
# generation with index bases, device-resident search, gather of the per-shard top-k, HIP merge, barrier/MAX timing.
#   /usr/local/graft/bin/gpurun -- 'bash tools/rehearse_2rank.sh'
set -e -o pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
INNR_BENCH_ALL_ON_DEVICE0=1 INNR_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
    --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --steps 2 --warmup 1 --n-per-gpu "${1:-1000000}" \
    --no-cpu-baseline 2>/dev/null | tail -1
