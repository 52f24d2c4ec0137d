#!/bin/bash
# Rehearsal of the N>1 bench path on a ONE-GPU box: 2 ranks share the card, gradients all-reduced through gloo.
# Checks the launch contract (torch.distributed.run env), balanced sharding, GradSync and the rank-0 JSON line;
# the throughput it prints means nothing (two processes on one GPU).
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT" || exit 1
mkdir -p gpurun_out
SVAE_DIST_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
  --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 10 --warmup 3 --nplans 4 \
  > gpurun_out/dp_rehearsal.json 2> gpurun_out/dp_rehearsal.err || { tail -30 gpurun_out/dp_rehearsal.err; exit 1; }
cat gpurun_out/dp_rehearsal.json
