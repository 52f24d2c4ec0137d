#!/bin/bash
# Rehearsal of the N>1 bench path on a ONE-GPU box, two legs:
#  1. RCCL leg: 1 rank under torch.distributed.run with SVAE_DP_REHEARSAL=1 -- process group on nccl (= RCCL),
#     the bucketed asynchronous all-reduces of GradSync issued from the backward streams, barrier, MAX/SUM
#     reductions of the timing: the exact call path of the driver's N>1 runs, at world size 1.
#  2. gloo leg: 2 ranks share the card, gradients staged through host memory: launch contract, balanced
#     sharding, matched collectives, rank-0 JSON line.  Its throughput means nothing.
set -o pipefail
export SVAE_TUNING=1      # the engine reads its SVAE_* tuning switches only with this set (engine.py, ENGINE_OPTIONS)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT" || exit 1
mkdir -p gpurun_out
SVAE_DP_REHEARSAL=1 timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 \
  --master-addr 127.0.0.1 --master-port 29516 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline \
  > gpurun_out/dp_rehearsal_rccl.json 2> gpurun_out/dp_rehearsal_rccl.err || { tail -30 gpurun_out/dp_rehearsal_rccl.err; exit 1; }
cat gpurun_out/dp_rehearsal_rccl.json
# 1b. the same leg with the collectives through the library's own RCCL wrappers (comm.hip), in stream order
SVAE_DP_REHEARSAL=1 SVAE_DP_NATIVE=1 timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 \
  --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-class \
  > gpurun_out/dp_rehearsal_rccl_native.json 2> gpurun_out/dp_rehearsal_rccl_native.err || { tail -30 gpurun_out/dp_rehearsal_rccl_native.err; exit 3; }
cat gpurun_out/dp_rehearsal_rccl_native.json
[ "$1" == "rccl-only" ] && exit 0
# (two processes on one card: on some boxes of the pool they are time-sliced with a long quantum -- 500 ms per step instead
#  of 15 -- so this leg is bounded and its failure is reported, not fatal to the caller)
echo "gloo leg: two ranks on the one card"
SVAE_DIST_BACKEND=gloo timeout -k 10 150 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
  --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 6 --warmup 2 --settle-steps 4 --nplans 4 --no-cpu-baseline --no-fp32-class --no-cold-figure \
  > gpurun_out/dp_rehearsal.json 2> gpurun_out/dp_rehearsal.err || { tail -5 gpurun_out/dp_rehearsal.err; echo "gloo leg failed or timed out"; exit 2; }
cat gpurun_out/dp_rehearsal.json
