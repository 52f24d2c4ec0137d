#!/bin/bash
# The round's secondary records in one GPU call (each step under its own time limit, chain stops at a failure):
#   gpurun --timeout 1100 -- 'bash tools/round_records.sh <tag>'
set -o pipefail
export SVAE_TUNING=1      # the engine reads its SVAE_* tuning switches only with this set (engine.py, ENGINE_OPTIONS)
TAG=${1:-x}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out
cd "$ROOT" || exit 1
mkdir -p "$O"
bash tools/dp_rehearsal.sh rccl-only > "$O/dp_rehearsal_$TAG.log" 2>&1 || { tail -20 "$O/dp_rehearsal_$TAG.log"; exit 1; }
cp "$O/dp_rehearsal_rccl.json" "$O/${TAG}_dp_rehearsal_rccl_world1.json.log"; cp "$O/dp_rehearsal_rccl_native.json" "$O/${TAG}_dp_rehearsal_rccl_native_world1.json.log"
echo "rccl legs done"
echo "epoch bench"; timeout -k 10 300 python tools/epoch_bench.py 2 > "$O/${TAG}_epoch_bench.json" 2> "$O/epoch.err" || { tail -20 "$O/epoch.err"; exit 3; }
timeout -k 10 300 python tools/b1_bench.py 1000 > "$O/${TAG}_b1_train.json" 2> "$O/b1.err" || { tail -20 "$O/b1.err"; exit 4; }
echo "large"; timeout -k 10 300 python bench.py --shape large --steps 8 --warmup 2 --nplans 4 --repeats 2 --no-cpu-baseline > "$O/${TAG}_bench_large.json.log" 2> "$O/large.err" || { tail -20 "$O/large.err"; exit 5; }
timeout -k 10 300 python bench.py --shape large --batch 1024 --steps 6 --warmup 2 --nplans 3 --repeats 1 --no-cpu-baseline > "$O/${TAG}_bench_large_B1024.json.log" 2> "$O/large2.err" || { tail -20 "$O/large2.err"; exit 6; }
echo "eval"; timeout -k 10 300 python tools/eval_bench.py > "$O/${TAG}_eval_bench.json.log" 2> "$O/eval.err" || { tail -20 "$O/eval.err"; exit 7; }
timeout -k 10 200 python bench.py --shape ml-1m --no-cpu-baseline --repeats 2 > "$O/${TAG}_bench_bf16_ml1m.json.log" 2> "$O/ml1m.err" || { tail -20 "$O/ml1m.err"; exit 8; }
timeout -k 10 200 python bench.py --shape netflix --no-cpu-baseline --repeats 2 > "$O/${TAG}_bench_bf16_netflix.json.log" 2> "$O/nf.err" || { tail -20 "$O/nf.err"; exit 9; }
echo "precision"; timeout -k 10 300 python tools/precision_report.py > "$O/${TAG}_precision_report.txt" 2> "$O/prec.err" || { tail -20 "$O/prec.err"; exit 10; }
timeout -k 10 300 python bench.py --eval-leg > "$O/${TAG}_bench_bf16_full.json.log" 2> "$O/full.err" || { tail -20 "$O/full.err"; exit 11; }
echo "single-process records ok"
# Two ranks sharing the one card (gloo), last and not fatal: on some boxes two processes are time-sliced on the GPU with a
# long quantum (500 ms per step instead of 15) -- see tools/dp_rehearsal.sh
SVAE_DIST_BACKEND=gloo timeout -k 10 150 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
  bench.py --gpus 2 --steps 6 --warmup 2 --settle-steps 4 --nplans 4 --no-cpu-baseline --no-fp32-class --no-cold-figure > "$O/${TAG}_dp_rehearsal_gloo2.json.log" 2> "$O/dp_gloo.err" \
  || echo "gloo torchrun leg failed or timed out"
echo "gloo torchrun leg done"
SVAE_DIST_BACKEND=gloo timeout -k 10 150 python bench.py --gpus 2 --steps 6 --warmup 2 --settle-steps 4 --nplans 4 --no-cpu-baseline --no-fp32-class --no-cold-figure --repeats 1 > "$O/${TAG}_bench_gpus2_spawn_gloo.json.log" 2> "$O/spawn.err" \
  || echo "gloo spawn leg failed or timed out"
echo records ok
