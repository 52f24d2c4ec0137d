#!/bin/bash
# A/B of the HIP runtime's hardware-queue count (GPU_MAX_HW_QUEUES, 4 on the box) on the three call paths of the bench:
# plain, torch.distributed RCCL at world size 1, the library's RCCL wrappers at world size 1.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT" || exit 1
mkdir -p gpurun_out
port=29530
for q in 4 8 4 8; do
  for leg in plain torch native; do
    port=$((port+1))
    case $leg in
      plain)  GPU_MAX_HW_QUEUES=$q timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-class --no-cold-figure --repeats 3 > gpurun_out/hwq.json 2> gpurun_out/hwq.err ;;
      torch)  GPU_MAX_HW_QUEUES=$q SVAE_DP_REHEARSAL=1 timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $port bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-class --no-cold-figure --repeats 3 > gpurun_out/hwq.json 2> gpurun_out/hwq.err ;;
      native) GPU_MAX_HW_QUEUES=$q SVAE_DP_REHEARSAL=1 SVAE_DP_NATIVE=1 timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $port bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-class --no-cold-figure --repeats 3 > gpurun_out/hwq.json 2> gpurun_out/hwq.err ;;
    esac
    [ $? -eq 0 ] || { tail -20 gpurun_out/hwq.err; exit 1; }
    python3 - "$q" "$leg" <<'PY'
import json, sys
d = json.loads([l for l in open("gpurun_out/hwq.json") if l.startswith("{")][-1])
print("queues %s %-6s ms/step %.4f (regions min %.4f median %.4f)  sides %s" % (sys.argv[1], sys.argv[2], d["ms_per_step"],
      d["ms_per_step_regions"]["min"], d["ms_per_step_regions"]["median"],
      [(s.get("tried"), s.get("held_frac")) for s in d["config"]["side_stream"]]))
PY
  done
done
