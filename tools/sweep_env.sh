#!/bin/bash
# One environment switch over several values on one box, N rounds: bash tools/sweep_env.sh VAR N v1 v2 v3 ...
set -o pipefail
export SVAE_TUNING=1      # the engine reads its SVAE_* tuning switches only with this set (engine.py, ENGINE_OPTIONS)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT" || exit 1
VAR=$1; N=$2; shift 2
mkdir -p gpurun_out
for i in $(seq 1 $N); do
  for v in "$@"; do
    env "$VAR=$v" timeout -k 10 200 python bench.py --no-cpu-baseline --repeats 3 > gpurun_out/sweep_$v.json 2> gpurun_out/sweep_$v.err || { tail -20 gpurun_out/sweep_$v.err; exit 1; }
    python3 - <<PY
import json
d=json.load(open("gpurun_out/sweep_$v.json"))
print("$i $VAR=$v ms/step %.4f (regions median %.4f) wgrad %.3f" % (d["ms_per_step"], d["ms_per_step_regions"]["median"], d["kernel_ms"].get("wgrad", 0)))
PY
  done
done
