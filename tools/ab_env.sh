#!/bin/bash
# A/B of one environment switch on one box: bench.py with VAR=A and VAR=B, alternating, N rounds.
# Usage (via gpurun): bash tools/ab_env.sh SVAE_LOSS_LATE 1 0 3 [bench args]
set -o pipefail
export SVAE_TUNING=1      # the engine reads its SVAE_* tuning switches only with this set (engine.py, ENGINE_OPTIONS)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT" || exit 1
VAR=$1; A=$2; B=$3; N=${4:-2}; shift 4
mkdir -p gpurun_out
for i in $(seq 1 $N); do
  for v in "$A" "$B"; do
    env "$VAR=$v" timeout -k 10 200 python bench.py --no-cpu-baseline --repeats 3 "$@" > gpurun_out/abenv_$v.json 2> gpurun_out/abenv_$v.err || { tail -20 gpurun_out/abenv_$v.err; exit 1; }
    python3 - <<PY
import json
d=json.load(open("gpurun_out/abenv_$v.json"))
print("$i $VAR=$v seq/s %.0f  ms/step %.4f (regions %s) loss %r" % (d["value"], d["ms_per_step"], d["ms_per_step_regions"]["median"], d.get("final_loss")))
PY
  done
done
