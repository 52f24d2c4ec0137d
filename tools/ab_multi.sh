#!/bin/bash
# Alternating A/B/C.. of environment settings on one box: bash tools/ab_multi.sh N "VAR=a VAR2=b" "VAR=c" ... -- [bench args]
set -o pipefail
export SVAE_TUNING=1      # the engine reads its SVAE_* tuning switches only with this set (engine.py, ENGINE_OPTIONS)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT" || exit 1
N=$1; shift
CFG=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do CFG+=("$1"); shift; done
[ "$1" == "--" ] && shift
mkdir -p gpurun_out
for i in $(seq 1 $N); do
  k=0
  for c in "${CFG[@]}"; do
    k=$((k+1))
    env $c timeout -k 10 200 python bench.py --no-cpu-baseline --no-fp32-class --repeats 3 "$@" > gpurun_out/abm_$k.json 2> gpurun_out/abm_$k.err || { tail -20 gpurun_out/abm_$k.err; exit 1; }
    python3 - "$i" "$c" gpurun_out/abm_$k.json <<'PY'
import json, sys
d = json.load(open(sys.argv[3]))
k = d.get("kernel_ms", {})
print("%s [%s] seq/s %.0f  ms/step %.4f (median of regions %.4f)  fdh %.3f dw %.3f loss %r" % (
    sys.argv[1], sys.argv[2], d["value"], d["ms_per_step"], d["ms_per_step_regions"]["median"],
    k.get("proj_fdh_fused", 0), k.get("proj_dw_fused", 0), d["config"].get("final_loss")))
PY
  done
done
