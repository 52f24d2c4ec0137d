#!/bin/bash
# A/B of two builds of the kernel library on one box: bench.py with the in-tree library and with SVAE_HIP_LIB=$1,
# alternating, $2 rounds.  Usage (via gpurun): bash tools/ab.sh svae_cf_amd/csrc/exp/lib_base.so 2 [bench args]
set -o pipefail
export SVAE_TUNING=1      # the engine reads its SVAE_* tuning switches only with this set (engine.py, ENGINE_OPTIONS)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT" || exit 1
ALT=$1; N=${2:-2}; shift 2
mkdir -p gpurun_out
for i in $(seq 1 $N); do
  timeout -k 10 200 python bench.py --no-cpu-baseline --repeats 3 "$@" > gpurun_out/ab_new_$i.json 2> gpurun_out/ab_new_$i.err || { tail -20 gpurun_out/ab_new_$i.err; exit 1; }
  SVAE_HIP_LIB=$ROOT/$ALT timeout -k 10 200 python bench.py --no-cpu-baseline --repeats 3 "$@" > gpurun_out/ab_alt_$i.json 2> gpurun_out/ab_alt_$i.err || { tail -20 gpurun_out/ab_alt_$i.err; exit 2; }
  python3 - <<PY
import json
for tag in ("new","alt"):
    d=json.load(open("gpurun_out/ab_%s_$i.json"%tag))
    k=d["kernel_ms"]
    print("$i", tag, "seq/s %.0f  ms/step %.4f (regions %s)  fdh %.4f dw %.4f gru %.3f/%.3f wgrad %.3f" % (d["value"], d["ms_per_step"], d["ms_per_step_regions"]["median"], k.get("proj_fdh_fused",0), k.get("proj_dw_fused",0), k.get("gru_fwd",0), k.get("gru_bwd",0), k.get("wgrad",0)))
PY
done
