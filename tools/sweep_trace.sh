#!/bin/bash
# Per-launch durations of the two projection sweeps inside the step, from a rocprofv3 kernel trace of the bench, for
# SVAE_FLAT_SCHED = 1 and 0 (one box, one call).  Usage (via gpurun): bash tools/sweep_trace.sh <tag>
set -o pipefail
export SVAE_TUNING=1      # the engine reads its SVAE_* tuning switches only with this set (engine.py, ENGINE_OPTIONS)
TAG=${1:-x}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for f in ${FLATS:-1 0}; do
  export SVAE_FLAT_SCHED=$f
  rm -rf "$OUT/trace_${TAG}_$f"
  timeout -k 10 240 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace_${TAG}_$f" -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-fp32-class --no-kernel-times --repeats 1 > "$OUT/trace_${TAG}_$f.log" 2>&1 || { tail -20 "$OUT/trace_${TAG}_$f.log"; exit 3; }
  python3 - "$(ls $OUT/trace_${TAG}_$f/*/*_kernel_trace.csv | tail -1)" "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
g = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if "proj_kernel" in n or "proj_fdh_finish" in n or "proj_dw_finish" in n:
        key = (n.split("(")[0][-40:], int(r["Grid_Size_X"] if "Grid_Size_X" in r else r["Grid_Size"]) )
        g[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("SVAE_FLAT_SCHED=%s" % sys.argv[2])
tot = collections.defaultdict(float)
for (n, grid), v in sorted(g.items()):
    v = v[len(v) // 3:]                       # steady state: drop the first third (settle / warm-up)
    print("  %-42s grid %7d  launches %4d  mean %.1f us  min %.1f  max %.1f" % (n, grid, len(v), sum(v) / len(v), min(v), max(v)))
PY
done
