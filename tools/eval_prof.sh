#!/bin/bash
# kernel trace of the scoring benchmark -> gpurun_out/prof_eval
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf "$ROOT/gpurun_out/prof_eval"
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/prof_eval" -- python3 "$ROOT/tools/eval_bench.py" > "$ROOT/gpurun_out/eval_bench.log" 2>&1 || { tail -20 "$ROOT/gpurun_out/eval_bench.log"; exit 3; }
tail -1 "$ROOT/gpurun_out/eval_bench.log"
