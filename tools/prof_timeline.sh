#!/bin/bash
# kernel trace of a short bench run -> gpurun_out/prof_<tag>; print with tools/timeline.py
TAG=${1:-t}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf "$ROOT/gpurun_out/prof_$TAG"
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/prof_$TAG" -- python3 "$ROOT/bench.py" --steps 20 --warmup 17 --no-cpu-baseline --no-kernel-times --no-fp32-class --no-cold-figure > "$ROOT/gpurun_out/prof_$TAG.log" 2>&1 || { tail -20 "$ROOT/gpurun_out/prof_$TAG.log"; exit 3; }
echo ok
