#!/bin/bash
# GPU-box check: parity tests -> bench -> rocprofv3 kernel trace.  Usage (from the repo root, via gpurun):
#   gpurun --timeout 900 -- 'bash tools/gpu_check.sh <tag>'
# Every GPU step has its own time limit and the chain stops at the first failure.
set -o pipefail
TAG=${1:-x}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd "$ROOT" || exit 1
timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > "$OUT/gpu_tests_$TAG.log" 2>&1 || { tail -30 "$OUT/gpu_tests_$TAG.log"; exit 1; }
tail -3 "$OUT/gpu_tests_$TAG.log"
timeout -k 10 240 python bench.py > "$OUT/bench_$TAG.json" 2> "$OUT/bench_$TAG.err" || { tail -20 "$OUT/bench_$TAG.err"; exit 2; }
cat "$OUT/bench_$TAG.json"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_$TAG" -- python3 "$ROOT/bench.py" --no-cpu-baseline > "$OUT/prof_$TAG.log" 2>&1 || { tail -20 "$OUT/prof_$TAG.log"; exit 3; }
echo "profile ok"
# the same command as the bench line above (default steps / warm-up): per-step kernel times must agree
grep "^{\"metric" "$OUT/prof_$TAG.log" > "$OUT/prof_$TAG.bench.json"
python3 "$ROOT/tools/prof_agree.py" "$OUT/prof_$TAG.bench.json" "$(ls $OUT/prof_$TAG/*/*_kernel_stats.csv | tail -1)" > "$OUT/prof_agree_$TAG.md" && cat "$OUT/prof_agree_$TAG.md"
