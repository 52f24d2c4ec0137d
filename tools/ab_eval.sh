#!/bin/bash
# A/B of the scoring path in one call: tools/eval_bench.py with the in-tree library and with svae_cf_amd/csrc/exp/lib_prev.so
# (a library built from another commit: git archive <commit> svae_cf_amd/csrc include | tar -x -C /tmp/x; make there; copy), twice each.
#   gpurun --timeout 900 -- 'bash tools/ab_eval.sh'
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT" || exit 1
for i in 1 2; do
  for tag in new prev; do
    if [ $tag = prev ]; then export SVAE_HIP_LIB=$ROOT/svae_cf_amd/csrc/exp/lib_prev.so; else unset SVAE_HIP_LIB; fi
    timeout -k 10 200 python tools/eval_bench.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d['scoring_session_two_lanes']
print('$i $tag call-by-call %.4f two-lane %.4f three-lane@192 %.4f' % (d['ms_per_batch'], t['ms_per_batch_by_sweep_workgroups']['192'], t['three_lanes_ms_per_batch_by_sweep_workgroups']['192']))" || exit 1
  done
done
