#!/bin/bash
export SVAE_TUNING=1      # the engine reads its SVAE_* tuning switches only with this set (engine.py, ENGINE_OPTIONS)
# experiment: forced split counts of the weight-gradient sweep
cd ${GRAFT_REPO_ROOT:-.}
for q in 3 4 6 8 9 12 16; do
  SVAE_DW_Q=$q timeout -k 10 120 python bench.py --no-cpu-baseline --steps 32 --warmup 8 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('q=$q', round(d['value']), {k: round(v,4) for k,v in d['kernel_ms'].items()})" || exit 1
done
for c in 192 224 256; do
  SVAE_DW_CUS=$c timeout -k 10 120 python bench.py --no-cpu-baseline --steps 32 --warmup 8 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('cus=$c', round(d['value']), {k: round(v,4) for k,v in d['kernel_ms'].items()})" || exit 1
done
