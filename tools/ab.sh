#!/bin/bash
# A/B of environment knobs: tools/ab.sh "VAR=a" "VAR=b" ... ; each runs bench.py (no CPU baseline)
cd ${GRAFT_REPO_ROOT:-.}
for kv in "$@"; do
  env_kv=$(echo $kv | tr ',' ' ')
  out=$(export $env_kv; timeout -k 10 120 python bench.py --no-cpu-baseline --steps 40 --warmup 8 2>/dev/null) || exit 1
  echo "$out" | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$kv', round(d['value']), round(d['ms_per_step'],4), round(d.get('host_enqueue_ms_per_step',0),3), {k: round(v,4) for k,v in d['kernel_ms'].items()}, d['config'].get('side_stream'))"
done
