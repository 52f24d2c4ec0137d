# time bench.py under each experimental library in svae_cf_amd/csrc/exp/ (names given as arguments)
for v in "" "$@"; do
  if [ -n "$v" ]; then export SVAE_HIP_LIB=$PWD/svae_cf_amd/csrc/exp/lib_$v.so; fi
  timeout -k 10 120 python bench.py --prec bf16 --no-cpu-baseline --steps 30 --warmup 6 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('variant=$v', round(d['value']), {k: round(v,4) for k,v in d['kernel_ms'].items()})" || exit 1
done
