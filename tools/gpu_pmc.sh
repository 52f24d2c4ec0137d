#!/bin/bash
# Hardware-counter passes over a short bench run, one rocprofv3 run per pass (counters are not split over
# passes by the tool; per-block slot limits: 8 SQ, 4 TCC (FETCH_SIZE = 3, WRITE_SIZE = 2), 2 GRBM).
#   gpurun --timeout 900 -- 'bash tools/gpu_pmc.sh <tag> [kernel-name-filter]'
# Output: gpurun_out/pmc_<tag>/<pass>/...counter_collection.csv and gpurun_out/pmc_<tag>.txt (summary).
set -o pipefail
TAG=${1:-x}
FILT=${2:-proj_kernel}   # "kernel" = every kernel
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
declare -A PASS
PASS[sq1]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
PASS[sq2]="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES"
PASS[fetch]="FETCH_SIZE TCC_HIT_sum"
PASS[write]="WRITE_SIZE TCC_MISS_sum TCC_REQ_sum"
: > "$ROOT/gpurun_out/pmc_$TAG.txt"
for p in sq1 sq2 fetch write; do
    timeout -s KILL 150 rocprofv3 --pmc ${PASS[$p]} --output-format csv -d "$OUT/$p" -- python3 "$ROOT/bench.py" --steps 4 --warmup 2 --no-cpu-baseline --no-fp32-class --no-cold-figure > "$OUT/$p.log" 2>&1
    rc=$?
    if [ $rc -ne 0 ]; then echo "pass $p failed rc=$rc"; tail -5 "$OUT/$p.log"; exit 1; fi
    f=$(ls "$OUT/$p"/*/*counter_collection.csv | tail -1)
    echo "== pass $p" >> "$ROOT/gpurun_out/pmc_$TAG.txt"
    python3 "$ROOT/tools/pmc_summary.py" "$f" "$FILT" >> "$ROOT/gpurun_out/pmc_$TAG.txt"
done
cat "$ROOT/gpurun_out/pmc_$TAG.txt"
