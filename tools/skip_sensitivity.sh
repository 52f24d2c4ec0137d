#!/bin/bash
# How much of the step does each kernel hold?  bench.py with one entry point at a time left out (tools/bench_skip.py -> _lib.install_skip):
# the drop in ms per step is that kernel's share of the critical path (results are wrong by construction; timing only).
# Usage (via gpurun): bash tools/skip_sensitivity.sh
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT" || exit 1
mkdir -p gpurun_out
: > gpurun_out/skip.txt
for s in ${SKIPS:-none svae_gru_fwd_range svae_gru_bwd_bf16out svae_rowgemm_bf16x svae_embed_bwd svae_rowgemm_bf16x,svae_embed_bwd svae_wgrad_cols_bf16dy svae_wgrad_cols svae_proj_nll_bwd_dw svae_proj_nll_fwd_dh svae_mlp_fwd_prep svae_mlp_bwd_prec svae_adam_step svae_rowgemm_ld_prec svae_embed_fwd svae_reduce_slabs svae_loss_reduce svae_mlp_prep_prec,svae_philox_normal,svae_transpose,svae_zero none}; do
  timeout -k 10 120 python tools/bench_skip.py $s --no-cpu-baseline --repeats 3 --no-kernel-times > gpurun_out/skip.json 2> gpurun_out/skip.err || { echo "$s failed"; tail -5 gpurun_out/skip.err; continue; }
  python3 -c "
import json; d=json.load(open('gpurun_out/skip.json')); print('%-28s ms/step %.4f (median of regions %.4f)' % ('$s', d['ms_per_step'], d['ms_per_step_regions']['median']))" | tee -a gpurun_out/skip.txt
done
