#!/usr/bin/env python3
"""End-to-end epoch throughput through the product's own entry point (notebook_api.train_batched: plans built from
the reader's per-user lists by the prefetcher's worker threads, uploaded, trained on) beside the kernel-pipeline
figure of bench.py (plans pre-built and resident).  ML-20M shape, 256 batches of 256 users.

    python tools/epoch_bench.py [workers]
"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import svae_cf_amd as S
from svae_cf_amd import synth, notebook_api as N
from svae_cf_amd.batch import build_plan
import svae_cf_amd.batch as _batch
if os.environ.get('SVAE_CONSUMER_UPLOADS'):
    _batch.CONSUMER_UPLOADS = os.environ['SVAE_CONSUMER_UPLOADS'] == '1'

workers = int(sys.argv[1]) if len(sys.argv) > 1 else 2
if os.environ.get("SVAE_SWITCH_INTERVAL"):
    sys.setswitchinterval(float(os.environ["SVAE_SWITCH_INTERVAL"]))
_, I, mu, sigma, lmax = synth.SHAPES["ml-20m"]
B, NB = 256, int(os.environ.get('SVAE_EPOCH_BATCHES', '256'))
users = synth.make_users(B * NB, I, mu, sigma, lmax, seed=98765)


class Reader:                       # the part of notebook_api.DataReader train_batched reads
    data = [u.tolist() for u in users]           # per-user Python lists, as DataReader holds them (nb:439-483)
    def _k(self): return 4


hp = dict(item_embed_size=256, rnn_size=200, hidden_size=150, latent_size=64, total_items=I, batch_size=1,
          weight_decay=5e-3, batch_log_interval=10 ** 9)
model = N.Model(hp, prec=S.PREC_BF16, seed=0)
N.make_optimizer(model, dict(hp, optimizer="adam"))
reader = Reader()
out = {}
for tag, w in (("warmup", workers), ("epoch", workers)):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = N.train_batched(reader, model, hp, batch_size=B, epoch=1, seed=0, workers=w)
    torch.cuda.synchronize()
    out[tag] = time.perf_counter() - t0
    out[tag + "_wait"] = N.train_batched.last_wait_s
# the kernel pipeline alone on the same batches (plans resident), and the plan builder alone (one thread)
from svae_cf_amd.dist import rank_batches
batches = rank_batches([len(u) for u in reader.data], B, np.random.default_rng(0 + 1000 * 1), 0, 1)
t0 = time.perf_counter()
plans = [build_plan([np.asarray(reader.data[i]) for i in idx], 4, user_keys=idx, device="cuda", total_items=I) for idx in batches]
torch.cuda.synchronize()
t_build = time.perf_counter() - t0
from svae_cf_amd.batch import build_plan_native, UserCSR
csr = UserCSR(reader.data)
t0 = time.perf_counter()
plans = [build_plan_native(csr, idx, 4, user_keys=idx, device="cuda", total_items=I, pin=True) for idx in batches]
torch.cuda.synchronize()
t_build_native = time.perf_counter() - t0
eng = model.engine
for p in plans:
    eng.prepare(p)
def resident():
    for p in plans[:8]:
        eng.train_step(p, 0.01, seed=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for p in plans:
        eng.train_step(p, 0.01, seed=1)
    torch.cuda.synchronize()
    return time.perf_counter() - t0


first_pass = resident()                          # first touch of these plans (uploads waited for, workspace growth)
t_res_default = min(resident(), resident())      # on the default stream
with torch.cuda.stream(eng.training_stream()):   # as train_batched runs its loop (and as bench.py does)
    t_res = min(resident(), resident())
nu = B * n
print(json.dumps({"metric": "end-to-end train sequences/s through train_batched (ML-20M shape)", "value": nu / out["epoch"],
                  "batches": n, "workers": workers, "ms_per_step_end_to_end": out["epoch"] / n * 1e3,
                  "resident_plans_seq_per_s": nu / t_res, "ms_per_step_resident": t_res / n * 1e3,
                  "ms_per_step_resident_default_stream": t_res_default / n * 1e3,
                  "ms_per_step_resident_first_pass": first_pass / n * 1e3,
                  "ratio": (nu / out["epoch"]) / (nu / t_res),
                  "build_plan_native_ms_per_batch_one_thread_incl_upload": t_build_native / n * 1e3,
                  "build_plan_ms_per_batch_one_thread_incl_upload": t_build / n * 1e3,
                  "side_streams": eng.side_stream_info, "first_epoch_s": out["warmup"], "loop_blocked_on_plans_ms_per_step": out["epoch_wait"] / n * 1e3}))
