#!/usr/bin/env python3
"""Where does a resident-plan loop on the DEFAULT stream lose time (epoch_bench: 1.17 ms per step against 0.85 on the
training stream, while bench.py with SVAE_TRAIN_HIPRIO=0 runs 0.856 on the default stream)?  The same loop at several
points of one process: before any other stream exists, after the high-priority stream was created, after an epoch of
train_batched (prefetcher threads, copy stream)."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import svae_cf_amd as S
from svae_cf_amd import synth, notebook_api as N
from svae_cf_amd.batch import build_plan
from svae_cf_amd.dist import rank_batches

_, I, mu, sigma, lmax = synth.SHAPES["ml-20m"]
B, NB = 256, 32
users = synth.make_users(B * NB, I, mu, sigma, lmax, seed=98765)


class Reader:
    data = [u.tolist() for u in users]
    def _k(self): return 4


hp = dict(item_embed_size=256, rnn_size=200, hidden_size=150, latent_size=64, total_items=I, batch_size=1,
          weight_decay=5e-3, batch_log_interval=10 ** 9)
model = N.Model(hp, prec=S.PREC_BF16, seed=0)
N.make_optimizer(model, dict(hp, optimizer="adam"))
eng = model.engine
reader = Reader()
batches = rank_batches([len(u) for u in reader.data], B, np.random.default_rng(1000), 0, 1)
plans = [build_plan([np.asarray(reader.data[i]) for i in idx], 4, user_keys=idx, device="cuda", total_items=I) for idx in batches]
for p in plans:
    eng.prepare(p)


def resident():
    for p in plans[:8]:
        eng.train_step(p, 0.01, seed=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4):
        for p in plans:
            eng.train_step(p, 0.01, seed=1)
    torch.cuda.synchronize()
    return round((time.perf_counter() - t0) / (4 * len(plans)) * 1e3, 4)


out = {}
if "train_first" in sys.argv:
    N.train_batched(reader, model, hp, batch_size=B, epoch=1, seed=0, workers=2)
    torch.cuda.synchronize()
if "native" in sys.argv:
    from svae_cf_amd.batch import build_plan_native, UserCSR
    csr = UserCSR(reader.data)
    plans = [build_plan_native(csr, idx, 4, user_keys=idx, device="cuda", total_items=I, pin=True) for idx in batches]
    for p in plans:
        eng.prepare(p)
out["1 default, nothing else"] = resident()
if "short" in sys.argv:
    print(json.dumps(out)); sys.exit(0)
hi = eng.training_stream()
out["2 default, high-priority stream created (idle)"] = resident()
with torch.cuda.stream(hi):
    out["3 training stream"] = resident()
out["4 default again"] = resident()
N.train_batched(reader, model, hp, batch_size=B, epoch=1, seed=0, workers=2)
torch.cuda.synchronize()
out["5 default after train_batched"] = resident()
with torch.cuda.stream(hi):
    out["6 training stream after train_batched"] = resident()
out["side_streams"] = eng.side_stream_info
print(json.dumps(out, indent=1))
