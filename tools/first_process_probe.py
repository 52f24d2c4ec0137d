"""Is the first GPU process on a fresh box slower, and where?  Per-step times (synchronised) of a bench-like loop."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import svae_cf_amd as S
from svae_cf_amd import synth
I, B = 20000, 256
_, _, mu, sigma, lmax = synth.SHAPES["ml-20m"]
users = synth.make_users(16 * B, I, mu, sigma, lmax, seed=98765)
batches = synth.bucketed_batches([len(u) for u in users], B, np.random.default_rng(98765))
plans = [S.build_plan([users[i] for i in idx], 4, user_keys=idx, device="cuda", total_items=I) for idx in batches]
eng = S.SVAEEngine(I, device="cuda", prec=S.PREC_BF16)
for p in plans:
    eng.prepare(p)
for i in range(8):
    eng.train_step(plans[i % 16], 0.01, seed=1)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for i in range(48):
        eng.train_step(plans[(8 + i) % 16], 0.01, seed=1)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("rep %d: enqueue %.3f ms/step, total %.3f ms/step" % (rep, (t1 - t0) / 48 * 1e3, (t2 - t0) / 48 * 1e3))
ts = []
for i in range(32):
    t0 = time.perf_counter()
    eng.train_step(plans[i % 16], 0.01, seed=1)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print("synchronised per-step ms:", " ".join("%.2f" % t for t in ts))
