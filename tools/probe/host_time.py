"""Probe: where the host's time goes per step (main thread: train_step launches; worker: plan build pieces)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import svae_cf_amd as S
from svae_cf_amd import synth, _lib
from svae_cf_amd.batch import build_plan_native, UserCSR, _PinnedPool
_, I, mu, sigma, lmax = synth.SHAPES["ml-20m"]
B, NB = 256, 32
users = synth.make_users(B * NB, I, mu, sigma, lmax, seed=5)
csr = UserCSR([u.tolist() for u in users])
eng = S.SVAEEngine(I, device="cuda", prec=S.PREC_BF16)
pool = _PinnedPool()
cs = torch.cuda.Stream()
T = dict(build=0.0, prepare=0.0)
plans = []
for b in range(NB):
    idx = np.arange(B * b, B * b + B)
    t0 = time.perf_counter()
    with torch.cuda.stream(cs):
        p = build_plan_native(csr, idx, 4, user_keys=idx, total_items=I, device=eng.device, stream=cs, host_alloc=pool.take, host_release=pool.give)
        t1 = time.perf_counter()
        eng.prepare(p)
        t2 = time.perf_counter()
    T["build"] += t1 - t0; T["prepare"] += t2 - t1
    plans.append(p)
torch.cuda.synchronize()
print({k: round(v / NB * 1e3, 3) for k, v in T.items()}, "ms per plan (host, one thread)")
for p in plans[:8]:
    eng.train_step(p, 0.01, seed=1)
torch.cuda.synchronize()
t0 = time.perf_counter()
for p in plans:
    eng.train_step(p, 0.01, seed=1)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("train_step host %.3f ms/step, with sync %.3f ms/step" % ((t1 - t0) / NB * 1e3, (t2 - t0) / NB * 1e3))
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for p in plans:
    eng.train_step(p, 0.01, seed=1)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
