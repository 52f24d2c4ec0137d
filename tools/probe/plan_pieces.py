"""Probe: time of the pieces of build_plan_native on the GPU box (one thread)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import svae_cf_amd as S
from svae_cf_amd import synth, _lib, batch
_, I, mu, sigma, lmax = synth.SHAPES["ml-20m"]
B, NB = 256, 64
users = synth.make_users(B * NB, I, mu, sigma, lmax, seed=5)
csr = batch.UserCSR([u.tolist() for u in users])
cr, pk, ds = _lib.plan_fns()
T = {"create": 0.0, "pack": 0.0, "destroy": 0.0, "to": 0.0, "total": 0.0}
def wrap(fn, tag):
    def f(*a):
        t0 = time.perf_counter(); r = fn(*a); T[tag] += time.perf_counter() - t0; return r
    return f
_lib._plan_fns = (wrap(cr, "create"), wrap(pk, "pack"), wrap(ds, "destroy"))
orig_to = batch.BatchPlan.to
def to(self, *a, **k):
    t0 = time.perf_counter(); r = orig_to(self, *a, **k); T["to"] += time.perf_counter() - t0; return r
batch.BatchPlan.to = to
pool = batch._PinnedPool()
cs = torch.cuda.Stream()
for rep in range(2):
    for k in T: T[k] = 0.0
    for b in range(NB):
        idx = np.arange(B * b, B * b + B)
        t0 = time.perf_counter()
        p = batch.build_plan_native(csr, idx, 4, user_keys=idx, total_items=I, device=torch.device("cuda:0"), stream=cs, host_alloc=pool.take, host_release=pool.give)
        T["total"] += time.perf_counter() - t0
    torch.cuda.synchronize()
    print({k: round(v / NB * 1e3, 3) for k, v in T.items()})

# ... and engine.prepare (the sweeps' schedule tables) per plan
eng = S.SVAEEngine(I, device="cuda", prec=S.PREC_BF16)
plans = []
for b in range(NB):
    idx = np.arange(B * b, B * b + B)
    plans.append(batch.build_plan_native(csr, idx, 4, user_keys=idx, total_items=I, device=torch.device("cuda:0"), stream=cs, host_alloc=pool.take, host_release=pool.give))
torch.cuda.synchronize()
t0 = time.perf_counter()
with torch.cuda.stream(cs):
    for p in plans:
        eng.prepare(p)
torch.cuda.synchronize()
print({"prepare_ms_per_plan": round((time.perf_counter() - t0) / NB * 1e3, 3)})
