"""Host (enqueue) time per training step vs GPU time: how far is the eager step from being host-bound?"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import svae_cf_amd as S
from svae_cf_amd import synth
I, B = 20000, 256
_, _, mu, sigma, lmax = synth.SHAPES["ml-20m"]
users = synth.make_users(16 * B, I, mu, sigma, lmax, seed=98765)
rng = np.random.default_rng(98765)
batches = synth.bucketed_batches([len(u) for u in users], B, rng)
plans = [S.build_plan([users[i] for i in idx], 4, user_keys=idx, device="cuda", total_items=I) for idx in batches]
eng = S.SVAEEngine(I, device="cuda", prec=S.PREC_BF16)
for p_ in plans:
    eng.prepare(p_)
if "--rccl" in sys.argv:      # world-size-1 process group on RCCL, all-reduces kept (needs MASTER_ADDR/PORT, RANK, WORLD_SIZE)
    import torch.distributed as dist
    from svae_cf_amd.dist import GradSync
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    eng.grad_sync = GradSync(eng.grads, eng.layout.buckets(), force=True)
for i in range(8):
    eng.train_step(plans[i % 16], 0.01, seed=1)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(40):
    eng.train_step(plans[i % 16], 0.01, seed=1)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("enqueue %.3f ms/step, total %.3f ms/step" % ((t1 - t0) / 40 * 1e3, (t2 - t0) / 40 * 1e3))
