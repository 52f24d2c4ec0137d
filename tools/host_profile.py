"""cProfile of the host side of train_step (where do the ~0.5 ms of enqueue time per step go?)."""
import cProfile, os, pstats, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import svae_cf_amd as S
from svae_cf_amd import synth
I, B = 20000, 256
_, _, mu, sigma, lmax = synth.SHAPES["ml-20m"]
users = synth.make_users(8 * B, I, mu, sigma, lmax, seed=98765)
rng = np.random.default_rng(98765)
batches = synth.bucketed_batches([len(u) for u in users], B, rng)
plans = [S.build_plan([users[i] for i in idx], 4, user_keys=idx, device="cuda", total_items=I) for idx in batches]
eng = S.SVAEEngine(I, device="cuda", prec=S.PREC_BF16)
for p_ in plans:
    eng.prepare(p_)
for i in range(8):
    eng.train_step(plans[i % 8], 0.01, seed=1)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(200):
    eng.train_step(plans[i % 8], 0.01, seed=1)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
