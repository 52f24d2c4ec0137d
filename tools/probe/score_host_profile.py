"""Host cost of one engine.score / engine.train_step call (enqueue only: the GPU is left to run behind), by function.

    python tools/probe/score_host_profile.py [score|train]
"""
import cProfile, os, pstats, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import svae_cf_amd as S
from svae_cf_amd import synth

what = sys.argv[1] if len(sys.argv) > 1 else "score"
I, B = 20000, 256
_, _, mu, sigma, lmax = synth.SHAPES["ml-20m"]
users = synth.make_users(8 * B, I, mu, sigma, lmax, seed=98765)
batches = synth.bucketed_batches([len(u) for u in users], B, np.random.default_rng(98765))
plans = [S.build_plan([users[i] for i in idx], 4, user_keys=idx, device="cuda", total_items=I) for idx in batches]
eng = S.SVAEEngine(I, device="cuda", prec=S.PREC_BF16)
call = (lambda p: eng.score(p, 100, seed=1)) if what == "score" else (lambda p: eng.train_step(p, 0.01, seed=1))
with torch.cuda.stream(eng.training_stream()):
    for p in plans:
        eng.prepare(p)
        call(p)
    torch.cuda.synchronize()
    # host time per call with the GPU idle in front (nothing to wait for) ...
    t0 = time.perf_counter()
    for p in plans:
        call(p)
    t_host = (time.perf_counter() - t0) / len(plans)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(4):
        for p in plans:
            call(p)
    pr.disable()
    torch.cuda.synchronize()
print("%s: host enqueue %.3f ms per call (8 calls, GPU idle in front)" % (what, t_host * 1e3))
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(18)
