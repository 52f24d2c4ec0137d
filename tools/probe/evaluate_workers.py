"""notebook_api.evaluate end to end by number of plan-builder threads and interpreter switch interval, with the host
time of the loop's engine.score calls (the rest of loop_s is waiting for plans, the stream contexts and the lock).

    python tools/probe/evaluate_workers.py
"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import svae_cf_amd as S
from svae_cf_amd import synth, notebook_api as N
import svae_cf_amd.batch as _batch
if os.environ.get('SVAE_CONSUMER_UPLOADS'):
    _batch.CONSUMER_UPLOADS = os.environ['SVAE_CONSUMER_UPLOADS'] == '1'
N.EVAL_PLAN_CACHE_BYTES = 0          # every call builds its plans (the first call on a reader)
I, B = 20000, 256
_, _, mu, sigma, lmax = synth.SHAPES["ml-20m"]
eng = S.SVAEEngine(I, device="cuda", prec=S.PREC_BF16)
torch.manual_seed(0)
eng.load_state_dict({n: torch.randn(s) * 0.05 for n, s in eng.layout.ref_shapes.items()})
acc = [0.0, 0]
_score = eng.score


def timed_score(*a, **k):
    t0 = time.perf_counter()
    r = _score(*a, **k)
    acc[0] += time.perf_counter() - t0
    acc[1] += 1
    return r


eng.score = timed_score


class Reader:
    def __init__(self, us):
        self.data = [u[:int(0.8 * len(u))].tolist() for u in us]
        self.data_te = [u[int(0.8 * len(u)):].tolist() for u in us]
    def eval_users(self): return list(zip(self.data, self.data_te))
    def _k(self): return 4


class M:
    engine = eng


NU = 64 * B
reader = Reader(synth.make_users(NU, I, mu, sigma, lmax, seed=4242))
N.evaluate(M, None, reader, dict(train_cp_users=200), False, batch_size=B, seed=1)
default_si = sys.getswitchinterval()
for si in (default_si, 2e-4):
    sys.setswitchinterval(si)
    for w in (1, 2, 3, 1, 2, 3):
        acc[0], acc[1] = 0.0, 0
        torch.cuda.synchronize(); t0 = time.perf_counter()
        N.evaluate(M, None, reader, dict(train_cp_users=200), False, batch_size=B, seed=1, workers=w)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("switch %.0e workers %d: %.1f ms, %.0f users/s; score() host %.3f ms per call; %s"
              % (si, w, dt * 1e3, NU / dt, acc[0] / max(1, acc[1]) * 1e3,
                 {k: round(v * 1e3, 2) for k, v in N.evaluate.last_timing.items()}))
sys.setswitchinterval(default_si)
