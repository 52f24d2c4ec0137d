"""Scoring throughput (evaluate()'s per-user work: forward with loss at anneal 0.2, last-step logits, seen-item
mask, top-100) on the ML-20M shape, and the masked top-k kernel alone against its HBM roofline.

    python tools/eval_bench.py [items] [batch]
"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import svae_cf_amd as S
from svae_cf_amd import synth, _lib

I = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
_, _, mu, sigma, lmax = synth.SHAPES["ml-20m"]
users = synth.make_users(8 * B, I, mu, sigma, lmax, seed=98765)
batches = synth.bucketed_batches([len(u) for u in users], B, np.random.default_rng(98765))
plans = [S.build_plan([users[i] for i in idx], 4, user_keys=idx, device="cuda", total_items=I) for idx in batches]
eng = S.SVAEEngine(I, device="cuda", prec=S.PREC_BF16)
torch.manual_seed(0)
eng.load_state_dict({n: torch.randn(s) * 0.05 for n, s in eng.layout.ref_shapes.items()})
for p in plans:
    eng.prepare(p)
    eng.score(p, 100, seed=1)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 0
for _ in range(3):
    for p in plans:
        eng.score(p, 100, seed=1)
        n += p.B
torch.cuda.synchronize()
dt = time.perf_counter() - t0


# the same plans inside a scoring session: parameter-derived operands made once, batches alternating between two
# workspaces on two streams (what notebook_api.evaluate does)
def session_pass(reps, sweep_cus=None, nlanes=2):
    cur = torch.cuda.current_stream()
    with eng.scoring_session(sweep_cus=sweep_cus):
        lanes = [cur] + [eng.scoring_stream(w) for w in range(nlanes - 1)]
        lanes = [s_ for s_ in lanes if s_ is not None]
        for s_ in lanes[1:]:
            s_.wait_stream(cur)
        k = 0
        for _ in range(reps):
            for p in plans:
                with torch.cuda.stream(lanes[k % len(lanes)]):
                    eng.score(p, 100, seed=1, slot=k % len(lanes))
                k += 1
        for s_ in lanes[1:]:
            cur.wait_stream(s_)


session_pass(1)
torch.cuda.synchronize()
t0 = time.perf_counter()
session_pass(3)
torch.cuda.synchronize()
dt_session = time.perf_counter() - t0
by_cus = {}
for cus in (160, 176, 192, 208, 224, 240):
    session_pass(1, cus)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    session_pass(3, cus)
    torch.cuda.synchronize()
    by_cus[cus] = (time.perf_counter() - t0) / (3 * len(plans)) * 1e3
three = {}
for cus in (128, 160, 192, 256):
    session_pass(1, cus, 3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    session_pass(3, cus, 3)
    torch.cuda.synchronize()
    three[cus] = (time.perf_counter() - t0) / (3 * len(plans)) * 1e3
# the top-k kernel alone
st = torch.cuda.current_stream().cuda_stream
p = plans[0]
scores = torch.randn(B, eng.IP, device="cuda")
work = torch.empty_like(scores)
idx = torch.empty(B * 100, dtype=torch.int32, device="cuda")
val = torch.empty(B * 100, device="cuda")
args = (scores.data_ptr(), eng.IP, p.seen.data_ptr(), p.seen_off.data_ptr(), B, I, 100, idx.data_ptr(), val.data_ptr(),
        work.data_ptr(), st)
for _ in range(3):
    _lib.call("svae_masked_topk", *args)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    _lib.call("svae_masked_topk", *args)
e1.record()
e1.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 20
# the same kernel over a larger scoring batch (8 plans' users at once: two workgroups per CU in flight)
BB = 8 * B
big = torch.randn(BB, eng.IP, device="cuda")
bwork = torch.empty_like(big)
seen_off = torch.cat([p.seen_off[:-1].to(torch.int64)] * 8 + [p.seen_off[-1:].to(torch.int64)]).to(torch.int32)   # reuse plan 0's lists
seen_off = (torch.arange(BB + 1, device="cuda") % (B + 0) * 0).to(torch.int32)      # no seen items: mask cost is negligible
bidx = torch.empty(BB * 100, dtype=torch.int32, device="cuda")
bval = torch.empty(BB * 100, device="cuda")
bargs = (big.data_ptr(), eng.IP, p.seen.data_ptr(), seen_off.data_ptr(), BB, I, 100, bidx.data_ptr(), bval.data_ptr(),
         bwork.data_ptr(), st)
for _ in range(3):
    _lib.call("svae_masked_topk", *bargs)
e0.record()
for _ in range(10):
    _lib.call("svae_masked_topk", *bargs)
e1.record()
e1.synchronize()
us_big = e0.elapsed_time(e1) * 1e3 / 10
ok = bool((torch.topk(big[:, :I], 100, dim=1).indices.to(torch.int32).reshape(-1) == bidx).all())
algo = B * I * 4 + B * 100 * 8

# END TO END through the public entry point: notebook_api.evaluate over a validation-type reader (fold-in + held-out
# lists per user, as DataReader holds them): plans built from the lists by the prefetcher, scored, ids back, the metric
# arithmetic of nb:626-654 -- against `value` above (engine.score on resident plans, nothing else).
from svae_cf_amd import notebook_api as N


class Reader:
    def __init__(self, us):
        self.data = [u[:int(0.8 * len(u))].tolist() for u in us]
        self.data_te = [u[int(0.8 * len(u)):].tolist() for u in us]
    def eval_users(self): return list(zip(self.data, self.data_te))
    def _k(self): return 4


class M:                       # the part of notebook_api.Model evaluate() reads
    engine = eng


NU = int(os.environ.get("SVAE_EVAL_USERS", 32 * B))
ev_users = synth.make_users(NU, I, mu, sigma, lmax, seed=4242)
reader = Reader(ev_users)
hp = dict(train_cp_users=200)
e2e = {}
for tag in ("warmup", "run", "run2"):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    metrics, _ = N.evaluate(M, None, reader, hp, False, batch_size=B, seed=1)
    torch.cuda.synchronize()
    e2e[tag] = time.perf_counter() - t0
    e2e[tag + "_timing"] = dict(N.evaluate.last_timing)
# the same users with plans resident and nothing but engine.score (the bar: 0.7 x this rate)
res_plans = [S.build_plan([np.asarray(reader.data[i]) for i in range(s, min(s + B, NU))], 4,
                          user_keys=np.arange(s, min(s + B, NU)), device="cuda", total_items=I) for s in range(0, NU, B)]
for p_ in res_plans:
    eng.prepare(p_, scoring=True)
    eng.score(p_, 100, seed=1, step=0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for p_ in res_plans:
    eng.score(p_, 100, seed=1, step=0)
torch.cuda.synchronize()
t_res = time.perf_counter() - t0
best = min(e2e["run"], e2e["run2"])
end_to_end = {"entry_point": "notebook_api.evaluate(batch_size=%d)" % B, "users": NU, "seconds": [e2e["run"], e2e["run2"]],
              "users_per_s": NU / best, "resident_score_users_per_s": NU / t_res, "ratio": t_res / best, "host_timing": [e2e["run_timing"], e2e["run2_timing"]],
              "metrics": metrics}
print(json.dumps({"metric": "users scored/s (ML-20M shape, top-100)", "value": n / dt, "batch": B, "items": I,
                  "evaluate_end_to_end": end_to_end,
                  "ms_per_batch": dt / (3 * len(plans)) * 1e3,
                  "scoring_session_two_lanes": {"users_per_s": n / dt_session, "ms_per_batch": dt_session / (3 * len(plans)) * 1e3,
                                                "ms_per_batch_by_sweep_workgroups": by_cus, "three_lanes_ms_per_batch_by_sweep_workgroups": three},
                  "masked_topk": {"us": us, "algorithmic_bytes": algo, "GB/s": algo / us / 1e3,
                                  "frac_of_8TB/s": algo / us / 1e3 / 8000.0},
                  "masked_topk_2048_users": {"us": us_big, "GB/s": 8 * algo / us_big / 1e3,
                                             "frac_of_8TB/s": 8 * algo / us_big / 1e3 / 8000.0, "equals_torch_topk": ok}}))
