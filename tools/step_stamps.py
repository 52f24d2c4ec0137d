#!/usr/bin/env python3
"""Where do the streams of a training step wait?  In-stream clock stamps (svae_stamp: a one-thread kernel, ~2 us on its
stream) at the join points of the step, WITHOUT a profiler attached (its launch cost makes the host the bottleneck at
exactly those points).  Prints, per stamp, the median time since the step's first stamp over the timed steps, and the
step period.     python tools/step_stamps.py [steps] [shape]"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import svae_cf_amd as S
from svae_cf_amd import synth

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 48
shape = sys.argv[2] if len(sys.argv) > 2 else "ml-20m"
_, I, mu, sigma, lmax = synth.SHAPES[shape]
B, NP = 256, 16
users = synth.make_users(B * NP, I, mu, sigma, lmax, seed=98765)
batches = synth.bucketed_batches([len(u) for u in users], B, np.random.default_rng(98765))
plans = [S.build_plan([users[i] for i in idx], 4, user_keys=idx, device="cuda", total_items=I) for idx in batches]
eng = S.SVAEEngine(I, device="cuda", prec=S.PREC_BF16)
torch.manual_seed(0)
eng.load_state_dict({n: torch.randn(s) * 0.05 for n, s in eng.layout.ref_shapes.items()})
eng.reserve(max(p.R for p in plans))
for p in plans:
    eng.prepare(p)
torch.cuda.set_stream(eng.training_stream())
for i in range(64):
    eng.train_step(plans[i % NP], 0.01, seed=1)
torch.cuda.synchronize()
buf = torch.zeros(steps * 64, dtype=torch.int64, device="cuda")
eng.stamps = dict(buf=buf, names={}, step=0)
for i in range(steps):
    eng.stamps["step"] = i
    eng.train_step(plans[i % NP], 0.01, seed=1)
torch.cuda.synchronize()
names = eng.stamps["names"]
eng.stamps = None
t = buf.cpu().numpy().reshape(steps, 64)[:, :len(names)].astype(np.float64) / 100.0     # us
t0 = t[:, 0:1]
rel = t - t0
period = np.diff(t[:, 0])
skip = 4
print("all plans: step period (us) median %.1f mean %.1f" % (np.median(period[skip:]), period[skip:].mean()))
# one plan (steps vary 2.5x in rows): the one given, else the plan with the median row count among the time-split ones
Rs = [p.R if p.split_t > 0 else -1 for p in plans]
which = int(sys.argv[3]) if len(sys.argv) > 3 else sorted(range(NP), key=lambda i: Rs[i])[NP // 2 + sum(r < 0 for r in Rs) // 2]
sel = np.array([i for i in range(skip, steps - 1) if i % NP == which])
print("plan %d: R = %d rows, split at t = %d (row %d), Tmax = %d; %d samples" % (which, plans[which].R, plans[which].split_t, plans[which].split_r, plans[which].Tmax, len(sel)))
period = (t[sel + 1, 0] - t[sel, 0])
rel = rel[sel]
skip = 0
print("step period (us): median %.1f  mean %.1f   (stamps cost ~2 us each on their streams, %d per step)" % (np.median(period[skip:]), period[skip:].mean(), len(names)))
order = sorted(names, key=lambda n: np.median(rel[skip:, names[n]]))
for n in order:
    v = rel[skip:, names[n]]
    print("%8.1f  (p10 %7.1f  p90 %7.1f)  %s" % (np.median(v), np.percentile(v, 10), np.percentile(v, 90), n))
# the floated work of step k ends inside step k + 1: relative to the NEXT step's beginning
j = names.get("w3 optimizer + operand copies of decoder.linear2 done")
if j is not None:
    late = t[:-1, j] - t[1:, 0]
    print("floated decoder.linear2 work of step k ends %.1f us (median) after step k+1 begins" % np.median(late[skip:]))
