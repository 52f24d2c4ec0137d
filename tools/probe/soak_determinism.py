#!/usr/bin/env python3
"""Soak: N training steps over the bench's plans, twice from the same state; the two runs must end bit-identical (a
missing stream dependency shows up as a difference sooner or later) and finite.   python tools/probe/soak_determinism.py [steps]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import svae_cf_amd as S
from svae_cf_amd import synth
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
I, B = 20000, 256
_, _, mu, sigma, lmax = synth.SHAPES["ml-20m"]
users = synth.make_users(4096, I, mu, sigma, lmax, seed=98765)
batches = synth.bucketed_batches([len(u) for u in users], B, np.random.default_rng(98765))[:16]
plans = [S.build_plan([users[i] for i in idx], 4, user_keys=idx, device="cuda", total_items=I) for idx in batches]
outs = []
for run in range(2):
    eng = S.SVAEEngine(I, device="cuda", prec=S.PREC_BF16)
    g = torch.Generator().manual_seed(1)
    eng.load_state_dict({n: torch.randn(s, generator=g) * 0.05 for n, s in eng.layout.ref_shapes.items()})
    eng.reserve(max(p.R for p in plans))
    for p in plans:
        eng.prepare(p)
    for i in range(steps):
        eng.train_step(plans[i % 16], 0.05, seed=3)
    torch.cuda.synchronize()
    outs.append((eng.params.clone(), eng.exp_avg.clone(), float(eng.loss_out[0])))
same = torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
print(json.dumps({"steps": steps, "bit_identical": bool(same), "finite": bool(torch.isfinite(outs[0][0]).all()),
                  "final_loss": outs[0][2], "final_loss_run2": outs[1][2]}))
sys.exit(0 if same else 1)
