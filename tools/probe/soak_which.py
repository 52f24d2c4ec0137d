#!/usr/bin/env python3
"""Which tensors differ between two runs of N steps from the same state?  (diagnostic for a nondeterministic step)"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import svae_cf_amd as S
from svae_cf_amd import synth
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
I, B = 20000, 256
_, _, mu, sigma, lmax = synth.SHAPES["ml-20m"]
users = synth.make_users(4096, I, mu, sigma, lmax, seed=98765)
batches = synth.bucketed_batches([len(u) for u in users], B, np.random.default_rng(98765))[:16]
plans = [S.build_plan([users[i] for i in idx], 4, user_keys=idx, device="cuda", total_items=I) for idx in batches]
for attempt in range(6):
    outs = []
    for run in range(2):
        eng = S.SVAEEngine(I, device="cuda", prec=S.PREC_BF16)
        g = torch.Generator().manual_seed(1)
        eng.load_state_dict({n: torch.randn(s, generator=g) * 0.05 for n, s in eng.layout.ref_shapes.items()})
        eng.reserve(max(p.R for p in plans))
        for p in plans:
            eng.prepare(p)
        hist = []
        for i in range(steps):
            eng.train_step(plans[i % 16], 0.05, seed=3)
            hist.append({k: v.clone() for k, v in eng.grad_dict().items()})      # (grad_dict settles the floated work)
        torch.cuda.synchronize()
        outs.append(hist)
    for i in range(steps):
        bad = [k for k in outs[0][i] if not torch.equal(outs[0][i][k], outs[1][i][k])]
        if bad:
            print("attempt", attempt, "first difference at step", i, "in", bad)
            for k in bad[:4]:
                a, b = outs[0][i][k], outs[1][i][k]
                idx = (a != b).nonzero()
                print("   ", k, tuple(a.shape), "elements", idx.shape[0], "first", idx[:5].tolist(), "max abs", float((a - b).abs().max()), "scale", float(a.abs().max()))
            break
    else:
        print("attempt", attempt, "identical")
