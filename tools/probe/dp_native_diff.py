"""Where does the native-collective step differ from the plain one at world size 1?  (diagnostic)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import svae_cf_amd as S
from svae_cf_amd import synth
from svae_cf_amd.dist import GradSync
I, B = 20000, 128
_, _, mu, sigma, lmax = synth.SHAPES["ml-20m"]
users = synth.make_users(B, I, mu, sigma, lmax, seed=77)
plan = S.build_plan(users, 4, user_keys=np.arange(B), device="cuda", total_items=I)
res = {}
for tag in ("plain", "plain2", "inactive", "native", "native2"):
    eng = S.SVAEEngine(I, device="cuda", prec=S.PREC_BF16)
    g = torch.Generator().manual_seed(3)
    eng.load_state_dict({n: (torch.randn(s_, generator=g) * (0.07 if n != "item_embed.weight" else 1.0))
                         for n, s_ in eng.layout.ref_shapes.items()})
    eng.prepare(plan)
    if tag == "inactive":
        eng.grad_sync = GradSync(eng.grads, eng.layout.buckets())
    if tag.startswith("native"):
        eng.grad_sync = GradSync(eng.grads, eng.layout.buckets(), force=True, native=True)
    outs = []
    for s_ in range(3):
        eng.train_step(plan, 0.05, seed=9, step=s_)
        torch.cuda.synchronize()
        outs.append((float(eng.loss_out[0]), float(eng.grads.double().abs().sum()), float(eng.params.double().abs().sum())))
    res[tag] = outs
    print(tag, outs, flush=True)
