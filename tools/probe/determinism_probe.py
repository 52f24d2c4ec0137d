"""Which array first differs between two engines running the same step?  (diagnostic)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import svae_cf_amd as S
from svae_cf_amd import synth
I, B = 20000, int(os.environ.get("PB", "128"))
_, _, mu, sigma, lmax = synth.SHAPES["ml-20m"]
users = synth.make_users(B, I, mu, sigma, lmax, seed=77)
plan = S.build_plan(users, 4, user_keys=np.arange(B), device="cuda", total_items=I)
print("R", plan.R, "split", plan.split_t, plan.split_r, flush=True)
names = ["hprev16", "hprev", "emb16", "gi", "hs", "gates4", "enc", "tmp", "eps", "z", "d1", "h2b", "h2s", "nll_row", "lse", "dd1", "dtmp", "denc", "dhs", "dg4b", "demb", "kl_row"]
snaps = []
for rep in range(4):
    eng = S.SVAEEngine(I, device="cuda", prec=S.PREC_BF16)
    g = torch.Generator().manual_seed(3)
    eng.load_state_dict({n: (torch.randn(s_, generator=g) * (0.07 if n != "item_embed.weight" else 1.0))
                         for n, s_ in eng.layout.ref_shapes.items()})
    eng.prepare(plan)
    if os.environ.get("PMODE", "step") == "step":
        eng.train_step(plan, 0.05, seed=9, step=0)
    else:
        eng.forward(plan, 0.05, seed=9, step=0)
        eng.backward()
    torch.cuda.synchronize()
    R = plan.R
    widths = dict(hprev16=224, hprev=224, emb16=256, gi=672, hs=224, gates4=896, enc=160, tmp=128, eps=64, z=64, d1=160, h2b=160, h2s=160, nll_row=1, lse=1,
                  dd1=160, dtmp=128, denc=160, dhs=224, dg4b=896, demb=256, kl_row=1)
    snap = {n: eng.ws[n][:R * widths[n]].clone() for n in names if eng.ws.get(n) is not None}
    snap["loss"] = eng.loss_out.clone()
    snap["grads"] = eng.grads.clone()
    for k_, v_ in eng.grad_dict().items():
        snap["g:" + k_] = v_.clone()
    snaps.append(snap)
for rep in range(1, 4):
    diff = [n for n in snaps[0] if not torch.equal(snaps[0][n].view(torch.uint8) if snaps[0][n].dtype == torch.bfloat16 else snaps[0][n],
                                                   snaps[rep][n].view(torch.uint8) if snaps[rep][n].dtype == torch.bfloat16 else snaps[rep][n])]
    print("rep", rep, "differs in:", diff, flush=True)
    for n in diff[:6]:
        a, b = snaps[0][n].float(), snaps[rep][n].float()
        w = a.numel() // plan.R if a.numel() % plan.R == 0 and a.numel() >= plan.R else 1
        idx = (a != b).nonzero().flatten()
        print("   ", n, "count", idx.numel(), "first rows", sorted(set((idx[:2000] // w).tolist()))[:12], "max abs", float((a - b).abs().max()))
