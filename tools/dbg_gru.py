"""Debug: bf16 GRU forward/backward outputs against the exact fp32 kernels on one ML-20M-shaped batch."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import svae_cf_amd as S
from svae_cf_amd import synth
from svae_cf_amd.layout import HP, G
I, B = 20000, 256
_, _, mu, sigma, lmax = synth.SHAPES["ml-20m"]
users = synth.make_users(2048, I, mu, sigma, lmax, seed=98765)
rng = np.random.default_rng(1)
idx = synth.bucketed_batches([len(u) for u in users], B, rng)[5]
plan = S.build_plan([users[i] for i in idx], 4, user_keys=idx, device="cuda", total_items=I)
ref = S.SVAEEngine(I, device="cuda", prec=S.PREC_FP32)
torch.manual_seed(0)
sd = {}
for name, shp in ref.layout.ref_shapes.items():
    sd[name] = torch.randn(shp) * (0.05 if not name.startswith("item") else 1.0)
ref.load_state_dict(sd)
fast = S.SVAEEngine(I, device="cuda", prec=S.PREC_BF16)
fast.load_state_dict(sd)
for e in (ref, fast):
    e.forward(plan, 0.05, seed=7, step=3)
    e.backward()
torch.cuda.synchronize()
R = plan.R
for name, w in (("hs", HP), ("hprev", HP), ("gates4", 4 * HP), ("dgi", G), ("dgh", G), ("dhs", HP)):
    a = ref.ws[name][:R * w].view(R, w).double()
    b = fast.ws[name][:R * w].view(R, w).double()
    d = (a - b).abs()
    rel = float((a - b).norm() / a.norm())
    worst = int(d.max(dim=1)[0].argmax())
    print("%-7s rel-L2 %.3e  max-abs %.3e at row %d col %d (R=%d)" % (name, rel, float(d.max()), worst, int(d[worst].argmax()), R))
    if name == "hs":
        percol = d.max(dim=0)[0]
        print("   per-16-unit-tile max err:", [round(float(percol[16 * t:16 * t + 16].max()), 4) for t in range(14)])
        rowerr = d.max(dim=1)[0]
        ro = plan.row_off.cpu().numpy(); na = plan.nact.cpu().numpy()
        t = int(np.searchsorted(ro, worst, side="right") - 1)
        print("   worst row: timestep", t, "user", worst - ro[t], "nact", na[t])
a = ref.ws["hs"][:R * HP].view(R, HP).double(); b = fast.ws["hs"][:R * HP].view(R, HP).double()
n0 = int(plan.nact[0])
e0 = (a[:n0] - b[:n0]).abs().max(dim=1)[0].cpu().numpy()
print("t=0 max err by user%4:", [round(float(e0[k::4].max()), 4) for k in range(4)], " first 12 users:", np.round(e0[:12], 4))
ga = ref.ws["gi"][:R * G].view(R, G).double(); gb = fast.ws["gi"][:R * G].view(R, G).double()
print("gi rel-L2", float((ga - gb).norm() / ga.norm()))
