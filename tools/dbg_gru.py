import sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import svae_cf_amd as S
from oracle import svae_numpy as O
from test_gpu_parity import _engine, _sd
rs = np.random.RandomState(6)
I = 500
users = [rs.permutation(I)[:rs.randint(2, 42)] for _ in range(70)]
epsl = [rs.standard_normal((len(u) - 1, 64)).astype(np.float32) for u in users]
p = O.make_params(I, 78)
res = {}
for prec in (0, 1):
    eng = _engine(I, prec=prec)
    eng.load_state_dict(_sd(p))
    plan = S.build_plan(users, 4, device="cuda", total_items=I)
    eps = torch.from_numpy(S.pack_rows(plan, epsl, 64)).cuda()
    for k in ("hs", "hprev", "gates4"):
        eng._ensure(plan.R); eng.ws[k].fill_(float("nan"))
    eng.forward(plan, 0.13, eps=eps); eng.backward(); torch.cuda.synchronize()
    R = plan.R
    res[prec] = {k: eng.ws[k][:R * n].view(R, n).cpu().numpy().copy() for k, n in (("hs", 224), ("hprev", 224), ("gates4", 896), ("dhs", 224), ("dgi", 672))}
    print(prec, "R", R, "B", plan.B, "Tmax", plan.Tmax, "nact", plan.nact.cpu().numpy()[:8])
for k in res[0]:
    a, b = res[0][k], res[1][k]
    bad = ~np.isfinite(b)
    d = np.abs(a - b); d[bad] = 1e9
    rows = np.where(d.max(axis=1) > 0.05)[0]
    print(k, "nan count", bad.sum(), "bad rows", len(rows), rows[:20], "max", d[~bad].max() if (~bad).any() else None)
