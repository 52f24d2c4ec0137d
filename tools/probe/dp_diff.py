"""Probe: gradients of one step (bf16 mode) for the union batch of tests/test_gpu_dp.py and for each rank's half,
written to gpurun_out/dp_diff_<tag>.npz -- run once per kernel library, then compare with `cmp`."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

if sys.argv[1] == "cmp":
    a, b = (np.load(os.path.join(ROOT, "gpurun_out", "dp_diff_%s.npz" % t)) for t in sys.argv[2:4])
    for k in a.files:
        d = np.linalg.norm(a[k].astype(np.float64) - b[k]) / max(np.linalg.norm(b[k]), 1e-30)
        if d > 2e-4 or k.endswith("loss"):
            print("%-44s rel %.3e" % (k, d))
    sys.exit(0)

import svae_cf_amd as S
from svae_cf_amd import synth
from svae_cf_amd.dist import rank_batches
import test_gpu_dp as T
users = T._users()
lens = [len(u) for u in users]
out = {}
steps = synth.bucketed_batches(lens, T.B, np.random.default_rng(11), world=T.WORLD)
for which in ("union", "r0", "r1"):
    eng = S.SVAEEngine(T.I, device="cuda", prec=1)
    eng.load_state_dict(T._init_sd(eng.layout))
    if which == "union":
        idx = np.concatenate(steps[0])
    else:
        idx = rank_batches(lens, T.B, np.random.default_rng(11), int(which[1]), T.WORLD)[0]
    plan = S.build_plan([users[i] for i in idx], 4, user_keys=idx, device="cuda", total_items=T.I)
    for rep in range(2):
        o = eng.forward(plan, 0.05, seed=77, step=0)
        eng.backward()
        torch.cuda.synchronize()
        g = {k: v.detach().cpu().numpy().copy() for k, v in eng.grad_dict().items()}
        if rep == 0:
            g0 = g
        else:
            for k in g:
                if not np.array_equal(g[k], g0[k]):
                    print(which, "NOT bit-reproducible:", k, np.abs(g[k] - g0[k]).max())
    for k, v in g.items():
        out[which + "/" + k] = v
    out[which + "/loss"] = np.asarray(torch.as_tensor(o[0] if isinstance(o, (tuple, list)) else o).detach().double().cpu().numpy()).ravel()[:1]
    print(which, "rows", plan.R if hasattr(plan, "R") else "?", "loss", out[which + "/loss"])
np.savez(os.path.join(ROOT, "gpurun_out", "dp_diff_%s.npz" % sys.argv[1]), **out)
