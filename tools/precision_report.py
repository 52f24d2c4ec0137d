#!/usr/bin/env python3
"""Measured deviation of the HIP path from the reference golden vectors, per precision mode.
Runs on the GPU box:  python tools/precision_report.py   (reads tests/golden only)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import svae_cf_amd as S  # noqa: E402
from oracle import svae_numpy as O  # noqa: E402  (checker)

G = os.path.join(ROOT, "tests", "golden")


def l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


def sd(p):
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in p.items()}


for prec, pname in ((0, "fp32"), (2, "split"), (1, "bf16")):
    print("=== prec", pname)
    g = np.load(os.path.join(G, "fwd_bwd_I97_k4.npz"))
    I, seed, anneal = int(g["total_items"]), int(g["seed"]), float(g["anneal"])
    eng = S.SVAEEngine(I, prec=prec)
    eng.load_state_dict(sd(O.make_params(I, seed)))
    plan = S.build_plan([g["u0_items"]], 4, device="cuda", total_items=I)
    eps = torch.from_numpy(S.pack_rows(plan, [g["u0_eps"]], 64)).cuda()
    out = eng.forward(plan, anneal, eps=eps, want_grad=False)
    print(" logits l2 %.2e  mu %.2e  loss rel %.2e" % (
        l2(eng.dense_logits()[:, :I].cpu().numpy(), g["u0_dec_out"]),
        l2(eng.mu_logsigma(plan.R)[0].cpu().numpy(), g["u0_mu"]),
        abs(float(out[0]) - float(g["u0_loss"])) / float(g["u0_loss"])))
    eng.forward(plan, anneal, eps=eps)
    eng.backward()
    gd = eng.grad_dict()
    print(" grads l2:", {k.replace("weight", "w").replace("bias", "b"): "%.1e" % l2(gd[k].cpu().numpy(), g["u0_grad/" + k]) for k in O.PARAM_NAMES})

    g = np.load(os.path.join(G, "train_I97.npz"))
    I, seed = int(g["total_items"]), int(g["seed"])
    eng = S.SVAEEngine(I, prec=prec)
    eng.load_state_dict(sd(O.make_params(I, seed)))
    losses = []
    for b, u in enumerate(g["order"], start=1):
        plan = S.build_plan([g["items%d" % u]], 4, device="cuda", total_items=I)
        eps = torch.from_numpy(S.pack_rows(plan, [g["eps%d" % (b - 1)]], 64)).cuda()
        losses.append(float(eng.train_step(plan, O.anneal_for_batch(b), eps=eps)[0]))
    print(" train 12 steps: max loss rel err %.2e" % np.max(np.abs(np.array(losses) - g["losses"]) / g["losses"]))
    st = eng.state_dict()
    print(" params l2 after 12 steps:", {k.replace("weight", "w").replace("bias", "b"): "%.1e" % l2(st[k].cpu().numpy(), g["final/" + k]) for k in O.PARAM_NAMES})
