#!/usr/bin/env python3
"""Timing experiment (results wrong by construction): the forward sweep with a FLAT work division -- equal ranges of
the flattened (x tile, y tile) space, cut at x-tile boundaries into workgroups, longest first -- against the
q / q+1 equal splits per x tile of svae_proj_schedule.  Needs csrc/exp/lib_any.so (-DSV_EXP_SCHED_ANY).
    SVAE_HIP_LIB=.../exp/lib_any.so python tools/probe/flat_sched_exp.py [flat|base] [cost_tiles]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import svae_cf_amd as S
from svae_cf_amd import synth
mode = sys.argv[1] if len(sys.argv) > 1 else "flat"
cost = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0          # a workgroup's fixed cost, in tiles
I, B = 20000, 256
_, _, mu, sigma, lmax = synth.SHAPES["ml-20m"]
users = synth.make_users(4096, I, mu, sigma, lmax, seed=98765)
batches = synth.bucketed_batches([len(u) for u in users], B, np.random.default_rng(98765))[:16]
plans = [S.build_plan([users[i] for i in idx], 4, user_keys=idx, device="cuda", total_items=I) for idx in batches]
eng = S.SVAEEngine(I, device="cuda", prec=S.PREC_BF16)
eng.reserve(max(p.R for p in plans))
nt = (I + 63) // 64

def flat_table(nx, ncu):
    xt = (nx + 255) // 256
    rounds = (4 * xt + ncu - 1) // ncu
    nr = min(rounds * ncu, 14 * xt)                 # (an x tile holds at most 16 slabs)
    # cost-aware: every piece pays `cost` tiles; ranges of equal cost
    ent = []
    total = xt * nt
    # iterate: first guess of pieces = nr + xt - 1
    target = (total + cost * (nr + xt - 1)) / nr
    x, y = 0, 0
    for r in range(nr):
        budget = target
        while budget > cost + 0.5 and x < xt:
            room = nt - y
            take = int(min(room, max(1, round(budget - cost))))
            if r == nr - 1:
                take = room
            ent.append([x, y, y + take, 0])
            budget -= cost + take
            y += take
            if y == nt:
                x, y = x + 1, 0
            if take < room:
                break
    while x < xt:                                   # leftovers (rounding): into the last pieces
        ent.append([x, y, nt, 0]); x, y = x + 1, 0
    cnt = {}
    for e in ent:
        e[3] = cnt.get(e[0], 0); cnt[e[0]] = e[3] + 1
    assert max(cnt.values()) <= 16, max(cnt.values())
    if os.environ.get("FLAT_ORDER", "len") == "len":
        ent.sort(key=lambda e: -(e[2] - e[1]))          # longest first
    else:
        # as svae_proj_schedule: by where the piece starts in Y, dealt to the XCDs (id % 8) in contiguous runs
        ent.sort(key=lambda e: (e[1], e[0]))
        n = len(ent)
        out = [None] * n
        p = 0
        for c in range(8):
            for i in range(c, n, 8):
                out[i] = ent[p]; p += 1
        ent = out
    return np.array(ent, dtype=np.int32)

orig = eng._proj_sched
def patched(plan, side, nx=None, ncu=None, host_only=False):
    if side != "rows" or mode != "flat":
        return orig(plan, side, nx=nx, ncu=ncu, host_only=host_only)
    ncu_ = eng._ncu if ncu is None else max(8, int(ncu))
    key = ("flat", nx, ncu_)
    if host_only:
        return None
    if key not in plan.proj_sched:
        tab = flat_table(plan.R if nx is None else nx, ncu_)
        plan.proj_sched[key] = (torch.from_numpy(tab).to("cuda"), int(tab.shape[0]), 1, 0)
    t, n, q, rem = plan.proj_sched[key]
    return (t.data_ptr(), n, q, rem)
eng._proj_sched = patched
for p in plans:
    eng.prepare(p)
for i in range(64):
    eng.train_step(plans[i % 16], 0.01, seed=1)
torch.cuda.synchronize()
import time
res = []
for rep in range(3):
    t0 = time.perf_counter()
    for i in range(48):
        eng.train_step(plans[i % 16], 0.01, seed=1)
    torch.cuda.synchronize()
    res.append((time.perf_counter() - t0) / 48 * 1e3)
eng.prof = {"proj_fdh_fused": []}
for i in range(48):
    eng.train_step(plans[i % 16], 0.01, seed=1)
torch.cuda.synchronize()
fdh = sum(a.elapsed_time(b) for a, b in eng.prof["proj_fdh_fused"]) / 48
print(json.dumps({"mode": mode, "cost_tiles": cost, "ms_per_step": res, "fdh_sweep_ms_per_step": fdh}))
