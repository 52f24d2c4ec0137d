import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import svae_cf_amd as S
from svae_cf_amd import synth, _lib
I, B = 20000, 128
_, _, mu, sigma, lmax = synth.SHAPES["ml-20m"]
users = synth.make_users(4 * B, I, mu, sigma, lmax, seed=77)
for prec in (S.PREC_BF16, S.PREC_SPLIT):
    for c in range(4):
        cand = np.arange(c * B, (c + 1) * B)
        p = S.build_plan([users[i] for i in cand], 4, user_keys=cand, device="cuda", total_items=I)
        if p.split_t <= 0:
            continue
        eng = S.SVAEEngine(I, device="cuda", prec=prec)
        g = torch.Generator().manual_seed(3)
        eng.load_state_dict({n: (torch.randn(s_, generator=g) * (0.07 if n != "item_embed.weight" else 1.0))
                             for n, s_ in eng.layout.ref_shapes.items()})
        eng.reserve(p.R); eng.prepare(p)
        for k, v in eng.ws.items():
            if v is not None and v.numel() and v.dtype == torch.float32:
                v.fill_(float("nan"))
        cur = torch.cuda.current_stream()
        for which in (0, 1):
            _lib.call("svae_spin", 3000, eng._side_stream(which, cur=cur).cuda_stream)
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        loss = eng.forward(p, 0.05, seed=9, step=4)
        t1.record()
        torch.cuda.synchronize()
        ro = int(p._row_off_host[p.split_t])
        hs = eng.ws["hs"][:p.R * 224].view(p.R, 224)
        print("prec", prec, "chunk", c, "R", p.R, "split_t", p.split_t, "split_r", p.split_r, "row_off[t1]", ro,
              "loss", float(loss[0]), "nan rows in hs", int(torch.isnan(hs).any(dim=1).sum()), "fwd ms", t0.elapsed_time(t1), flush=True)
