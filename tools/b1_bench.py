#!/usr/bin/env python3
"""The notebook's own loop, one user per step (nb:1182-1233: model(x), VAELoss, loss.backward(), optimizer.step()
through notebook_api's Model / VAELoss / train), timed on the ML-1M shape: milliseconds per user beside the
13.8 ms per batch (= per user, batch_size 1) the reference's committed log shows (nb:882-886, hardware unstated).

    python tools/b1_bench.py [n_users]
"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import svae_cf_amd as S
from svae_cf_amd import synth, notebook_api as N

n_users = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
_, I, mu, sigma, lmax = synth.SHAPES["ml-1m"]
users = synth.make_users(n_users, I, mu, sigma, lmax, seed=4321)
hp = dict(item_embed_size=256, rnn_size=200, hidden_size=150, latent_size=64, total_items=I, batch_size=1,
          loss_type="next_k", next_k=4, number_users_to_keep=10 ** 9, weight_decay=5e-3, optimizer="adam",
          batch_log_interval=10 ** 9, learning_rate=0.01)
lines = ["%d,%d" % (u, it) for u, seq in enumerate(users) for it in seq]
out = {}
for tag, prec in (("fp32", S.PREC_FP32), ("bf16", S.PREC_BF16)):
    reader = N.DataReader(hp, lines, None, I, True)
    model = N.Model(hp, prec=prec, seed=0)
    opt = N.make_optimizer(model, hp)
    crit = N.VAELoss(hp, model)
    logs = []
    N.train(reader, model, opt, crit, dict(hp, number_users_to_keep=50), epoch=0, log=logs.append)   # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nb = N.train(reader, model, opt, crit, hp, epoch=1, log=logs.append)
    torch.cuda.synchronize()
    out[tag] = (time.perf_counter() - t0) / nb * 1e3
    out[tag + "_users"] = nb
print(json.dumps({"metric": "notebook-style B=1 train(): ms per user (ML-1M shape, synthetic)", "ms_per_user": out,
                  "reference_ms_per_batch_nb882": 13.8, "mean_len": float(np.mean([len(u) for u in users]))}))
