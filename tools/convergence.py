#!/usr/bin/env python3
"""Does training in the TIMED mode (bf16 operands) reach the metrics of the fp32-class mode?

north_star: "Recall/NDCG within 0.5 % absolute ... matching reference loss and metrics".  The parity tests compare one
step (or twelve) against the reference; this trains the same model twice -- PREC_BF16 (what bench.py times) and
PREC_SPLIT (the same kernel templates at fp32-class precision, pinned to the reference at the fp32 gates) -- from the
same initial parameters, over the same batches, with the same Philox noise, through the notebook's own loop
(`train_batched` per epoch, `evaluate` on held-out users after every epoch: nb:1274-1307, nb:594-666), and sets the two
validation curves side by side.

    python tools/convergence.py [--steps 320] [--train-users 2560] [--val-users 320] [--batch 64] [--out FILE]

ML-1M-shaped synthetic users (synth.SHAPES["ml-1m"]: 3 483 items, lognormal(4.0, 1.0) lengths); held-out users are
split 80 / 20 chronologically as the reference's preprocessing does (nb:180-207).
"""
import argparse
import json
import os
import re
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def hyper_params(I, **kw):
    hp = {"history_split_test": [0.8, 0.2], "weight_decay": 5e-3, "batch_size": 1, "item_embed_size": 256,
          "rnn_size": 200, "hidden_size": 150, "latent_size": 64, "loss_type": "next_k", "next_k": 4,
          "number_users_to_keep": 10 ** 9, "batch_log_interval": 10 ** 9, "train_cp_users": 200, "total_items": I,
          "optimizer": "adam"}
    hp.update(kw)
    return hp


def _lines(users, first_uid=0):
    return ["%d,%d,1\n" % (first_uid + u, int(it)) for u, items in enumerate(users) for it in items]


def make_readers(n_train, n_val, I, hp, lmax=200, seed=98765):
    from svae_cf_amd import notebook_api as N
    from svae_cf_amd import synth
    _, _, mu, sigma, _ = synth.SHAPES["ml-1m"]
    users = synth.make_users(n_train + n_val, I, mu, sigma, lmax, seed=seed)
    tr, va = users[:n_train], users[n_train:]
    fold = [u[:int(0.8 * len(u))] for u in va]
    held = [u[int(0.8 * len(u)):] for u in va]
    train_reader = N.DataReader(hp, _lines(tr), None, I, True)
    val_reader = N.DataReader(hp, _lines(fold, n_train), _lines(held, n_train), I, False)
    return train_reader, val_reader


def run_mode(prec, train_reader, val_reader, hp, steps, batch, init_seed=0, noise_seed=5, say=None, eval_every=1,
             eval_seed=5):
    """Trains `steps` optimizer steps (whole epochs of train_batched) in `prec`; after every epoch evaluates the
    validation users.  Returns dict(step_loss=[...], epochs=[{step, loss, NDCG@100, Rec@100, ...}], params=state)."""
    from svae_cf_amd import notebook_api as N
    model = N.Model(hp, prec=prec, seed=init_seed)
    crit = N.VAELoss(hp, model)
    N.make_optimizer(model, hp)
    hp_log = dict(hp, batch_log_interval=batch)           # train_batched logs every batch_log_interval // B steps
    step_loss, epochs, done, epoch = [], [], 0, 0
    t0 = time.time()
    import faulthandler
    while done < steps:
        epoch += 1
        # an epoch takes well under a second: one that has not ended after 60 s is a hang -- say where, then leave
        faulthandler.dump_traceback_later(60, exit=True)
        logs = []
        n = N.train_batched(train_reader, model, hp_log, batch_size=batch, epoch=epoch, seed=noise_seed, log=logs.append)
        step_loss += [float(re.search(r"loss +([-\d.einf]+)", s).group(1)) for s in logs]
        done += n
        if epoch % eval_every and done < steps:
            continue
        metrics, _ = N.evaluate(model, crit, val_reader, hp, False, batch_size=256, seed=eval_seed)
        rec = dict(step=done, epoch=epoch, **metrics)
        epochs.append(rec)
        if say is not None:
            say("prec %d epoch %2d step %4d train loss %.4f | val loss %.4f NDCG@100 %.4f Rec@100 %.4f (%.1f s)"
                % (prec, epoch, done, step_loss[-1], metrics["loss"], metrics["NDCG@100"], metrics["Rec@100"],
                   time.time() - t0))
    faulthandler.cancel_dump_traceback_later()
    torch.cuda.synchronize()
    return dict(step_loss=step_loss, epochs=epochs, params={k: v.cpu() for k, v in model.state_dict().items()})


def plateau(run, first_step, keys=("loss", "NDCG@10", "Rec@10", "NDCG@100", "Rec@100")):
    """Mean and standard deviation of the validation metrics over the evaluations at steps >= first_step."""
    ep = [e for e in run["epochs"] if e["step"] >= first_step]
    return {k: dict(mean=float(np.mean([e[k] for e in ep])), std=float(np.std([e[k] for e in ep], ddof=1)), n=len(ep))
            for k in keys}


def compare(a, b):
    """a = timed mode (bf16), b = fp32-class mode: the figures the gate is written on."""
    la, lb = np.asarray(a["step_loss"]), np.asarray(b["step_loss"])
    out = dict(steps=int(len(la)),
               train_loss_rel_max=float(np.max(np.abs(la - lb) / np.abs(lb))),
               train_loss_rel_last=float(abs(la[-1] - lb[-1]) / abs(lb[-1])),
               val_loss_rel_max=max(abs(x["loss"] - y["loss"]) / abs(y["loss"]) for x, y in zip(a["epochs"], b["epochs"])))
    for key in ("NDCG@10", "Rec@10", "Prec@10", "NDCG@100", "Rec@100", "Prec@100"):
        out["abs_diff_max " + key] = max(abs(x[key] - y[key]) for x, y in zip(a["epochs"], b["epochs"]))
        out["abs_diff_last " + key] = abs(a["epochs"][-1][key] - b["epochs"][-1][key])
    num = sum(float((a["params"][k].double() - b["params"][k].double()).pow(2).sum()) for k in a["params"])
    den = sum(float(b["params"][k].double().pow(2).sum()) for k in b["params"])
    out["params_rel_l2"] = float(np.sqrt(num / den))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=320)
    ap.add_argument("--train-users", type=int, default=2560)
    ap.add_argument("--val-users", type=int, default=320)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--items", type=int, default=3483)
    ap.add_argument("--lmax", type=int, default=200)
    ap.add_argument("--modes", default="bf16,split,fp32",
                    help="bf16 = the timed mode; split = its kernel templates at fp32-class precision; fp32 = the generic "
                         "exact-fp32 kernels with materialised logits (a second fp32-class implementation: how far two "
                         "of THOSE drift apart over the same steps is the yardstick for bf16's drift)")
    ap.add_argument("--eval-every", type=int, default=1, help="validation every this many epochs (and after the last)")
    ap.add_argument("--plateau-from", type=int, default=2560, help="first step of the plateau statistics")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import svae_cf_amd as S
    hp = hyper_params(args.items)
    tr, va = make_readers(args.train_users, args.val_users, args.items, hp, lmax=args.lmax)
    res = {}
    precs = dict(bf16=S.PREC_BF16, split=S.PREC_SPLIT, fp32=S.PREC_FP32)
    for name in args.modes.split(","):
        # "split@6": the same mode with another seed of the TRAINING noise (the latent eps of nb:744; validation noise
        # unchanged) -- how far two runs of one fp32-class implementation sit apart is the yardstick for bf16's offset
        mode, _, sd = name.partition("@")
        res[name] = run_mode(precs[mode], tr, va, hp, args.steps, args.batch, say=lambda s: print(s, flush=True),
                             eval_every=args.eval_every, noise_seed=int(sd) if sd else 5)
    base = "split" if "split" in res else sorted(res)[-1]
    cmp_ = {"%s vs %s" % (m, base): compare(res[m], res[base]) for m in res if m != base}
    plat = {m: plateau(res[m], args.plateau_from) for m in res} if args.steps > args.plateau_from else {}
    for m in plat:
        print("plateau %-8s " % m + "  ".join("%s %.3f +- %.3f" % (k, v["mean"], v["std"]) for k, v in plat[m].items()))
    rep = dict(plateau=plat, config=dict(vars(args), shape="ML-1M-shaped synthetic", optimizer="adam, lr 1e-3, L2 5e-3 (nb:1260-1263)",
                           anneal="per user, reference schedule (train_batched anneal_by='users')"),
               comparison=cmp_,
               curves={m: dict(step_loss=res[m]["step_loss"], epochs=res[m]["epochs"]) for m in res})
    print(json.dumps(cmp_, indent=1))
    if args.out:
        with open(args.out, "w") as f:
            json.dump(rep, f, indent=1)


if __name__ == "__main__":
    main()
