"""CPU oracle for the SVAE training / scoring step  --  TEST INFRASTRUCTURE ONLY.

This file is a plain-numpy restatement of the algorithm of the reference
notebook ``main_svae.ipynb`` (citations ``nb:N`` = raw line N of that file).
It is the checker the HIP path is compared against.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it; the product package ``svae_cf_amd`` never does.

Parity status: PINNED.  ``oracle/make_golden.py`` executes the reference's own
``Model`` / ``VAELoss`` / ``DataReader`` / ``evaluate`` / ``train`` code (from
the notebook, in the build container) and stores inputs+outputs under
``tests/golden/``; ``tests/test_oracle_golden.py`` checks every function here
against those vectors.

Everything is written for ONE user sequence (the reference runs batch_size 1,
nb:80); ``batch_loss_and_grads`` gives the batched contract used by the GPU
path: a step over a set S of users == mean over u in S of the reference's
single-user loss (and of its gradients).
"""
import math

import numpy as np

# state_dict names / shapes exactly as the reference registers them (nb:684-732)
PARAM_NAMES = (
    "encoder.linear1.weight", "encoder.linear1.bias",
    "decoder.linear1.weight", "decoder.linear1.bias",
    "decoder.linear2.weight", "decoder.linear2.bias",
    "item_embed.weight",
    "gru.weight_ih_l0", "gru.weight_hh_l0", "gru.bias_ih_l0", "gru.bias_hh_l0",
    "linear1.weight", "linear1.bias",
)


def param_shapes(total_items, embed=256, rnn=200, hidden=150, latent=64):
    I, E, H, D, L = total_items, embed, rnn, hidden, latent
    return {
        "encoder.linear1.weight": (D, H), "encoder.linear1.bias": (D,),
        "decoder.linear1.weight": (D, L), "decoder.linear1.bias": (D,),
        "decoder.linear2.weight": (I, D), "decoder.linear2.bias": (I,),
        "item_embed.weight": (I, E),
        "gru.weight_ih_l0": (3 * H, E), "gru.weight_hh_l0": (3 * H, H),
        "gru.bias_ih_l0": (3 * H,), "gru.bias_hh_l0": (3 * H,),
        "linear1.weight": (2 * L, D), "linear1.bias": (2 * L,),
    }


def make_params(total_items, seed, dtype=np.float32, **dims):
    """Deterministic parameter set (legacy RandomState: stable across numpy versions).

    Scales resemble the reference init (nb:690,703,704,730 xavier-normal for
    Linear weights, N(0,1) embedding, U(+-1/sqrt(H)) GRU) but are drawn from our
    own stream; parity tests load these values into the reference modules.
    """
    rs = np.random.RandomState(seed)
    shapes = param_shapes(total_items, **dims)
    H = shapes["gru.weight_hh_l0"][1]
    out = {}
    for name in PARAM_NAMES:
        shp = shapes[name]
        if name == "item_embed.weight":
            v = rs.standard_normal(shp)
        elif name.startswith("gru."):
            v = rs.uniform(-1.0, 1.0, shp) / math.sqrt(H)
        elif name.endswith(".weight"):
            # a bit larger than xavier so logits / latents are not degenerate
            v = rs.standard_normal(shp) * math.sqrt(2.0 / (shp[0] + shp[1])) * 1.5
        else:
            v = rs.uniform(-1.0, 1.0, shp) * 0.1
        out[name] = np.ascontiguousarray(v.astype(dtype))
    return out


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


# --------------------------------------------------------------------------
# targets (DataReader.iter, nb:485-526; next_k branch nb:509-513)
# --------------------------------------------------------------------------
def build_inputs_targets(items, next_k):
    """items: the user's chronological item list (len n >= 2).

    Returns x [T] (= items[:-1], nb:521) and tgt [T, k] int32 where row t holds
    items[t+1 : t+1+k] (nb:512) padded with -1.  ``loss_type`` predict_next is
    next_k = 1 (nb:503-507); postfix is next_k >= n (nb:515-519).
    """
    items = np.asarray(items, dtype=np.int64)
    T = len(items) - 1
    k = int(min(next_k, T))
    x = items[:-1].copy()
    tgt = np.full((T, max(k, 1)), -1, dtype=np.int32)
    for t in range(T):
        nxt = items[t + 1: t + 1 + next_k]
        tgt[t, :len(nxt)] = nxt
    return x, tgt


def dense_targets(tgt, total_items, dtype=np.float32):
    """The reference's dense one-hot y_s [T, I] (scatter_ of 1.0, nb:510-513)."""
    T = tgt.shape[0]
    y = np.zeros((T, total_items), dtype=dtype)
    for t in range(T):
        idx = tgt[t][tgt[t] >= 0]
        y[t, idx] = 1.0          # scatter_ writes 1.0: duplicates do not add up
    return y


# --------------------------------------------------------------------------
# forward (Model.forward nb:752-768)
# --------------------------------------------------------------------------
def gru_forward(emb, p, dtype):
    """nn.GRU semantics (1 layer, h0 = 0, gate order r,z,n; nb:724-727, nb:759)."""
    W_ih, W_hh = p["gru.weight_ih_l0"], p["gru.weight_hh_l0"]
    b_ih, b_hh = p["gru.bias_ih_l0"], p["gru.bias_hh_l0"]
    H = W_hh.shape[1]
    T = emb.shape[0]
    gi = emb @ W_ih.T + b_ih                    # [T, 3H]
    h = np.zeros(H, dtype=dtype)
    hs = np.zeros((T, H), dtype=dtype)
    r_all = np.zeros((T, H), dtype=dtype)
    z_all = np.zeros((T, H), dtype=dtype)
    n_all = np.zeros((T, H), dtype=dtype)
    ghn_all = np.zeros((T, H), dtype=dtype)
    for t in range(T):
        gh = W_hh @ h + b_hh
        r = sigmoid(gi[t, :H] + gh[:H])
        z = sigmoid(gi[t, H:2 * H] + gh[H:2 * H])
        n = np.tanh(gi[t, 2 * H:] + r * gh[2 * H:])
        h = (1.0 - z) * n + z * h
        hs[t], r_all[t], z_all[t], n_all[t], ghn_all[t] = h, r, z, n, gh[2 * H:]
    return hs, dict(r=r_all, z=z_all, n=n_all, ghn=ghn_all)


def forward(p, x, eps, dtype=np.float32):
    """Returns dict with dec_out [T,I], mu [T,L], log_sigma [T,L] and the cache.

    eps is the N(0,1) noise the reference draws with np.random.normal (nb:744).
    """
    p = {k: v.astype(dtype) for k, v in p.items()}
    eps = np.asarray(eps, dtype=dtype)
    L = p["linear1.weight"].shape[0] // 2
    emb = p["item_embed.weight"][x]                                     # nb:756
    hs, gc = gru_forward(emb, p, dtype)                                 # nb:759
    enc = np.tanh(hs @ p["encoder.linear1.weight"].T + p["encoder.linear1.bias"])   # nb:694-695
    tmp = enc @ p["linear1.weight"].T + p["linear1.bias"]               # nb:738
    mu, log_sigma = tmp[:, :L], tmp[:, L:]                              # nb:740-741
    sigma = np.exp(log_sigma)                                           # nb:743
    z = mu + sigma * eps                                                # nb:750
    d1 = np.tanh(z @ p["decoder.linear1.weight"].T + p["decoder.linear1.bias"])     # nb:708-709
    dec_out = d1 @ p["decoder.linear2.weight"].T + p["decoder.linear2.bias"]        # nb:710
    cache = dict(x=x, emb=emb, hs=hs, gru=gc, enc=enc, sigma=sigma, eps=eps, z=z, d1=d1, p=p)
    return dict(dec_out=dec_out, mu=mu, log_sigma=log_sigma, cache=cache)


def log_softmax(a):
    m = a.max(axis=-1, keepdims=True)
    s = a - m
    return s - np.log(np.exp(s).sum(axis=-1, keepdims=True))


# --------------------------------------------------------------------------
# loss (VAELoss.forward nb:795-812)
# --------------------------------------------------------------------------
def vae_loss(dec_out, mu, log_sigma, tgt, anneal, batch_size=1):
    """tgt [T,k] int (-1 pad).  Returns (loss, kld, likelihood, num_ones)."""
    kld = np.mean(np.sum(0.5 * (-log_sigma + np.exp(log_sigma) + mu ** 2 - 1.0), axis=-1))  # nb:797
    logp = log_softmax(dec_out)                                                             # nb:802
    y = dense_targets(tgt, dec_out.shape[-1], dtype=dec_out.dtype)
    num_ones = float(y[0].sum())                                                            # nb:803
    lik = float(np.sum(-1.0 * y * logp)) / (float(batch_size) * num_ones)                   # nb:805-808
    return anneal * kld + lik, kld, lik, num_ones                                           # nb:810


# --------------------------------------------------------------------------
# backward (what autograd does at nb:1205), written out by hand
# --------------------------------------------------------------------------
def backward(fwd, tgt, anneal, scale=1.0, batch_size=1):
    """Gradient of scale * VAELoss w.r.t. every parameter.  Returns dict name->grad."""
    c = fwd["cache"]
    p = c["p"]
    dtype = fwd["dec_out"].dtype
    T, I = fwd["dec_out"].shape
    L = fwd["mu"].shape[1]
    H = p["gru.weight_hh_l0"].shape[1]
    g = {}

    y = dense_targets(tgt, I, dtype=dtype)
    num_ones = float(y[0].sum())
    n_r = y.sum(axis=1, keepdims=True)
    sm = np.exp(log_softmax(fwd["dec_out"]))
    dlogits = (n_r * sm - y) * dtype.type(scale / (batch_size * num_ones))

    g["decoder.linear2.weight"] = dlogits.T @ c["d1"]
    g["decoder.linear2.bias"] = dlogits.sum(axis=0)
    dd1 = dlogits @ p["decoder.linear2.weight"]
    dd1p = dd1 * (1.0 - c["d1"] ** 2)
    g["decoder.linear1.weight"] = dd1p.T @ c["z"]
    g["decoder.linear1.bias"] = dd1p.sum(axis=0)
    dz = dd1p @ p["decoder.linear1.weight"]

    kscale = dtype.type(scale * anneal / T)
    dmu = dz + kscale * fwd["mu"]
    dls = dz * c["sigma"] * c["eps"] + kscale * 0.5 * (np.exp(fwd["log_sigma"]) - 1.0)
    dtmp = np.concatenate([dmu, dls], axis=1)
    g["linear1.weight"] = dtmp.T @ c["enc"]
    g["linear1.bias"] = dtmp.sum(axis=0)
    denc = dtmp @ p["linear1.weight"]
    dencp = denc * (1.0 - c["enc"] ** 2)
    g["encoder.linear1.weight"] = dencp.T @ c["hs"]
    g["encoder.linear1.bias"] = dencp.sum(axis=0)
    dhs = dencp @ p["encoder.linear1.weight"]

    # BPTT through the GRU
    W_hh = p["gru.weight_hh_l0"]
    gc = c["gru"]
    dgi = np.zeros((T, 3 * H), dtype=dtype)
    dgh = np.zeros((T, 3 * H), dtype=dtype)
    dh = np.zeros(H, dtype=dtype)
    for t in range(T - 1, -1, -1):
        dh = dh + dhs[t]
        r, z, n, ghn = gc["r"][t], gc["z"][t], gc["n"][t], gc["ghn"][t]
        hprev = c["hs"][t - 1] if t > 0 else np.zeros(H, dtype=dtype)
        dn = dh * (1.0 - z)
        dzg = dh * (hprev - n)
        dnp = dn * (1.0 - n * n)
        drg = dnp * ghn
        dzp = dzg * z * (1.0 - z)
        drp = drg * r * (1.0 - r)
        dgi[t] = np.concatenate([drp, dzp, dnp])
        dgh[t] = np.concatenate([drp, dzp, dnp * r])
        dh = dh * z + dgh[t] @ W_hh
    hprev_all = np.concatenate([np.zeros((1, H), dtype=dtype), c["hs"][:-1]], axis=0)
    g["gru.weight_ih_l0"] = dgi.T @ c["emb"]
    g["gru.weight_hh_l0"] = dgh.T @ hprev_all
    g["gru.bias_ih_l0"] = dgi.sum(axis=0)
    g["gru.bias_hh_l0"] = dgh.sum(axis=0)
    demb = dgi @ p["gru.weight_ih_l0"]
    ge = np.zeros_like(p["item_embed.weight"])
    np.add.at(ge, c["x"], demb)                 # dense grad, duplicates accumulate
    g["item_embed.weight"] = ge
    return g


def loss_and_grads(p, items, eps, anneal, next_k=4, dtype=np.float32, scale=1.0):
    x, tgt = build_inputs_targets(items, next_k)
    fwd = forward(p, x, eps, dtype)
    loss, kld, lik, _ = vae_loss(fwd["dec_out"], fwd["mu"], fwd["log_sigma"], tgt, anneal)
    grads = backward(fwd, tgt, anneal, scale=scale)
    return loss, grads, fwd


def batch_loss_and_grads(p, users, eps_list, anneal, next_k=4, dtype=np.float32):
    """Batched contract: mean over users of the reference single-user loss / grads."""
    B = len(users)
    tot, G = 0.0, None
    for items, eps in zip(users, eps_list):
        loss, g, _ = loss_and_grads(p, items, eps, anneal, next_k, dtype, scale=1.0 / B)
        tot += float(loss) / B
        G = g if G is None else {k: G[k] + g[k] for k in G}
    return tot, G


# --------------------------------------------------------------------------
# Adam with L2 weight decay (torch.optim.Adam defaults, nb:1260-1263, step nb:1206)
# --------------------------------------------------------------------------
def adam_init(p):
    return dict(step=0, m={k: np.zeros_like(v) for k, v in p.items()},
                v={k: np.zeros_like(v) for k, v in p.items()})


def adam_step(p, grads, st, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8, weight_decay=5e-3):
    st["step"] += 1
    t = st["step"]
    bc1 = 1.0 - b1 ** t
    bc2 = 1.0 - b2 ** t
    for k in p:
        dt = p[k].dtype.type
        g = grads[k] + dt(weight_decay) * p[k]
        st["m"][k] = dt(b1) * st["m"][k] + dt(1.0 - b1) * g
        st["v"][k] = dt(b2) * st["v"][k] + dt(1.0 - b2) * g * g
        denom = np.sqrt(st["v"][k]) / dt(math.sqrt(bc2)) + dt(eps)
        p[k] = p[k] - dt(lr / bc1) * (st["m"][k] / denom)
    return p


# --------------------------------------------------------------------------
# the notebook's other optimizers (nb:1252-1259, nb:1264-1267): torch.optim.Adagrad(lr=hyper_params
# ['learning_rate'], weight_decay), Adadelta(weight_decay), RMSprop(weight_decay) -- torch's defaults
# otherwise (Adagrad lr_decay 0, eps 1e-10; Adadelta lr 1.0, rho 0.9, eps 1e-6; RMSprop lr 0.01,
# alpha 0.99, eps 1e-8, no momentum, not centered).  All add L2 decay to the gradient first.
# --------------------------------------------------------------------------
def opt_init(p):
    return dict(step=0, a={k: np.zeros_like(v) for k, v in p.items()},
                b={k: np.zeros_like(v) for k, v in p.items()})


def adagrad_step(p, grads, st, lr=1e-2, lr_decay=0.0, eps=1e-10, weight_decay=5e-3):
    st["step"] += 1
    clr = lr / (1.0 + (st["step"] - 1) * lr_decay)
    for k in p:
        dt = p[k].dtype.type
        g = grads[k] + dt(weight_decay) * p[k]
        st["a"][k] = st["a"][k] + g * g                       # state_sum
        p[k] = p[k] - dt(clr) * (g / (np.sqrt(st["a"][k]) + dt(eps)))
    return p


def adadelta_step(p, grads, st, lr=1.0, rho=0.9, eps=1e-6, weight_decay=5e-3):
    st["step"] += 1
    for k in p:
        dt = p[k].dtype.type
        g = grads[k] + dt(weight_decay) * p[k]
        st["a"][k] = dt(rho) * st["a"][k] + dt(1.0 - rho) * g * g          # square_avg
        std = np.sqrt(st["a"][k] + dt(eps))
        delta = np.sqrt(st["b"][k] + dt(eps)) / std * g
        st["b"][k] = dt(rho) * st["b"][k] + dt(1.0 - rho) * delta * delta  # acc_delta
        p[k] = p[k] - dt(lr) * delta
    return p


def rmsprop_step(p, grads, st, lr=1e-2, alpha=0.99, eps=1e-8, weight_decay=5e-3):
    st["step"] += 1
    for k in p:
        dt = p[k].dtype.type
        g = grads[k] + dt(weight_decay) * p[k]
        st["a"][k] = dt(alpha) * st["a"][k] + dt(1.0 - alpha) * g * g      # square_avg
        p[k] = p[k] - dt(lr) * (g / (np.sqrt(st["a"][k]) + dt(eps)))
    return p


def anneal_for_batch(b, total_anneal_steps=200000, cap=0.2):
    """Anneal used by the b-th step (1-based) of an epoch (nb:1188-1191, nb:1211-1215).

    The schedule's locals restart every epoch; the value is updated AFTER use, so
    steps 1 and 2 both see 0.0 and step b sees min(cap, (b-2)/total).
    """
    if b <= 2:
        return 0.0
    return min(cap, 1.0 * float(b - 2) / total_anneal_steps)


# --------------------------------------------------------------------------
# scoring (evaluate nb:594-666)
# --------------------------------------------------------------------------
def masked_scores(last_logits, x):
    """s - |s * mask| * 1e8 with mask = one-hot of the INPUT items (nb:621-624)."""
    m = np.zeros_like(last_logits)
    m[np.asarray(x)] = 1.0
    return last_logits - np.abs(last_logits * m) * last_logits.dtype.type(100000000)


def topk_desc(scores, k):
    """Descending order by score, ties by lower index (torch.sort(-s) is stable here)."""
    order = np.argsort(-scores, kind="stable")
    return order[:k]


def user_metrics(rec_sorted, heldout, Ks=(10, 100)):
    """Loop body nb:633-648 for one user.  rec_sorted = descending item ranking."""
    out = {}
    for k in Ks:
        rec_list = list(rec_sorted[:k])
        best, now_at, dcg, hits = 0.0, 0.0, 0.0, 0.0
        for movie in heldout:
            now_at += 1.0
            if now_at <= k:
                best += 1.0 / float(np.log2(now_at + 1))
            if movie not in rec_list:
                continue
            hits += 1.0
            dcg += 1.0 / float(np.log2(float(rec_list.index(movie) + 2)))
        out["NDCG@%d" % k] = float(dcg) / float(best)
        out["Rec@%d" % k] = float(hits) / float(len(heldout))
        out["Prec@%d" % k] = float(hits) / float(k)
    return out


def evaluate_users(p, users_foldin, users_heldout, eps_list, next_k=4, dtype=np.float32,
                   Ks=(10, 100), loss_divisor=None):
    """evaluate() (nb:594-666): loss at anneal 0.2, masked top-k metrics.

    loss_divisor: the reference divides the loss sum by ``batch`` (nb:658), which is
    one more than the users scored when the train-set early break fired (nb:612-613).
    Also returns len_to_ndcg_at_100_map keyed by n_heldout + len(x) + 1 (nb:651-654).
    """
    metrics = {"loss": 0.0}
    for k in Ks:
        for nm in ("NDCG", "Rec", "Prec"):
            metrics["%s@%d" % (nm, k)] = 0.0
    n = 0
    len_map = {}
    for items, held, eps in zip(users_foldin, users_heldout, eps_list):
        x, tgt = build_inputs_targets(items, next_k)
        fwd = forward(p, x, eps, dtype)
        loss, _, _, _ = vae_loss(fwd["dec_out"], fwd["mu"], fwd["log_sigma"], tgt, 0.2)
        metrics["loss"] += float(loss)
        s = masked_scores(fwd["dec_out"][-1], x)
        rec = topk_desc(s, max(Ks))
        um = user_metrics(rec, held, Ks)
        for kk, v in um.items():
            metrics[kk] += v
        len_map.setdefault(len(held) + len(x) + 1, []).append(um["NDCG@100"])
        n += 1
    metrics["loss"] = round(metrics["loss"] / float(loss_divisor or n), 4)  # nb:658-659
    for k in Ks:
        for nm in ("NDCG", "Rec", "Prec"):
            key = "%s@%d" % (nm, k)
            metrics[key] = round(100.0 * metrics[key] / float(n), 4)        # nb:661-664
    return metrics, len_map
