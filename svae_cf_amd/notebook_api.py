"""The notebook's own surface -- Model / VAELoss / DataReader / train / evaluate / load_data -- on
top of the HIP engine, so the reference's driver cell (nb:1235-1307, nb:1366-1382) runs unchanged
except for its imports.  Same names, argument meaning and return values as main_svae.ipynb.

Two ways to train:
  * notebook style (batch_size 1, nb:80): ``train(reader)`` -> per user ``model(x)``,
    ``criterion(...)``, ``loss.backward()``, ``optimizer.step()`` -- each of them runs HIP kernels
    through torch.autograd.Function bridges (dense ``dec_out`` as in nb:768);
  * batched: ``train_batched(...)`` packs B users per step (engine.train_step, fused kernels).
``evaluate`` always batches users for scoring and reproduces the reference's metric arithmetic
(nb:626-664) on the top-100 lists.
"""
import math
import time

import numpy as np
import torch

from . import _lib
from .batch import build_plan
from .engine import PREC_BF16, PREC_FP32, SVAEEngine, _stream
from .layout import L


def _xavier_normal(shape, gen):
    std = math.sqrt(2.0 / (shape[0] + shape[1]))
    return torch.randn(shape, generator=gen) * std


def reference_init(total_items, hyper_params, seed=None):
    """Parameter init with the reference's distributions (nb:690, 703-704, 730 xavier-normal Linear
    weights; torch defaults elsewhere: Linear bias U(+-1/sqrt(fan_in)), Embedding N(0,1), GRU
    U(+-1/sqrt(H)))."""
    # seed None: draw from torch's GLOBAL generator, as the reference's Model.__init__ does (reproducible under a
    # preceding torch.manual_seed, and the caller's RNG state is advanced, not replaced)
    gen = torch.Generator().manual_seed(int(seed)) if seed is not None else None
    E, H = hyper_params["item_embed_size"], hyper_params["rnn_size"]
    D, Lz = hyper_params["hidden_size"], hyper_params["latent_size"]
    I = total_items

    def u(shape, bound):
        return (torch.rand(shape, generator=gen) * 2 - 1) * bound
    k = 1.0 / math.sqrt(H)
    return {
        "encoder.linear1.weight": _xavier_normal((D, H), gen), "encoder.linear1.bias": u((D,), 1 / math.sqrt(H)),
        "decoder.linear1.weight": _xavier_normal((D, Lz), gen), "decoder.linear1.bias": u((D,), 1 / math.sqrt(Lz)),
        "decoder.linear2.weight": _xavier_normal((I, D), gen), "decoder.linear2.bias": u((I,), 1 / math.sqrt(D)),
        "item_embed.weight": torch.randn((I, E), generator=gen),
        "gru.weight_ih_l0": u((3 * H, E), k), "gru.weight_hh_l0": u((3 * H, H), k),
        "gru.bias_ih_l0": u((3 * H,), k), "gru.bias_hh_l0": u((3 * H,), k),
        "linear1.weight": _xavier_normal((2 * Lz, D), gen), "linear1.bias": u((2 * Lz,), 1 / math.sqrt(D)),
    }


class SparseTargets:
    """What DataReader yields instead of the dense one-hot y_s [1,T,I] (nb:500-513): the int32
    table tgt [T, k] (-1 padded).  ``dense()`` gives the reference tensor when a caller wants it."""

    def __init__(self, tgt, total_items):
        self.tgt = tgt
        self.total_items = total_items

    def dense(self, device="cpu"):
        T, K = self.tgt.shape
        y = torch.zeros(1, T, self.total_items, device=device)
        t = torch.as_tensor(self.tgt, device=device).long()
        rows = torch.arange(T, device=device)[:, None].expand(T, K)
        m = t >= 0
        y[0, rows[m], t[m]] = 1.0
        return y


def _targets_from_dense(y):
    """Dense one-hot [1,T,I] (any device) -> int32 [T,K] table, -1 padded."""
    y2 = y[0]
    K = int(y2.sum(-1).max().item())
    vals, idx = torch.topk(y2, max(K, 1), dim=-1)
    idx = idx.to(torch.int32)
    idx[vals <= 0] = -1
    return idx.contiguous()


# --------------------------------------------------------------------------------------------
# autograd bridges
# --------------------------------------------------------------------------------------------
class _ForwardFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, flat, model, plan, eps):
        eng = model.engine
        eng.forward(plan, 0.0, eps=eps, seed=model.noise_seed, step=model._calls, want_loss=False)
        R = plan.R
        dec = eng.dense_logits()[:, :eng.I].clone()
        mu, ls = eng.mu_logsigma(R)
        ctx.model, ctx.plan = model, plan
        # the engine keeps ONE set of activations (its workspace): backward is only valid for the forward that
        # wrote them.  Stamp this forward; backward refuses if another forward ran in between.
        ctx.generation = eng.generation
        return dec.view(1, R, eng.I), mu.clone(), ls.clone()

    @staticmethod
    def backward(ctx, d_dec, d_mu, d_ls):
        eng, plan = ctx.model.engine, ctx.plan
        if eng.generation != ctx.generation:
            raise RuntimeError(
                "Model.forward was called again (or the engine ran another batch) before this backward: the engine "
                "keeps the activations of its LAST forward only, and gradients are written, not accumulated -- one "
                "model(x) per loss.backward() (the notebook's train loop, nb:1201-1206); two forwards before a "
                "backward, or gradient accumulation over several calls, are not supported on this surface")
        eng.backward_dense(plan, d_dec.reshape(plan.R, eng.I), d_mu, d_ls)
        flat = ctx.model.flat
        return eng.grads, None, None, None


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dec, mu, ls, model, tgt, anneal, batch_size):
        eng = model.engine
        R, I = dec.shape[1], dec.shape[2]
        loss, dlogits, dtmp = eng.dense_loss(dec.reshape(R, I), mu, ls, tgt, float(anneal), batch_size)
        ctx.save_for_backward(dlogits, dtmp)
        ctx.shape = dec.shape
        return loss

    @staticmethod
    def backward(ctx, g):
        dlogits, dtmp = ctx.saved_tensors
        return (g * dlogits).view(ctx.shape), g * dtmp[:, :L], g * dtmp[:, L:], None, None, None, None


class Encoder:      # kept for name parity with nb:684-696; the layers live inside Model's engine
    pass


class Decoder:      # nb:698-711
    pass


class Model(torch.nn.Module):
    """Drop-in for the notebook's Model (nb:713-768).  ``hyper_params`` needs total_items and, if
    present, must carry the default sizes (256/200/150/64) the kernels are compiled for."""

    def __init__(self, hyper_params, prec=PREC_FP32, device="cuda", seed=None):
        super().__init__()
        self.hyper_params = hyper_params
        want = dict(item_embed_size=256, rnn_size=200, hidden_size=150, latent_size=64)
        for k, v in want.items():
            if hyper_params.get(k, v) != v:
                raise ValueError("kernels are built for %s=%d (reference default), got %r" % (k, v, hyper_params[k]))
        hp = dict(want)
        hp.update({k: hyper_params[k] for k in want if k in hyper_params})
        self.engine = SVAEEngine(hyper_params["total_items"], device=device, prec=prec,
                                 weight_decay=hyper_params.get("weight_decay", 5e-3))
        self.engine.fused = False if prec == PREC_FP32 else self.engine.fused
        self.engine.load_state_dict(reference_init(hyper_params["total_items"], hp, seed))
        self.flat = torch.nn.Parameter(self.engine.params, requires_grad=True)
        self.noise_seed = 0 if seed is None else int(seed)
        self._calls = 0
        self.z_mean = self.z_log_sigma = None
        self.eps_source = None      # optional callable (T, L) -> numpy noise (parity tests replay nb:744)

    # reference-shaped state_dict (names / shapes of nb:718-729)
    def state_dict(self, *a, **k):
        return self.engine.state_dict()

    def load_state_dict(self, sd, strict=True):
        self.engine.load_state_dict(sd)

    def cuda(self, device=None):
        return self

    def forward(self, x):
        """x: int64 [1, T].  Returns (dec_out [1,T,I], z_mean [T,64], z_log_sigma [T,64]) as nb:768."""
        if x.dim() != 2 or x.shape[0] != 1:
            raise ValueError("batch_size must be 1 on this surface (nb:80); use train_batched for more")
        items = np.concatenate([x[0].detach().cpu().numpy().astype(np.int64), [0]])   # last item is not an input
        plan = build_plan([items], 1, device=self.engine.device)
        eps = None
        if self.eps_source is not None:
            eps = torch.as_tensor(np.asarray(self.eps_source(plan.R, L), dtype=np.float32)).to(self.engine.device)
        self._calls += 1
        self._last_x = x
        dec, mu, ls = _ForwardFn.apply(self.flat, self, plan, eps)
        self.z_mean, self.z_log_sigma = mu, ls
        return dec, mu, ls


class VAELoss(torch.nn.Module):
    """Drop-in for nb:790-812.  ``y_true_s`` may be the reference's dense one-hot or SparseTargets."""

    def __init__(self, hyper_params, model=None):
        super().__init__()
        self.hyper_params = hyper_params
        self.model = model

    def forward(self, decoder_output, mu_q, logvar_q, y_true_s, anneal):
        model = self.model
        if model is None:
            raise RuntimeError("VAELoss needs the Model whose engine runs the kernels: VAELoss(hp, model)")
        if isinstance(y_true_s, SparseTargets):
            tgt = torch.as_tensor(y_true_s.tgt, dtype=torch.int32, device=decoder_output.device)
        else:
            tgt = _targets_from_dense(y_true_s).to(decoder_output.device)
        return _LossFn.apply(decoder_output, mu_q, logvar_q, model, tgt, anneal,
                             float(self.hyper_params.get("batch_size", 1)))


class Adam:
    """torch.optim.Adam(model.parameters(), weight_decay=wd) look-alike (nb:1260-1263) that runs the fused
    HIP Adam over the engine's flat buffers."""

    def __init__(self, model_or_params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, model=None):
        self.model = model if model is not None else model_or_params
        if not isinstance(self.model, Model):
            raise TypeError("pass the Model: Adam(model, weight_decay=...)")
        e = self.model.engine
        e.set_optimizer("adam", lr, weight_decay, eps=eps, betas=betas)

    def zero_grad(self):
        pass        # gradients are written, never accumulated (nb:1197-1198 become no-ops)

    def step(self):
        self.model.engine.adam_step()


class _EngineOptimizer:
    """Shared shape of the look-alikes below: construction selects the engine's optimizer kernel and clears its state,
    zero_grad is a no-op (gradients are written, never accumulated), step runs one fused launch over the flat buffers."""

    def __init__(self, model):
        if not isinstance(model, Model):
            raise TypeError("pass the Model: %s(model, weight_decay=...)" % type(self).__name__)
        self.model = model

    def zero_grad(self):
        pass

    def step(self):
        self.model.engine.adam_step()


class Adagrad(_EngineOptimizer):
    """torch.optim.Adagrad(model.parameters(), weight_decay=wd, lr=hyper_params['learning_rate']) look-alike
    (nb:1252-1255); torch's defaults otherwise (initial_accumulator_value 0)."""

    def __init__(self, model, lr=1e-2, lr_decay=0.0, weight_decay=0.0, eps=1e-10):
        super().__init__(model)
        model.engine.set_optimizer("adagrad", lr, weight_decay, decay=lr_decay, eps=eps)


class Adadelta(_EngineOptimizer):
    """torch.optim.Adadelta(model.parameters(), weight_decay=wd) look-alike (nb:1256-1259)."""

    def __init__(self, model, lr=1.0, rho=0.9, eps=1e-6, weight_decay=0.0):
        super().__init__(model)
        model.engine.set_optimizer("adadelta", lr, weight_decay, decay=rho, eps=eps)


class RMSprop(_EngineOptimizer):
    """torch.optim.RMSprop(model.parameters(), weight_decay=wd) look-alike (nb:1264-1267): plain RMSprop, as the
    notebook constructs it (momentum and centered, which it never sets, are refused)."""

    def __init__(self, model, lr=1e-2, alpha=0.99, eps=1e-8, weight_decay=0.0, momentum=0, centered=False):
        super().__init__(model)
        if momentum or centered:
            raise NotImplementedError("RMSprop with momentum / centered: the reference never asks for them")
        model.engine.set_optimizer("rmsprop", lr, weight_decay, decay=alpha, eps=eps)


def make_optimizer(model, hyper_params):
    """The notebook's optimizer selection nb:1252-1267: hyper_params['optimizer'] in adagrad / adadelta / adam /
    rmsprop; 'learning_rate' is read for adagrad only (nb:75), everything else runs at torch's defaults."""
    kind, wd = hyper_params.get("optimizer", "adam"), hyper_params["weight_decay"]
    if kind == "adagrad":
        return Adagrad(model, weight_decay=wd, lr=hyper_params["learning_rate"])
    if kind == "adadelta":
        return Adadelta(model, weight_decay=wd)
    if kind == "adam":
        return Adam(model, weight_decay=wd)
    if kind == "rmsprop":
        return RMSprop(model, weight_decay=wd)
    raise ValueError("hyper_params['optimizer'] = %r: the notebook defines adagrad, adadelta, adam, rmsprop "
                     "(any other value leaves its `optimizer` undefined, nb:1252-1267)" % (kind,))


# --------------------------------------------------------------------------------------------
# data (nb:406-576)
# --------------------------------------------------------------------------------------------
class DataReader:
    """Same constructor and semantics as nb:439-576; yields SparseTargets instead of dense y_s unless
    hyper_params['dense_targets'] is true."""

    def __init__(self, hyper_params, a, b, num_items, is_training):
        self.hyper_params = hyper_params
        self.batch_size = hyper_params["batch_size"]
        self.num_items = num_items
        self.is_training = is_training
        ua = np.array([[int(v) for v in line.strip().split(",")[:2]] for line in a], dtype=np.int64).reshape(-1, 2)
        self.min_user = int(ua[:, 0].min())
        self.num_users = int(ua[:, 0].max()) - self.min_user + 1                   # nb:444-450
        self.data = [[] for _ in range(self.num_users)]
        for uid, sid in ua:
            self.data[uid - self.min_user].append(int(sid))
        self.data_te = None
        if not is_training:
            self.data_te = [[] for _ in range(self.num_users)]
            for line in b:
                uid, sid = [int(v) for v in line.strip().split(",")[:2]]
                self.data_te[uid - self.min_user].append(sid)
        self.num_b = int(min(len(self.data), hyper_params["number_users_to_keep"]) / self.batch_size)  # nb:482

    def _k(self):
        lt = self.hyper_params["loss_type"]
        if lt == "predict_next":
            return 1
        if lt == "next_k":
            return int(self.hyper_params["next_k"])
        if lt == "postfix":
            return 10 ** 9
        raise ValueError("loss_type %r is not implemented by the reference reader either (nb:503-519)" % lt)

    def _targets(self, items):
        from .batch import build_plan as _bp
        T = len(items) - 1
        k = min(self._k(), T)
        tgt = np.full((T, k), -1, dtype=np.int32)
        arr = np.asarray(items, dtype=np.int64)
        for j in range(k):
            n = T - j
            tgt[:n, j] = arr[1 + j:1 + j + n]
        st = SparseTargets(tgt, self.num_items)
        return st.dense() if self.hyper_params.get("dense_targets") else st

    def user_order(self):
        order = list(range(len(self.data)))
        np.random.shuffle(order)                                                  # nb:493 (global numpy stream)
        return order

    def iter(self):
        users_done = 0
        for user in self.user_order():
            if users_done > self.hyper_params["number_users_to_keep"]:
                break
            users_done += 1
            items = self.data[user]
            yield torch.as_tensor([items[:-1]], dtype=torch.int64), self._targets(items)

    def eval_users(self):
        """(fold_in items, held-out items) per user, as iter_eval splits them (nb:536-546)."""
        out = []
        for user in range(len(self.data)):
            if len(out) >= self.hyper_params["number_users_to_keep"]:
                break
            if self.is_training:
                split = float(self.hyper_params["history_split_test"][0])
                n = int(split * len(self.data[user]))
                out.append((self.data[user][:n], self.data[user][n:]))
            else:
                out.append((self.data[user], self.data_te[user]))
        return out

    def iter_eval(self):
        for fold, held in self.eval_users():
            yield (torch.as_tensor([fold[:-1]], dtype=torch.int64), self._targets(fold), [list(held)],
                   [[1] * len(held)])


def load_data(hyper_params):
    """nb:406-437: reads pro_sg/{train,validation_tr,validation_te,test_tr,test_te}.csv + unique_sid.txt."""
    base = hyper_params["data_base"]

    def lines(name):
        with open(base + name) as f:
            return f.readlines()[1:]
    with open(base + "unique_sid.txt") as f:
        num_items = sum(1 for _ in f)
    train_reader = DataReader(hyper_params, lines("train.csv"), None, num_items, True)
    val_reader = DataReader(hyper_params, lines("validation_tr.csv"), lines("validation_te.csv"), num_items, False)
    test_reader = DataReader(hyper_params, lines("test_tr.csv"), lines("test_te.csv"), num_items, False)
    return train_reader, val_reader, test_reader, num_items


# --------------------------------------------------------------------------------------------
# train / evaluate (nb:1182-1233, nb:594-666)
# --------------------------------------------------------------------------------------------
def anneal_for_batch(b, total_anneal_steps=200000, cap=0.2):
    """Anneal seen by the b-th step (1-based) of an epoch: the schedule's locals restart every epoch and
    the value is updated AFTER use (nb:1188-1191, nb:1211-1215)."""
    if b <= 2:
        return 0.0
    return min(cap, float(b - 2) / total_anneal_steps) if total_anneal_steps > 0 else cap


def format_batch_log(epoch, batch, batch_limit, ms_per_batch, loss):
    """The running-loss line of train() (format string of nb:1226)."""
    return "| epoch {:3d} | {:5d}/{:5d} batches | ms/batch {:5.2f} | loss {:5.4f}".format(
        epoch, batch, batch_limit, ms_per_batch, loss)


def train(reader, model, optimizer, criterion, hyper_params, epoch=1, log=print):
    """The notebook's train(reader) (nb:1182-1233) with its globals as arguments; same log line."""
    model.train()
    total_loss, start_time, batch = 0.0, time.time(), 0
    batch_limit = int(reader.num_b)
    interval = hyper_params["batch_log_interval"]
    losses = []
    for x, y_s in reader.iter():
        batch += 1
        model.zero_grad()
        optimizer.zero_grad()
        dec, mu, ls = model(x)
        loss = criterion(dec, mu, ls, y_s, anneal_for_batch(batch))
        loss.backward()
        optimizer.step()
        losses.append(loss.detach())
        if (batch % interval == 0 and batch > 0) or batch == batch_limit:
            div = interval
            if batch == batch_limit:
                div = (batch_limit % interval) - 1                                # nb:1220 (off by one, kept)
            if div <= 0:
                div = 1
            cur = float(torch.stack(losses).sum()) / div
            log(format_batch_log(epoch, batch, batch_limit, (time.time() - start_time) * 1000 / div, cur))
            losses, start_time = [], time.time()
    return batch


def users_anneal(users_before, total_anneal_steps=200000, cap=0.2):
    """Anneal the reference applies to the user that follows `users_before` users of the epoch (its schedule
    advances once per USER: batch_size is 1, nb:80, nb:1211-1215)."""
    return anneal_for_batch(users_before + 1, total_anneal_steps, cap)


def train_batched(reader, model, hyper_params, batch_size=256, epoch=1, seed=0, log=None, grad_sync=None,
                  rank=0, world=1, anneal_by="users", workers=2):
    """One epoch with B users per step (length-bucketed, user-sharded over ranks).

    KL anneal: the reference's schedule advances once per user (one user = one optimizer step there).  With
    B*world users per optimizer step the default ``anneal_by="users"`` keeps the KL weight a function of the USERS
    consumed so far in the epoch -- step b uses the weight the reference gives the first user of that global batch,
    ``anneal_for_batch(1 + (b-1)*B*world)`` -- so a pass over the data sees the same KL weights as the notebook's
    pass (up to 0.022 on ML-1M, the 0.2 cap on ML-20M).  ``anneal_by="steps"`` counts optimizer steps instead (B*world
    times smaller weights over the same pass; the behaviour of round 1).  Either way the batched objective is the
    MEAN over the batch's users of the reference's per-user loss at that weight, and one Adam step is taken per
    batch, not per user ("batched objective")."""
    from .dist import rank_batches
    if anneal_by not in ("users", "steps"):
        raise ValueError("anneal_by must be 'users' or 'steps'")
    eng = model.engine
    rng = np.random.default_rng(seed + 1000 * epoch)
    eng.grad_sync = grad_sync
    k = reader._k()
    users_before = 0
    # plans (the per-user work of DataReader.iter, nb:485-526) are built and uploaded ahead of the GPU by worker
    # threads (the library's host-side builder, over ONE flat copy of the reader's per-user lists made on the first
    # epoch); the step only waits for an event
    from .batch import PlanPrefetcher, UserCSR
    csr = getattr(reader, "_svae_csr", None)
    if csr is None or csr[0] is not reader.data or len(csr[1]) != len(reader.data):
        csr = (reader.data, UserCSR(reader.data))
        try:
            reader._svae_csr = csr
        except AttributeError:
            pass
    lens_all = np.diff(csr[1].off)
    batches = rank_batches(lens_all, batch_size, rng, rank, world)
    if len(batches):
        eng.reserve(max(int(lens_all[np.asarray(b)].sum()) - len(b) for b in batches))   # rows = sum of (length - 1)
    plans = PlanPrefetcher(batches, csr[1], k, eng.device, total_items=eng.I, prepare=(eng.prepare_host, eng.prepare_upload),
                           workers=workers, depth=4 * workers)
    # the loop runs on the engine's high-priority stream (SVAEEngine.training_stream): the caller's chain of a step is
    # its critical path and shares the GPU with the engine's own side streams
    outer, hi = torch.cuda.current_stream(), eng.training_stream()
    hi.wait_stream(outer)
    with torch.cuda.stream(hi):
        for b, (idx, plan) in enumerate(zip(batches, plans), start=1):
            anneal = users_anneal(users_before) if anneal_by == "users" else anneal_for_batch(b)
            eng.train_step(plan, anneal, seed=seed + epoch)
            users_before += len(idx) * world
            if log is not None and b % max(1, hyper_params["batch_log_interval"] // batch_size) == 0:
                log("| epoch {:3d} | {:5d}/{:5d} batches | loss {:5.4f}".format(epoch, b, len(batches), float(eng.loss_out[0])))
    outer.wait_stream(hi)
    train_batched.last_wait_s = plans.wait_s          # (diagnostic: time the loop spent waiting for plans)
    return len(batches)


# 1 / log2(r) for r = 2 .. : the discount of a hit at 0-based rank r - 2, and the running ideal DCG.  Built with the
# same scalar calls the reference makes (nb:640, nb:644) and accumulated in the same order, so the vectorised metric
# arithmetic below is float-identical to its Python loop.
_KMAX = 100
_INV_LOG2 = np.array([0.0, 0.0] + [1.0 / float(np.log2(float(r))) for r in range(2, _KMAX + 2)])
_IDEAL = np.zeros(_KMAX + 1)
for _j in range(1, _KMAX + 1):
    _IDEAL[_j] = _IDEAL[_j - 1] + _INV_LOG2[_j + 1]


def _user_metrics(rec, held_lists, Ks=(10, 100)):
    """Per-user NDCG / Rec / Prec @K from descending rankings: rec [U, max(Ks)] item ids (distinct within a row),
    held_lists[u] = the user's held-out items.  Semantics of the reference's loop (nb:633-648): every held-out item
    found among the first K recommendations is a hit discounted by 1/log2(rank+2); the ideal DCG covers the first
    min(n_held, K) positions; Rec divides by the full held-out count, Prec by K.
    All users at once: one [held entries x max(Ks)] comparison finds each held-out item's rank, ``np.bincount`` with
    weights sums a user's hits and discounts in held-out order (the same float64 additions, in the same order, as
    the reference's accumulation).  Returns {"NDCG@10": [U] float64, ...}."""
    U = len(held_lists)
    n_held = np.fromiter((len(h) for h in held_lists), dtype=np.int64, count=U)
    flat = (np.concatenate([np.asarray(h, dtype=np.int64) for h in held_lists]) if U else np.zeros(0, dtype=np.int64))
    return _user_metrics_flat(rec, flat, n_held, Ks)


def _user_metrics_flat(rec, flat, n_held, Ks=(10, 100)):
    """_user_metrics with the held-out items of all users as ONE array `flat` and their counts `n_held` (what
    evaluate() keeps per reader, so that a batch's metric arithmetic has no per-user Python work)."""
    rec = np.asarray(rec)
    U = len(n_held)
    if max(Ks) > _KMAX or rec.shape[1] < max(Ks):
        raise ValueError("rankings must hold max(Ks) <= %d columns" % _KMAX)
    if (n_held == 0).any():
        raise ZeroDivisionError("a user without held-out items (the reference divides by zero there too, nb:646)")
    owner = np.repeat(np.arange(U), n_held)
    if rec.dtype != flat.dtype:
        flat = flat.astype(rec.dtype) if (len(flat) == 0 or flat.max() <= np.iinfo(rec.dtype).max) else flat
    match = rec[owner] == flat[:, None]                                   # [N, kmax]
    rank = np.where(match.any(axis=1), match.argmax(axis=1), -1)          # 0-based position in the ranking, -1: absent
    return _metrics_from_ranks(rank, n_held, Ks)


def _metrics_from_ranks(rank, n_held, Ks=(10, 100)):
    """The float64 arithmetic of nb:637-648 given, for every held-out item (users concatenated, `n_held` per user),
    its 0-based position in the user's ranking or -1 (what svae_heldout_ranks computes on the device)."""
    rank = np.asarray(rank, dtype=np.int64)
    n_held = np.asarray(n_held, dtype=np.int64)
    U = len(n_held)
    if (n_held == 0).any():
        raise ZeroDivisionError("a user without held-out items (the reference divides by zero there too, nb:646)")
    owner = np.repeat(np.arange(U), n_held)
    listed = rank >= 0                       # only items in the ranking add anything (x + 0.0 == x: same sums without the rest)
    owner, rank = owner[listed], rank[listed]
    out = {}
    for k in Ks:
        hit = rank < k
        own_k = owner[hit]
        hits = np.bincount(own_k, minlength=U).astype(np.float64)
        dcg = np.bincount(own_k, weights=_INV_LOG2[rank[hit] + 2], minlength=U)
        best = _IDEAL[np.minimum(n_held, k)]
        out["NDCG@%d" % k] = dcg / best
        out["Rec@%d" % k] = hits / n_held.astype(np.float64)
        out["Prec@%d" % k] = hits / float(k)
    return out


def _running_sum(total, values):
    """total + values[0] + values[1] + ... added left to right (np.sum adds pairwise: other last bits)."""
    return float(np.cumsum(np.concatenate(([total], np.asarray(values, dtype=np.float64))))[-1])


def _metrics_from_rankings(rec, held_lists, n_inputs, Ks=(10, 100)):
    """Sums over users of the six metrics, and len_to_ndcg_at_100_map keyed by n_held + n_inputs + 1 (nb:649-654)."""
    per = _user_metrics(rec, held_lists, Ks)
    sums = {key: _running_sum(0.0, v) for key, v in per.items()}
    len_map = {}
    if 100 in Ks:
        for u, held in enumerate(held_lists):
            len_map.setdefault(int(len(held)) + int(n_inputs[u]) + 1, []).append(float(per["NDCG@100"][u]))
    return sums, len_map


EVAL_PLAN_CACHE_BYTES = 1 << 30      # device bytes of resident evaluation plans kept per reader (evaluate)


class _ResidentPlans:
    """The iterable evaluate() reads its plans from when they are already on the device (PlanPrefetcher's face)."""
    wait_s = 0.0

    def __init__(self, plans):
        self.plans = plans

    def __iter__(self):
        return iter(self.plans)


def evaluate(model, criterion, reader, hyper_params, is_train_set, batch_size=256, eps_list=None, seed=0,
             rank=0, world=1, group=None, workers=2):
    """nb:594-666.  Returns (metrics dict with the reference's 7 keys, len_to_ndcg_at_100_map).
    Users are scored in batches; the loss is the mean of per-user reference losses at anneal 0.2
    (nb:617), divided by ``batch`` exactly as the reference does (nb:658, one more than the users
    scored when the train-set early break fired, nb:612-613).

    Pipelined (round 3): the plans of the user batches are built by the library's host-side builder on worker
    threads and uploaded ahead of the GPU (``batch.PlanPrefetcher`` over ONE flat copy of the fold-in lists, as
    ``train_batched`` does); the top-100 ids of a batch come back through a pinned buffer with an asynchronous copy
    and its metric arithmetic (nb:626-654) runs on the host while the NEXT batch is being scored; the per-batch losses
    stay on the device until the end, and so does the hit search of nb:636-645 (``svae_heldout_ranks``: the position of
    every held-out item in its user's top-100); one read-back at the end feeds the float64 metric arithmetic of
    nb:637-654 for all users.  Nothing synchronises per batch.

    Later calls on the same reader (fit() evaluates every epoch) find their plans resident on the device
    (``reader._svae_eval_plans``, up to EVAL_PLAN_CACHE_BYTES) and start no builder threads; the batches are scored
    three at a time on three streams inside one ``engine.scoring_session()``.

    world > 1 (data parallel, torch.distributed initialised): the user batches are dealt round-robin to the ranks,
    every rank scores its share, and the per-user metric rows (7 float64 per user) and per-batch loss sums are
    all-gathered; every rank then adds them up in the single-process order, so the result is bit-identical to a
    world-size-1 call whatever the number of ranks (noise is keyed by user, not by rank)."""
    from .batch import PlanPrefetcher, UserCSR, pack_rows
    eng = model.engine
    # the reader's (fold-in, held-out) split (nb:536-546) and its flat copies are made once per reader and split kind
    ckey = ("_svae_eval_csr", bool(is_train_set), hyper_params.get("train_cp_users"),
            reader.hyper_params.get("number_users_to_keep") if hasattr(reader, "hyper_params") else None)
    cache = getattr(reader, "_svae_eval_cache", None)
    if cache is None or cache[0] != ckey or cache[1] is not reader.data:
        pairs = reader.eval_users()
        batch = len(pairs)
        if is_train_set and len(pairs) > hyper_params["train_cp_users"]:
            pairs = pairs[:hyper_params["train_cp_users"]]
            batch = len(pairs) + 1
        hcnt = np.fromiter((len(h) for _, h in pairs), dtype=np.int64, count=len(pairs))
        hoff = np.zeros(len(pairs) + 1, dtype=np.int64)
        np.cumsum(hcnt, out=hoff[1:])
        hflat = np.fromiter((i for _, h in pairs for i in h), dtype=np.int32, count=int(hoff[-1]))
        ulen = hcnt + np.fromiter((len(f) for f, _ in pairs), dtype=np.int64, count=len(pairs))   # nb:653's map key
        cache = (ckey, reader.data, UserCSR([f for f, _ in pairs]), hflat, hoff, pairs, batch, ulen)
        try:
            reader._svae_eval_cache = cache
        except AttributeError:
            pass
    pairs, batch = cache[5], cache[6]
    k = reader._k()
    Ks = (10, 100)
    names = ["%s@%d" % (nm, kk) for kk in Ks for nm in ("NDCG", "Rec", "Prec")]
    mine = {}                        # chunk start -> (loss * users, per-user metric rows [n, 6])
    starts = list(range(0, len(pairs), batch_size))
    my_starts = [s for ci, s in enumerate(starts) if ci % world == rank]
    csr, hflat, hoff = cache[2], cache[3], cache[4]
    batches = [np.arange(s, min(s + batch_size, len(pairs)), dtype=np.int64) for s in my_starts]
    # An evaluation's batches are the same at every call (the users in reader order): their plans stay ON THE DEVICE,
    # kept on the reader beside the lists above, and every call after the first scores resident plans (fit() evaluates
    # two readers per epoch).  Up to EVAL_PLAN_CACHE_BYTES per reader; larger readers build their plans at every call.
    pkey = (int(batch_size), int(k), int(rank), int(world), int(eng.I), str(eng.device), id(eng))
    held = getattr(reader, "_svae_eval_plans", None)
    if held is not None and held[0] == pkey and held[1] is cache:
        plans = _ResidentPlans(held[2])
    else:
        plans = PlanPrefetcher(batches, csr, k, eng.device, total_items=eng.I,
                               prepare=(lambda p_: eng.prepare_host(p_, scoring=True), eng.prepare_upload), workers=workers, depth=2 * max(1, workers))
        held = None
    keep, keep_bytes = ([], 0) if held is None else (None, 0)
    kmax = max(Ks)
    t_begin = time.perf_counter()
    loss_dev = torch.zeros(max(1, len(my_starts)), device=eng.device, dtype=torch.float32)
    # the hit search of nb:636-645 runs on the device (svae_heldout_ranks): every held-out item's position in its
    # user's top-100, written into ONE array for the whole evaluation and read back once at the end
    if held is not None and len(held) > 3:           # (the held-out lists stay on the device with the plans)
        hflat_d, hoff_d = held[3], held[4]
    else:
        hflat_d = torch.from_numpy(hflat).to(eng.device)
        hoff_d = torch.from_numpy(hoff).to(eng.device)
    rank_dev = torch.full((max(1, len(hflat)),), -2, device=eng.device, dtype=torch.int32)
    # The parameters do not change in here: one scoring session (the parameter-derived operands made once) and up to
    # three workspaces -- the batches alternate between the caller's stream and the engine's two side streams, so that
    # the recurrence of one batch (a quarter of the chip) runs beside the sweeps of the others.
    cur = torch.cuda.current_stream()
    lanes = eng.scoring_lanes(3)
    # (the sweep's width does not depend on how many batches THIS rank scores: a batch's loss has the same bits wherever
    #  it is scored -- the sharded evaluation equals the single-process one exactly)
    with eng.scoring_session(sweep_cus=None if len(lanes) == 1 else eng.two_lane_sweep_cus(batch_size)):
        lanes = lanes[:max(1, min(len(lanes), len(my_starts)))]
        for s_ in lanes[1:]:
            s_.wait_stream(cur)                   # (the arrays above were filled on the caller's stream)
        for ci, (s0, plan) in enumerate(zip(my_starts, plans)):
            n = plan.B
            slot = ci % len(lanes)
            if keep is not None:
                keep_bytes += int(plan._buf.numel()) * plan._buf.element_size() if plan._buf is not None else 0
                keep = keep + [plan] if keep_bytes <= EVAL_PLAN_CACHE_BYTES else None
            with torch.cuda.stream(lanes[slot]):
                eps = None
                if eps_list is not None:
                    eps = torch.from_numpy(pack_rows(plan, eps_list[s0:s0 + n], L)).to(eng.device)
                out, idx, _, _ = eng.score(plan, kmax, eps=eps, seed=seed, step=s0, slot=slot)
                loss_dev[ci].copy_(out[0])
                _lib.call("svae_heldout_ranks", idx.data_ptr(), n, int(idx.shape[1]), plan.row_key.data_ptr(),
                          plan.last_row.data_ptr(), hflat_d.data_ptr(), hoff_d.data_ptr(), len(pairs), rank_dev.data_ptr(),
                          _stream())
        for s_ in lanes[1:]:
            cur.wait_stream(s_)
    if keep is not None and len(keep) == len(my_starts) and eps_list is None:
        try:
            reader._svae_eval_plans = (pkey, cache, keep, hflat_d, hoff_d)
        except AttributeError:
            pass
    t_loop = time.perf_counter()
    rank_all = rank_dev.cpu().numpy()
    t_sync = time.perf_counter()
    # one pass of the metric arithmetic over all of this rank's users (every user's value depends on that user's
    # ranks alone, so the grouping into batches does not show in the result), then cut back into the batches
    spans = [(s0, min(batch_size, len(pairs) - s0)) for s0 in my_starts]
    if world == 1:
        r_mine, held_mine = rank_all[:int(hoff[-1])], np.diff(hoff)
    else:
        r_mine = np.concatenate([rank_all[hoff[s0]:hoff[s0 + n]] for s0, n in spans] or [np.zeros(0, np.int32)])
        held_mine = np.concatenate([np.diff(hoff[s0:s0 + n + 1]) for s0, n in spans] or [np.zeros(0, np.int64)])
    if len(r_mine) and r_mine.min() < -1:
        raise RuntimeError("evaluate: a scored user's held-out ranks were not written (user keys of the plan?)")
    if spans:
        per = _metrics_from_ranks(r_mine, held_mine, Ks)
        rows_mine = np.stack([per[nm] for nm in names], axis=1)
        at = 0
        for s0, n in spans:
            mine[s0] = rows_mine[at:at + n]
            at += n
    losses = loss_dev.cpu().numpy() if my_starts else np.zeros(0, dtype=np.float32)
    # (diagnostic: host time of the scoring loop, of it blocked on plans, then the wait for the GPU, then the metrics)
    evaluate.last_timing = dict(loop_s=t_loop - t_begin, plans_wait_s=plans.wait_s, drain_s=t_sync - t_loop,
                                metrics_s=time.perf_counter() - t_sync)
    # engine loss = mean over the batch's users
    mine = {s0: (float(losses[ci]) * len(pairs[s0:s0 + batch_size]), mine[s0]) for ci, s0 in enumerate(my_starts)}
    if world > 1:
        import torch.distributed as dist
        parts = [None] * world
        dist.all_gather_object(parts, mine, group=group)
        mine = {}
        for p_ in parts:
            mine.update(p_)
    loss_sum, sums, len_map = 0.0, dict.fromkeys(names, 0.0), {}
    for s in starts:                                           # the single-process order of additions
        loss_sum += mine[s][0]
    rows = np.concatenate([mine[s][1] for s in starts]) if starts else np.zeros((0, len(names)))
    for j, n in enumerate(names):                              # one left-to-right sum over the users in their order
        sums[n] = _running_sum(0.0, rows[:, j])
    # len_to_ndcg_at_100_map (nb:653): key = held-out + fold-in items of the user (len(h) + (len(f) - 1) + 1), the
    # users' NDCG@100 in user order under each key, keys in order of first appearance
    ulen, ndcg100 = cache[7], rows[:, names.index("NDCG@100")]
    order = np.argsort(ulen, kind="stable")
    keys, first, count = np.unique(ulen, return_index=True, return_counts=True)
    begin = np.concatenate(([0], np.cumsum(count)))
    for g in np.argsort(first):
        len_map[int(keys[g])] = ndcg100[order[begin[g]:begin[g + 1]]].tolist()
    total_users = float(len(pairs))
    metrics = {"loss": round(float(loss_sum) / float(batch), 4)}
    for n in names:
        metrics[n] = round((100.0 * sums[n]) / total_users, 4)
    return metrics, len_map


# --------------------------------------------------------------------------------------------
# epoch driver (nb:1271-1307, nb:1366-1382): SURVEY 8f row 2
# --------------------------------------------------------------------------------------------
_RULE = 89          # width of the notebook's separator rules


def _metrics_string(metrics, tag):
    """' | name = value' per metric in dict order, then ' (TAG)' (nb:1283-1284, nb:1289-1290, nb:1373-1379)."""
    return "".join(" | %s = %s" % (m, metrics[m]) for m in metrics) + " (" + tag + ")"


def format_epoch_end(epoch, seconds, train_metrics, val_metrics):
    """The block the driver logs after every epoch (nb:1292-1301): rule, TRAIN line, rule, VAL line, rule."""
    rule = "-" * _RULE
    head = "| end of epoch {:3d} | time: {:5.2f}s".format(epoch, seconds)
    return "\n".join([rule, head + _metrics_string(train_metrics, "TRAIN"), rule,
                      head + _metrics_string(val_metrics, "VAL"), rule])


def format_end_of_training(test_metrics):
    """The final block (nb:1376-1381)."""
    rule = "=" * _RULE
    return "\n".join([rule, "| End of training" + _metrics_string(test_metrics, "TEST"), rule])


def fit(model, criterion, train_reader, val_reader, test_reader, hyper_params, optimizer=None, batch_size=None,
        log=print, checkpoint_path=None, seed=0, grad_sync=None, rank=0, world=1):
    """The notebook's epoch loop and final test evaluation (nb:1271-1307, nb:1366-1382) with its log lines:
    per epoch train -> evaluate(train, first `train_cp_users` users) -> evaluate(val) -> the two
    '| end of epoch' lines; the model with the best validation NDCG@100 so far (>=, nb:1303) is kept; at the end
    it is restored and scored on the test reader ('| End of training ... (TEST)').

    Differences, on purpose: the best model is kept as a ``state_dict`` (reference names and shapes; written to
    ``checkpoint_path`` with torch.save when given) instead of a pickled module (nb:1304); ``batch_size`` switches
    the per-epoch training from the notebook's per-user ``train`` to ``train_batched``.  Data parallel (world > 1):
    training and the three evaluations are sharded over the ranks (identical results on every rank); only rank 0
    logs and writes the checkpoint (to a temporary file, then os.replace: no reader ever sees a partial file).
    Returns (test metrics, len_to_ndcg_at_100_map, best validation NDCG@100)."""
    import os
    if optimizer is None:             # nb:1252-1267; it also configures the engine's step for train_batched
        optimizer = make_optimizer(model, hyper_params)
    if world > 1 and batch_size is None:
        raise ValueError("data parallel training needs batch_size (train_batched); the per-user train() is single-process")
    say = log if rank == 0 else (lambda *_: None)
    ev = dict(rank=rank, world=world)
    best_val_ndcg, best_state = None, None
    for epoch in range(1, hyper_params["epochs"] + 1):
        epoch_start_time = time.time()
        if batch_size is None:
            train(train_reader, model, optimizer, criterion, hyper_params, epoch, log=say)
        else:
            train_batched(train_reader, model, hyper_params, batch_size=batch_size, epoch=epoch, seed=seed, log=say,
                          grad_sync=grad_sync, rank=rank, world=world)
        train_metrics, _ = evaluate(model, criterion, train_reader, hyper_params, True, **ev)
        metrics, _ = evaluate(model, criterion, val_reader, hyper_params, False, **ev)
        say(format_epoch_end(epoch, time.time() - epoch_start_time, train_metrics, metrics))
        if not best_val_ndcg or metrics["NDCG@100"] >= best_val_ndcg:          # nb:1303
            best_state = {k: v.detach().clone() for k, v in model.state_dict().items()}
            if checkpoint_path is not None and rank == 0:
                tmp = "%s.tmp.%d" % (checkpoint_path, os.getpid())
                torch.save(best_state, tmp)
                os.replace(tmp, checkpoint_path)
            best_val_ndcg = metrics["NDCG@100"]
    if best_state is not None:
        model.load_state_dict(best_state)                                       # nb:1367
    metrics, len_to_ndcg_at_100_map = evaluate(model, criterion, test_reader, hyper_params, False, **ev)
    say(format_end_of_training(metrics))
    return metrics, len_to_ndcg_at_100_map, best_val_ndcg
