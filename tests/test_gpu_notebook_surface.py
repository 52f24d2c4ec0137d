"""The notebook's own surface (Model / VAELoss / DataReader / train / evaluate) on the HIP engine, checked
against golden vectors produced by running the reference's train() and evaluate() themselves."""
import os

import numpy as np
import pytest
import torch

from oracle import svae_numpy as O

pytestmark = pytest.mark.gpu


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"), allow_pickle=False)


def _hp(I, **kw):
    hp = {"history_split_test": [0.8, 0.2], "weight_decay": 5e-3, "batch_size": 1, "item_embed_size": 256,
          "rnn_size": 200, "hidden_size": 150, "latent_size": 64, "loss_type": "next_k", "next_k": 4,
          "number_users_to_keep": 1000000000, "batch_log_interval": 10 ** 9, "train_cp_users": 200,
          "total_items": I}
    hp.update(kw)
    return hp


def _lines(users):
    return ["%d,%d,1\n" % (u, int(it)) for u, items in enumerate(users) for it in items]


def _gate(got, ref, name, tol=1e-4):
    d = np.asarray(got, np.float64) - np.asarray(ref, np.float64)
    assert np.linalg.norm(d) / np.linalg.norm(ref) <= tol, name


def test_notebook_train_loop_matches_reference(golden_dir):
    """train(reader): shuffled order and latent noise come from numpy's global stream exactly as in the
    reference (nb:493, nb:744), so seeding it replays the reference run."""
    from svae_cf_amd import notebook_api as N
    g = _load(golden_dir, "train_I97")
    I, seed, n = int(g["total_items"]), int(g["seed"]), int(g["n_users"])
    hp = _hp(I)
    reader = N.DataReader(hp, _lines([g["items%d" % u] for u in range(n)]), None, I, True)
    model = N.Model(hp)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in O.make_params(I, seed).items()})
    model.eps_source = lambda T, L: np.random.normal(0, 1, size=(T, L))
    crit = N.VAELoss(hp, model)
    opt = N.Adam(model, weight_decay=hp["weight_decay"])
    np.random.seed(seed)
    steps = N.train(reader, model, opt, crit, hp)
    assert steps == n
    sd = model.state_dict()
    for k in O.PARAM_NAMES:
        _gate(sd[k].cpu().numpy(), g["final/" + k], k)


def test_notebook_forward_and_loss_single_user(golden_dir):
    from svae_cf_amd import notebook_api as N
    g = _load(golden_dir, "fwd_bwd_I97_k4")
    I, seed, anneal = int(g["total_items"]), int(g["seed"]), float(g["anneal"])
    hp = _hp(I)
    model = N.Model(hp)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in O.make_params(I, seed).items()})
    eps = g["u0_eps"]
    model.eps_source = lambda T, L: eps
    crit = N.VAELoss(hp, model)
    x = torch.from_numpy(g["u0_x"])[None]
    dec, mu, ls = model(x)
    assert dec.shape == (1, len(eps), I)
    np.testing.assert_allclose(dec[0].detach().cpu().numpy(), g["u0_dec_out"], rtol=2e-4, atol=2e-5)
    # dense one-hot targets exactly as the reference reader hands them over
    y = torch.zeros(1, len(eps), I)
    nz = g["u0_y_nz"]
    y[0, nz[:, 0], nz[:, 1]] = 1.0
    loss = crit(dec, mu, ls, y.cuda(), anneal)
    assert abs(float(loss) - float(g["u0_loss"])) <= 1e-5 * float(g["u0_loss"])
    loss.backward()
    gd = model.engine.grad_dict()
    for k in O.PARAM_NAMES:
        _gate(gd[k].cpu().numpy(), g["u0_grad/" + k], k)


@pytest.mark.parametrize("name", ["eval_val_I211", "eval_train_I211"])
def test_notebook_evaluate_matches_reference(golden_dir, name):
    from svae_cf_amd import notebook_api as N
    g = _load(golden_dir, name)
    I, seed, n = int(g["total_items"]), int(g["seed"]), int(g["n_users"])
    is_train = bool(g["is_train_set"])
    hp = _hp(I, train_cp_users=4)
    fold = [g["fold%d" % u] for u in range(n)]
    held = [g["held%d" % u] for u in range(n)]
    if is_train:
        full = [np.concatenate([f, h]) for f, h in zip(fold, held)]
        reader = N.DataReader(hp, _lines(full), None, I, True)
    else:
        reader = N.DataReader(hp, _lines(fold), _lines(held), I, False)
    p = O.make_params(I, seed)
    p["decoder.linear2.weight"] = p["decoder.linear2.weight"] * np.float32(g["w3_scale"])
    model = N.Model(hp)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in p.items()})
    eps_list = [g["eps%d" % u] for u in range(int(g["n_eval"]))]
    metrics, len_map = N.evaluate(model, N.VAELoss(hp, model), reader, hp, is_train, batch_size=3, eps_list=eps_list)
    ref = dict(zip([str(s) for s in g["metric_names"]], g["metric_values"]))
    assert list(metrics.keys()) == list(ref.keys())
    assert abs(metrics["loss"] - ref["loss"]) <= 1e-3 * abs(ref["loss"])
    for k in ref:
        if k != "loss":
            assert abs(metrics[k] - ref[k]) <= 0.5, (k, metrics[k], ref[k])       # north-star gate: 0.5 absolute
            assert metrics[k] == pytest.approx(ref[k], abs=1e-9), k               # and in fact identical here
    assert sorted(len_map.keys()) == list(g["len_map_keys"])
    for kk in len_map:
        np.testing.assert_allclose(len_map[kk], g["len_map/%d" % kk], atol=1e-12)


@pytest.mark.parametrize("batch_size", [None, 16])
def test_fit_epoch_driver_log_lines_and_best_checkpoint(tmp_path, batch_size):
    """Epoch driver (nb:1271-1307, nb:1366-1382): log-line formats the reference's plotting cell parses
    (nb:1310-1347), best-validation selection with >=, restore before the test evaluation, state_dict checkpoint
    with the reference's parameter names.  Per-user training (batch_size None) and the batched fast path."""
    import re
    from svae_cf_amd import notebook_api as N
    from svae_cf_amd import synth
    I = 120
    users = synth.make_users(60, I, 2.6, 0.4, 40, seed=3)
    tr, va, te = users[:40], users[40:50], users[50:]

    def split(us):
        return [u[:int(0.8 * len(u))] for u in us], [u[int(0.8 * len(u)):] for u in us]
    hp = _hp(I, epochs=2, batch_log_interval=20, train_cp_users=8)
    train_reader = N.DataReader(hp, _lines(tr), None, I, True)
    vf, vh = split(va)
    tf, th = split(te)
    val_reader = N.DataReader(hp, _lines(vf), _lines(vh), I, False)
    test_reader = N.DataReader(hp, _lines(tf), _lines(th), I, False)
    torch.manual_seed(0)
    model = N.Model(hp)
    crit = N.VAELoss(hp, model)
    logs = []
    ckpt = str(tmp_path / "best.pt")
    metrics, len_map, best = N.fit(model, crit, train_reader, val_reader, test_reader, hp, batch_size=batch_size,
                                   log=logs.append, checkpoint_path=ckpt)
    text = "\n".join(logs)
    ends = re.findall(r"\| end of epoch +(\d+) \| time: +[\d.]+s \| loss = [\d.]+ \| NDCG@10 = [\d.]+ .*\((TRAIN|VAL)\)", text)
    assert ends == [("1", "TRAIN"), ("1", "VAL"), ("2", "TRAIN"), ("2", "VAL")]
    assert re.search(r"\| End of training \| loss = [\d.]+ \| NDCG@10 = .* \| Prec@100 = [\d.]+ \(TEST\)", text)
    assert list(metrics.keys()) == ["loss", "NDCG@10", "Rec@10", "Prec@10", "NDCG@100", "Rec@100", "Prec@100"]
    vals = [float(v) for v in re.findall(r"NDCG@100 = ([\d.]+) \| Rec@100 = [\d.]+ \| Prec@100 = [\d.]+ \(VAL\)", text)]
    assert best == max(vals) and len(vals) == 2
    sd = torch.load(ckpt, weights_only=True)
    assert set(sd.keys()) == set(O.PARAM_NAMES)
    now = model.state_dict()
    for k in sd:                      # the best model was restored before the test evaluation
        assert torch.equal(sd[k].cpu(), now[k].cpu()), k
    assert all(np.isfinite(v) for v in metrics.values()) and sum(len(v) for v in len_map.values()) == len(te)


def test_model_init_reproducible_under_manual_seed_and_stale_backward_is_refused():
    """Model(hp) draws its parameters from torch's global generator (like the reference's Model.__init__), and the
    autograd bridge refuses a backward whose forward is no longer the engine's last one (the engine keeps one set of
    activations; the reference's autograd would keep both)."""
    from svae_cf_amd import notebook_api as N
    hp = _hp(60)
    torch.manual_seed(7)
    a = N.Model(hp).state_dict()
    torch.manual_seed(7)
    model = N.Model(hp)
    b = model.state_dict()
    assert all(torch.equal(a[k], b[k]) for k in a)
    crit = N.VAELoss(hp, model)
    x1 = torch.tensor([[1, 2, 3, 4, 5]])
    x2 = torch.tensor([[7, 8, 9]])
    tg = lambda x: N.SparseTargets(np.concatenate([x[0, 1:].numpy(), [0]]).astype(np.int32)[:, None], 60)
    d1, m1, l1 = model(x1)
    loss1 = crit(d1, m1, l1, tg(x1), 0.1)
    d2, m2, l2 = model(x2)                       # a second forward before the first backward
    loss2 = crit(d2, m2, l2, tg(x2), 0.1)
    with pytest.raises(RuntimeError, match="LAST forward"):
        loss1.backward()
    loss2.backward()                             # the last forward's backward is fine
    assert torch.isfinite(model.engine.grads).all()
