"""The numpy oracle (oracle/svae_numpy.py) against vectors produced by EXECUTING the
reference notebook (oracle/make_golden.py -> tests/golden/*.npz).  CPU only."""
import os

import numpy as np
import pytest

from oracle import svae_numpy as O

FWD_CASES = ["fwd_bwd_I97_k4", "fwd_bwd_I211_k1", "fwd_bwd_I64_postfix"]


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"), allow_pickle=False)


def _k(g):
    lt = str(g["loss_type"]) if "loss_type" in g.files else "next_k"
    return {"next_k": int(g["next_k"]), "predict_next": 1, "postfix": 10 ** 9}[lt]


def _close(a, b, rtol, atol_rel=1e-6):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    scale = max(np.abs(b).max(), 1e-30)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol_rel * scale)


def _param_gate(got, ref, name, tol=1e-4):
    """Parameter gate after N Adam steps: per-tensor relative L2 error <= 1e-4.

    Max-norm is NOT used: in the first Adam steps the update is lr*g/(|g|+1e-8), so an
    element whose gradient is ~1e-8 moves by anything in [-lr, lr] depending on the last
    bit of g (two fp32 implementations of the same math -- torch vs this oracle -- already
    differ by ~lr on about 1 element in 1e5).  Such outliers must stay rarer than 1e-4.
    """
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    d = got - ref
    l2 = np.linalg.norm(d) / np.linalg.norm(ref)
    assert l2 <= tol, (name, l2)
    nout = int((np.abs(d) > tol * np.abs(ref).max()).sum())      # (one element is allowed in a tensor under 10^4)
    assert nout <= max(1, int(1e-4 * d.size)), (name, nout, d.size)


@pytest.mark.parametrize("name", FWD_CASES)
def test_targets_match_reference_reader(golden_dir, name):
    g = _load(golden_dir, name)
    I = int(g["total_items"])
    for j in range(int(g["n_users"])):
        items = g["u%d_items" % j]
        x, tgt = O.build_inputs_targets(items, _k(g))
        np.testing.assert_array_equal(x, g["u%d_x" % j])
        y = O.dense_targets(tgt, I)
        nz = np.argwhere(y > 0).astype(np.int32)
        np.testing.assert_array_equal(nz, g["u%d_y_nz" % j])


@pytest.mark.parametrize("name", FWD_CASES)
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_forward_loss_match_reference(golden_dir, name, dtype):
    g = _load(golden_dir, name)
    I, seed, anneal = int(g["total_items"]), int(g["seed"]), float(g["anneal"])
    p = O.make_params(I, seed)
    for j in range(int(g["n_users"])):
        x, tgt = O.build_inputs_targets(g["u%d_items" % j], _k(g))
        fwd = O.forward(p, x, g["u%d_eps" % j], dtype)
        _close(fwd["dec_out"], g["u%d_dec_out" % j], 2e-4, 2e-6)
        _close(fwd["mu"], g["u%d_mu" % j], 2e-4, 2e-6)
        _close(fwd["log_sigma"], g["u%d_log_sigma" % j], 2e-4, 2e-6)
        loss = O.vae_loss(fwd["dec_out"], fwd["mu"], fwd["log_sigma"], tgt, anneal)[0]
        assert abs(loss - float(g["u%d_loss" % j])) <= 2e-6 * abs(float(g["u%d_loss" % j]))


@pytest.mark.parametrize("name", FWD_CASES)
def test_backward_matches_reference_autograd(golden_dir, name):
    g = _load(golden_dir, name)
    I, seed, anneal = int(g["total_items"]), int(g["seed"]), float(g["anneal"])
    p = O.make_params(I, seed)
    _, grads, _ = O.loss_and_grads(p, g["u0_items"], g["u0_eps"], anneal, _k(g), np.float64)
    for k in O.PARAM_NAMES:
        _close(grads[k], g["u0_grad/" + k], 1e-3, 2e-6)


def test_train_loop_matches_reference_train(golden_dir):
    """Replays the reference's train() (nb:1182-1233): shuffled order, per-step anneal, Adam+L2."""
    g = _load(golden_dir, "train_I97")
    I, seed = int(g["total_items"]), int(g["seed"])
    p = O.make_params(I, seed)
    st = O.adam_init(p)
    losses = []
    for b, u in enumerate(g["order"], start=1):
        anneal = O.anneal_for_batch(b)
        assert anneal == float(g["anneals"][b - 1])
        loss, grads, _ = O.loss_and_grads(p, g["items%d" % u], g["eps%d" % (b - 1)], anneal, 4, np.float32)
        losses.append(loss)
        p = O.adam_step(p, grads, st)
    np.testing.assert_allclose(losses, g["losses"], rtol=1e-4)
    for k in O.PARAM_NAMES:
        _param_gate(p[k], g["final/" + k], k)


def test_batched_contract_matches_reference(golden_dir):
    g = _load(golden_dir, "batch_adam_I97")
    I, seed = int(g["total_items"]), int(g["seed"])
    p = O.make_params(I, seed)
    st = O.adam_init(p)
    losses = []
    for s in range(int(g["n_steps"])):
        nu = int(g["s%d_n_users" % s])
        users = [g["s%d_u%d_items" % (s, j)] for j in range(nu)]
        eps = [g["s%d_u%d_eps" % (s, j)] for j in range(nu)]
        loss, grads = O.batch_loss_and_grads(p, users, eps, float(g["s%d_anneal" % s]), 4, np.float32)
        if s == 0:
            for k in O.PARAM_NAMES:
                _close(grads[k], g["s0_grad/" + k], 2e-3, 5e-6)
        losses.append(loss)
        p = O.adam_step(p, grads, st)
    np.testing.assert_allclose(losses, g["losses"], rtol=1e-4)
    for k in O.PARAM_NAMES:
        _param_gate(p[k], g["final/" + k], k)


@pytest.mark.parametrize("kind", ["adam", "adagrad", "adadelta", "rmsprop"])
def test_optimizer_arithmetic_matches_torch_optim(golden_dir, kind):
    """The optimizers alone, as nb:1252-1267 constructs them, on given gradients (fixture made by torch.optim):
    the parameter after every one of 6 steps, to 2 ulp of the largest element (fp32 arithmetic on both sides)."""
    g = _load(golden_dir, "optimizer_steps")
    p = {"w": g["p0"].copy()}
    st = O.adam_init(p) if kind == "adam" else O.opt_init(p)
    step = getattr(O, kind + "_step")
    for s in range(int(g["n_steps"])):
        p = step(p, {"w": g["grads"][s]}, st)
        ref = g[kind][s]
        assert np.abs(p["w"].astype(np.float64) - ref).max() <= 2.5e-7 * np.abs(ref).max(), (kind, s)
        np.testing.assert_array_equal(p["w"][:16], 0.0)        # no gradient, no weight: stays exactly zero


@pytest.mark.parametrize("kind", ["adagrad", "adadelta", "rmsprop"])
def test_other_optimizers_match_reference(golden_dir, kind):
    """Batched steps under the notebook's other optimizer choices (nb:1252-1259, nb:1264-1267).  RMSprop's case is
    one step plus the loss of a probe batch at the new parameters (see oracle/make_golden.py for why)."""
    g = _load(golden_dir, "batch_%s_I53" % kind)
    assert str(g["optimizer"]) == kind
    I, seed = int(g["total_items"]), int(g["seed"])
    p = O.make_params(I, seed)
    st = O.opt_init(p)
    step = getattr(O, kind + "_step")
    losses = []
    for s in range(int(g["n_steps"])):
        nu = int(g["s%d_n_users" % s])
        users = [g["s%d_u%d_items" % (s, j)] for j in range(nu)]
        eps = [g["s%d_u%d_eps" % (s, j)] for j in range(nu)]
        loss, grads = O.batch_loss_and_grads(p, users, eps, float(g["s%d_anneal" % s]), 4, np.float32)
        losses.append(loss)
        p = step(p, grads, st)
    np.testing.assert_allclose(losses, g["losses"], rtol=1e-4)
    # RMSprop's first step is 10 lr g / (|g| + 1e-7): elements with |g| ~ 1e-7 (where fp32 summation order decides
    # the second digit of g) move by anything up to lr * 10, so its gate is 1e-3; the arithmetic itself is pinned to
    # 2 ulp by test_optimizer_arithmetic_matches_torch_optim
    for k in O.PARAM_NAMES:
        _param_gate(p[k], g["final/" + k], k, tol=1e-3 if kind == "rmsprop" else 1e-4)
    if "probe_loss" in g.files:
        nu = int(g["probe_n_users"])
        users = [g["probe_u%d_items" % j] for j in range(nu)]
        eps = [g["probe_u%d_eps" % j] for j in range(nu)]
        loss, _ = O.batch_loss_and_grads(p, users, eps, 0.2, 4, np.float32)
        assert abs(loss - float(g["probe_loss"])) <= 1e-3 * abs(float(g["probe_loss"]))     # north-star loss gate


@pytest.mark.parametrize("name", ["eval_val_I211", "eval_train_I211"])
def test_evaluate_matches_reference_evaluate(golden_dir, name):
    g = _load(golden_dir, name)
    I, seed = int(g["total_items"]), int(g["seed"])
    p = O.make_params(I, seed)
    p["decoder.linear2.weight"] = p["decoder.linear2.weight"] * np.float32(g["w3_scale"])
    n_eval = int(g["n_eval"])
    fold = [g["fold%d" % u] for u in range(n_eval)]
    held = [list(g["held%d" % u]) for u in range(n_eval)]
    eps = [g["eps%d" % u] for u in range(n_eval)]
    div = n_eval + 1 if bool(g["is_train_set"]) and int(g["n_users"]) > n_eval else None
    metrics, len_map = O.evaluate_users(p, fold, held, eps, 4, np.float32, loss_divisor=div)
    ref = dict(zip([str(s) for s in g["metric_names"]], g["metric_values"]))
    assert abs(metrics["loss"] - ref["loss"]) <= 1e-3 * abs(ref["loss"])
    for k in ref:
        if k != "loss":
            assert metrics[k] == pytest.approx(ref[k], abs=1e-9), k
    assert sorted(len_map.keys()) == list(g["len_map_keys"])
    for kk in len_map:
        np.testing.assert_allclose(len_map[kk], g["len_map/%d" % kk], atol=1e-12)
    # ranking: top-100 indices bit-exact where scores are untied
    for u in range(n_eval):
        x = fold[u][:-1]
        fwd = O.forward(p, x, eps[u], np.float32)
        _close(fwd["dec_out"][-1], g["last_logits"][u], 2e-4, 2e-6)
        s = O.masked_scores(g["last_logits"][u], x)
        np.testing.assert_array_equal(O.topk_desc(s, 100), g["top100"][u])


def test_torch_port_matches_reference(golden_dir):
    """oracle/svae_torch.py (the cpu_baseline port) replays the reference's train() run."""
    import torch
    from oracle import svae_torch as TP
    g = _load(golden_dir, "train_I97")
    I, seed = int(g["total_items"]), int(g["seed"])
    model = TP.TorchSVAE(I)
    model.load_reference_state(O.make_params(I, seed))
    opt = torch.optim.Adam(model.parameters(), weight_decay=5e-3)
    users = [g["items%d" % u] for u in g["order"]]
    eps = [g["eps%d" % j] for j in range(len(users))]
    losses = TP.train_users(model, opt, users, 4, I, eps_list=eps, anneal_fn=O.anneal_for_batch)
    np.testing.assert_allclose(losses, g["losses"], rtol=1e-5)
    sd = model.reference_state()
    for k in O.PARAM_NAMES:
        _param_gate(sd[k], g["final/" + k], k)
