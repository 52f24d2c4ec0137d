"""The FUSED kernels the benchmark times (projection sweeps that never materialise logits, MLP chain, row-panel and
weight-gradient GEMMs, with the engine's fused orchestration: side streams, floated decoder.linear2 bucket) under the
fp32-class gates of test_gpu_parity.py, through their split instantiation PREC_SPLIT (same kernel templates, operands
as bf16 hi + lo parts, products hi*hi + hi*lo + lo*hi; csrc/proj.hip header).  Golden vectors come from executing the
reference notebook (oracle/make_golden.py).  Gates: loss 1e-5 .. 1e-4 relative, gradients per-tensor rel-L2 1e-4,
parameters after N Adam steps rel-L2 1e-4 (north_star).  Since round 3 the GRU recurrence runs here through the split
instantiation of the TIMED persistent kernel template too (gru.hip, prec 2: same work division, branch-free step, lane-swap
re-deal, two time ranges; h and W_hh as hi + mid + lo parts, exact sigmoid / tanh), not through the generic fp32 kernels:
`_engine` asserts it, and test_split_gru_template_vs_generic_fp32_kernels compares the two."""
import os

import numpy as np
import pytest
import torch

from oracle import svae_numpy as O

pytestmark = pytest.mark.gpu


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"), allow_pickle=False)


def _sd(p):
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in p.items()}


def _l2(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


def _param_gate(got, ref, name, tol=1e-4):
    d = np.asarray(got, dtype=np.float64) - np.asarray(ref, dtype=np.float64)
    l2 = np.linalg.norm(d) / np.linalg.norm(ref)
    assert l2 <= tol, (name, l2)
    nout = int((np.abs(d) > tol * np.abs(ref).max()).sum())
    assert nout <= max(1, int(1e-4 * d.size)), (name, nout, d.size)


def _engine(I):
    import svae_cf_amd as S
    eng = S.SVAEEngine(I, device="cuda", prec=S.PREC_SPLIT)
    assert eng.fused and eng.split
    assert eng.gru_prec == S.PREC_SPLIT, "PREC_SPLIT must run the timed GRU template, not fall back to gru_*_f32"
    return eng


def _kk(g):
    lt = str(g["loss_type"]) if "loss_type" in g.files else "next_k"
    return {"next_k": int(g["next_k"]), "predict_next": 1, "postfix": 10 ** 9}[lt]


@pytest.mark.parametrize("name", ["fwd_bwd_I97_k4", "fwd_bwd_I211_k1", "fwd_bwd_I64_postfix"])
def test_split_forward_loss_grads_vs_reference(golden_dir, name):
    import svae_cf_amd as S
    g = _load(golden_dir, name)
    I, seed, anneal = int(g["total_items"]), int(g["seed"]), float(g["anneal"])
    eng = _engine(I)
    eng.load_state_dict(_sd(O.make_params(I, seed)))
    for j in range(int(g["n_users"])):
        plan = S.build_plan([g["u%d_items" % j]], _kk(g), device="cuda", total_items=I)
        eps = torch.from_numpy(S.pack_rows(plan, [g["u%d_eps" % j]], 64)).cuda()
        out = eng.forward(plan, anneal, eps=eps)
        eng.backward()
        torch.cuda.synchronize()
        mu, ls = [t.cpu().numpy() for t in eng.mu_logsigma(plan.R)]
        # (operands carry 16 significant bits: elementwise 3e-5 of the tensor's scale, 2e-5 in rel-L2)
        np.testing.assert_allclose(mu, g["u%d_mu" % j], rtol=2e-4, atol=3e-5 * np.abs(g["u%d_mu" % j]).max())
        np.testing.assert_allclose(ls, g["u%d_log_sigma" % j], rtol=2e-4, atol=3e-5 * np.abs(g["u%d_log_sigma" % j]).max())
        assert _l2(mu, g["u%d_mu" % j]) <= 2e-5 and _l2(ls, g["u%d_log_sigma" % j]) <= 2e-5
        ref_loss = float(g["u%d_loss" % j])
        assert abs(float(out[0]) - ref_loss) <= 1e-5 * abs(ref_loss), (float(out[0]), ref_loss)
        # the loss-only sweep (scoring) gives the same loss
        out2 = float(eng.forward(plan, anneal, eps=eps, want_grad=False)[0])
        assert abs(out2 - ref_loss) <= 1e-5 * abs(ref_loss)
        if j == 0:
            eng.forward(plan, anneal, eps=eps)
            eng.backward()
            torch.cuda.synchronize()
            gd = eng.grad_dict()
            for k in O.PARAM_NAMES:
                assert _l2(gd[k].cpu().numpy(), g["u0_grad/" + k]) <= 1e-4, (k, _l2(gd[k].cpu().numpy(), g["u0_grad/" + k]))


def test_split_train_steps_vs_reference_train(golden_dir):
    """12 B=1 steps of the reference's own train() (order, anneal schedule, Adam + L2) through train_step."""
    import svae_cf_amd as S
    g = _load(golden_dir, "train_I97")
    I, seed = int(g["total_items"]), int(g["seed"])
    eng = _engine(I)
    eng.load_state_dict(_sd(O.make_params(I, seed)))
    losses = []
    for b, u in enumerate(g["order"], start=1):
        plan = S.build_plan([g["items%d" % u]], 4, device="cuda", total_items=I)
        eps = torch.from_numpy(S.pack_rows(plan, [g["eps%d" % (b - 1)]], 64)).cuda()
        losses.append(float(eng.train_step(plan, O.anneal_for_batch(b), eps=eps)[0]))
    np.testing.assert_allclose(losses, g["losses"], rtol=1e-4)
    sd = eng.state_dict()
    for k in O.PARAM_NAMES:
        _param_gate(sd[k].cpu().numpy(), g["final/" + k], k)


def test_split_batched_steps_vs_reference(golden_dir):
    import svae_cf_amd as S
    g = _load(golden_dir, "batch_adam_I97")
    I, seed = int(g["total_items"]), int(g["seed"])
    eng = _engine(I)
    eng.load_state_dict(_sd(O.make_params(I, seed)))
    losses = []
    for s in range(int(g["n_steps"])):
        nu = int(g["s%d_n_users" % s])
        users = [g["s%d_u%d_items" % (s, j)] for j in range(nu)]
        epsl = [g["s%d_u%d_eps" % (s, j)] for j in range(nu)]
        plan = S.build_plan(users, 4, device="cuda", total_items=I)
        eps = torch.from_numpy(S.pack_rows(plan, epsl, 64)).cuda()
        if s == 0:
            eng.forward(plan, float(g["s%d_anneal" % s]), eps=eps)
            eng.backward()
            gd = eng.grad_dict()
            for k in O.PARAM_NAMES:
                assert _l2(gd[k].cpu().numpy(), g["s0_grad/" + k]) <= 1e-4, k
        losses.append(float(eng.train_step(plan, float(g["s%d_anneal" % s]), eps=eps)[0]))
    np.testing.assert_allclose(losses, g["losses"], rtol=1e-4)
    sd = eng.state_dict()
    for k in O.PARAM_NAMES:
        _param_gate(sd[k].cpu().numpy(), g["final/" + k], k)


def test_split_larger_batch_vs_oracle():
    """70 ragged users, I = 500 (item count not a tile multiple, several x tiles and splits); oracle in fp64."""
    import svae_cf_amd as S
    rs = np.random.RandomState(6)
    I = 500
    users = [rs.permutation(I)[:rs.randint(2, 42)] for _ in range(70)]
    epsl = [rs.standard_normal((len(u) - 1, 64)).astype(np.float32) for u in users]
    p = O.make_params(I, 78)
    loss_ref, grads_ref = O.batch_loss_and_grads(p, users, epsl, 0.13, 4, np.float64)
    eng = _engine(I)
    eng.load_state_dict(_sd(p))
    plan = S.build_plan(users, 4, device="cuda", total_items=I)
    eps = torch.from_numpy(S.pack_rows(plan, epsl, 64)).cuda()
    out = eng.forward(plan, 0.13, eps=eps)
    eng.backward()
    torch.cuda.synchronize()
    assert abs(float(out[0]) - loss_ref) <= 1e-5 * abs(loss_ref), (float(out[0]), loss_ref)
    gd = eng.grad_dict()
    for k in O.PARAM_NAMES:
        assert _l2(gd[k].cpu().numpy(), grads_ref[k]) <= 1e-4, (k, _l2(gd[k].cpu().numpy(), grads_ref[k]))
    g1 = eng.grads.clone()
    eng.forward(plan, 0.13, eps=eps)
    eng.backward()
    torch.cuda.synchronize()
    assert torch.equal(g1, eng.grads)


def test_ml1m_shape_subset_vs_oracle():
    """configs[1]'s own shape: 3 483 items, sequences up to 1000 items (synth.SHAPES['ml-1m']), 16 users of a
    1024-user set, drawn from its two longest 256-user length buckets (the oracle stays affordable: 5 s): split
    instantiation against the fp64 oracle at the fp32 gates, bf16 (the timed mode) at its measured gates."""
    import svae_cf_amd as S
    from svae_cf_amd import synth
    _, I, mu, sigma, lmax = synth.SHAPES["ml-1m"]
    users_all = synth.make_users(1024, I, mu, sigma, lmax, seed=98765)
    lens = np.array([len(u) for u in users_all])
    buckets = sorted(synth.bucketed_batches(lens, 256, np.random.default_rng(2)), key=lambda b: -lens[b].mean())
    idx = np.concatenate((buckets[0][::37][:7], buckets[1][::29][:9]))      # lengths 1000, 305, ... 59
    users = [users_all[i] for i in idx]
    assert max(len(u) for u in users) == 1000 and len(users) == 16
    rs = np.random.RandomState(4)
    epsl = [rs.standard_normal((len(u) - 1, 64)).astype(np.float32) for u in users]
    p = O.make_params(I, 5)
    loss_ref, grads_ref = O.batch_loss_and_grads(p, users, epsl, 0.02, 4, np.float64)
    plan = S.build_plan(users, 4, device="cuda", total_items=I)
    eps = torch.from_numpy(S.pack_rows(plan, epsl, 64)).cuda()
    for prec, ltol, gtol in ((S.PREC_SPLIT, 1e-5, 1e-4), (S.PREC_BF16, 1e-3, 3e-2)):
        eng = S.SVAEEngine(I, device="cuda", prec=prec)
        eng.load_state_dict(_sd(p))
        out = eng.forward(plan, 0.02, eps=eps)
        eng.backward()
        torch.cuda.synchronize()
        assert abs(float(out[0]) - loss_ref) <= ltol * abs(loss_ref), (prec, float(out[0]), loss_ref)
        gd = eng.grad_dict()
        for k in O.PARAM_NAMES:
            assert _l2(gd[k].cpu().numpy(), grads_ref[k]) <= gtol, (prec, k, _l2(gd[k].cpu().numpy(), grads_ref[k]))


def test_split_gru_template_vs_generic_fp32_kernels():
    """The split instantiation of the persistent GRU kernels (forward in two time ranges, BPTT with the merged fp32
    gate-gradient array) against the generic fp32 recurrence kernels of rounds 1-2 inside the same PREC_SPLIT step, on an
    ML-20M-shaped batch with a time split: hidden states and saved gates to 2e-6 of the largest entry, every gradient
    to 1e-5 relative L2."""
    import svae_cf_amd as S
    from svae_cf_amd import synth
    from svae_cf_amd.layout import HP
    I, B = 20000, 256
    _, _, mu, sigma, lmax = synth.SHAPES["ml-20m"]
    users = synth.make_users(2048, I, mu, sigma, lmax, seed=4321)
    plan = None
    for cand in synth.bucketed_batches([len(u) for u in users], B, np.random.default_rng(2)):
        p = S.build_plan([users[i] for i in cand], 4, user_keys=cand, device="cuda", total_items=I)
        if p.split_t > 0 and p.R % 64 != 0:
            plan = p
            break
    assert plan is not None
    got = {}
    for generic in (False, True):
        eng = _engine(I)
        g = torch.Generator().manual_seed(5)
        eng.load_state_dict({n: (torch.randn(s_, generator=g) * (0.07 if n != "item_embed.weight" else 1.0))
                             for n, s_ in eng.layout.ref_shapes.items()})
        if generic:
            eng.gru_prec = S.PREC_FP32
        eng.prepare(plan)
        loss = float(eng.forward(plan, 0.05, seed=9, step=2)[0])
        eng.backward()
        torch.cuda.synchronize()
        got[generic] = dict(loss=loss, hs=eng.ws["hs"][:plan.R * HP].clone(), g4=eng.ws["gates4"][:plan.R * HP * 4].clone(),
                            grads={k: v.clone() for k, v in eng.grad_dict().items()})
    a, b = got[False], got[True]
    assert abs(a["loss"] - b["loss"]) <= 1e-6 * abs(b["loss"])
    for k in ("hs", "g4"):
        assert float((a[k] - b[k]).abs().max()) <= 2e-6 * float(b[k].abs().max()), k
    for k in b["grads"]:
        assert _l2(a["grads"][k].cpu().numpy(), b["grads"][k].cpu().numpy()) <= 1e-5, k
