"""GPU parity: the HIP path (through the C-ABI) against the numpy oracle and against the golden
vectors produced by executing the reference notebook.  Tolerances (fp32 mode):
  forward tensors   2e-4 relative (+2e-6 of the tensor's max)       -- fp32 round-off of K<=672 dot products
  loss              1e-5 relative   (north-star gate: 1e-3)
  gradients         per-tensor relative L2 <= 1e-4
  params after N Adam steps: per-tensor relative L2 <= 1e-4 (north-star gate), outliers < 1e-4 of elements
"""
import os

import numpy as np
import pytest
import torch

from oracle import svae_numpy as O

pytestmark = pytest.mark.gpu


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"), allow_pickle=False)


def _engine(I, prec=0):
    import svae_cf_amd as S
    return S.SVAEEngine(I, device="cuda", prec=prec)


def _sd(p):
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in p.items()}


def _close(a, b, rtol, atol_rel):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol_rel * max(np.abs(b).max(), 1e-30))


def _l2(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


def _param_gate(got, ref, name, tol=1e-4):
    d = np.asarray(got, dtype=np.float64) - np.asarray(ref, dtype=np.float64)
    l2 = np.linalg.norm(d) / np.linalg.norm(ref)
    assert l2 <= tol, (name, l2)
    nout = int((np.abs(d) > tol * np.abs(ref).max()).sum())      # (one element is allowed in a tensor under 10^4)
    assert nout <= max(1, int(1e-4 * d.size)), (name, nout, d.size)


def _kk(g):
    lt = str(g["loss_type"]) if "loss_type" in g.files else "next_k"
    return {"next_k": int(g["next_k"]), "predict_next": 1, "postfix": 10 ** 9}[lt]


@pytest.mark.parametrize("name", ["fwd_bwd_I97_k4", "fwd_bwd_I211_k1", "fwd_bwd_I64_postfix"])
def test_forward_loss_grads_vs_reference(golden_dir, name):
    import svae_cf_amd as S
    g = _load(golden_dir, name)
    I, seed, anneal = int(g["total_items"]), int(g["seed"]), float(g["anneal"])
    p = O.make_params(I, seed)
    eng = _engine(I)
    eng.load_state_dict(_sd(p))
    for j in range(int(g["n_users"])):
        items = g["u%d_items" % j]
        plan = S.build_plan([items], _kk(g), device="cuda")
        eps = torch.from_numpy(S.pack_rows(plan, [g["u%d_eps" % j]], 64)).cuda()
        out = eng.forward(plan, anneal, eps=eps, want_grad=False)
        torch.cuda.synchronize()
        logits = eng.logits_rows(plan.R).cpu().numpy()
        mu, ls = [t.cpu().numpy() for t in eng.mu_logsigma(plan.R)]
        _close(logits, g["u%d_dec_out" % j], 2e-4, 2e-6)
        _close(mu, g["u%d_mu" % j], 2e-4, 2e-6)
        _close(ls, g["u%d_log_sigma" % j], 2e-4, 2e-6)
        ref_loss = float(g["u%d_loss" % j])
        assert abs(float(out[0]) - ref_loss) <= 1e-5 * abs(ref_loss)
        if j == 0:
            eng.forward(plan, anneal, eps=eps, want_grad=True)
            eng.backward()
            torch.cuda.synchronize()
            gd = eng.grad_dict()
            for k in O.PARAM_NAMES:
                assert _l2(gd[k].cpu().numpy(), g["u0_grad/" + k]) <= 1e-4, k


def test_train_steps_vs_reference_train(golden_dir):
    """B=1 steps in the reference's shuffled order with its anneal schedule and Adam(L2)."""
    import svae_cf_amd as S
    g = _load(golden_dir, "train_I97")
    I, seed = int(g["total_items"]), int(g["seed"])
    eng = _engine(I)
    eng.load_state_dict(_sd(O.make_params(I, seed)))
    losses = []
    for b, u in enumerate(g["order"], start=1):
        plan = S.build_plan([g["items%d" % u]], 4, device="cuda")
        eps = torch.from_numpy(S.pack_rows(plan, [g["eps%d" % (b - 1)]], 64)).cuda()
        out = eng.train_step(plan, O.anneal_for_batch(b), eps=eps)
        losses.append(float(out[0]))
    np.testing.assert_allclose(losses, g["losses"], rtol=1e-4)
    sd = eng.state_dict()
    for k in O.PARAM_NAMES:
        _param_gate(sd[k].cpu().numpy(), g["final/" + k], k)


def test_batched_steps_vs_reference(golden_dir):
    """Ragged batches (mean over users of the reference loss), 3 Adam steps."""
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
        plan = S.build_plan(users, 4, device="cuda")
        eps = torch.from_numpy(S.pack_rows(plan, epsl, 64)).cuda()
        anneal = float(g["s%d_anneal" % s])
        eng.forward(plan, anneal, eps=eps)
        eng.backward()
        if s == 0:
            gd = eng.grad_dict()
            for k in O.PARAM_NAMES:
                assert _l2(gd[k].cpu().numpy(), g["s0_grad/" + k]) <= 1e-4, k
        eng.adam_step()
        losses.append(float(eng.loss_out[0]))
    np.testing.assert_allclose(losses, g["losses"], rtol=1e-4)
    sd = eng.state_dict()
    for k in O.PARAM_NAMES:
        _param_gate(sd[k].cpu().numpy(), g["final/" + k], k)


@pytest.mark.parametrize("kind", ["adam", "adagrad", "adadelta", "rmsprop"])
def test_optimizer_kernels_vs_torch_optim(golden_dir, kind):
    """svae_adam_step / svae_optimizer_step through the C-ABI on given gradients against torch.optim's trajectory
    (fixture optimizer_steps, constructed as nb:1252-1267): every step to 4 ulp of the largest element."""
    from svae_cf_amd import _lib
    g = _load(golden_dir, "optimizer_steps")
    dev = torch.device("cuda")
    p = torch.from_numpy(g["p0"].copy()).to(dev)
    a, b = torch.zeros_like(p), torch.zeros_like(p)
    st = torch.cuda.current_stream().cuda_stream
    hp = {"adagrad": (0, 1e-2, 0.0, 1e-10), "adadelta": (1, 1.0, 0.9, 1e-6), "rmsprop": (2, 1e-2, 0.99, 1e-8)}
    for s in range(int(g["n_steps"])):
        gr = torch.from_numpy(g["grads"][s].copy()).to(dev)
        if kind == "adam":
            _lib.call("svae_adam_step", p.data_ptr(), gr.data_ptr(), b.data_ptr(), a.data_ptr(), p.numel(), s + 1,
                      1e-3, 0.9, 0.999, 1e-8, 5e-3, 1.0, st)
        else:
            k, lr, decay, eps = hp[kind]
            _lib.call("svae_optimizer_step", k, p.data_ptr(), gr.data_ptr(), a.data_ptr(), b.data_ptr(), p.numel(),
                      s + 1, lr, decay, eps, 5e-3, 1.0, st)
        torch.cuda.synchronize()
        ref = g[kind][s]
        got = p.cpu().numpy()
        assert np.abs(got.astype(np.float64) - ref).max() <= 5e-7 * np.abs(ref).max(), (kind, s)
        np.testing.assert_array_equal(got[:16], 0.0)


@pytest.mark.parametrize("kind", ["adagrad", "adadelta", "rmsprop"])
def test_batched_steps_other_optimizers_vs_reference(golden_dir, kind):
    """Ragged batches under the notebook's other optimizer choices (nb:1252-1259, nb:1264-1267) through the notebook
    surface's make_optimizer; same gates as the Adam case (loss 1e-4, parameters rel-L2 1e-4; RMSprop 1e-3).  RMSprop: one step,
    then the loss of a probe batch at the new parameters within the north-star 1e-3."""
    import svae_cf_amd as S
    from svae_cf_amd import notebook_api as N
    g = _load(golden_dir, "batch_%s_I53" % kind)
    I, seed = int(g["total_items"]), int(g["seed"])
    hp = dict(item_embed_size=256, rnn_size=200, hidden_size=150, latent_size=64, total_items=I, batch_size=1,
              weight_decay=5e-3, learning_rate=0.01, optimizer=kind)
    model = N.Model(hp)
    eng = model.engine
    eng.load_state_dict(_sd(O.make_params(I, seed)))
    opt = N.make_optimizer(model, hp)
    losses = []
    for s in range(int(g["n_steps"])):
        nu = int(g["s%d_n_users" % s])
        users = [g["s%d_u%d_items" % (s, j)] for j in range(nu)]
        epsl = [g["s%d_u%d_eps" % (s, j)] for j in range(nu)]
        plan = S.build_plan(users, 4, device="cuda")
        eps = torch.from_numpy(S.pack_rows(plan, epsl, 64)).cuda()
        eng.forward(plan, float(g["s%d_anneal" % s]), eps=eps)
        eng.backward()
        opt.step()
        losses.append(float(eng.loss_out[0]))
    np.testing.assert_allclose(losses, g["losses"], rtol=1e-4)
    sd = eng.state_dict()
    # (RMSprop: 1e-3, see tests/test_oracle_golden.py::test_other_optimizers_match_reference)
    for k in O.PARAM_NAMES:
        _param_gate(sd[k].cpu().numpy(), g["final/" + k], k, tol=1e-3 if kind == "rmsprop" else 1e-4)
    if "probe_loss" in g.files:
        nu = int(g["probe_n_users"])
        plan = S.build_plan([g["probe_u%d_items" % j] for j in range(nu)], 4, device="cuda")
        eps = torch.from_numpy(S.pack_rows(plan, [g["probe_u%d_eps" % j] for j in range(nu)], 64)).cuda()
        out = eng.forward(plan, 0.2, eps=eps, want_grad=False)
        ref = float(g["probe_loss"])
        assert abs(float(out[0]) - ref) <= 1e-3 * abs(ref)


def test_unknown_optimizer_is_refused():
    from svae_cf_amd import notebook_api as N
    hp = dict(item_embed_size=256, rnn_size=200, hidden_size=150, latent_size=64, total_items=40, batch_size=1,
              weight_decay=5e-3, optimizer="sgd")
    with pytest.raises(ValueError):
        N.make_optimizer(N.Model(hp), hp)


def test_larger_batch_vs_oracle():
    """64 ragged users, I=500, T up to 40 -- several GRU workgroups and GEMM tiles; oracle in fp64."""
    import svae_cf_amd as S
    rs = np.random.RandomState(5)
    I = 500
    users = [rs.permutation(I)[:rs.randint(2, 42)] for _ in range(64)]
    epsl = [rs.standard_normal((len(u) - 1, 64)).astype(np.float32) for u in users]
    p = O.make_params(I, 77)
    loss_ref, grads_ref = O.batch_loss_and_grads(p, users, epsl, 0.13, 4, np.float64)
    eng = _engine(I)
    eng.load_state_dict(_sd(p))
    plan = S.build_plan(users, 4, device="cuda")
    eps = torch.from_numpy(S.pack_rows(plan, epsl, 64)).cuda()
    out = eng.forward(plan, 0.13, eps=eps)
    eng.backward()
    torch.cuda.synchronize()
    assert abs(float(out[0]) - loss_ref) <= 1e-5 * abs(loss_ref)
    gd = eng.grad_dict()
    for k in O.PARAM_NAMES:
        assert _l2(gd[k].cpu().numpy(), grads_ref[k]) <= 1e-4, k
    # determinism: same inputs -> same bits
    g1 = eng.grads.clone()
    eng.forward(plan, 0.13, eps=eps)
    eng.backward()
    torch.cuda.synchronize()
    assert torch.equal(g1, eng.grads)


@pytest.mark.parametrize("name", ["eval_val_I211", "eval_train_I211"])
def test_scoring_vs_reference_evaluate(golden_dir, name):
    import svae_cf_amd as S
    g = _load(golden_dir, name)
    I, seed = int(g["total_items"]), int(g["seed"])
    p = O.make_params(I, seed)
    p["decoder.linear2.weight"] = p["decoder.linear2.weight"] * np.float32(g["w3_scale"])
    n_eval = int(g["n_eval"])
    fold = [g["fold%d" % u] for u in range(n_eval)]
    epsl = [g["eps%d" % u] for u in range(n_eval)]
    eng = _engine(I)
    eng.load_state_dict(_sd(p))
    plan = S.build_plan(fold, 4, device="cuda")
    eps = torch.from_numpy(S.pack_rows(plan, epsl, 64)).cuda()
    _, idx, val, last = eng.score(plan, 100, eps=eps)
    torch.cuda.synchronize()
    idx, last = idx.cpu().numpy(), last.cpu().numpy()
    for su in range(n_eval):
        u = int(plan.order[su])
        _close(last[su], g["last_logits"][u], 2e-4, 2e-6)
        # top-100 indices bit-exact (scores are untied and separated far above fp32 round-off here)
        ref_top = g["top100"][u]
        s = O.masked_scores(g["last_logits"][u], fold[u][:-1])
        gaps = np.abs(np.diff(s[ref_top]))
        stable = np.concatenate(([True], gaps > 1e-4)) & np.concatenate((gaps > 1e-4, [True]))
        np.testing.assert_array_equal(idx[su][stable], ref_top[stable])
        assert set(idx[su]) == set(ref_top) or (~stable).any()


# ---------------------------------------------------------------------------------------------
# bf16 MFMA mode (the throughput mode of bench.py).  Operands are rounded to bf16 (2^-9 relative),
# accumulation and every non-GEMM op stay fp32.  Measured deviation from the reference golden
# vectors (tools/precision_report.py, MI355X): logits/mu 5e-3 rel-L2, loss 2e-4..9e-4 relative,
# gradients ~1e-2 rel-L2, parameters after 12 Adam steps 1e-4..3e-3 rel-L2.  Gates below: the
# north-star loss gate (1e-3) holds; the 1e-4 parameter gate is an fp32-class tolerance and is
# only claimed for prec=fp32 (tests above).
# ---------------------------------------------------------------------------------------------
def test_bf16_mode_train_steps_vs_reference(golden_dir):
    import svae_cf_amd as S
    g = _load(golden_dir, "train_I97")
    I, seed = int(g["total_items"]), int(g["seed"])
    eng = _engine(I, prec=1)
    eng.load_state_dict(_sd(O.make_params(I, seed)))
    losses = []
    for b, u in enumerate(g["order"], start=1):
        plan = S.build_plan([g["items%d" % u]], 4, device="cuda", total_items=I)
        eps = torch.from_numpy(S.pack_rows(plan, [g["eps%d" % (b - 1)]], 64)).cuda()
        losses.append(float(eng.train_step(plan, O.anneal_for_batch(b), eps=eps)[0]))
    # north_star's loss gate (1e-3); parameters: bf16 operands (2^-9 relative) put the gradients at ~1e-2 and the
    # parameters after 12 Adam steps at 1e-4 .. 3e-3 rel-L2 (tools/precision_report.py) -- a stated exception to
    # the 1e-4 parameter gate, which the SAME kernels meet in their split instantiation
    # (tests/test_gpu_split_parity.py::test_split_train_steps_vs_reference_train).  Gate: 5e-3.
    np.testing.assert_allclose(losses, g["losses"], rtol=1e-3)
    sd = eng.state_dict()
    for k in O.PARAM_NAMES:
        assert _l2(sd[k].cpu().numpy(), g["final/" + k]) <= 5e-3, k


def test_bf16_mode_larger_batch_vs_oracle():
    import svae_cf_amd as S
    rs = np.random.RandomState(6)
    I = 500
    users = [rs.permutation(I)[:rs.randint(2, 42)] for _ in range(70)]
    epsl = [rs.standard_normal((len(u) - 1, 64)).astype(np.float32) for u in users]
    p = O.make_params(I, 78)
    loss_ref, grads_ref = O.batch_loss_and_grads(p, users, epsl, 0.13, 4, np.float64)
    eng = _engine(I, prec=1)
    eng.load_state_dict(_sd(p))
    plan = S.build_plan(users, 4, device="cuda", total_items=I)
    eps = torch.from_numpy(S.pack_rows(plan, epsl, 64)).cuda()
    out = eng.forward(plan, 0.13, eps=eps)
    eng.backward()
    torch.cuda.synchronize()
    assert abs(float(out[0]) - loss_ref) <= 1e-3 * abs(loss_ref)
    gd = eng.grad_dict()
    for k in O.PARAM_NAMES:
        assert _l2(gd[k].cpu().numpy(), grads_ref[k]) <= 3e-2, k
    g1 = eng.grads.clone()
    eng.forward(plan, 0.13, eps=eps)
    eng.backward()
    torch.cuda.synchronize()
    assert torch.equal(g1, eng.grads)


@pytest.mark.parametrize("R", [1, 127, 1000, 4097])
def test_rowgemm_matches_fp32_product_of_bf16_operands(R):
    """Row-panel GEMM of the GRU input side (csrc/rowgemm.hip) against torch on the same bf16-rounded operands:
    gi = emb W_ih^T + b (which 0) and demb = dgi W_ih (which 1).  The kernel accumulates exact bf16 x bf16
    products in fp32, so only the summation order differs: 2e-5 of the result's scale (K <= 672 terms)."""
    from svae_cf_amd import _lib
    from svae_cf_amd.layout import E, G
    g = torch.Generator().manual_seed(R)
    wih = (torch.randn(G, E, generator=g) * 0.1).cuda()
    bias = torch.randn(G, generator=g).cuda()
    st = torch.cuda.current_stream().cuda_stream
    wb = wih.bfloat16().float()
    for which, (K, N) in enumerate(((E, G), (G, E))):
        x = torch.randn(R, K, generator=g).cuda()
        frags = torch.zeros(_lib.load().svae_rowgemm_frag_elems(which), dtype=torch.bfloat16, device="cuda")
        y = torch.full((R, N), float("nan"), device="cuda")
        _lib.call("svae_rowgemm_prep", wih.data_ptr(), which, frags.data_ptr(), st)
        _lib.call("svae_rowgemm", x.data_ptr(), frags.data_ptr(), bias.data_ptr() if which == 0 else None,
                  y.data_ptr(), R, which, st)
        torch.cuda.synchronize()
        xb = x.bfloat16().float()
        ref = xb @ wb.t() + bias if which == 0 else xb @ wb
        err = float((y - ref).abs().max())
        assert err <= 2e-5 * float(ref.abs().max()) + 1e-6, (which, R, err)


def test_masked_topk_equals_stable_descending_sort_on_edge_cases():
    """svae_masked_topk against torch.sort(descending=True, stable=True) of the masked scores (the reference's full
    sort nb:632 with ties by lower item index): random rows, heavy ties (pre-filter overflows -> selection over all
    scores), a constant row (more ties at the threshold than the LDS sort holds -> arg-max rounds), -inf entries,
    K = 1 and K = I.  Bit-exact indices and values."""
    from svae_cf_amd import _lib
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(5)

    def run(scores, seen_lists, K):
        n, I = scores.shape
        ld = (I + 3) // 4 * 4
        buf = torch.zeros(n, ld)
        buf[:, :I] = scores
        buf = buf.cuda()
        off = np.zeros(n + 1, dtype=np.int32)
        off[1:] = np.cumsum([len(x) for x in seen_lists])
        seen = torch.from_numpy(np.concatenate([np.asarray(x, dtype=np.int32) for x in seen_lists] + [np.zeros(1, np.int32)])).cuda()
        off_t = torch.from_numpy(off).cuda()
        idx = torch.empty(n * K, dtype=torch.int32, device="cuda")
        val = torch.empty(n * K, device="cuda")
        work = torch.empty_like(buf)
        _lib.call("svae_masked_topk", buf.data_ptr(), ld, seen.data_ptr(), off_t.data_ptr(), n, I, K, idx.data_ptr(),
                  val.data_ptr(), work.data_ptr(), st)
        torch.cuda.synchronize()
        ref = scores.clone()
        for u, lst in enumerate(seen_lists):
            for it in lst:
                ref[u, it] = ref[u, it] - (ref[u, it] * 1.0).abs() * 1e8
        rv, ri = torch.sort(ref, dim=1, descending=True, stable=True)
        assert torch.equal(idx.view(n, K).cpu().long(), ri[:, :K]), "indices"
        assert torch.equal(val.view(n, K).cpu(), rv[:, :K]), "values"

    I = 5000
    rnd = torch.randn(4, I, generator=g)
    run(rnd, [[1, 7, 4999], [], list(range(0, 3000, 3)), [0]], 100)
    few = torch.randint(0, 3, (3, I), generator=g).float()             # three distinct values: thousands of ties
    run(few, [[], [5], [6, 7]], 100)
    run(torch.full((2, I), 0.25), [[], [3]], 100)                       # constant rows
    inf = rnd.clone()
    inf[:, ::2] = float("-inf")
    run(inf, [[], [], [], []], 100)
    run(rnd, [[], [], [], []], 1)
    small = torch.randn(3, 50, generator=g)
    run(small, [[], [1], [2, 3]], 50)
    run(torch.randn(2, 3000, generator=g), [[], []], 2000)             # K above the LDS sort: arg-max rounds
    # ---- the register-resident form's bound (per-wave ballot search, r = ceil(K / full waves) <= 64, else the radix
    # select) and its places by counting (up to 1 024 candidates, else select + sort): both sides of every switch
    for I, K in ((20000, 100), (2000, 100), (256, 100), (253, 60), (252, 60), (700, 120), (600, 128), (600, 129),
                 (20000, 1000), (24576, 1024), (24576, 7)):
        r = torch.randn(3, I, generator=g)
        best = torch.topk(r[1], min(200, I // 2)).indices.tolist()
        run(r, [[], best, list(range(0, I, 5))], K)
    I = 20000
    run((torch.randint(0, 2000, (3, I), generator=g).float() - 1000.0) / 7.0, [[], [3, 4], list(range(0, 600, 2))], 100)   # ~10 ties per value
    desc = -torch.arange(I, dtype=torch.float32).repeat(2, 1)          # the best items all in wave 0's lanes: the other
    run(desc, [[], [0, 1, 2, 50]], 100)                                 # waves pull the bound down (thousands of candidates)
    block = torch.randn(2, I, generator=g)
    block[:, 4096:8192] += 6.0                                          # one float4 slot of every thread far above the rest
    run(block, [[], list(range(4096, 4300))], 100)
    # ---- catalogues beyond 24 576 items: the STREAMED path (round 3: the row read once in register-resident chunks, the
    # seen items in a hash set, the bound tightened by selecting over the candidate list)
    for I in (24580, 30000, 131072 + 4, 200000):
        big = torch.randn(3, I, generator=g)
        top = torch.topk(big[1], 300).indices.tolist()                  # a user who has SEEN its 300 best items
        run(big, [[0, I - 1, 12345], top, list(range(0, 2700, 3))], 100)
    I = 100000
    asc = torch.arange(I, dtype=torch.float32).repeat(2, 1) * 1e-3      # ascending rows: every chunk beats the bound of
    run(asc, [[], [I - 1, I - 2, I - 50]], 100)                         # the chunks before it (the list is tightened)
    quant = torch.randint(0, 50, (2, I), generator=g).float()           # 2 000 ties per value: a flood at the bound
    run(quant, [[], [7, 8, 9]], 100)                                    # (-> starts again on the legacy path)
    run(torch.full((2, 40000), -1.5), [[], [3]], 100)                   # constant rows
    many = torch.randn(2, 60000, generator=g)
    run(many, [list(range(0, 925)), list(range(0, 2000, 2))], 100)      # K + seen = 1025 and 1100 > 1024: legacy path
    run(many, [list(range(0, 924)), [5]], 100)                          # K + seen = 1024: the streamed path's limit
    run(torch.randn(2, 50000, generator=g), [[], [1]], 1)
    run(torch.randn(2, 50000, generator=g), [[], [1]], 1000)
    neg = -torch.rand(2, 70000, generator=g) - 1.0                      # all scores negative; zeros among the seen
    neg[1, 11] = 0.0
    run(neg, [[], [11, 12]], 100)


def test_postfix_targets_beyond_256_per_row_vs_oracle():
    """loss_type 'postfix' (nb:515-519): row t targets EVERY later item, so a 300-item user has rows with up to 299
    targets -- more than one target per thread of the row kernel.  fp32 path (svae_softmax_nll over dense logits,
    the notebook surface's path) against the fp64 oracle: loss 1e-5, gradients 1e-4; bf16 fused path: loss 1e-3."""
    import svae_cf_amd as S
    rs = np.random.RandomState(21)
    I = 400
    users = [rs.permutation(I)[:300], rs.permutation(I)[:9], rs.permutation(I)[:270]]
    epsl = [rs.standard_normal((len(u) - 1, 64)).astype(np.float32) for u in users]
    p = O.make_params(I, 31)
    K = 10 ** 9
    loss_ref, grads_ref = O.batch_loss_and_grads(p, users, epsl, 0.07, K, np.float64)
    eng = _engine(I)
    eng.load_state_dict(_sd(p))
    plan = S.build_plan(users, K, device="cuda", total_items=I)
    assert plan.K == 299
    eps = torch.from_numpy(S.pack_rows(plan, epsl, 64)).cuda()
    out = eng.forward(plan, 0.07, eps=eps)
    eng.backward()
    torch.cuda.synchronize()
    assert abs(float(out[0]) - loss_ref) <= 1e-5 * abs(loss_ref)
    gd = eng.grad_dict()
    for k in O.PARAM_NAMES:
        assert _l2(gd[k].cpu().numpy(), grads_ref[k]) <= 1e-4, k
    eng16 = _engine(I, prec=1)
    eng16.load_state_dict(_sd(p))
    out16 = eng16.forward(plan, 0.07, eps=eps)
    eng16.backward()
    torch.cuda.synchronize()
    assert abs(float(out16[0]) - loss_ref) <= 1e-3 * abs(loss_ref)
    gd = eng16.grad_dict()
    for k in O.PARAM_NAMES:
        assert _l2(gd[k].cpu().numpy(), grads_ref[k]) <= 3e-2, k


def test_philox_normal_stream_properties():
    """svae_philox_normal (the counter-based replacement of np.random.normal, nb:744) through the C-ABI:
    same (seed, step, key) -> same bits on every launch and wherever the row sits in the batch (shard invariance);
    another step or seed -> another stream; 10^6 draws look N(0,1) (mean, variance, 4th moment, tail fractions);
    rows with adjacent keys and adjacent columns are uncorrelated."""
    from svae_cf_amd import _lib
    st = torch.cuda.current_stream().cuda_stream
    R = 15625                                   # x 64 columns = 10^6 draws
    keys = (torch.arange(R, dtype=torch.int64) // 50 << 20) + torch.arange(R, dtype=torch.int64) % 50   # (user << 20) + t
    keys = keys.cuda()

    def draw(k, seed, step):
        out = torch.empty(k.numel(), 64, device="cuda")
        _lib.call("svae_philox_normal", out.data_ptr(), k.data_ptr(), k.numel(), seed, step, st)
        torch.cuda.synchronize()
        return out
    a = draw(keys, 1234, 7)
    assert torch.equal(a, draw(keys, 1234, 7))
    perm = torch.randperm(R, generator=torch.Generator().manual_seed(0)).cuda()
    assert torch.equal(draw(keys[perm], 1234, 7), a[perm])             # a row's noise depends on its key only
    assert torch.equal(draw(keys[100:300], 1234, 7), a[100:300])
    b, c = draw(keys, 1234, 8), draw(keys, 1235, 7)
    assert not torch.equal(a, b) and not torch.equal(a, c)
    x = a.double().cpu().numpy()
    n = x.size
    assert np.isfinite(x).all()
    assert abs(x.mean()) < 5 / np.sqrt(n)                               # 5 sigma of the mean
    assert abs(x.var() - 1.0) < 5 * np.sqrt(2.0 / n)
    assert abs((x ** 4).mean() - 3.0) < 5 * np.sqrt(96.0 / n)
    for t, pr in ((1.0, 0.31731), (2.0, 0.04550), (3.0, 0.002700), (4.0, 6.334e-5)):
        frac = (np.abs(x) > t).mean()
        assert abs(frac - pr) < 5 * np.sqrt(pr * (1 - pr) / n) + 1e-6, (t, frac)
    corr = lambda u, v: float(np.mean(u * v))
    assert abs(corr(x[:-1], x[1:])) < 5 / np.sqrt(n)                    # adjacent keys (t, t + 1), same column
    assert abs(corr(x[:, :-1], x[:, 1:])) < 5 / np.sqrt(n)              # adjacent columns (same Philox block or next)
    assert abs(corr(x[:, 0::4], x[:, 1::4])) < 5 / np.sqrt(n / 4)       # the two outputs of one Box-Muller pair
    y, z = b.double().cpu().numpy(), c.double().cpu().numpy()
    assert abs(corr(x, y)) < 5 / np.sqrt(n) and abs(corr(x, z)) < 5 / np.sqrt(n)
