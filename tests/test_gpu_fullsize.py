"""GPU tests at BASELINE.json's full shapes (configs[2] ML-20M, configs[3] Netflix, configs[4] 1M-item catalogue).

The numpy oracle cannot run these sizes in seconds, so parity here is carried by
  * the SPLIT instantiation of the fused kernels (PREC_SPLIT: the same kernel templates at fp32-class precision,
    pinned to the oracle and to the reference's golden vectors at the fp32 gates in test_gpu_split_parity.py) as
    the comparator of the fused bf16 path on identical inputs (same parameters, same Philox noise: keyed by user
    id and timestep); where the logits fit (configs[2], [3]) the exact-fp32 path with materialised logits is
    checked against the split instantiation too, at 1e-5 / 1e-4; and
  * size-independent properties: rows of dlogits sum to zero  =>  sum_i dL/db3[i] = 0; bitwise determinism;
    top-k == a full descending sort of the same masked scores (what the reference does, nb:632).
Tolerances (bf16 operands, fp32 accumulation): loss 1e-3 relative (the north-star gate), gradients 3e-2
relative L2 per tensor (measured ~1e-2, tools/precision_report.py).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _l2(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def _init(eng, seed):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shp in eng.layout.ref_shapes.items():
        if name == "item_embed.weight":
            sd[name] = torch.randn(shp, generator=g)
        elif name.startswith("gru."):
            sd[name] = (torch.rand(shp, generator=g) * 2 - 1) / (200 ** 0.5)
        elif name.endswith("weight"):
            sd[name] = torch.randn(shp, generator=g) * (2.0 / (shp[0] + shp[1])) ** 0.5
        else:
            sd[name] = (torch.rand(shp, generator=g) * 2 - 1) * 0.05
    eng.load_state_dict(sd)
    return sd


def _pair(I, users, keys, anneal=0.05, dense_too=True):
    """One forward+backward of the same batch in the split instantiation (comparator) and in fused bf16 mode;
    dense_too: the exact-fp32 path with materialised logits must agree with the comparator at the fp32 gates."""
    import svae_cf_amd as S
    plan = S.build_plan(users, 4, user_keys=keys, device="cuda", total_items=I)
    ref = S.SVAEEngine(I, device="cuda", prec=S.PREC_SPLIT)
    sd = _init(ref, 11)
    if dense_too:
        dense = S.SVAEEngine(I, device="cuda", prec=S.PREC_FP32)
        dense.load_state_dict(sd)
        l_dense = float(dense.forward(plan, anneal, seed=7, step=3)[0])
        dense.backward()
        l_split = float(ref.forward(plan, anneal, seed=7, step=3)[0])
        ref.backward()
        torch.cuda.synchronize()
        assert abs(l_split - l_dense) <= 1e-5 * abs(l_dense), (l_split, l_dense)
        gd, gs = dense.grad_dict(), ref.grad_dict()
        for k in gd:
            assert _l2(gs[k], gd[k]) <= 1e-4, ("split vs exact fp32", k, _l2(gs[k], gd[k]))
        del dense, gd, gs
        torch.cuda.empty_cache()
    fast = S.SVAEEngine(I, device="cuda", prec=S.PREC_BF16)
    fast.load_state_dict(sd)
    l_ref = float(ref.forward(plan, anneal, seed=7, step=3)[0])
    ref.backward()
    l_fast = float(fast.forward(plan, anneal, seed=7, step=3)[0])
    fast.backward()
    torch.cuda.synchronize()
    return plan, ref, fast, l_ref, l_fast


def _check_pair(ref, fast, l_ref, l_fast):
    assert abs(l_fast - l_ref) <= 1e-3 * abs(l_ref), (l_fast, l_ref)
    gr, gf = ref.grad_dict(), fast.grad_dict()
    for k in gr:
        assert _l2(gf[k], gr[k]) <= 3e-2, (k, _l2(gf[k], gr[k]))
    gb = gf["decoder.linear2.bias"].double()
    assert abs(float(gb.sum())) <= 2e-3 * float(gb.abs().sum()), "rows of dlogits must sum to zero"


def test_ml20m_shape_fused_bf16_vs_exact_fp32():
    """configs[2]: 20 000 items, 256 length-bucketed users, sequences <= 200."""
    from svae_cf_amd import synth
    I, B = 20000, 256
    _, _, mu, sigma, lmax = synth.SHAPES["ml-20m"]
    users = synth.make_users(2048, I, mu, sigma, lmax, seed=98765)
    rng = np.random.default_rng(1)
    idx = synth.bucketed_batches([len(u) for u in users], B, rng)[5]
    plan, ref, fast, l_ref, l_fast = _pair(I, [users[i] for i in idx], idx)
    _check_pair(ref, fast, l_ref, l_fast)
    R = plan.R
    lse_ref, lse_fast = ref.ws["lse"][:R], fast.ws["lse"][:R]
    assert float((lse_ref - lse_fast).abs().max()) <= 5e-3
    # bitwise determinism of the whole fused step
    g1 = fast.grads.clone()
    fast.forward(plan, 0.05, seed=7, step=3)
    fast.backward()
    torch.cuda.synchronize()
    assert torch.equal(g1, fast.grads)


def test_netflix_shape_item_count_not_a_tile_multiple():
    """configs[3]: 17 770 items (17 770 % 64 != 0: pad items must contribute nothing), sequences up to 1000."""
    from svae_cf_amd import synth
    I = 17770
    _, _, mu, sigma, lmax = synth.SHAPES["netflix"]
    users = synth.make_users(256, I, mu, sigma, lmax, seed=5)
    users = sorted(users, key=len)[-48:]          # the longest ones: T up to ~1000
    plan, ref, fast, l_ref, l_fast = _pair(I, users, np.arange(48))
    assert plan.Tmax > 400
    _check_pair(ref, fast, l_ref, l_fast)


def test_large_catalogue_one_million_items():
    """configs[4]: 1M items, projection / log-softmax dominated, plus top-100 scoring."""
    import svae_cf_amd as S
    I = 1_000_000
    rs = np.random.RandomState(3)
    users = [rs.choice(I, size=rs.randint(3, 14), replace=False) for _ in range(8)]
    plan, ref, fast, l_ref, l_fast = _pair(I, users, np.arange(8))
    _check_pair(ref, fast, l_ref, l_fast)
    # scoring: loss-only sweep (bf16) against the split instantiation, and top-100 against a full sort of the same scores
    l_s_ref, idx_ref, val_ref, last_ref = ref.score(plan, 100, seed=7, step=3)
    l_s_ref = float(l_s_ref[0])
    l_s_fast, idx_fast, val_fast, last_fast = fast.score(plan, 100, seed=7, step=3)
    torch.cuda.synchronize()
    assert abs(float(l_s_fast[0]) - l_s_ref) <= 1e-3 * abs(l_s_ref)
    for eng_idx, last in ((idx_ref, last_ref), (idx_fast, last_fast)):
        for u in range(plan.B):
            s = last[u].clone()
            seen = plan.seen[plan.seen_off[u]:plan.seen_off[u + 1]].long()
            m = torch.zeros_like(s)
            m[seen] = 1.0
            s = s - (s * m).abs() * 1e8                      # the reference's mask form, nb:624
            top = torch.sort(-s, stable=True)[1][:100]
            assert torch.equal(top.int(), eng_idx[u]), u


def test_bf16_scoring_loss_matches_oracle():
    """The loss-only sweep (MODE_FWD of the fused projection) against the numpy oracle."""
    import svae_cf_amd as S
    from oracle import svae_numpy as O
    rs = np.random.RandomState(9)
    I = 700
    users = [rs.permutation(I)[:rs.randint(3, 30)] for _ in range(40)]
    epsl = [rs.standard_normal((len(u) - 1, 64)).astype(np.float32) for u in users]
    p = O.make_params(I, 21)
    loss_ref, _ = O.batch_loss_and_grads(p, users, epsl, 0.2, 4, np.float64)
    eng = S.SVAEEngine(I, device="cuda", prec=S.PREC_BF16)
    eng.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in p.items()})
    plan = S.build_plan(users, 4, device="cuda", total_items=I)
    eps = torch.from_numpy(S.pack_rows(plan, epsl, 64)).cuda()
    out = eng.score(plan, 10, eps=eps)[0]
    torch.cuda.synchronize()
    assert abs(float(out[0]) - loss_ref) <= 1e-3 * abs(loss_ref)


def test_gru_gradient_array_forms_agree():
    """The BPTT kernel can hand its gate gradients over as two fp32 arrays (dgi, dgh), as one merged fp32 array
    (dr, dz, dn, dn*r) or as that array in bf16.  Every MFMA consumer rounds them to bf16 (RNE) before multiplying,
    so the three forms must give BIT-IDENTICAL weight and embedding gradients; the two GRU bias gradients are column
    sums of the array itself and differ by the bf16 rounding of their terms in the third form (<= 1e-3 of the
    largest entry; measured 2e-4).  A fourth run takes the 256 x 256-tile kernel for the GRU's weight gradients (engine.wgrad_big,
    bf16 hidden states): equal to fp32 summation order."""
    import svae_cf_amd as S
    from svae_cf_amd import synth
    I, B = 3000, 64
    users = synth.make_users(B, I, 3.6, 0.6, 120, seed=11)
    plan = S.build_plan(users, 4, user_keys=np.arange(B), device="cuda", total_items=I)
    grads = []
    for merged, bf16, big, w16 in ((False, False, False, False), (True, False, False, False), (True, True, False, False),
                                   (True, True, True, False), (True, True, False, True)):
        eng = S.SVAEEngine(I, device="cuda", prec=S.PREC_BF16)
        torch.manual_seed(1)
        eng.load_state_dict({n: torch.randn(s) * 0.1 for n, s in eng.layout.ref_shapes.items()})
        eng.merged_dg, eng.dg_bf16 = merged, bf16
        eng.wgrad_big = big        # (the 256 x 256-tile kernel sums in another order: compared separately below)
        eng.wgrad16 = w16          # (the production kernel of the GRU's weight gradients: hand-counted waits, bf16 hidden states)
        eng.forward(plan, 0.05, seed=3, step=1)
        eng.backward()
        torch.cuda.synchronize()
        grads.append(eng.grad_dict())
    for k in grads[0]:
        assert float(grads[0][k].abs().max()) > 0, k
        assert torch.equal(grads[0][k], grads[1][k]), ("merged fp32 array", k)
        if k in ("gru.bias_ih_l0", "gru.bias_hh_l0"):
            d = float((grads[1][k] - grads[2][k]).abs().max())
            assert d <= 1e-3 * float(grads[1][k].abs().max()), (k, d)
        else:
            assert torch.equal(grads[1][k], grads[2][k]), ("merged bf16 array", k)
        # the alternative form (engine.wgrad_big): the GRU's two weight gradients on 256 x 256 tiles with a bf16 copy of the hidden states --
        # the same bf16 products, another summation order (tiles, split count)
        d = float((grads[2][k] - grads[3][k]).abs().max())
        assert d <= 3e-6 * float(grads[2][k].abs().max()), ("256-tile weight gradients", k, d)
        # the production form (engine.wgrad16): the same tiles, stages and MFMA order on a bf16 copy of the hidden states
        # -- every weight gradient bit-identical; the two GRU bias gradients (column sums) in another summation order
        if k in ("gru.bias_ih_l0", "gru.bias_hh_l0"):
            d = float((grads[2][k] - grads[4][k]).abs().max())
            assert d <= 3e-6 * float(grads[2][k].abs().max()), ("hand-waited kernel, bias", k, d)
        else:
            assert torch.equal(grads[2][k], grads[4][k]), ("hand-waited kernel", k)


def test_fused_operand_copies_equal_the_two_kernel_path():
    """svae_mlp_fwd_prep (MLP chain + the sweeps' bf16 operand copies in one kernel) against svae_mlp_fwd followed by
    svae_proj_prep_rows_range, on a two-range forward of an ML-20M-shaped batch whose row count is not a multiple of
    64: h2b / h2s, d1 and the loss bit-identical, gradients bit-identical.  (The transposed copy h2tb is made for the
    split instantiation only: the bf16 sweeps read the transpose out of the row-major image.)"""
    import svae_cf_amd as S
    from svae_cf_amd import synth
    I, B = 20000, 256
    _, _, mu, sigma, lmax = synth.SHAPES["ml-20m"]
    users = synth.make_users(2048, I, mu, sigma, lmax, seed=98765)
    idx = None
    for cand in synth.bucketed_batches([len(u) for u in users], B, np.random.default_rng(1)):
        plan = S.build_plan([users[i] for i in cand], 4, user_keys=cand, device="cuda", total_items=I)
        if plan.split_t > 0 and plan.R % 64 != 0:
            idx = cand
            break
    assert idx is not None, "no batch with a time split and a ragged row count"
    eng = S.SVAEEngine(I, device="cuda", prec=S.PREC_BF16)
    _init(eng, 11)
    eng.prepare(plan)
    Rp = (plan.R + 63) // 64 * 64
    got = {}
    for fused in (False, True):
        eng.fused_prep = fused
        for name in ("h2b", "h2s", "d1"):
            eng.ws[name].zero_() if name in eng.ws else None
        loss = eng.forward(plan, 0.05, seed=7, step=3).clone()
        eng.backward()
        torch.cuda.synchronize()
        assert eng.ws["h2tb"] is None
        got[fused] = dict(loss=loss, grads=eng.grads.clone(), d1=eng.ws["d1"][:plan.R * 160].clone(),
                          h2b=eng.ws["h2b"][:Rp * 160].clone(), h2s=eng.ws["h2s"][:Rp * 160].clone())
    for k in got[False]:
        assert torch.equal(got[False][k], got[True][k]), k


def test_step_order_options_leave_the_step_bit_identical():
    """The late scheduling moves of round 2 change where and when kernels run, not what they compute: the gather and
    the W_ih fragments in one launch (svae_embed_fwd_wih_prep) against the two kernels, and the second time range's
    MLP chain on the side stream against the caller's, and the gathered embedding rows kept as bf16 (their consumers
    round them to bf16 anyway) against fp32 rows -- loss, every gradient and the parameters after the optimizer step
    are bit-identical, over three train steps of an ML-20M-shaped batch with a time split."""
    import svae_cf_amd as S
    from svae_cf_amd import synth
    I, B = 20000, 256
    _, _, mu, sigma, lmax = synth.SHAPES["ml-20m"]
    users = synth.make_users(2048, I, mu, sigma, lmax, seed=4321)
    plan = None
    for cand in synth.bucketed_batches([len(u) for u in users], B, np.random.default_rng(2)):
        p = S.build_plan([users[i] for i in cand], 4, user_keys=cand, device="cuda", total_items=I)
        if p.split_t > 0:
            plan = p
            break
    assert plan is not None
    got = {}
    for tag, head, mlp2, e16, pair in (("old", False, False, False, False), ("new", True, True, False, True),
                                       ("emb16", True, True, True, True), ("emb16_single", True, True, True, False)):
        eng = S.SVAEEngine(I, device="cuda", prec=S.PREC_BF16)
        _init(eng, 5)
        eng.head_fused, eng.mlp2_side, eng.emb16 = head, mlp2, e16
        eng.wgrad_pair = pair      # the two GRU weight gradients in one launch (svae_wgrad_pair_bf16dy) or in two
        eng.wgrad_big = False      # (the 256 x 256-tile kernel sums in another order: test_gru_gradient_array_forms_agree)
        eng.wgrad16 = False        # (the hand-waited kernel sums the two GRU BIAS gradients in another order: same test)
        eng.prepare(plan)
        losses = []
        for s in range(3):
            losses.append(eng.train_step(plan, 0.05, seed=9, step=s).clone())
        torch.cuda.synchronize()
        got[tag] = dict(losses=torch.stack(losses), grads=eng.grads.clone(), params=eng.params.clone(),
                        frags=eng.wih_frags[0].clone())
        if e16 and pair:      # the bf16 rows are the fp32 rows of the other runs, rounded to nearest even
            emb16 = eng.ws["emb16"][:plan.R * 256].clone()
        elif not e16:
            got[tag]["emb"] = eng.ws["emb"][:plan.R * 256].clone()
    for tag in ("new", "emb16", "emb16_single"):
        for k in got[tag]:
            assert torch.equal(got["old"][k], got[tag][k]), (tag, k)
    assert torch.equal(emb16, got["old"]["emb"].to(torch.bfloat16))


def test_two_range_bptt_equals_one_launch():
    """svae_gru_bwd_range_bf16out: the BPTT as two launches (timesteps >= tb, then < tb, the recurrent part of dL/dh
    handed over in dh_carry) against the single launch, on an ML-20M-shaped batch with a time split: the gate-gradient
    array bit-identical; the gradients equal up to the summation order of the weight gradients' row slabs (the two
    row ranges write disjoint slab sets)."""
    import svae_cf_amd as S
    from svae_cf_amd import synth
    I, B = 20000, 256
    _, _, mu, sigma, lmax = synth.SHAPES["ml-20m"]
    users = synth.make_users(2048, I, mu, sigma, lmax, seed=4242)
    plan = None
    for cand in synth.bucketed_batches([len(u) for u in users], B, np.random.default_rng(2)):
        p = S.build_plan([users[i] for i in cand], 4, user_keys=cand, device="cuda", total_items=I)
        if p.split_t > 0 and p.R % 64 != 0:
            plan = p
            break
    assert plan is not None
    got = {}
    for split in (False, True):
        eng = S.SVAEEngine(I, device="cuda", prec=S.PREC_BF16)
        _init(eng, 5)
        eng.bptt_split = split
        eng.prepare(plan)
        eng.forward(plan, 0.05, seed=9, step=2)
        eng.backward()
        torch.cuda.synchronize()
        got[split] = (eng.ws["dg4b"][:plan.R * 896].clone(), eng.grad_dict())
    assert torch.equal(got[False][0], got[True][0])
    for k in got[False][1]:
        a, b = got[True][1][k], got[False][1][k]
        assert _l2(a, b) <= 2e-6, (k, _l2(a, b))


def test_caller_supplied_noise_on_a_time_split_plan():
    """`eps` handed in by the caller (parity mode) on a plan with a time split: the second range's MLP chain runs on
    the side stream and reads the noise buffer, so the caller's noise must be in place before that stream is released
    (round 2's advisor finding: it was copied in afterwards, on the caller's stream).  Three steps with a DIFFERENT
    noise array each (a stale buffer would then show), the side-stream MLP on and off: bit-identical losses, gradients
    and parameters; a wrong-sized array is refused before anything is enqueued."""
    import svae_cf_amd as S
    from svae_cf_amd import synth
    I, B = 20000, 256
    _, _, mu, sigma, lmax = synth.SHAPES["ml-20m"]
    users = synth.make_users(2048, I, mu, sigma, lmax, seed=4321)
    plan = None
    for cand in synth.bucketed_batches([len(u) for u in users], B, np.random.default_rng(2)):
        p = S.build_plan([users[i] for i in cand], 4, user_keys=cand, device="cuda", total_items=I)
        if p.split_t > 0:
            plan = p
            break
    assert plan is not None
    g = torch.Generator(device="cuda").manual_seed(3)
    noise = [torch.randn(plan.R, 64, device="cuda", generator=g) for _ in range(3)]
    got = {}
    for side in (False, True):
        eng = S.SVAEEngine(I, device="cuda", prec=S.PREC_BF16)
        _init(eng, 5)
        eng.mlp2_side = side
        eng.prepare(plan)
        losses = [eng.train_step(plan, 0.05, eps=noise[s]).clone() for s in range(3)]
        torch.cuda.synchronize()
        got[side] = dict(losses=torch.stack(losses), grads=eng.grads.clone(), params=eng.params.clone())
        with pytest.raises(ValueError):
            eng.train_step(plan, 0.05, eps=noise[0][:-1])
        torch.cuda.synchronize()
        assert torch.equal(eng.params, got[side]["params"]), "a refused call must not have run anything"
    for k in got[False]:
        assert torch.equal(got[False][k], got[True][k]), k


@pytest.mark.parametrize("prec_name", ["bf16", "split"])
def test_a_step_does_not_depend_on_the_batch_before_it(prec_name):
    """Every kernel of a step must read what THIS step wrote: the workspace is reused from step to step, and a read that
    runs ahead of its producer (streams!) sees the batch before -- silently, in every test that repeats one plan.  Here
    an engine runs batch A and then batch B (other users, other lengths), a second engine of the same parameters runs
    batch B with its workspace poisoned (NaN in every fp32 array, 0x7fff patterns in the bf16 ones): loss, every
    gradient and every saved activation of batch B must be bit-identical -- and finite.  Forward + backward without an
    optimizer step in between, so that the two engines hold the same parameters; time-split plans, both precisions
    whose recurrence runs in two ranges; the second engine's side streams are held back for 3 ms first, so that a race
    between the streams is lost for certain.  (Round 3: this catches the first recurrence reading input gates that the
    side stream had not written yet -- rows between the decoder-side cut and row_off[t1].)"""
    import svae_cf_amd as S
    from svae_cf_amd import synth
    I, B = 20000, 128
    prec = S.PREC_BF16 if prec_name == "bf16" else S.PREC_SPLIT
    _, _, mu, sigma, lmax = synth.SHAPES["ml-20m"]
    users = synth.make_users(4 * B, I, mu, sigma, lmax, seed=77)
    plans = []
    for c in range(4):          # (unsorted chunks: mixed lengths, so that the plans have a time split)
        cand = np.arange(c * B, (c + 1) * B)
        p = S.build_plan([users[i] for i in cand], 4, user_keys=cand, device="cuda", total_items=I)
        if p.split_t > 0 and int(p._row_off_host[p.split_t]) % 64 != 0:
            plans.append(p)
    assert len(plans) >= 2, "need two time-split plans whose cut is not a multiple of 64 rows"
    pa, pb = plans[0], plans[1]
    assert pa.R != pb.R

    def engine():
        eng = S.SVAEEngine(I, device="cuda", prec=prec)
        _init(eng, 21)
        eng.reserve(max(pa.R, pb.R))
        eng.prepare(pa)
        eng.prepare(pb)
        return eng

    def run(eng, plan, hold_side_us=0):
        if hold_side_us:
            # race detector: the engine's side streams are held back, so that whatever the caller's stream reads from
            # them without waiting for them is still the poison (a lost race, on demand)
            from svae_cf_amd import _lib
            cur = torch.cuda.current_stream()
            sides = [eng._side_stream(which, cur=cur) for which in (0, 1)]     # (choosing one synchronises: first)
            for sd in sides:
                _lib.call("svae_spin", hold_side_us, sd.cuda_stream)
        loss = eng.forward(plan, 0.05, seed=9, step=4).clone()
        eng.backward()
        torch.cuda.synchronize()
        keep = {k: v[:plan.R * (v.numel() // eng.cap)].clone() for k, v in eng.ws.items()
                if k in ("gi", "hs", "gates4", "enc", "tmp", "z", "d1", "dd1", "dhs", "demb", "nll_row", "lse", "kl_row")}
        return loss, eng.grads.clone(), keep

    e1 = engine()
    # (on the engine's training stream, as train_batched runs its steps -- not the legacy default stream)
    with torch.cuda.stream(e1.training_stream()):
        run(e1, pa)
        got1 = run(e1, pb)
        e2 = engine()
        run(e2, pa)                                    # (a first, plain step: its side streams exist and are warm)
        for k, v in e2.ws.items():
            if v is None or v.numel() == 0:
                continue
            if v.dtype == torch.float32:
                v.fill_(float("nan"))
            elif k in ("dg4b", "emb16"):
                v.view(torch.int16).fill_(0x7fc0)      # bf16 NaN
        torch.cuda.synchronize()
        got2 = run(e2, pb, hold_side_us=3000)
    assert torch.isfinite(got2[0]).all() and torch.isfinite(got2[1]).all()
    assert torch.equal(got1[0], got2[0]), (got1[0], got2[0])
    for k in got1[2]:
        assert torch.equal(got1[2][k], got2[2][k]), k
    assert torch.equal(got1[1], got2[1])


def test_a_training_step_does_not_depend_on_the_batch_before_it():
    """The same detector over the whole `train_step` (optimizer and the floated decoder.linear2 work included): engine 1
    trains on batch A then batch B; engine 2 trains on A, then its workspace is poisoned, its side streams are held back
    for 3 ms, and it trains on B: the parameters and optimizer moments after B must be bit-identical and finite."""
    import svae_cf_amd as S
    from svae_cf_amd import synth, _lib
    I, B = 20000, 128
    _, _, mu, sigma, lmax = synth.SHAPES["ml-20m"]
    users = synth.make_users(4 * B, I, mu, sigma, lmax, seed=77)
    plans = []
    for c in range(4):
        cand = np.arange(c * B, (c + 1) * B)
        p = S.build_plan([users[i] for i in cand], 4, user_keys=cand, device="cuda", total_items=I)
        if p.split_t > 0:
            plans.append(p)
    assert len(plans) >= 2
    pa, pb = plans[0], plans[1]
    out = []
    for poisoned in (False, True):
        eng = S.SVAEEngine(I, device="cuda", prec=S.PREC_BF16)
        _init(eng, 21)
        eng.reserve(max(pa.R, pb.R))
        eng.prepare(pa)
        eng.prepare(pb)
        with torch.cuda.stream(eng.training_stream()):
            eng.train_step(pa, 0.05, seed=9, step=0)
            if poisoned:
                eng._settle()
                torch.cuda.synchronize()
                for k, v in eng.ws.items():
                    if v is None or v.numel() == 0:
                        continue
                    if v.dtype == torch.float32:
                        v.fill_(float("nan"))
                    elif k in ("dg4b", "emb16"):
                        v.view(torch.int16).fill_(0x7fc0)
                for t in (eng.slabs, eng.slabs_side, eng.slabs_hh):
                    t.fill_(float("nan"))
                torch.cuda.synchronize()
                cur = torch.cuda.current_stream()
                sides = [eng._side_stream(which, cur=cur) for which in (0, 1)]
                for sd in sides:
                    _lib.call("svae_spin", 3000, sd.cuda_stream)
            eng.train_step(pb, 0.05, seed=9, step=1)
            eng._settle()
        torch.cuda.synchronize()
        out.append((eng.params.clone(), eng.exp_avg.clone(), eng.exp_avg_sq.clone(), eng.loss_out.clone()))
    for a, b in zip(*out):
        assert torch.isfinite(b).all()
        assert torch.equal(a, b)


def test_scoring_does_not_depend_on_the_batch_before_it():
    """The detector over `score` (evaluate()'s per-batch work): batch A then batch B against batch B on a poisoned
    workspace with the side streams held back -- loss, top-100 ids and values bit-identical."""
    import svae_cf_amd as S
    from svae_cf_amd import synth, _lib
    I, B = 20000, 128
    _, _, mu, sigma, lmax = synth.SHAPES["ml-20m"]
    users = synth.make_users(2 * B, I, mu, sigma, lmax, seed=78)
    pa, pb = (S.build_plan(users[c * B:(c + 1) * B], 4, user_keys=np.arange(c * B, (c + 1) * B), device="cuda",
                           total_items=I) for c in range(2))
    out = []
    for poisoned in (False, True):
        eng = S.SVAEEngine(I, device="cuda", prec=S.PREC_BF16)
        _init(eng, 21)
        eng.reserve(max(pa.R, pb.R))
        eng.prepare(pa, scoring=True)
        eng.prepare(pb, scoring=True)
        eng.score(pa, 100, seed=3, step=0)
        torch.cuda.synchronize()
        if poisoned:
            for k, v in eng.ws.items():
                if v is not None and v.numel() and v.dtype == torch.float32:
                    v.fill_(float("nan"))
            torch.cuda.synchronize()
            cur = torch.cuda.current_stream()
            sides = [eng._side_stream(which, cur=cur) for which in (0, 1)]
            for sd in sides:
                _lib.call("svae_spin", 3000, sd.cuda_stream)
        loss, idx, val, _ = eng.score(pb, 100, seed=3, step=1)
        torch.cuda.synchronize()
        out.append((loss.clone(), idx.clone(), val.clone()))
    for a, b in zip(*out):
        assert torch.isfinite(b.float()).all()
        assert torch.equal(a, b)


def test_scoring_session_on_three_lanes_equals_call_by_call_scoring():
    """engine.scoring_session(): parameter-derived operands made once, batches alternating between the engine's three
    workspaces on three streams (what notebook_api.evaluate does).  Six batches of unequal size -- the largest LAST, so the
    workspaces grow inside the session -- against the same batches scored one call at a time outside a session: loss,
    top-100 ids and values bit-identical; the poisoned variant fills both workspaces with NaN between two passes (a
    slot must not read what the batch before left).  A training step after the session sees fresh operands again."""
    import svae_cf_amd as S
    from svae_cf_amd import synth
    I = 20000
    _, _, mu, sigma, lmax = synth.SHAPES["ml-20m"]
    sizes = [96, 64, 128, 32, 80, 200]
    users = synth.make_users(sum(sizes), I, mu, sigma, lmax, seed=79)
    plans, at = [], 0
    for n in sizes:
        plans.append(S.build_plan(users[at:at + n], 4, user_keys=np.arange(at, at + n), device="cuda", total_items=I))
        at += n
    eng = S.SVAEEngine(I, device="cuda", prec=S.PREC_BF16)
    _init(eng, 22)
    for p in plans:
        eng.prepare(p, scoring=True)
    ref = []
    for k, p in enumerate(plans):
        loss, idx, val, _ = eng.score(p, 100, seed=3, step=k)
        ref.append((loss.clone(), idx.clone(), val.clone()))
    torch.cuda.synchronize()
    eng2 = S.SVAEEngine(I, device="cuda", prec=S.PREC_BF16)        # a fresh engine: its workspaces start empty
    _init(eng2, 22)
    with pytest.raises(ValueError):
        eng2.score(plans[0], 100, slot=1)                          # slot 1 outside a session
    cur = torch.cuda.current_stream()
    for poisoned in (False, True):
        got = []
        with eng2.scoring_session():
            lanes = eng2.scoring_lanes(3)
            assert len(lanes) == 3 and lanes[0] is not lanes[1] and lanes[1] is not lanes[2]
            for s_ in lanes[1:]:
                s_.wait_stream(cur)
            for k, p in enumerate(plans):
                with torch.cuda.stream(lanes[k % 3]):
                    loss, idx, val, _ = eng2.score(p, 100, seed=3, step=k, slot=k % 3)
                    got.append((loss.clone(), idx.clone(), val.clone()))
            for s_ in lanes[1:]:
                cur.wait_stream(s_)
        torch.cuda.synchronize()
        for k, (a, b) in enumerate(zip(ref, got)):
            for x, y in zip(a, b):
                assert torch.isfinite(y.float()).all(), (poisoned, k)
                assert torch.equal(x, y), (poisoned, k)
        if not poisoned:
            for ws in (eng2.ws, eng2._slots[1][0], eng2._slots[2][0]):
                for v in ws.values():
                    if v is not None and v.numel() and v.dtype == torch.float32:
                        v.fill_(float("nan"))
            torch.cuda.synchronize()
    assert eng2._slot == 0 and not eng2._session
    # the session's cached operands do not outlive it: a training step, then call-by-call scoring = the same on `eng`
    tp = S.build_plan(users[:64], 4, user_keys=np.arange(64), device="cuda", total_items=I)
    for e in (eng, eng2):
        e.train_step(tp, 0.1, seed=5, step=0)
    a = eng.score(plans[1], 100, seed=3, step=1)
    with eng2.scoring_session():
        b = eng2.score(plans[1], 100, seed=3, step=1)
    torch.cuda.synchronize()
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    assert not torch.equal(a[0], ref[1][0])                        # (the step did change the parameters)


def test_side_streams_are_not_held_behind_the_callers_dispatch():
    """HIP maps a process's streams onto 4 hardware queues; two streams on one queue still run CONCURRENTLY (no barrier
    between their packets) but are DISPATCHED in order, so a 4 000-workgroup finish kernel or optimizer launch on one
    holds every launch of the other for its whole length (round 3: a loop on the default stream after an epoch on the
    training stream ran 1.17 instead of 0.85 ms per step).  The engine's choice of side streams tests for it: whatever
    stream the caller uses, in whatever order, the streams chosen beside it pass both tests."""
    import svae_cf_amd as S
    from svae_cf_amd import synth
    I = 2000
    users = synth.make_users(64, I, 3.0, 0.6, 60, seed=5)
    plan = S.build_plan(users, 4, user_keys=np.arange(64), device="cuda", total_items=I)
    eng = S.SVAEEngine(I, device="cuda", prec=S.PREC_BF16)
    _init(eng, 3)
    with torch.cuda.stream(eng.training_stream()):
        eng.train_step(plan, 0.1, seed=1)
    eng.train_step(plan, 0.1, seed=1)                       # the default stream, AFTER the training stream
    with torch.cuda.stream(torch.cuda.Stream()):
        eng.train_step(plan, 0.1, seed=1)                   # and a stream of the caller's own
    torch.cuda.synchronize()
    assert len(eng.side_stream_info) >= 6
    for info in eng.side_stream_info:
        assert info["concurrent"] and info["dispatch_independent"], eng.side_stream_info


def test_first_time_range_keeps_half_the_chip_at_large_batches():
    """The first range's forward sweep leaves the second range's recurrence its CUs (one workgroup per 4 users) -- at
    B = 1024 that difference is 0, and with the flat schedule it IS the workgroup count: 8 workgroups swept the whole
    range (round 3, 1M items: 125 ms instead of 21).  Never under half the chip."""
    import types
    import svae_cf_amd as S
    eng = S.SVAEEngine(500, device="cuda", prec=S.PREC_BF16)
    for B in (1, 64, 256, 512, 1024, 4096):
        n = eng._range1_cus(types.SimpleNamespace(B=B))
        assert eng._ncu // 2 <= n <= eng._ncu, (B, n)
    assert eng._range1_cus(types.SimpleNamespace(B=256)) == eng._ncu - 64
    # the weight-gradient sweep's share of the chip grows with the catalogue (measured optima, engine._dw_workgroups)
    big = S.SVAEEngine(200000, device="cuda", prec=S.PREC_BF16)
    assert eng._dw_workgroups(20000, eng._ncu) == min(eng._ncu, eng.dw_wgs)
    assert big._dw_workgroups(20000, big._ncu) == big._ncu - 32 and big._range1_cus(types.SimpleNamespace(B=256)) == big._ncu


def test_training_is_bit_reproducible_over_many_steps():
    """160 training steps over 16 ML-20M-shaped plans, three times from the same state: parameters and optimizer state end
    BIT-IDENTICAL.  Every reduction in the step has a fixed order (slabs, no atomics), so any difference is a missing
    dependency between streams or inside a kernel -- round 3 found one this way: an LDS reduction behind a bare s_barrier
    (no wait for the LDS writes in front of it) made two bias gradients differ in the last bits every few dozen steps."""
    import svae_cf_amd as S
    from svae_cf_amd import synth
    I, B = 20000, 256
    _, _, mu, sigma, lmax = synth.SHAPES["ml-20m"]
    users = synth.make_users(4096, I, mu, sigma, lmax, seed=98765)
    batches = synth.bucketed_batches([len(u) for u in users], B, np.random.default_rng(98765))[:16]
    plans = [S.build_plan([users[i] for i in idx], 4, user_keys=idx, device="cuda", total_items=I) for idx in batches]
    outs = []
    for run in range(3):
        eng = S.SVAEEngine(I, device="cuda", prec=S.PREC_BF16)
        _init(eng, 7)
        eng.reserve(max(p.R for p in plans))
        for p in plans:
            eng.prepare(p)
        with torch.cuda.stream(eng.training_stream()):
            for i in range(160):
                eng.train_step(plans[i % 16], 0.05, seed=3)
        torch.cuda.synchronize()
        outs.append((eng.params.clone(), eng.exp_avg.clone(), eng.exp_avg_sq.clone()))
        assert bool(torch.isfinite(outs[-1][0]).all())
    for run in (1, 2):
        for a, b, what in zip(outs[0], outs[run], ("parameters", "first moments", "second moments")):
            assert torch.equal(a, b), (run, what, int((a != b).sum()))
