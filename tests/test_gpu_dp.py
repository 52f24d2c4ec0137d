"""Data-parallel step on the GPU: 2 ranks that share the card (gloo all-reduce of the flat gradient buffer
through GradSync, exactly the path bench.py and train_batched take for N>1) must reproduce the
single-process step over the UNION of the two ranks' users.

Contract checked: step over S = mean over u in S of the reference's single-user loss, so
sum-all-reduce x 1/world of the per-rank gradients = the gradient of the union batch; Philox noise is keyed
by (user id, t), so a user's noise does not depend on which rank holds it.
Tolerance: parameters after 3 Adam steps, per-tensor relative L2 <= 1e-4 (the north-star gate; the two runs
differ only in fp32 summation order, and in bf16 mode in nothing else either).
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
I, B, WORLD, STEPS = 300, 24, 2, 3


def _users():
    from svae_cf_amd import synth
    return synth.make_users(STEPS * B * WORLD, I, 2.8, 0.6, 60, seed=5)


def _init_sd(layout):
    g = torch.Generator().manual_seed(3)
    return {n: torch.randn(s, generator=g) * 0.1 for n, s in layout.ref_shapes.items()}


def _run(rank, world, prec, port, q, opt="adam", sparse=False, shard=False):
    sys.path.insert(0, ROOT)
    if q is not None:
        # a rank that stops making progress dumps its stacks and exits instead of hanging the test run
        import faulthandler
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        fh = open(os.path.join(ROOT, "gpurun_out", "dp_test_rank%d_prec%d.trace" % (rank, prec)), "w")
        faulthandler.dump_traceback_later(70, exit=True, file=fh)
    import torch.distributed as dist
    import svae_cf_amd as S
    from svae_cf_amd import synth
    from svae_cf_amd.dist import GradSync, rank_batches
    if world > 1:
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    users = _users()
    lens = [len(u) for u in users]
    eng = S.SVAEEngine(I, device="cuda", prec=prec)
    eng.load_state_dict(_init_sd(eng.layout))
    if opt == "adagrad":        # nb:1252-1255: the gradient scale 1/world goes through svae_optimizer_step as well
        eng.set_optimizer("adagrad", 0.01, 5e-3, decay=0.0, eps=1e-10)
    steps = synth.bucketed_batches(lens, B, np.random.default_rng(11), world=WORLD)
    mine = rank_batches(lens, B, np.random.default_rng(11), rank, WORLD) if world > 1 else None
    if world > 1:
        eng.grad_sync = GradSync(eng.grads, eng.layout.buckets())
        if sparse:      # the item-embedding gradient as (row id, row) pairs: the large-catalogue exchange
            from svae_cf_amd.layout import E
            eng.grad_sync.enable_sparse_rows(2, I, E, capacity=max(len(u) for u in users) * B)
        if shard:       # decoder.linear2: reduce-scatter, optimizer over the rank's shard, parameters all-gathered
            eng.grad_sync.enable_sharded(0)
    losses = []
    for s in range(STEPS):
        idx = mine[s] if world > 1 else np.concatenate(steps[s])
        plan = S.build_plan([users[i] for i in idx], 4, user_keys=idx, device="cuda", total_items=I)
        out = eng.train_step(plan, 0.05, seed=77, step=s)
        losses.append(float(out[0]))
    torch.cuda.synchronize()
    sd = {k: v.cpu().numpy() for k, v in eng.state_dict().items()}
    if world > 1:
        t = torch.tensor(losses, dtype=torch.float64)
        dist.all_reduce(t)
        losses = (t / world).tolist()
        dist.destroy_process_group()
    if q is None:
        return sd, losses
    if rank == 0:
        q.put((sd, losses))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("prec,opt,sparse,shard", [(0, "adam", False, False), (1, "adam", False, False),
                                                   (0, "adagrad", False, False), (0, "adam", True, False),
                                                   (1, "adam", False, True), (0, "adagrad", True, True)])
def test_two_ranks_equal_union_batch(prec, opt, sparse, shard):
    import torch.multiprocessing as mp
    ref_sd, ref_losses = _run(0, 1, prec, 0, None, opt)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_run, args=(r, WORLD, prec, port, q, opt, sparse, shard), daemon=True) for r in range(WORLD)]
    for p in ps:
        p.start()
    try:
        sd, losses = q.get(timeout=100)
        for p in ps:
            p.join(30)
            assert p.exitcode == 0
    finally:
        for p in ps:
            if p.is_alive():
                p.kill()
    np.testing.assert_allclose(losses, ref_losses, rtol=1e-5 if prec == 0 else 1e-3)
    for k, v in ref_sd.items():
        d = np.linalg.norm(sd[k].astype(np.float64) - v) / np.linalg.norm(v)
        assert d <= 1e-4, (k, d)


# ---------------------------------------------------------------------------------------------
# sharded evaluate(): users dealt to the ranks, per-user metric rows gathered -> the single-process result
# ---------------------------------------------------------------------------------------------
def _eval_setup():
    from svae_cf_amd import notebook_api as N
    from svae_cf_amd import synth
    hp = {"history_split_test": [0.8, 0.2], "weight_decay": 5e-3, "batch_size": 1, "item_embed_size": 256,
          "rnn_size": 200, "hidden_size": 150, "latent_size": 64, "loss_type": "next_k", "next_k": 4,
          "number_users_to_keep": 10 ** 9, "batch_log_interval": 10 ** 9, "train_cp_users": 20, "total_items": I}
    users = synth.make_users(70, I, 2.8, 0.6, 60, seed=9)
    lines = lambda us: ["%d,%d,1\n" % (u, int(it)) for u, items in enumerate(us) for it in items]
    fold = [u[:int(0.8 * len(u))] for u in users[:37]]
    held = [u[int(0.8 * len(u)):] for u in users[:37]]
    val_reader = N.DataReader(hp, lines(fold), lines(held), I, False)
    train_reader = N.DataReader(hp, lines(users[37:]), None, I, True)
    model = N.Model(hp, seed=4)
    return N, hp, model, val_reader, train_reader


def _eval_run(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import faulthandler
    faulthandler.dump_traceback_later(70, exit=True)
    import torch.distributed as dist
    if world > 1:
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    N, hp, model, val_reader, train_reader = _eval_setup()
    crit = N.VAELoss(hp, model)
    out = [N.evaluate(model, crit, val_reader, hp, False, batch_size=8, seed=3, rank=rank, world=world),
           N.evaluate(model, crit, train_reader, hp, True, batch_size=8, seed=3, rank=rank, world=world)]
    if world > 1:
        dist.destroy_process_group()
    if q is None:
        return out
    q.put((rank, out))


def test_sharded_evaluate_equals_single_process():
    """evaluate(..., rank, world) on 2 ranks (5 and 3 user batches dealt round-robin; the train-set early break
    included): every rank returns the metric dict and len_to_ndcg_at_100_map of the single-process call, exactly."""
    import torch.multiprocessing as mp
    ref = _eval_run(0, 1, 0, None)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_eval_run, args=(r, WORLD, port, q), daemon=True) for r in range(WORLD)]
    for p in ps:
        p.start()
    try:
        got = dict(q.get(timeout=100) for _ in ps)
        for p in ps:
            p.join(30)
            assert p.exitcode == 0
    finally:
        for p in ps:
            if p.is_alive():
                p.kill()
    for r in range(WORLD):
        for (m, lm), (m0, lm0) in zip(got[r], ref):
            assert m == m0, (r, m, m0)
            assert lm == lm0
    assert ref[0][0]["NDCG@100"] > 0 and ref[1][0]["loss"] > 0


def test_evaluate_keeps_its_plans_on_the_device_between_calls():
    """evaluate() on the same reader again (fit() does, every epoch): the second call reads the plans the first one left
    on the device (no builder threads) and returns the same metrics and length map to the last digit; a training step in
    between changes the metrics but not the plans; another batch size builds new plans."""
    import svae_cf_amd as S
    N, hp, model, val_reader, train_reader = _eval_setup()
    crit = N.VAELoss(hp, model)
    a = N.evaluate(model, crit, val_reader, hp, False, batch_size=8, seed=3)
    held = val_reader._svae_eval_plans
    assert len(held[2]) == 5 and all(p.x.is_cuda for p in held[2])
    b = N.evaluate(model, crit, val_reader, hp, False, batch_size=8, seed=3)
    assert val_reader._svae_eval_plans is held                       # (nothing was rebuilt)
    assert a[0] == b[0] and a[1] == b[1]
    c = N.evaluate(model, crit, val_reader, hp, False, batch_size=16, seed=3)
    assert val_reader._svae_eval_plans is not held and len(val_reader._svae_eval_plans[2]) == 3
    # (c differs from a: the latent noise is keyed by (seed, first user of the batch, user), so it moves with the batching)
    users = [np.asarray(f) for f in train_reader.data[:8]]
    plan = S.build_plan(users, 4, device="cuda", total_items=I)
    model.engine.train_step(plan, 0.1, seed=1, step=0)
    d = N.evaluate(model, crit, val_reader, hp, False, batch_size=16, seed=3)
    assert d[0]["loss"] != c[0]["loss"]


@pytest.mark.parametrize("consumer_uploads", [True, False])
def test_prefetched_plans_equal_directly_built_ones(consumer_uploads):
    """PlanPrefetcher (builder threads; the uploads issued by the consumer's thread or by the builders) hands out, in
    order, plans byte-identical to build_plan_native on the same users, with the engine's tables attached and an event to
    wait for; leaving the loop early neither hangs nor leaks a worker."""
    import svae_cf_amd as S
    from svae_cf_amd import synth
    from svae_cf_amd.batch import PlanPrefetcher, UserCSR, build_plan_native
    Ibig = 5000
    _, _, mu, sigma, lmax = synth.SHAPES["ml-1m"]
    users = synth.make_users(400, Ibig, mu, sigma, lmax, seed=12)
    csr = UserCSR([u.tolist() for u in users])
    rng = np.random.default_rng(3)
    batches = [np.sort(rng.choice(400, size=n, replace=False)).astype(np.int64) for n in (64, 7, 128, 33, 64, 90)]
    eng = S.SVAEEngine(Ibig, device="cuda", prec=S.PREC_BF16)
    pf = PlanPrefetcher(batches, csr, 4, eng.device, total_items=Ibig, prepare=(eng.prepare_host, eng.prepare_upload),
                        workers=2, depth=4, consumer_uploads=consumer_uploads)
    assert pf.consumer_uploads == consumer_uploads
    cur = torch.cuda.current_stream()
    n = 0
    for idx, plan in zip(batches, pf):
        assert plan.ready is not None and plan.x.is_cuda
        cur.wait_event(plan.ready)
        ref = build_plan_native(csr, idx, 4, user_keys=idx, device="cuda", total_items=Ibig)
        assert (plan.B, plan.R, plan.Tmax, plan.K, plan.split_t, plan.split_r) == (ref.B, ref.R, ref.Tmax, ref.K, ref.split_t, ref.split_r)
        assert set(plan._offs) == set(ref._offs)
        for name in plan._offs:                                   # (the gaps between the packed arrays are not written)
            assert torch.equal(getattr(plan, name), getattr(ref, name)), name
        assert np.array_equal(plan.order, ref.order) and np.array_equal(plan._row_off_host, ref._row_off_host)
        assert any(k[0] == "rows" for k in plan.proj_sched) or eng._sched_cache
        out = eng.train_step(plan, 0.05, seed=1, step=n)          # (the plan is usable as handed out)
        n += 1
    torch.cuda.synchronize()
    assert n == len(batches) and bool(torch.isfinite(out[0]))
    pf2 = PlanPrefetcher(batches, csr, 4, eng.device, total_items=Ibig, prepare=eng.prepare, workers=2, depth=4,
                         consumer_uploads=consumer_uploads)
    for k, plan in enumerate(pf2):
        if k == 1:
            break                                                  # the generator's clean-up shuts the pool down
    torch.cuda.synchronize()


# ---------------------------------------------------------------------------------------------
# the C-ABI's own collectives (svae_rccl_* / svae_allreduce_sum_f32 / reduce-scatter / all-gather): RCCL at world size 1
# ---------------------------------------------------------------------------------------------
def test_c_abi_rccl_wrappers_world_size_one():
    """The thin RCCL wrappers of the C-ABI (hosts without torch.distributed), on the one GPU there is: a communicator
    of one rank from a unique id, the in-place sum all-reduce of a bucket (identity at one rank), and the
    reduce-scatter / all-gather pair of the sharded bucket (one shard = the bucket).  Checks the call path -- library
    loading, argument order, stream ordering -- not a multi-GPU result, which needs hardware this box does not have."""
    import ctypes
    sys.path.insert(0, ROOT)
    from svae_cf_amd import _lib
    uid = (ctypes.c_char * 128)()
    _lib.call("svae_rccl_unique_id", ctypes.cast(uid, ctypes.c_void_p))
    assert any(b != b"\x00" for b in uid)
    comm = ctypes.c_void_p()
    _lib.call("svae_rccl_comm_create", ctypes.byref(comm), 1, 0, ctypes.cast(uid, ctypes.c_void_p))
    try:
        st = torch.cuda.current_stream().cuda_stream
        g = torch.randn(4096 + 512, device="cuda")
        want = g.clone()
        _lib.call("svae_allreduce_sum_f32", comm, g.data_ptr(), g.numel(), st)
        _lib.call("svae_reduce_scatter_sum_f32", comm, g.data_ptr(), g.numel(), 0, st)
        _lib.call("svae_allgather_f32", comm, g.data_ptr(), g.numel(), 0, st)
        torch.cuda.synchronize()
        assert torch.equal(g, want)
        with pytest.raises(_lib.SvaeHipError):
            _lib.call("svae_allreduce_sum_f32", comm, g.data_ptr(), 0, st)
    finally:
        _lib.call("svae_rccl_comm_destroy", comm)


def test_sparse_rows_rebuild_kernel_equals_the_rank_ordered_sum():
    """svae_sparse_rows_rebuild (the rebuild behind the sparse exchange of the embedding gradient): own rows zeroed, then
    every rank's (id, row) pairs added in rank order, pad entries (-1) skipped whatever their values hold (Inf / NaN
    left over from an earlier step must not leak: round 2's advisor finding).  Bit-identical to the same additions in
    the same order on the host."""
    sys.path.insert(0, ROOT)
    from svae_cf_amd import _lib
    rs = np.random.RandomState(4)
    rows, width, world, cap = 5000, 256, 3, 700
    table = rs.standard_normal((rows, width)).astype(np.float32)
    all_ids = np.full((world, cap), -1, dtype=np.int64)
    all_vals = rs.standard_normal((world, cap, width)).astype(np.float32)
    counts = [650, 0, 700]
    for r, n in enumerate(counts):
        all_ids[r, :n] = rs.permutation(rows)[:n]
        all_vals[r, n:] = np.float32("nan") if r == 0 else np.float32("inf")       # stale contents behind the count
    own = all_ids[0, :counts[0]].copy()
    ref = table.copy()
    ref[own] = 0.0
    for r in range(world):
        for i in range(counts[r]):
            ref[all_ids[r, i]] = ref[all_ids[r, i]] + all_vals[r, i]
    t = torch.from_numpy(table).cuda()
    ids_d, vals_d, own_d = torch.from_numpy(all_ids).cuda(), torch.from_numpy(all_vals).cuda(), torch.from_numpy(own).cuda()
    _lib.call("svae_sparse_rows_rebuild", t.data_ptr(), rows, width, ids_d.data_ptr(), vals_d.data_ptr(), world, cap,
              own_d.data_ptr(), len(own), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(t.cpu(), torch.from_numpy(ref))


@pytest.mark.parametrize("sparse,sharded", [(False, False), (True, True)])
def test_native_collectives_at_world_size_one_reproduce_the_plain_step(sparse, sharded):
    """dist.GradSync(native=True): the buckets' collectives through the library's own RCCL wrappers, issued in order on
    the streams that produce them (one communicator per stream), at world size 1 -- all there is on a one-GPU box.  The
    step with them (dense all-reduce, or the embedding gradient as (id, row) pairs + the decoder.linear2 bucket as
    reduce-scatter / sharded optimizer / all-gather) must leave the parameters of the plain single-process step, bit
    for bit, over three steps."""
    sys.path.insert(0, ROOT)
    import svae_cf_amd as S
    from svae_cf_amd import synth
    from svae_cf_amd.dist import GradSync
    from svae_cf_amd.layout import E
    I, B = 20000, 128
    _, _, mu, sigma, lmax = synth.SHAPES["ml-20m"]
    users = synth.make_users(B, I, mu, sigma, lmax, seed=77)
    plan = S.build_plan(users, 4, user_keys=np.arange(B), device="cuda", total_items=I)
    got = []
    for native in (False, True):
        eng = S.SVAEEngine(I, device="cuda", prec=S.PREC_BF16)
        g = torch.Generator().manual_seed(3)
        eng.load_state_dict({n: (torch.randn(s_, generator=g) * (0.07 if n != "item_embed.weight" else 1.0))
                             for n, s_ in eng.layout.ref_shapes.items()})
        eng.prepare(plan)
        gs = None
        if native:
            gs = eng.grad_sync = GradSync(eng.grads, eng.layout.buckets(), force=True, native=True)
            assert gs.native and gs.active and gs.world == 1
            if sparse:
                gs.enable_sparse_rows(2, I, E, capacity=plan.R)
            if sharded:
                gs.enable_sharded(0)
        for s_ in range(3):
            eng.train_step(plan, 0.05, seed=9, step=s_)
        torch.cuda.synchronize()
        got.append((eng.params.clone(), eng.loss_out.clone()))
        if gs is not None:
            assert len(gs._comms) >= 1
            gs.close()
    assert torch.equal(got[0][1], got[1][1])
    assert torch.equal(got[0][0], got[1][0])
