"""CPU-only checks of the host side: batch plans, parameter layout, the notebook-surface data reader and
metric arithmetic (against the reference golden vectors), the C-ABI library's exported symbols, and the
data-parallel gradient sync over gloo (world_size 2)."""
import ctypes
import os
import re
import socket
import sys

import numpy as np
import pytest
import torch

from oracle import svae_numpy as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"), allow_pickle=False)


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "svae_hip.h")).read()
    names = sorted(set(re.findall(r"\b(svae_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 25
    lib = ctypes.CDLL(os.path.join(ROOT, "svae_cf_amd", "csrc", "libsvae_hip.so"))
    for n in names:
        assert hasattr(lib, n), n
    from svae_cf_amd import _lib
    for n in _lib.SIGNATURES:
        assert n in names, "bound but not declared in include/svae_hip.h: " + n
    assert lib.svae_abi_version() == 1


def test_missing_library_fails_loudly(monkeypatch):
    from svae_cf_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libsvae_hip.so")
    with pytest.raises(_lib.SvaeHipError):
        _lib.load()


def test_engine_refuses_to_run_without_gpu():
    import svae_cf_amd as S
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        S.SVAEEngine(100)


def test_layout_pack_unpack_roundtrip():
    from svae_cf_amd.layout import ParamLayout
    lay = ParamLayout(97)
    p = O.make_params(97, 3)
    flat = torch.zeros(lay.total)
    lay.pack({k: torch.from_numpy(v) for k, v in p.items()}, flat)
    out = lay.unpack(flat)
    assert list(out.keys())[0] == "encoder.linear1.weight"
    for k in O.PARAM_NAMES:
        np.testing.assert_array_equal(out[k].numpy(), p[k])
    # pads are zero; buckets tile the buffer in backward-readiness order
    assert float(flat.abs().sum()) == pytest.approx(sum(float(np.abs(v).sum()) for v in p.values()), rel=1e-5)
    b = lay.buckets()
    assert b[0][0] == 0 and b[-1][1] == lay.total and all(b[i][1] == b[i + 1][0] for i in range(len(b) - 1))
    assert lay.n_params == 407 * 97 + 334028


@pytest.mark.parametrize("k", [1, 4, 10 ** 9])
def test_plan_rows_and_targets_match_oracle(k):
    import svae_cf_amd as S
    rs = np.random.RandomState(1)
    I = 60
    users = [rs.permutation(I)[:n] for n in (7, 2, 12, 7, 3)]
    plan = S.build_plan(users, k, total_items=I)
    assert plan.R == sum(len(u) - 1 for u in users) and plan.B == 5
    assert np.all(np.diff(plan.nact.numpy()) <= 0)
    x, tgt = plan.x.numpy(), plan.tgt.numpy()
    for su in range(plan.B):
        u = int(plan.order[su])
        ox, ot = O.build_inputs_targets(users[u], k)
        rows = plan.row_of[su]
        np.testing.assert_array_equal(x[rows], ox)
        np.testing.assert_array_equal(tgt[rows][:, :ot.shape[1]], ot)
        assert np.all(tgt[rows][:, ot.shape[1]:] == -1)
        n1 = (ot[0] >= 0).sum()
        np.testing.assert_allclose(plan.w_nll.numpy()[rows], 1.0 / (plan.B * n1))
        np.testing.assert_allclose(plan.w_kl.numpy()[rows], 1.0 / (plan.B * len(ox)))
        assert plan.last_row.numpy()[su] == rows[-1]
    # embedding scatter segments and target occurrences cover every row / target exactly once
    perm, so, si = plan.perm.numpy(), plan.seg_off.numpy(), plan.seg_item.numpy()
    assert sorted(perm.tolist()) == list(range(plan.R))
    for s in range(plan.nseg):
        assert np.all(x[perm[so[s]:so[s + 1]]] == si[s])
    occ_off, occ_row = plan.occ_off.numpy(), plan.occ_row.numpy()
    assert occ_off[-1] == (tgt >= 0).sum()
    for it in range(I):
        for r in occ_row[occ_off[it]:occ_off[it + 1]]:
            assert it in tgt[r]


def test_plan_rejects_bad_input():
    import svae_cf_amd as S
    with pytest.raises(ValueError):
        S.build_plan([], 4)
    with pytest.raises(ValueError):
        S.build_plan([np.array([3])], 4)


def test_duplicate_targets_count_once():
    import svae_cf_amd as S
    plan = S.build_plan([np.array([1, 2, 2, 2, 5])], 4)
    t0 = plan.tgt.numpy()[0]
    assert sorted(t0[t0 >= 0].tolist()) == [2, 5]
    y = O.dense_targets(O.build_inputs_targets(np.array([1, 2, 2, 2, 5]), 4)[1], 8)
    assert y[0].sum() == 2


@pytest.mark.parametrize("name", ["fwd_bwd_I97_k4", "fwd_bwd_I211_k1", "fwd_bwd_I64_postfix"])
def test_datareader_matches_reference_reader(golden_dir, name):
    """Same shuffled order (numpy global stream, nb:493), same x, same targets as the reference's DataReader.iter."""
    from svae_cf_amd import notebook_api as N
    g = _load(golden_dir, name)
    I, seed, n = int(g["total_items"]), int(g["seed"]), int(g["n_users"])
    rs = np.random.RandomState(seed)
    lens = {"fwd_bwd_I97_k4": [18, 5, 9], "fwd_bwd_I211_k1": [51, 3], "fwd_bwd_I64_postfix": [7, 6]}[name]
    users = [rs.permutation(I)[:m].astype(np.int64) for m in lens]
    hp = {"batch_size": 1, "number_users_to_keep": 10 ** 9, "loss_type": str(g["loss_type"]),
          "next_k": int(g["next_k"]), "history_split_test": [0.8, 0.2]}
    lines = ["%d,%d,1\n" % (u, int(it)) for u, items in enumerate(users) for it in items]
    reader = N.DataReader(hp, lines, None, I, True)
    assert reader.num_b == n
    np.random.seed(seed)
    got = list(reader.iter())
    assert len(got) == n
    for j, (x, y) in enumerate(got):
        np.testing.assert_array_equal(x[0].numpy(), g["u%d_x" % j])
        nz = torch.nonzero(y.dense()[0]).numpy()
        np.testing.assert_array_equal(nz, g["u%d_y_nz" % j])


@pytest.mark.parametrize("name", ["eval_val_I211", "eval_train_I211"])
def test_metric_arithmetic_matches_reference_evaluate(golden_dir, name):
    from svae_cf_amd import notebook_api as N
    g = _load(golden_dir, name)
    n_eval = int(g["n_eval"])
    held = [list(g["held%d" % u]) for u in range(n_eval)]
    n_in = [len(g["fold%d" % u]) - 1 for u in range(n_eval)]
    sums, len_map = N._metrics_from_rankings([g["top100"][u] for u in range(n_eval)], held, n_in)
    ref = dict(zip([str(s) for s in g["metric_names"]], g["metric_values"]))
    for k, v in sums.items():
        assert round(100.0 * v / n_eval, 4) == pytest.approx(ref[k], abs=1e-9)
    assert sorted(len_map.keys()) == list(g["len_map_keys"])


def test_anneal_schedule_matches_reference(golden_dir):
    from svae_cf_amd import notebook_api as N
    g = _load(golden_dir, "train_I97")
    for b, a in enumerate(g["anneals"], start=1):
        assert N.anneal_for_batch(b) == float(a)


def test_shard_batches_partitions_users():
    from svae_cf_amd.dist import shard_batches
    from svae_cf_amd.synth import bucketed_batches
    lens = np.random.default_rng(0).integers(5, 100, 1000)
    batches = bucketed_batches(lens, 32, np.random.default_rng(1))
    assert sorted(np.concatenate(batches).tolist()) == list(range(1000))
    parts = [shard_batches(batches, r, 3) for r in range(3)]
    assert len({len(p) for p in parts}) == 1
    allu = np.concatenate([np.concatenate(p) for p in parts])
    assert len(set(allu.tolist())) == len(allu)


def test_rank_batches_balanced_and_disjoint():
    """DP sharding: every step gives all ranks the same user count, near-equal rows and Tmax; users are
    never shared between ranks; world=1 reproduces the plain bucketed order."""
    from svae_cf_amd.dist import rank_batches
    from svae_cf_amd.synth import bucketed_batches
    lens = np.clip(np.rint(np.random.default_rng(0).lognormal(4.2, 0.9, 16 * 64 * 4)), 5, 200).astype(int)
    world, B = 4, 64
    parts = [rank_batches(lens, B, np.random.default_rng(7), r, world) for r in range(world)]
    assert len({len(p) for p in parts}) == 1 and len(parts[0]) == 16
    allu = np.concatenate([np.concatenate(p) for p in parts])
    assert len(set(allu.tolist())) == len(allu) == len(lens)
    for step in range(len(parts[0])):
        rows = [int((lens[parts[r][step]] - 1).sum()) for r in range(world)]
        tmax = [int(lens[parts[r][step]].max()) for r in range(world)]
        assert all(len(parts[r][step]) == B for r in range(world))
        assert max(rows) - min(rows) <= max(tmax) and max(rows) <= 1.02 * min(rows) + 8
        assert max(tmax) - min(tmax) <= max(2, 0.1 * max(tmax))
    one = rank_batches(lens, B, np.random.default_rng(7), 0, 1)
    ref = bucketed_batches(lens, B, np.random.default_rng(7))
    assert all((a == b).all() for a, b in zip(one, ref))


@pytest.mark.parametrize("uniform", [False, True])
def test_proj_schedule_covers_every_tile_once(uniform):
    """svae_proj_schedule (host function of the C-ABI library): every (x tile, y tile) pair belongs to exactly
    one workgroup, an x tile's slabs are 0..ns_j-1, split counts are q or q+1 (uniform: all q), and the balanced
    form launches a whole number of rounds on 256 CUs when the 16-slab cap does not bind."""
    from svae_cf_amd import _lib
    for nx, ny in ((21814, 20000), (20000, 21814), (300, 97), (97, 300), (2957, 20000), (50944, 20000), (257, 63), (1, 1)):
        tab, nwg, q, rem = _lib.proj_schedule(nx, ny, 256, uniform)
        assert nwg == len(tab)
        xt, nt = (nx + 255) // 256, (ny + 63) // 64
        cov = np.zeros((xt, nt), dtype=np.int64)
        for j, b, e, s in tab:
            assert 0 <= b <= e <= nt
            cov[j, b:e] += 1
        assert (cov == 1).all(), (nx, ny)
        ns = np.bincount(tab[:, 0], minlength=xt)
        assert (ns[:rem] == q + 1).all() and (ns[rem:] == q).all() and q + (rem > 0) <= _lib.load().svae_proj_ns_max()
        for j in range(xt):
            assert sorted(tab[tab[:, 0] == j][:, 3].tolist()) == list(range(ns[j]))
        if uniform:
            assert rem == 0 and q == min(8, nt)
        elif q < min(16, nt):
            assert len(tab) % 256 == 0
        ln = tab[:, 2] - tab[:, 1]
        assert ln.max() - ln.min() <= nt // q - nt // (q + 1) + 1


@pytest.mark.parametrize("boundary", [0, 5])
def test_flat_proj_schedule_covers_every_tile_once_at_equal_cost(boundary):
    """svae_proj_schedule_flat (round 3): contiguous ranges of the x-major (x tile, y tile) space, one per workgroup:
    every tile pair belongs to exactly one segment, an x tile's slabs are 0..ns_j-1 with ns_j <= 16 and equal to the
    table's slab counts, the workgroups' segment lists partition the segments in order, every workgroup has work, and
    -- the point of it -- no workgroup's cost (tiles + `boundary` per further segment) exceeds the mean by more than a
    tile and a boundary (the balanced form's longest piece: up to 23 % above the mean)."""
    from svae_cf_amd import _lib
    nsm = _lib.load().svae_proj_ns_max()
    for nx, ny, want in ((21814, 20000, 256), (15000, 20000, 192), (6800, 20000, 256), (20000, 21814, 192),
                         (20000, 21814, 158), (300, 20000, 256), (50769, 20000, 256), (97, 300, 256), (257, 63, 7),
                         (1, 1, 256), (1000000, 11200, 256), (11200, 1000000, 256)):
        tab, w, q, nseg = _lib.proj_schedule_flat(nx, ny, want, boundary)
        xt, nt = (nx + 255) // 256, (ny + 63) // 64
        assert q == 0 and 1 <= w <= want and w <= nseg <= w + xt
        flat = tab.reshape(-1)
        seg = tab[:nseg]
        off = flat[4 * nseg:4 * nseg + w + 1]
        ns = flat[4 * nseg + 4 * ((w + 1 + 3) // 4):][:xt]
        cov = np.zeros((xt, nt), dtype=np.int64)
        for j, b, e, _ in seg:
            assert 0 <= b < e <= nt
            cov[j, b:e] += 1
        assert (cov == 1).all(), (nx, ny)
        assert ns.min() >= 1 and ns.max() <= nsm
        for j in range(xt):
            assert sorted(seg[seg[:, 0] == j][:, 3].tolist()) == list(range(ns[j]))
        assert off[0] == 0 and off[w] == nseg and (np.diff(off) >= 1).all()
        ln = seg[:, 2] - seg[:, 1]
        cost = np.array([ln[off[i]:off[i + 1]].sum() + boundary * (off[i + 1] - off[i] - 1) for i in range(w)])
        assert cost.max() <= cost.sum() / w + 1 + boundary, (nx, ny, cost.max(), cost.mean())
        # a workgroup's segments are consecutive in the flat order: next x tile, from its first y tile
        for i in range(w):
            for k in range(off[i] + 1, off[i + 1]):
                assert seg[k][0] == seg[k - 1][0] + 1 and seg[k][1] == 0 and seg[k - 1][2] == nt


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _dp_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from svae_cf_amd.dist import GradSync, allreduce_metrics
    from svae_cf_amd.layout import ParamLayout
    lay = ParamLayout(50)
    g = torch.full((lay.total,), float(rank + 1))
    sync = GradSync(g, lay.buckets())
    for i in range(len(lay.buckets())):
        sync.bucket_ready(i)
    sync.wait()
    m = allreduce_metrics([1.0 + rank, 10.0], "cpu")
    q.put((rank, float(g.min()), float(g.max()), sync.gscale, m))
    dist.destroy_process_group()


def test_grad_sync_gloo_world2():
    """Bucketed all-reduce covers the whole flat buffer; gscale turns the sum into the mean over ranks."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(60)
    for rank, lo, hi, gs, m in res:
        assert lo == hi == 3.0 and gs == 0.5 and m == [3.0, 20.0]


def _sparse_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from svae_cf_amd.dist import GradSync
    from svae_cf_amd.layout import ParamLayout, E
    I = 500
    lay = ParamLayout(I)
    s, e = lay.buckets()[2]
    out = {}
    for mode in ("dense", "sparse"):
        gen = torch.Generator().manual_seed(100 + rank)
        g = torch.zeros(lay.total)
        # a step's embedding gradient: a few touched rows (overlapping between the ranks), everything else zero
        ids = torch.randperm(I, generator=gen)[:37 + 5 * rank].to(torch.int32)
        ids[0] = 3                                             # both ranks touch row 3, rank 1 also row 0
        if rank == 1:
            ids[1] = 0
        ids = torch.unique(ids)
        table = g[s:s + I * E].view(I, E)
        table[ids.long()] = torch.randn(len(ids), E, generator=gen)
        g[:s] = float(rank + 1)                                # the other buckets: dense as before
        sync = GradSync(g, lay.buckets())
        if mode == "sparse":
            sync.enable_sparse_rows(2, I, E, capacity=64)
        sync.bucket_ready(0)
        sync.bucket_ready(1)
        if mode == "sparse":
            sync.rows_ready(ids, len(ids))
        else:
            sync.bucket_ready(2)
        sync.wait()
        out[mode] = g.clone()
    q.put((rank, torch.equal(out["dense"], out["sparse"]), float(out["sparse"][s:e].abs().sum()),
           out["sparse"].numpy().tobytes()))
    dist.destroy_process_group()


def test_sparse_row_exchange_equals_dense_allreduce_gloo_world2():
    """GradSync.enable_sparse_rows: the item-embedding gradient exchanged as (row id, row) pairs (all-gather, rebuilt
    in rank order) gives exactly the dense all-reduce's result at world size 2 -- rows touched by one rank, by both,
    row 0 (where the pad entries add their zeros) -- and the two ranks end bit-identical.  The other buckets stay
    dense.  This is the large-catalogue path (at 1M items the dense bucket is 1 GB per step)."""
    import torch.multiprocessing as mp
    from svae_cf_amd.dist import GradSync
    assert GradSync.sparse_rows_pay(1_000_000, 45_000, 8) and not GradSync.sparse_rows_pay(20_000, 24_000, 8)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_sparse_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(60)
    assert all(same for _, same, _, _ in res)
    assert res[0][2] > 0 and res[0][3] == res[1][3]


def _sharded_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from svae_cf_amd.dist import GradSync
    from svae_cf_amd.layout import ParamLayout
    lay = ParamLayout(501)
    (s0, e0) = lay.buckets()[0]
    gen = torch.Generator().manual_seed(7 + rank)
    g = torch.randn(lay.total, generator=gen)
    want = None
    if True:
        # the sum every rank must see on its shard = what the dense all-reduce gives
        ref = g.clone()
        dist.all_reduce(ref)
        want = ref
    params = torch.full((lay.total,), -1.0)
    sync = GradSync(g, lay.buckets())
    sync.enable_sharded(0)
    a, b = sync.shard_range()
    for i in range(3):
        sync.bucket_ready(i)
    sync.wait()
    shard_ok = torch.equal(g[a:b], want[a:b]) and torch.equal(g[e0:], want[e0:])
    # "optimizer step" over the own shard only, then the parameters go round
    params[a:b] = g[a:b] * 0.5
    wk = sync.gather_params(params)
    if wk is not None:
        wk.wait()
    gathered_ok = torch.equal(params[s0:e0], want[s0:e0] * 0.5) and bool((params[e0:] == -1.0).all())
    q.put((rank, (a, b), (s0, e0), shard_ok, gathered_ok))
    dist.destroy_process_group()


def test_sharded_bucket_reduce_scatter_allgather_gloo_world2():
    """GradSync.enable_sharded: the decoder.linear2 bucket (padded by ParamLayout to a multiple of 512 floats) splits
    into `world` equal shards; after the exchange a rank's shard holds the all-reduce's sum, an update of that shard
    alone followed by ``gather_params`` leaves every rank with the whole updated bucket and nothing beyond it
    touched.  (gloo has no reduce-scatter: the bucket is all-reduced there, the rank's shard of it is the same sum;
    RCCL runs the in-place reduce_scatter_tensor / all_gather_into_tensor pair.)"""
    import torch.multiprocessing as mp
    from svae_cf_amd.layout import ParamLayout
    for items in (97, 501, 20000, 26744):
        s0, e0 = ParamLayout(items).buckets()[0]
        assert s0 == 0 and e0 % 512 == 0 and e0 >= 151 * items
        assert ParamLayout(items).entries["decoder.linear1.weight"].offset == e0
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_sharded_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(60)
    (s0, e0) = res[0][2]
    assert res[0][1][0] == s0 and res[0][1][1] == res[1][1][0] and res[1][1][1] == e0      # shards tile the bucket
    assert all(r[3] and r[4] for r in res), res


# ---------------------------------------------------------------------------------------------
# epoch driver's log lines against the reference's OWN printed output
# ---------------------------------------------------------------------------------------------
def test_log_lines_byte_identical_to_reference_output(golden_dir):
    """tests/golden/ref_log_lines.json holds the lines the reference's run printed (embedded cell output nb:882-1175,
    extracted by oracle/make_golden_logs.py) and the numbers in them; the same numbers through notebook_api's
    formatters must give the same bytes: 125 running lines (format nb:1226), 25 end-of-epoch blocks (nb:1292-1301),
    the end-of-training block (nb:1376-1381)."""
    import json
    from svae_cf_amd import notebook_api as N
    with open(os.path.join(golden_dir, "ref_log_lines.json")) as f:
        g = json.load(f)
    assert len(g["batch_lines"]) == 125 and len(g["epoch_blocks"]) == 25
    for b in g["batch_lines"]:
        assert N.format_batch_log(b["epoch"], b["batch"], b["batch_limit"], b["ms_per_batch"], b["loss"]) == b["line"]
    keys = ["loss", "NDCG@10", "Rec@10", "Prec@10", "NDCG@100", "Rec@100", "Prec@100"]       # nb:597-603
    for e in g["epoch_blocks"]:
        assert [k for k, _ in e["train"]] == keys
        assert N.format_epoch_end(e["epoch"], e["seconds"], dict(e["train"]), dict(e["val"])) == e["text"]
    assert N.format_end_of_training(dict(g["end_block"]["test"])) == g["end_block"]["text"]


def test_fit_logs_through_the_pinned_formatters(monkeypatch):
    """fit() and train() emit exactly what the pinned formatters return (no second copy of the format strings)."""
    import inspect
    from svae_cf_amd import notebook_api as N
    src_fit, src_train = inspect.getsource(N.fit), inspect.getsource(N.train)
    assert "format_epoch_end(" in src_fit and "format_end_of_training(" in src_fit and "{:5.2f}" not in src_fit
    assert "format_batch_log(" in src_train and "{:5.2f}" not in src_train


def test_evaluate_returns_metric_keys_in_reference_order():
    from svae_cf_amd import notebook_api as N
    per = N._user_metrics(np.arange(100)[None, :], [[3, 500]])
    assert list(per) == ["NDCG@10", "Rec@10", "Prec@10", "NDCG@100", "Rec@100", "Prec@100"]


# ---------------------------------------------------------------------------------------------
# vectorised metric arithmetic == the reference's loop, to the last bit
# ---------------------------------------------------------------------------------------------
def test_user_metrics_float_identical_to_reference_loop():
    """notebook_api._user_metrics (all users at once, bincount accumulation) against oracle.user_metrics (the
    reference's per-user Python loop nb:633-648): exactly equal float64 values -- random catalogues, held-out lists
    from 1 to 400 items, held-out items repeated, no hit at all, every item a hit."""
    from oracle import svae_numpy as O
    from svae_cf_amd import notebook_api as N
    rs = np.random.RandomState(3)
    rec, held = [], []
    for u in range(300):
        I = int(rs.choice([120, 500, 5000]))
        rec.append(rs.permutation(I)[:100])
        n = int(rs.choice([1, 2, 5, 30, 99, 100, 101, 400]))
        h = rs.randint(0, I, n)                     # with replacement: repeated held-out items occur
        if u % 7 == 0:
            h = rec[-1][rs.permutation(100)[:min(n, 100)]]      # all hits
        if u % 11 == 0:
            h = np.full(n, I + 5)                   # no hit
        held.append([int(v) for v in h])
    per = N._user_metrics(np.stack(rec), held)
    for u in range(300):
        ref = O.user_metrics(rec[u], held[u])
        for k, v in ref.items():
            assert per[k][u] == v, (u, k, per[k][u], v)
    sums, len_map = N._metrics_from_rankings(np.stack(rec), held, [7] * 300)
    for k in sums:
        acc = 0.0
        for u in range(300):
            acc += O.user_metrics(rec[u], held[u])[k]
        assert sums[k] == acc
    assert sum(len(v) for v in len_map.values()) == 300
    with pytest.raises(ZeroDivisionError):
        N._user_metrics(np.stack(rec[:2]), [[1], []])


# ---------------------------------------------------------------------------------------------
# KL anneal of the batched path counts USERS, as the reference's schedule does
# ---------------------------------------------------------------------------------------------
def test_batched_anneal_follows_users_consumed():
    """The reference advances its KL weight once per user (batch_size 1).  train_batched's default gives step b the
    weight the reference gives the first user of that global batch; B = 1, world = 1 is the reference schedule."""
    from svae_cf_amd import notebook_api as N
    for B, world in ((1, 1), (256, 1), (256, 8)):
        before = 0
        for b in range(1, 60):
            a = N.users_anneal(before)
            assert a == N.anneal_for_batch(1 + (b - 1) * B * world)
            if B == 1:
                assert a == N.anneal_for_batch(b)
            before += B * world
    # one pass over ML-1M's 4534 users ends near the reference's last weight (0.0227), not 256x below it
    assert abs(N.users_anneal(17 * 256) - N.anneal_for_batch(17 * 256 + 1)) == 0.0
    assert N.users_anneal(17 * 256) > 0.02 and N.anneal_for_batch(18) < 1e-4
    # ML-20M-sized pass reaches the cap
    assert N.users_anneal(137000) == 0.2


def test_reference_init_draws_from_global_generator():
    """Model(hyper_params) without a seed must be reproducible under torch.manual_seed, like the reference's
    Model.__init__ (global generator), and must not reseed the caller's generator."""
    from svae_cf_amd import notebook_api as N
    hp = dict(item_embed_size=256, rnn_size=200, hidden_size=150, latent_size=64)
    torch.manual_seed(123)
    a = N.reference_init(40, hp)
    after_a = torch.rand(1)
    torch.manual_seed(123)
    b = N.reference_init(40, hp)
    after_b = torch.rand(1)
    assert all(torch.equal(a[k], b[k]) for k in a) and torch.equal(after_a, after_b)
    c = N.reference_init(40, hp)
    assert not torch.equal(a["item_embed.weight"], c["item_embed.weight"])
    s1, s2 = N.reference_init(40, hp, seed=5), N.reference_init(40, hp, seed=5)
    assert all(torch.equal(s1[k], s2[k]) for k in s1)


@pytest.mark.parametrize("I,nu,mu,sg,lmax,k,occ", [(300, 48, 2.8, 0.6, 60, 4, True), (300, 17, 2.8, 0.6, 60, 1, True),
                                                  (97, 9, 2.0, 0.5, 40, 10 ** 6, True), (2000, 64, 3.5, 0.8, 300, 4, True),
                                                  (50, 30, 3.0, 0.5, 80, 4, False), (20, 33, 3.0, 0.5, 300, 10 ** 6, True)])
def test_native_plan_builder_equals_numpy_builder(I, nu, mu, sg, lmax, k, occ):
    """svae_plan_create / svae_plan_pack (csrc/plan.hip, host code) against batch.build_plan, byte for byte: every
    table, the host-side order / lengths / row_of, the time split; next_k 1, 4 and postfix; duplicate targets inside
    a row; with and without the occurrence tables; with and without user keys."""
    from svae_cf_amd import synth
    from svae_cf_amd.batch import build_plan, build_plan_native, UserCSR
    rng = np.random.default_rng(I + nu)
    users = [np.asarray(u).copy() for u in synth.make_users(nu, I, mu, sg, lmax, seed=I)]
    users[0][1:3] = users[0][0]                       # duplicates among the next-k targets
    csr = UserCSR([u.tolist() for u in users])
    assert len(csr) == nu and csr.items.dtype == np.int32
    idx = rng.permutation(nu)[:max(2, nu - 3)]
    for keys in (idx, None):
        a = build_plan([users[i] for i in idx], k, user_keys=keys, total_items=I if occ else None)
        b = build_plan_native(csr, idx, k, user_keys=keys, total_items=I if occ else None)
        for n in ("B", "R", "Tmax", "K", "nseg", "nseg_long", "nocc_long", "split_t", "split_r"):
            assert getattr(a, n) == getattr(b, n), n
        assert np.array_equal(a.order, b.order) and np.array_equal(a.lengths, b.lengths)
        for n in ("x", "tgt", "w_nll", "w_kl", "row_off", "nact", "last_row", "perm", "seg_off", "seg_item", "row_key",
                  "seen", "seen_off", "occ_off", "occ_row", "occ_items"):
            x, y = getattr(a, n), getattr(b, n)
            if x is None or y is None:
                assert x is None and y is None, n
                continue
            assert x.dtype == y.dtype and x.shape == y.shape, n
            assert np.array_equal(x.numpy().view(np.uint8), y.numpy().view(np.uint8)), n
        for u in range(a.B):
            assert np.array_equal(a.row_of[u], b.row_of[u])


def test_native_plan_builder_errors():
    from svae_cf_amd.batch import build_plan_native, UserCSR
    with pytest.raises(ValueError):
        build_plan_native(UserCSR([[1], [1, 2]]), [0, 1], 4)          # a user with one item (nb:803 divides by zero)
    with pytest.raises(IndexError):
        build_plan_native(UserCSR([[1, 2], [1, 2]]), [0, 2], 4)
    with pytest.raises(Exception):
        build_plan_native(UserCSR([[1, 2, 7], [1, 2]]), [0, 1], 4, total_items=5)   # item id beyond the catalogue
