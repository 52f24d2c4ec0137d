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
        tab, q, rem = _lib.proj_schedule(nx, ny, 256, uniform)
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
