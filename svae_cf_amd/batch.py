"""Batch plans: how a set of user sequences is laid out for the GPU step.

The reference feeds ONE user per step as ``x [1,T]`` plus a dense one-hot target
``y_s [1,T,I]`` built by T ``scatter_`` calls (DataReader.iter nb:485-526, next_k branch
nb:509-513).  Here a batch of B users is packed TIME-MAJOR:

  * users sorted by length (descending); ``nact[t]`` users are still running at timestep t;
  * row r = row_off[t] + u holds timestep t of (sorted) user u  -> the GRU kernel sees the
    active users of a timestep as one contiguous block, every other kernel sees R = sum_u T_u
    independent rows;
  * targets are an int32 ``tgt [R, K]`` table (-1 = padding) instead of the dense one-hot;
  * per-row weights carry the batched loss contract (mean over users of the reference's
    single-user loss): ``w_nll[r] = 1/(B*num_ones_u)`` (nb:803, nb:808),
    ``w_kl[r] = 1/(B*T_u)`` (the ``torch.mean`` over rows of nb:797).
"""
import numpy as np
import torch

EMBED_LONG_ROWS = 16     # = svae_embed_long_rows(): longer item segments get a whole workgroup in svae_embed_bwd
OCC_LONG = 32            # = svae_proj_occ_long(): same for an item's target occurrences in svae_proj_nll_bwd_dw


class BatchPlan:
    """Device-resident description of one batch (all int32 / fp32 tensors)."""

    __slots__ = ("B", "R", "Tmax", "K", "order", "lengths", "x", "tgt", "w_nll", "w_kl", "row_off",
                 "nact", "last_row", "perm", "seg_off", "seg_item", "nseg", "row_key", "seen",
                 "seen_off", "row_of", "device", "occ_off", "occ_row", "total_items", "nseg_long", "occ_items",
                 "nocc_long", "proj_sched", "split_t", "split_r")

    def to(self, device):
        for name in ("x", "tgt", "w_nll", "w_kl", "row_off", "nact", "last_row", "perm", "seg_off",
                     "seg_item", "row_key", "seen", "seen_off", "occ_off", "occ_row", "occ_items"):
            v = getattr(self, name)
            if v is not None:
                setattr(self, name, v.to(device, non_blocking=True))
        self.device = device
        return self


def choose_time_split(row_off, us_per_step=2.4, us_launch=15.0, us_chain=40.0, us_per_row=0.011, cu_share=0.75):
    """Where to cut the forward in two time ranges (engine.forward, fused mode): the GRU recurrence is a chain of
    Tmax latency-bound steps on a quarter of the CUs, the decoder side of a row (MLP chain + projection sweep) is
    throughput-bound and only needs the row's h_t -- so the rows of timesteps < t1 go through the decoder side
    while the recurrence runs steps >= t1.  Rows are time-major, so they are the first row_off[t1] rows; the cut
    is rounded down to a multiple of 64 rows (the rows in between ride with the second range).
    Cost model (measured on MI355X, ML-20M shape): a recurrence launch = 15 us + 2.4 us per step; the decoder side
    of r rows = 40 us + 11 us per 1000 rows, 1/0.75 of that while the recurrence holds a quarter of the CUs.
    Returns (t1, r1), or (0, 0) when no cut beats the unsplit forward by 5 %."""
    row_off = np.asarray(row_off, dtype=np.int64)
    Tmax, R = len(row_off) - 1, int(row_off[-1])
    best, arg = us_launch + us_per_step * Tmax + us_chain + us_per_row * R, (0, 0)
    base = best
    for t1 in range(1, Tmax):
        r1 = int(row_off[t1]) // 64 * 64
        if r1 < 2048 or R - r1 < 2048:
            continue
        g1 = us_launch + us_per_step * t1
        g2 = us_launch + us_per_step * (Tmax - t1)
        c1 = us_chain + us_per_row * r1 / cu_share
        c2 = us_chain + us_per_row * (R - r1)
        cost = g1 + max(g2, c1) + c2
        if cost < best:
            best, arg = cost, (t1, r1)
    return arg if best < 0.95 * base else (0, 0)


def build_plan(users, next_k, user_keys=None, device=None, pin=False, total_items=None):
    """users: list of 1-D int arrays (chronological item ids, len >= 2).

    next_k: number of following items that are targets of a timestep (reference
    ``hyper_params['next_k']``; ``predict_next`` is 1, ``postfix`` is any value >= max length).
    user_keys: int64 id per user for the counter-based noise stream (defaults to position).
    total_items: catalogue size; when given, the plan also carries the target occurrences grouped by
    item (``occ_off [I+1]``, ``occ_row``) that the fused projection backward needs for dL/dW3.
    """
    B = len(users)
    if B == 0:
        raise ValueError("empty batch")
    lens = np.array([len(u) - 1 for u in users], dtype=np.int64)
    if lens.min() < 1:
        raise ValueError("every user needs at least 2 items (the reference crashes on T=0, nb:803)")
    order = np.argsort(-lens, kind="stable")
    T = lens[order]
    Tmax = int(T[0])
    K = int(min(next_k, Tmax))
    nact = (T[None, :] > np.arange(Tmax)[:, None]).sum(axis=1).astype(np.int32)
    row_off = np.zeros(Tmax + 1, dtype=np.int32)
    np.cumsum(nact, out=row_off[1:])
    R = int(row_off[-1])
    if user_keys is None:
        user_keys = np.arange(B, dtype=np.int64)
    user_keys = np.asarray(user_keys, dtype=np.int64)

    x = np.empty(R, dtype=np.int32)
    tgt = np.full((R, K), -1, dtype=np.int32)
    w_nll = np.empty(R, dtype=np.float32)
    w_kl = np.empty(R, dtype=np.float32)
    row_key = np.empty(R, dtype=np.int64)
    last_row = np.empty(B, dtype=np.int32)
    seen_off = np.zeros(B + 1, dtype=np.int32)
    np.cumsum(T, out=seen_off[1:])
    seen = np.empty(int(seen_off[-1]), dtype=np.int32)
    row_of = []
    for u in range(B):
        items = np.asarray(users[order[u]], dtype=np.int64)
        Tu = int(T[u])
        rows = row_off[:Tu] + u
        row_of.append(rows)
        x[rows] = items[:-1]
        seen[seen_off[u]:seen_off[u + 1]] = items[:-1]
        for j in range(K):
            n = Tu - j
            if n > 0:
                tgt[rows[:n], j] = items[1 + j:1 + j + n]
        row_key[rows] = (user_keys[order[u]] << 20) + np.arange(Tu)
        last_row[u] = rows[-1]
        w_kl[rows] = 1.0 / (B * Tu)
    # a duplicate target inside a row counts once (scatter_ of 1.0, nb:510-513)
    for j in range(1, K):
        dup = (tgt[:, j:j + 1] == tgt[:, :j]).any(axis=1) & (tgt[:, j] >= 0)
        tgt[dup, j] = -1
    n_r = (tgt >= 0).sum(axis=1)
    for u in range(B):
        num_ones = float(n_r[row_of[u][0]])          # first timestep's target count (nb:803)
        w_nll[row_of[u]] = 1.0 / (B * num_ones)
    # rows sorted by item for the deterministic embedding scatter-add
    # (segments ordered by length, longest first: the kernel gives the long ones a workgroup each)
    perm = np.argsort(x, kind="stable")
    xs = x[perm]
    starts = np.flatnonzero(np.concatenate(([True], xs[1:] != xs[:-1])))
    seg_len = np.diff(np.concatenate((starts, [R])))
    by_len = np.argsort(-seg_len, kind="stable")
    seg_len = seg_len[by_len]
    seg_item = xs[starts[by_len]].astype(np.int32)
    seg_off = np.zeros(len(by_len) + 1, dtype=np.int32)
    np.cumsum(seg_len, out=seg_off[1:])
    perm = perm[np.repeat(starts[by_len] - seg_off[:-1], seg_len) + np.arange(R)].astype(np.int32)
    nseg_long = int((seg_len > EMBED_LONG_ROWS).sum())

    p = BatchPlan()
    p.proj_sched = {}          # engine-side cache: work division of the projection sweeps over this plan
    p.B, p.R, p.Tmax, p.K = B, R, Tmax, K
    p.order = order
    p.lengths = T
    p.split_t, p.split_r = choose_time_split(row_off)
    p.nseg = int(len(seg_item))
    p.nseg_long = nseg_long
    p.row_of = row_of
    p.device = None

    def t(a):
        tt = torch.from_numpy(a)
        return tt.pin_memory() if pin else tt
    p.x, p.tgt, p.w_nll, p.w_kl = t(x), t(tgt), t(w_nll), t(w_kl)
    p.row_off, p.nact, p.last_row = t(row_off), t(nact), t(last_row)
    p.perm, p.seg_off, p.seg_item = t(perm), t(seg_off), t(seg_item)
    p.row_key, p.seen, p.seen_off = t(row_key), t(seen), t(seen_off)
    p.total_items = total_items
    p.occ_off = p.occ_row = p.occ_items = None
    p.nocc_long = 0
    if total_items is not None:
        rr, kk = np.nonzero(tgt >= 0)
        it = tgt[rr, kk]
        o = np.argsort(it, kind="stable")
        occ_off = np.zeros(total_items + 1, dtype=np.int32)
        np.cumsum(np.bincount(it, minlength=total_items), out=occ_off[1:])
        p.occ_off, p.occ_row = t(occ_off), t(rr[o].astype(np.int32))
        cnt = np.diff(occ_off)
        occ_items = np.argsort(-cnt, kind="stable").astype(np.int32)      # items by occurrence count, busiest first
        p.occ_items, p.nocc_long = t(occ_items), int((cnt > OCC_LONG).sum())
    if device is not None:
        p.to(device)
    return p


def pack_rows(plan, per_user, width, dtype=np.float32):
    """Scatter per-user [T_u, width] arrays (ORIGINAL user order) into the plan's row order."""
    out = np.zeros((plan.R, width), dtype=dtype)
    for u in range(plan.B):
        out[plan.row_of[u]] = np.asarray(per_user[plan.order[u]], dtype=dtype)
    return out


def unpack_rows(plan, rows_array):
    """Inverse of pack_rows: list (original user order) of [T_u, width] arrays."""
    out = [None] * plan.B
    for u in range(plan.B):
        out[plan.order[u]] = rows_array[plan.row_of[u]]
    return out
