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
import time

import numpy as np
import torch

EMBED_LONG_ROWS = 16     # = svae_embed_long_rows(): longer item segments get a whole workgroup in svae_embed_bwd
OCC_LONG = 32            # = svae_proj_occ_long(): same for an item's target occurrences in svae_proj_nll_bwd_dw


_TORCH_DT = {np.dtype(np.int32): torch.int32, np.dtype(np.int64): torch.int64, np.dtype(np.float32): torch.float32}


class BatchPlan:
    """Description of one batch (int32 / fp32 / int64 tensors, all views of ONE packed buffer: host until ``to``)."""

    __slots__ = ("B", "R", "Tmax", "K", "order", "lengths", "x", "tgt", "w_nll", "w_kl", "row_off",
                 "nact", "last_row", "perm", "seg_off", "seg_item", "nseg", "row_key", "seen",
                 "seen_off", "_row_of", "device", "occ_off", "occ_row", "total_items", "nseg_long", "occ_items",
                 "nocc_long", "proj_sched", "split_t", "split_r", "_host", "_offs", "_buf", "_row_off_host", "ready")

    def _bind(self, buf):
        """Point every tensor attribute at its part of the packed buffer `buf` (host or device)."""
        for name, (off, shape, dt) in self._offs.items():
            n = int(np.prod(shape)) * dt.itemsize
            setattr(self, name, buf[off:off + n].view(_TORCH_DT[dt]).view(*shape))
        self._buf = buf
        self.ready = None

    def to(self, device, stream=None):
        """Upload (one copy).  stream: a torch stream to copy on (pinned host buffer -> asynchronous); an event
        recorded behind the copy is kept in ``ready`` for the consumer's stream to wait on."""
        if stream is not None:
            with torch.cuda.stream(stream):
                buf = self._host.to(device, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(stream)
        else:
            buf, ev = self._host.to(device, non_blocking=True), None
        self._bind(buf)
        self.ready = ev
        self.device = device
        return self

    @property
    def row_of(self):
        """Per sorted user: the rows of its timesteps (pack_rows / unpack_rows; built on first use)."""
        if self._row_of is None:
            ro = self._row_off_host
            self._row_of = [ro[:int(t)] + u for u, t in enumerate(self.lengths)]
        return self._row_of


def choose_time_split(row_off, total_items=0, us_per_step=2.4, us_launch=15.0, us_chain=40.0, us_per_row=None, cu_share=0.75):
    """Where to cut the forward in two time ranges (engine.forward, fused mode): the GRU recurrence is a chain of
    Tmax latency-bound steps on a quarter of the CUs, the decoder side of a row (MLP chain + projection sweep) is
    throughput-bound and only needs the row's h_t -- so the rows of timesteps < t1 go through the decoder side
    while the recurrence runs steps >= t1.  Rows are time-major, so they are the first row_off[t1] rows; the cut
    is rounded down to a multiple of 64 rows (the rows in between ride with the second range).
    Cost model (measured on MI355X, ML-20M shape): a recurrence launch = 15 us + 2.4 us per step; the decoder side
    of r rows = 40 us + 11 us per 1000 rows (at 20 000 items; by catalogue size, below), 1/0.75 of that while the
    recurrence holds a quarter of the CUs.
    (Round 2, faster sweeps: the constants re-checked by moving each 20-30 % either way -- every variant 0.5-2 % slower;
    again with the second range's MLP chain behind its recurrence on the side stream: +-25 / +50 us on the second
    recurrence's side of the model, 0.3-1 % slower.)
    Returns (t1, r1), or (0, 0) when no cut beats the unsplit forward by 5 %."""
    if us_per_row is None:
        # the decoder side's cost per row grows with the catalogue (the sweep's share of it): 11 us per 1000 rows at 20 000
        # items (rounds 1-2), 2.5 at the ML-1M shape's 3 483 (round 3: 0.011 -> 0.7306 ms per step, 0.006 -> 0.7236,
        # 0.003 -> 0.7208, 0.002 -> 0.7188, 0.001 and below -> no cut, 0.7395); linear in between and beyond
        us_per_row = 0.0007 + 5.15e-7 * float(total_items) if total_items and total_items > 0 else 0.011
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


def _dedupe_targets(tgt):
    """A duplicate target inside a row counts once (scatter_ of 1.0, nb:510-513): every occurrence but the first
    becomes -1, in place.  Few columns: pairwise column compares; many (postfix rows): one stable sort per row."""
    R, K = tgt.shape
    if K <= 8:
        for j in range(1, K):
            dup = (tgt[:, j:j + 1] == tgt[:, :j]).any(axis=1) & (tgt[:, j] >= 0)
            tgt[dup, j] = -1
        return
    o = np.argsort(tgt, axis=1, kind="stable")                 # equal items: in column order
    srt = np.take_along_axis(tgt, o, axis=1)
    dup = np.zeros((R, K), dtype=bool)
    dup[:, 1:] = (srt[:, 1:] == srt[:, :-1]) & (srt[:, 1:] >= 0)
    rr, cc = np.nonzero(dup)
    tgt[rr, o[rr, cc]] = -1


def build_plan(users, next_k, user_keys=None, device=None, pin=False, total_items=None, stream=None):
    """users: list of 1-D int arrays (chronological item ids, len >= 2).

    next_k: number of following items that are targets of a timestep (reference
    ``hyper_params['next_k']``; ``predict_next`` is 1, ``postfix`` is any value >= max length).
    user_keys: int64 id per user for the counter-based noise stream (defaults to position).
    total_items: catalogue size; when given, the plan also carries the target occurrences grouped by
    item (``occ_off [I+1]``, ``occ_row``) that the fused projection backward needs for dL/dW3.
    Everything is built with whole-array numpy operations (no per-user loop: 1-2 ms for a 256-user ML-20M-shaped
    batch, where the per-user form took 44 ms) and goes to the device as ONE packed buffer.
    """
    B = len(users)
    if B == 0:
        raise ValueError("empty batch")
    nitems = np.fromiter((len(u) for u in users), dtype=np.int64, count=B)
    lens = nitems - 1
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

    # all items of the sorted users in one array; (uu, tt) = (sorted user, timestep) of every row, user-major
    flat = np.concatenate([np.asarray(users[i], dtype=np.int64).ravel() for i in order])
    ioff = np.zeros(B + 1, dtype=np.int64)
    np.cumsum(T + 1, out=ioff[1:])
    seen_off = np.zeros(B + 1, dtype=np.int32)
    np.cumsum(T, out=seen_off[1:])
    uu = np.repeat(np.arange(B, dtype=np.int64), T)
    tt = np.arange(R, dtype=np.int64) - np.repeat(seen_off[:-1].astype(np.int64), T)
    rows = row_off[tt].astype(np.int64) + uu                    # row of (user, timestep)
    pos = ioff[uu] + tt                                         # where that timestep's input item sits in `flat`
    # everything is made in user-major order (contiguous) and brought into row order by ONE gather each:
    # inv[r] = position of row r in user-major order
    inv = np.empty(R, dtype=np.int64)
    inv[rows] = np.arange(R, dtype=np.int64)
    xu = flat[pos]
    seen = xu.astype(np.int32)                                  # per user: its input items (items[:-1]), CSR by seen_off
    x = seen[inv]
    Tu = T[uu]
    tgt_u = np.empty((R, K), dtype=np.int32)
    last = len(flat) - 1
    for j in range(K):
        tgt_u[:, j] = np.where(tt + j < Tu, flat[np.minimum(pos + (1 + j), last)], -1)   # item t+1+j, if it exists
    _dedupe_targets(tgt_u)
    tgt = tgt_u[inv]
    row_key = ((user_keys[order][uu] << 20) + tt)[inv]
    w_kl = (1.0 / (B * Tu)).astype(np.float32)[inv]
    last_row = (row_off[T - 1] + np.arange(B)).astype(np.int32)
    # first timestep's target count (nb:803): user-major position seen_off[u]
    n0 = (tgt_u[seen_off[:-1]] >= 0).sum(axis=1).astype(np.float64)
    w_nll = (1.0 / (B * n0[uu])).astype(np.float32)[inv]
    # rows sorted by item for the deterministic embedding scatter-add
    # (segments ordered by length, longest first: the kernel gives the long ones a workgroup each).
    # Stable orders come from plain sorts of unique composite keys (value * n + index): several times faster than
    # numpy's stable argsort.
    perm = np.sort(x.astype(np.int64) * R + np.arange(R, dtype=np.int64)) % R
    xs = x[perm]
    starts = np.flatnonzero(np.concatenate(([True], xs[1:] != xs[:-1])))
    seg_len = np.diff(np.concatenate((starts, [R])))
    ns_ = len(seg_len)
    by_len = np.sort((int(seg_len.max()) - seg_len) * ns_ + np.arange(ns_, dtype=np.int64)) % ns_
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
    p.split_t, p.split_r = choose_time_split(row_off, total_items or 0)
    p.nseg = int(len(seg_item))
    p.nseg_long = nseg_long
    p._row_off_host = row_off
    p._row_of = None
    p.device = None
    p._buf = None
    arrays = dict(x=x, tgt=tgt, w_nll=w_nll, w_kl=w_kl, row_off=row_off, nact=nact, last_row=last_row, perm=perm,
                  seg_off=seg_off, seg_item=seg_item, row_key=row_key, seen=seen, seen_off=seen_off)
    p.total_items = total_items
    p.occ_off = p.occ_row = p.occ_items = None
    p.nocc_long = 0
    if total_items is not None:
        # target occurrences grouped by item, rows ascending inside an item (ties: target column ascending)
        tf = tgt.ravel().astype(np.int64)
        live = np.flatnonzero(tf >= 0)
        nk = R * K
        key = np.sort(tf[live] * nk + live)
        occ_row = ((key % nk) // K).astype(np.int32)
        occ_off = np.zeros(total_items + 1, dtype=np.int32)
        np.cumsum(np.bincount(key // nk, minlength=total_items), out=occ_off[1:])
        cnt = np.diff(occ_off).astype(np.int64)
        # items by occurrence count, busiest first (ties: item ascending)
        occ_items = (np.sort((int(cnt.max()) - cnt) * total_items + np.arange(total_items, dtype=np.int64))
                     % total_items).astype(np.int32)
        arrays.update(occ_off=occ_off, occ_row=occ_row, occ_items=occ_items)
        p.nocc_long = int((cnt > OCC_LONG).sum())
    # one packed host buffer (256-byte aligned parts): ONE host-to-device copy per plan instead of sixteen
    offs, total = {}, 0
    for name, a in arrays.items():
        offs[name] = total
        total += (a.nbytes + 255) // 256 * 256
    host = torch.empty(max(total, 256), dtype=torch.uint8, pin_memory=bool(pin))
    hv = host.numpy()
    for name, a in arrays.items():
        hv[offs[name]:offs[name] + a.nbytes] = np.ascontiguousarray(a).view(np.uint8).ravel()
    p._host, p._offs = host, {n: (offs[n], arrays[n].shape, arrays[n].dtype) for n in arrays}
    p._bind(host)
    if device is not None:
        p.to(device, stream=stream)
    return p


class UserCSR:
    """Every user's item list in ONE int32 array plus int64 offsets: what the native plan builder reads.  Made once
    per dataset (the reference's DataReader holds a Python list of per-user lists, nb:439-483; walking those lists is
    interpreter work that would otherwise be repeated for every batch of every epoch)."""

    __slots__ = ("items", "off", "n")

    def __init__(self, users):
        import itertools
        n = len(users)
        lens = np.fromiter((len(u) for u in users), dtype=np.int64, count=n)
        self.off = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(lens, out=self.off[1:])
        if n and isinstance(users[0], np.ndarray):
            items = np.concatenate([np.asarray(u).ravel() for u in users])
        else:
            items = np.fromiter(itertools.chain.from_iterable(users), dtype=np.int64, count=int(self.off[-1]))
        if len(items) and (items.min() < 0 or items.max() >= 2 ** 31):
            raise ValueError("item ids must fit int32")
        self.items = np.ascontiguousarray(items, dtype=np.int32)
        self.n = n

    def __len__(self):
        return self.n


_NATIVE_ARRAYS = (("x", np.int32), ("tgt", np.int32), ("w_nll", np.float32), ("w_kl", np.float32),
                  ("row_off", np.int32), ("nact", np.int32), ("last_row", np.int32), ("perm", np.int32),
                  ("seg_off", np.int32), ("seg_item", np.int32), ("row_key", np.int64), ("seen", np.int32),
                  ("seen_off", np.int32), ("occ_off", np.int32), ("occ_row", np.int32), ("occ_items", np.int32))


def build_plan_native(csr, idx, next_k, user_keys=None, device=None, pin=False, total_items=None, stream=None,
                      host_alloc=None, host_release=None, bind=True):
    """build_plan through the library's host-side builder (svae_plan_create / svae_plan_pack, csrc/plan.hip): same
    plan, byte for byte (tests/test_host_logic.py), for the users ``idx`` of a UserCSR -- one C call that holds no
    interpreter lock.  host_alloc(nbytes) -> uint8 host tensor to pack into (a pinned-buffer pool); host_release(
    tensor, event) is called once the upload has been issued (event None: not uploaded)."""
    import ctypes
    from . import _lib
    create, pack, destroy = _lib.plan_fns()
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    B = len(idx)
    if B == 0:
        raise ValueError("empty batch")
    if idx.min() < 0 or idx.max() >= len(csr):
        raise IndexError("user index out of range")
    keys = None if user_keys is None else np.ascontiguousarray(user_keys, dtype=np.int64)
    info = _lib.PlanInfo()
    h = create(csr.items.ctypes.data, csr.off.ctypes.data, idx.ctypes.data, None if keys is None else keys.ctypes.data,
               B, int(min(next_k, 2 ** 31 - 1)), int(total_items or 0), ctypes.byref(info))
    if not h:
        msg = _lib.last_error()
        raise (ValueError if "at least 2 items" in msg else _lib.SvaeHipError)(msg)
    try:
        n = int(info.total_bytes)
        host = host_alloc(n) if host_alloc is not None else torch.empty(n, dtype=torch.uint8, pin_memory=bool(pin))
        order = np.empty(B, dtype=np.int64)
        lengths = np.empty(B, dtype=np.int64)
        row_off = np.empty(info.Tmax + 1, dtype=np.int32)
        rc = pack(h, host.data_ptr(), host.numel(), order.ctypes.data, lengths.ctypes.data, row_off.ctypes.data)
        if rc != 0:
            raise _lib.SvaeHipError(_lib.last_error())
    finally:
        destroy(h)
    R, K, I = info.R, info.K, info.total_items
    shapes = dict(x=(R,), tgt=(R, K), w_nll=(R,), w_kl=(R,), row_off=(info.Tmax + 1,), nact=(info.Tmax,),
                  last_row=(B,), perm=(R,), seg_off=(info.nseg + 1,), seg_item=(info.nseg,), row_key=(R,), seen=(R,),
                  seen_off=(B + 1,), occ_off=(I + 1,), occ_row=(info.nocc,), occ_items=(I,))
    p = BatchPlan()
    p.proj_sched = {}
    p.B, p.R, p.Tmax, p.K = B, R, info.Tmax, K
    p.order, p.lengths = order, lengths
    p.split_t, p.split_r = info.split_t, info.split_r
    p.nseg, p.nseg_long, p.nocc_long = info.nseg, info.nseg_long, info.nocc_long
    p._row_off_host, p._row_of, p.device, p._buf = row_off, None, None, None
    p.total_items = total_items
    p.occ_off = p.occ_row = p.occ_items = None
    p._offs = {name: (int(info.off[i]), shapes[name], np.dtype(dt)) for i, (name, dt) in enumerate(_NATIVE_ARRAYS)
               if info.off[i] >= 0}
    p._host = host[:n]
    if device is None:
        if bind:                  # (bind=False: the caller uploads it next and only then needs the tensor views)
            p._bind(p._host)
        return p
    p.to(device, stream=stream)
    p._host = None
    if host_release is not None:
        host_release(host, p.ready)
    return p


class _PinnedPool:
    """Pinned host buffers for plan uploads, reused: allocating pinned memory costs about as much as building a plan.
    A buffer goes back with the event recorded behind its upload and is handed out again once that has completed."""

    def __init__(self):
        import threading
        self.lock, self.free = threading.Lock(), []

    def take(self, nbytes):
        with self.lock:
            for i, (t, ev) in enumerate(self.free):
                if t.numel() >= nbytes and (ev is None or ev.query()):
                    del self.free[i]
                    return t
        return torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8, pin_memory=True)

    def give(self, t, ev):
        with self.lock:
            self.free.append((t, ev))


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


_COPY_STREAMS = {}


def _copy_stream(device):
    key = device.index if device.index is not None else torch.cuda.current_device()
    st = _COPY_STREAMS.get(key)
    if st is None:
        st = _COPY_STREAMS[key] = torch.cuda.Stream(device=device)
    return st


CONSUMER_UPLOADS = True      # PlanPrefetcher's default (tools flip it for A/B runs)


class PlanPrefetcher:
    """Builds the plans of an epoch AHEAD of the GPU: worker threads run the library's plan builder over a UserCSR
    (one C call of about 0.45 ms for a 256-user ML-20M-shaped batch, made without the interpreter lock; a plain list
    of per-user arrays goes through the numpy ``build_plan``, 4 ms) and pack each plan into a pooled pinned host buffer;
    the copy goes out on the process's copy stream -- issued by the consumer's thread a few plans ahead
    (``consumer_uploads``, below), or by the builders themselves -- and the consumer gets the plans in order, each with an event
    (``plan.ready``) that ``SVAEEngine.forward`` makes its stream wait for.  With the reference's data model -- a
    Python list of per-user item lists, nb:439-483 -- the per-user work of DataReader.iter (nb:485-526: x, the
    next-k targets) is what this replaces; it has to run somewhere, and here it runs beside the step before.

        for plan in PlanPrefetcher(batches, users, k, device, total_items, prepare=engine.prepare): ...

    batches: list of index arrays into `users`; prepare: optional callable(plan) run in the worker with the copy
    stream current (the engine's per-plan tables); depth: plans in flight ahead of the consumer.
    Measured on MI355X, ML-20M shape, 256 batches of 256 users through train_batched: 1.03 ms per step with one or two
    workers against 0.95 ms with every plan resident beforehand (0.92x); three and more workers were SLOWER (1.20-1.24
    ms: the loop is then never blocked on plans, the step itself stretches), hence the default of two."""

    def __init__(self, batches, users, next_k, device, total_items=None, prepare=None, workers=2, depth=8,
                 consumer_uploads=None, ahead=2):
        """prepare: callable(plan), or a pair (host_half(plan) -> made, upload_half(plan, made)) as
        ``(engine.prepare_host, engine.prepare_upload)``.
        consumer_uploads (UserCSR input on a GPU): the builder threads make NO device call -- they pack the plan into
        a pinned buffer (and run the host half of `prepare`); the CONSUMER's thread issues the copy on the copy stream,
        `ahead` plans ahead of the one it hands out.  With uploads issued from the builder threads the consumer's
        kernel launches and the builders' copies met inside the HIP runtime: a scoring call's host time rose from
        0.11 to 0.25-0.35 ms beside two or three builders (tools/probe/evaluate_workers.py)."""
        from concurrent.futures import ThreadPoolExecutor
        self.batches, self.users, self.k = list(batches), users, next_k
        self.pool_pinned = _PinnedPool()
        self.wait_s = 0.0
        self.device, self.total_items = torch.device(device), total_items
        if isinstance(prepare, tuple):
            self.prepare_host, self.prepare_upload = prepare
            self.prepare = lambda p_: self.prepare_upload(p_, self.prepare_host(p_))
        else:
            self.prepare, self.prepare_host, self.prepare_upload = prepare, None, None
        self.depth = max(1, int(depth))
        self.ahead = max(1, int(ahead))
        self.pool = ThreadPoolExecutor(max_workers=max(1, int(workers)))
        # ONE copy stream per device and process (a prefetcher is made per epoch; streams are a finite resource)
        self.copy_stream = _copy_stream(self.device) if self.device.type == "cuda" else None
        if consumer_uploads is None:
            consumer_uploads = CONSUMER_UPLOADS
        self.consumer_uploads = bool(consumer_uploads and self.copy_stream is not None and isinstance(users, UserCSR))

    def _build_host(self, idx):
        """Builder thread, consumer_uploads: the plan in a pinned host buffer, the engine's tables on the host."""
        taken = []

        def alloc(n):
            taken.append(self.pool_pinned.take(n))
            return taken[-1]
        plan = build_plan_native(self.users, idx, self.k, user_keys=idx, total_items=self.total_items, host_alloc=alloc,
                                 bind=False)
        made = self.prepare_host(plan) if self.prepare_host is not None else None
        return plan, made, taken[0]

    def _upload(self, plan, made, host):
        """Consumer's thread: the copy (and the tables') on the copy stream, the event the consumer's stream waits for."""
        with torch.cuda.stream(self.copy_stream):
            plan.to(self.device, stream=self.copy_stream)
            plan._host = None
            if self.prepare_upload is not None:
                self.prepare_upload(plan, made)
            elif self.prepare is not None:
                self.prepare(plan)
            if self.prepare is not None:
                plan.ready = torch.cuda.Event()
                plan.ready.record(self.copy_stream)
        self.pool_pinned.give(host, plan.ready)
        return plan

    def _make(self, idx):
        users = self.users
        native = isinstance(users, UserCSR)
        if self.copy_stream is None:
            if native:
                return build_plan_native(users, idx, self.k, user_keys=idx, total_items=self.total_items)
            return build_plan([np.asarray(users[i]) for i in idx], self.k, user_keys=idx, total_items=self.total_items)
        if self.device.index is not None:
            torch.cuda.set_device(self.device)
        if native:
            plan = build_plan_native(users, idx, self.k, user_keys=idx, total_items=self.total_items,
                                     device=self.device, stream=self.copy_stream,
                                     host_alloc=self.pool_pinned.take, host_release=self.pool_pinned.give)
        else:
            plan = build_plan([np.asarray(users[i]) for i in idx], self.k, user_keys=idx, pin=True,
                              total_items=self.total_items)
        with torch.cuda.stream(self.copy_stream):
            if not native:
                plan.to(self.device, stream=self.copy_stream)
            if self.prepare is not None:
                self.prepare(plan)
                plan.ready = torch.cuda.Event()
                plan.ready.record(self.copy_stream)
        return plan

    def __iter__(self):
        from collections import deque
        q, it = deque(), iter(self.batches)
        make = self._build_host if self.consumer_uploads else self._make
        up = deque()                                   # consumer_uploads: plans whose copy has been issued
        try:
            for idx in it:
                q.append(self.pool.submit(make, idx))
                if len(q) >= self.depth:
                    break
            while q or up:
                if self.consumer_uploads:
                    # issue the copies of the plans that are built, up to `ahead` beyond the one handed out; wait only
                    # when there is nothing to hand out
                    while q and len(up) <= self.ahead and (not up or q[0].done()):
                        t0 = time.perf_counter()
                        built = q.popleft().result()
                        if not up:
                            self.wait_s += time.perf_counter() - t0
                        up.append(self._upload(*built))
                        nxt = next(it, None)
                        if nxt is not None:
                            q.append(self.pool.submit(make, nxt))
                    yield up.popleft()
                    continue
                t0 = time.perf_counter()
                plan = q.popleft().result()
                self.wait_s += time.perf_counter() - t0          # the consumer's time blocked on the builders
                nxt = next(it, None)
                if nxt is not None:
                    q.append(self.pool.submit(make, nxt))
                yield plan
        finally:
            for f in q:
                f.cancel()
            self.pool.shutdown(wait=True)

    def __len__(self):
        return len(self.batches)

