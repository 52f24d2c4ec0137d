"""User-sharded data parallelism (NEW in this build: the reference is single-process).

One process per GPU.  Users (independent sequences) are sharded: every rank builds the same seeded
global batch order; a global batch (B*world users of similar length) is dealt round-robin to the ranks
(synth.bucketed_batches(world=...)), so all ranks do the same amount of work at every step.  Parameters and Adam state are replicated; each
step does ONE exchange: a sum all-reduce of the flat gradient buffer, issued per bucket in
backward-readiness order (decoder.linear2 first, item_embed last) so RCCL works over xGMI while the
rest of backward still runs; Adam then applies gscale = 1/world (mean over the union of users).
"""
import torch
import torch.distributed as dist


class GradSync:
    """Bucketed all-reduce over a flat gradient buffer.

    Backend nccl (= RCCL, the production path): asynchronous; the collective is ordered behind the stream that
    is current when ``bucket_ready`` is called and runs on RCCL's own stream beside the rest of backward;
    ``wait`` orders the caller's stream behind all of them (no host block).
    Any other backend (gloo: the CPU tests and the one-GPU rehearsal, where ranks share a card): the bucket is
    staged through host memory synchronously.  gloo's own device-tensor all-reduce is not used: on this ROCm
    build it never completed (both ranks blocked in Work.wait(); seen with tests/test_gpu_dp.py).
    ``force`` keeps the collectives at world size 1 (rehearsal of the RCCL call path on one GPU)."""

    def __init__(self, flat_grads, buckets, group=None, force=False, native=False):
        self.flat = flat_grads
        self.buckets = list(buckets)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.active = dist.is_initialized() and (self.world > 1 or force)
        self.device_collectives = self.active and dist.get_backend(group) == "nccl"
        # native (opt-in, round 3): the collectives are issued through the library's own RCCL wrappers
        # (svae_allreduce_sum_f32 / svae_reduce_scatter_sum_f32 / svae_allgather_f32, csrc/comm.hip) IN ORDER on the
        # stream that is current when a bucket is announced -- the stream that produced it -- with one communicator per
        # such stream, instead of torch.distributed's collectives on its own stream with an event hop on either side
        # (43 us per step at world size 1, profiles/r02n_dp_rehearsal_rccl_world1.json.log).  `wait` then has nothing to
        # do: the consumers follow on the same stream.  torch.distributed is only used to hand rank 0's communicator id
        # to the other ranks.  NOT the default: it has only ever run at world size 1 (one GPU per box here).
        self.native = bool(native) and flat_grads.is_cuda and (self.device_collectives or (force and self.world == 1))
        if self.native:
            self.active = True
        self._comms = {}
        self.pending = []
        self.sparse = None        # (bucket, rows, width, capacity, buffers): see enable_sparse_rows
        self.sharded = None       # (bucket, shard length, rank): see enable_sharded
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0

    def enable_sharded(self, bucket):
        """Exchange bucket `bucket` with a reduce-scatter instead of an all-reduce: rank r ends up with the summed
        gradient of the r-th of `world` equal shards only, runs the optimizer over that shard (its moments live on
        that rank alone) and the updated parameters go round with an all-gather (``gather_params``).  The same bytes
        on the links as a ring all-reduce (which is these two phases), but the optimizer's 28 bytes per parameter are
        touched once per job instead of once per rank -- decoder.linear2 is 151 I floats: 4.2 GB of optimizer
        traffic per step and rank at 1M items -- and every replica receives the SAME updated shard, so the replicas
        cannot drift.  Meant for the decoder.linear2 bucket (ParamLayout pads it to a multiple of 512 floats)."""
        s, e = self.buckets[bucket]
        if (e - s) % (64 * self.world):
            raise ValueError("bucket %d (%d floats) does not split into %d shards of whole 256-byte lines"
                             % (bucket, e - s, self.world))
        self.sharded = dict(bucket=bucket, shard=(e - s) // self.world)

    def shard_range(self):
        """[start, end) of this rank's shard of the sharded bucket, in elements of the flat buffers."""
        s, _ = self.buckets[self.sharded["bucket"]]
        n = self.sharded["shard"]
        return s + self.rank * n, s + (self.rank + 1) * n

    def gather_params(self, flat_params):
        """After the optimizer step over this rank's shard: every rank's shard of the parameters to every rank.
        Returns a work object with ``wait()`` (orders the current stream behind the all-gather), or None."""
        if not self.active:
            return None
        s, e = self.buckets[self.sharded["bucket"]]
        a, b = self.shard_range()
        if self.native:
            from . import _lib
            st = torch.cuda.current_stream().cuda_stream
            _lib.call("svae_allgather_f32", self._comm(st), flat_params[s:e].data_ptr(), b - a, self.rank, st)
            return None
        if self.device_collectives or not flat_params.is_cuda:
            # in place: a rank's shard is its part of the gathered buffer
            src = flat_params[a:b] if self.device_collectives else flat_params[a:b].clone()
            return dist.all_gather_into_tensor(flat_params[s:e], src, group=self.group, async_op=True)
        host = flat_params[a:b].to("cpu")                   # (gloo with device tensors: staged through the host)
        full = torch.empty(e - s, dtype=host.dtype)
        dist.all_gather_into_tensor(full, host, group=self.group)
        flat_params[s:e].copy_(full)
        return None

    def enable_sparse_rows(self, bucket, n_rows, width, capacity):
        """Exchange bucket `bucket` -- the gradient of an [n_rows, width] table of which a step touches at most
        `capacity` rows per rank: the item embedding, whose gradient has one non-zero row per distinct input item of
        the batch -- as (row id, row) pairs with an all-gather instead of a dense all-reduce.  At 1M items the dense
        bucket is 1 GB per step, the pairs of a 256-user batch 11-45 MB per rank.  Every rank then rebuilds the
        summed rows in rank order (own rows zeroed first, then rank 0, 1, ...: the same additions in the same order
        everywhere, so the replicas stay bit-identical), the untouched rows are zero on every rank as they were.
        Worth it when world * capacity rows are fewer than the table has (`sparse_rows_pay`).  The engine announces
        the touched rows of a step with ``rows_ready`` in place of ``bucket_ready(bucket)``."""
        s, e = self.buckets[bucket]
        if e - s < n_rows * width:
            raise ValueError("bucket %d holds %d floats, the table %d" % (bucket, e - s, n_rows * width))
        dev = self.flat.device
        self.sparse = dict(bucket=bucket, rows=n_rows, width=width, cap=int(capacity),
                           ids=torch.full((capacity,), -1, dtype=torch.int64, device=dev),
                           vals=torch.zeros(capacity, width, dtype=self.flat.dtype, device=dev),
                           all_ids=torch.full((self.world * capacity,), -1, dtype=torch.int64, device=dev),
                           all_vals=torch.zeros(self.world * capacity, width, dtype=self.flat.dtype, device=dev),
                           spread=torch.arange(capacity, dtype=torch.int64, device=dev) % n_rows)

    def _comm(self, stream_ptr):
        """The communicator of the stream `stream_ptr` (native mode): created on first use -- collectively, every rank
        reaches this point in the same order -- from a 128-byte id that rank 0 makes and torch.distributed hands round."""
        c = self._comms.get(stream_ptr)
        if c is None:
            import ctypes
            from . import _lib
            ident = ctypes.create_string_buffer(128)
            if self.rank == 0:
                _lib.call("svae_rccl_unique_id", ident)
            if self.world > 1:
                box = [bytes(ident.raw)]
                dist.broadcast_object_list(box, src=0, group=self.group)
                ident = ctypes.create_string_buffer(box[0], 128)
            comm = ctypes.c_void_p()
            _lib.call("svae_rccl_comm_create", ctypes.byref(comm), self.world, self.rank, ident)
            c = self._comms[stream_ptr] = comm.value
        return c

    def close(self):
        """Destroy the native communicators (call with the device idle)."""
        from . import _lib
        for c in self._comms.values():
            _lib.call("svae_rccl_comm_destroy", c)
        self._comms = {}

    @staticmethod
    def sparse_rows_pay(n_rows, capacity, world):
        """True when the (id, row) exchange moves fewer bytes than the dense all-reduce of the table's gradient
        (ring all-reduce: 2 (w-1)/w of the table per rank; all-gather: (w-1) capacities)."""
        return world > 1 and (world - 1) * capacity * 1.05 < 2.0 * (world - 1) / world * n_rows

    def rows_ready(self, ids, n):
        """The sparse bucket's gradient is complete on the current stream; `ids[:n]` (device int tensor, distinct) are
        the table rows this rank's step touched."""
        sp = self.sparse
        if not self.active:
            return
        if n > sp["cap"]:
            raise ValueError("%d touched rows exceed the sparse exchange's capacity %d" % (n, sp["cap"]))
        s, _ = self.buckets[sp["bucket"]]
        table = self.flat[s:s + sp["rows"] * sp["width"]].view(sp["rows"], sp["width"])
        own = ids[:n].to(torch.int64)
        if self.native:
            # in place: this rank's pairs are its part of the gathered buffers (ids as pairs of 4-byte words)
            from . import _lib
            cap, r, st = sp["cap"], self.rank, torch.cuda.current_stream().cuda_stream
            mine_i, mine_v = sp["all_ids"][r * cap:(r + 1) * cap], sp["all_vals"][r * cap:(r + 1) * cap]
            mine_i.fill_(-1)
            mine_i[:n] = own
            mine_v[:n] = table.index_select(0, own)
            comm = self._comm(st)
            _lib.call("svae_allgather_f32", comm, sp["all_ids"].data_ptr(), 2 * cap, r, st)
            _lib.call("svae_allgather_f32", comm, sp["all_vals"].data_ptr(), cap * sp["width"], r, st)
            self._sparse_rebuild(table, own)
            return
        sp["ids"].fill_(-1)
        sp["ids"][:n] = own
        sp["vals"][:n] = table.index_select(0, own)
        if self.device_collectives or not self.flat.is_cuda:
            w1 = dist.all_gather_into_tensor(sp["all_ids"], sp["ids"], group=self.group, async_op=True)
            w2 = dist.all_gather_into_tensor(sp["all_vals"], sp["vals"], group=self.group, async_op=True)
            self.pending.append((sp["bucket"], _SparseWork(self, (w1, w2), table, own)))
            return
        hi, hv = sp["ids"].to("cpu"), sp["vals"].to("cpu")     # (gloo with device tensors: staged through the host)
        ai = torch.empty(self.world * sp["cap"], dtype=hi.dtype)
        av = torch.empty(self.world * sp["cap"], sp["width"], dtype=hv.dtype)
        dist.all_gather_into_tensor(ai, hi, group=self.group)
        dist.all_gather_into_tensor(av, hv, group=self.group)
        sp["all_ids"].copy_(ai)
        sp["all_vals"].copy_(av)
        self._sparse_rebuild(table, own)

    def _sparse_rebuild(self, table, own):
        sp = self.sparse
        if table.is_cuda:
            # one zeroing launch and one adding launch per rank, in rank order (svae_sparse_rows_rebuild)
            from . import _lib
            _lib.call("svae_sparse_rows_rebuild", table.data_ptr(), sp["rows"], sp["width"], sp["all_ids"].data_ptr(),
                      sp["all_vals"].data_ptr(), self.world, sp["cap"], own.data_ptr(), int(own.numel()),
                      torch.cuda.current_stream().cuda_stream)
            own.record_stream(torch.cuda.current_stream())
            return
        table.index_fill_(0, own, 0.0)
        cap = sp["cap"]
        for r in range(self.world):                       # fixed order: the replicas stay bit-identical
            ids = sp["all_ids"][r * cap:(r + 1) * cap]
            vals = sp["all_vals"][r * cap:(r + 1) * cap]
            live = ids >= 0
            # pad entries add zeros (a rank's count is not known on the host, and is not asked for: no sync) -- spread
            # over the table: thousands of atomic adds to ONE row took a millisecond
            # (selected, not multiplied: rows beyond a rank's count keep earlier steps' contents, and 0 * Inf is NaN)
            table.index_add_(0, torch.where(live, ids, sp["spread"]),
                             torch.where(live.unsqueeze(1), vals, torch.zeros((), dtype=vals.dtype, device=vals.device)))

    def bucket_ready(self, i):
        """Call when every gradient of bucket i has been enqueued on the current stream."""
        if not self.active:
            return
        s, e = self.buckets[i]
        if self.native:
            from . import _lib
            st = torch.cuda.current_stream().cuda_stream
            if self.sharded is not None and self.sharded["bucket"] == i:
                _lib.call("svae_reduce_scatter_sum_f32", self._comm(st), self.flat[s:e].data_ptr(), self.sharded["shard"],
                          self.rank, st)
            else:
                _lib.call("svae_allreduce_sum_f32", self._comm(st), self.flat[s:e].data_ptr(), e - s, st)
            return
        if self.sharded is not None and self.sharded["bucket"] == i and self.device_collectives:
            a, b = self.shard_range()          # in place: the rank's shard of the sum lands in its part of the bucket
            self.pending.append((i, dist.reduce_scatter_tensor(self.flat[a:b], self.flat[s:e], op=dist.ReduceOp.SUM,
                                                               group=self.group, async_op=True)))
            return
        # (backends without a reduce-scatter -- gloo, the CPU tests and the one-GPU rehearsal -- all-reduce the sharded
        #  bucket: the rank's shard of the result is the same sum)
        if self.device_collectives or not self.flat.is_cuda:
            self.pending.append((i, dist.all_reduce(self.flat[s:e], op=dist.ReduceOp.SUM, group=self.group,
                                                    async_op=True)))
            return
        host = self.flat[s:e].to("cpu")            # ordered behind the current stream, blocks until copied
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
        self.flat[s:e].copy_(host)                 # on the current stream as well

    def wait(self, which=None):
        """Order the current stream behind the all-reduces of the buckets in `which` (None: all of them)."""
        keep = []
        for i, w in self.pending:
            if which is None or i in which:
                w.wait()
            else:
                keep.append((i, w))
        self.pending = keep

    @property
    def gscale(self):
        return 1.0 / self.world


class _SparseWork:
    """The two all-gathers of a sparse bucket and the rebuild behind them, with the ``wait`` of a collective's work."""

    def __init__(self, sync, works, table, own):
        self.sync, self.works, self.table, self.own = sync, works, table, own

    def wait(self):
        for w in self.works:
            w.wait()
        self.sync._sparse_rebuild(self.table, self.own)


def shard_batches(batches, rank, world):
    """Rank r takes batches r, r+world, ...; the tail that does not fill all ranks is dropped so
    every rank runs the same number of steps (collectives stay matched).  Training uses the balanced form
    ``rank_batches`` instead: with shuffled length-bucketed batches this one hands the ranks of a step
    batches of very different lengths, and every step then lasts as long as its longest batch."""
    n = len(batches) // world * world
    return [batches[i] for i in range(rank, n, world)]


def rank_batches(lengths, batch_size, rng, rank, world, window=16):
    """This rank's batch of every step, balanced over ranks (see synth.bucketed_batches)."""
    from .synth import bucketed_batches
    steps = bucketed_batches(lengths, batch_size, rng, window=window, world=world)
    if world == 1:
        return steps
    return [s[rank] for s in steps]


def allreduce_metrics(values, device, group=None):
    """Sum a small list of python floats over ranks (eval: loss sum, 6 metric sums, user count)."""
    t = torch.tensor(values, dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t.tolist()
