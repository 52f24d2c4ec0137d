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

    def __init__(self, flat_grads, buckets, group=None, force=False):
        self.flat = flat_grads
        self.buckets = list(buckets)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.active = dist.is_initialized() and (self.world > 1 or force)
        self.device_collectives = self.active and dist.get_backend(group) == "nccl"
        self.pending = []

    def bucket_ready(self, i):
        """Call when every gradient of bucket i has been enqueued on the current stream."""
        if not self.active:
            return
        s, e = self.buckets[i]
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
