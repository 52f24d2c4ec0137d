"""Parameter layout in HBM: ONE flat fp32 buffer, padded, in backward-readiness order.

Why one buffer: Adam is a single fused launch over (p, g, m, v) and the data-parallel
gradient all-reduce is a handful of contiguous buckets.  Order = the order in which
backward finishes the gradients (decoder.linear2 first ... item_embed last), so a bucket
can be reduced on a side stream while the rest of backward still runs.

Padding (K dims to multiples of 32, see csrc/common.h): pad entries are zero and stay zero
under Adam with L2 decay because their gradient is exactly zero.

The reference's state_dict names / shapes (nb:718-729) are kept at the
boundary: ``pack`` / ``unpack`` convert between that dict and the flat buffer.
"""
import collections

import torch

E, H, HP, D, DP, L = 256, 200, 224, 150, 160, 64
G = 3 * HP

Entry = collections.namedtuple("Entry", "name offset groups rows cols prows pcols size")


def _round_up(x, m):
    return (x + m - 1) // m * m


class ParamLayout:
    def __init__(self, total_items):
        self.I = int(total_items)
        self.IP = _round_up(self.I, 4)          # item rows / logits columns padded to a float4
        I, IP = self.I, self.IP
        spec = [
            # name, groups, rows, cols, padded rows, padded cols   (tensor viewed as [groups, rows, cols])
            ("decoder.linear2.weight", 1, I, D, IP, DP),
            ("decoder.linear2.bias", 1, 1, I, 1, IP),
            ("decoder.linear1.weight", 1, D, L, DP, L),
            ("decoder.linear1.bias", 1, 1, D, 1, DP),
            ("linear1.weight", 1, 2 * L, D, 2 * L, DP),
            ("linear1.bias", 1, 1, 2 * L, 1, 2 * L),
            ("encoder.linear1.weight", 1, D, H, DP, HP),
            ("encoder.linear1.bias", 1, 1, D, 1, DP),
            ("gru.weight_hh_l0", 3, H, H, HP, HP),
            ("gru.bias_hh_l0", 3, 1, H, 1, HP),
            ("gru.weight_ih_l0", 3, H, E, HP, E),
            ("gru.bias_ih_l0", 3, 1, H, 1, HP),
            ("item_embed.weight", 1, I, E, I, E),
        ]
        self.entries = collections.OrderedDict()
        off = 0
        for name, g, r, c, pr, pc in spec:
            size = g * pr * pc
            if name.endswith(".bias") or name.startswith("gru.bias"):
                # a bias sits right behind its weight: the weight-gradient GEMM writes [gW | gb] in one slab
                prev = list(self.entries.values())[-1]
                assert off == prev.offset + prev.size, name
            self.entries[name] = Entry(name, off, g, r, c, pr, pc, size)
            off = _round_up(off + size, 64)     # every tensor starts 256-byte aligned
            if name == "decoder.linear2.bias":
                # bucket 0 (decoder.linear2.*) ends on a multiple of 512 floats: it splits into 1, 2, 4 or 8 equal
                # shards of whole 256-byte lines (dist.GradSync.enable_sharded: reduce-scatter, sharded optimizer
                # step, all-gather).  The pad floats are zero and stay zero like every other pad.
                off = _round_up(off, 512)
        self.total = off
        # reference shapes at the boundary
        self.ref_shapes = {
            "decoder.linear2.weight": (I, D), "decoder.linear2.bias": (I,),
            "decoder.linear1.weight": (D, L), "decoder.linear1.bias": (D,),
            "linear1.weight": (2 * L, D), "linear1.bias": (2 * L,),
            "encoder.linear1.weight": (D, H), "encoder.linear1.bias": (D,),
            "gru.weight_hh_l0": (3 * H, H), "gru.bias_hh_l0": (3 * H,),
            "gru.weight_ih_l0": (3 * H, E), "gru.bias_ih_l0": (3 * H,),
            "item_embed.weight": (I, E),
        }
        self.n_params = sum(int(torch.Size(s).numel()) for s in self.ref_shapes.values())

    def view(self, flat, name):
        """Padded [groups, prows, pcols] view of tensor ``name`` inside a flat buffer."""
        e = self.entries[name]
        return flat[e.offset:e.offset + e.size].view(e.groups, e.prows, e.pcols)

    def offset(self, name):
        return self.entries[name].offset

    def pack(self, state_dict, flat):
        """state_dict (reference names/shapes, any device) -> flat padded buffer (in place)."""
        flat.zero_()
        for name, e in self.entries.items():
            t = state_dict[name]
            if tuple(t.shape) != self.ref_shapes[name]:
                raise ValueError("%s: expected shape %s, got %s" % (name, self.ref_shapes[name], tuple(t.shape)))
            src = t.detach().to(device=flat.device, dtype=flat.dtype).reshape(e.groups, e.rows, e.cols)
            self.view(flat, name)[:, :e.rows, :e.cols].copy_(src)
        return flat

    def unpack(self, flat):
        """flat padded buffer -> OrderedDict of contiguous tensors with the reference names/shapes,
        in the reference's registration order."""
        order = ["encoder.linear1.weight", "encoder.linear1.bias", "decoder.linear1.weight",
                 "decoder.linear1.bias", "decoder.linear2.weight", "decoder.linear2.bias",
                 "item_embed.weight", "gru.weight_ih_l0", "gru.weight_hh_l0", "gru.bias_ih_l0",
                 "gru.bias_hh_l0", "linear1.weight", "linear1.bias"]
        out = collections.OrderedDict()
        for name in order:
            e = self.entries[name]
            v = self.view(flat, name)[:, :e.rows, :e.cols]
            out[name] = v.reshape(self.ref_shapes[name]).clone()
        return out

    def buckets(self, n_buckets=3):
        """Contiguous [start, end) slices of the flat buffer for bucketed gradient all-reduce.
        Bucket 0 = decoder.linear2.* (151 I floats; ready when the weight-gradient sweep of the projection has
        finished, on the side stream), bucket 1 = the small dense layers and the GRU (0.33 M floats; the last
        gradients of the step), bucket 2 = item_embed (256 I floats; ready before the GRU weight gradients, see
        engine.backward).  Three collectives per step: every one costs host time and a stream hand-over."""
        names = list(self.entries.keys())
        groups = [names[0:2], names[2:12], names[12:13]]
        out = []
        for gi, gnames in enumerate(groups):
            s = self.entries[gnames[0]].offset
            last = self.entries[gnames[-1]]
            end = _round_up(last.offset + last.size, 64)
            if gi == 0:
                end = self.entries[groups[1][0]].offset        # incl. the pad up to a multiple of 512 floats
            out.append((s, end))
        return out
