#!/usr/bin/env python3
"""Are the launches of one stream held up behind a long DISPATCH on another (two hardware queues on one dispatcher
pipe)?  A = 8192 one-wave workgroups of 10 us with 64 KB of LDS each (2 per CU: ~160 us of dispatching, chip mostly
free) on stream i; B = one workgroup of 10 us on stream j, released at the same moment.  Prints B's completion time
after the release for every ordered pair: ~20 us = independent, ~A's length = held up."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from svae_cf_amd import _lib

order = sys.argv[1] if len(sys.argv) > 1 else "normal"
torch.zeros(1, device="cuda")
streams = [("null", torch.cuda.default_stream())]
if order == "hi_first":
    streams.append(("hi", torch.cuda.Stream(priority=-1)))
for i in range(6):
    streams.append(("s%d" % i, torch.cuda.Stream()))
if order != "hi_first":
    streams.append(("hi", torch.cuda.Stream(priority=-1)))
for _, s in streams:                          # first use creates the queue, in this order
    _lib.call("svae_spin", 10, s.cuda_stream)
torch.cuda.synchronize()


def held(sa, sb):
    e0, eb, ea = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    torch.cuda.synchronize()
    e0.record(sa)
    sb.wait_event(e0)
    _lib.call("svae_spin_grid", 10, 8192, 65536, sa.cuda_stream)
    _lib.call("svae_spin", 10, sb.cuda_stream)
    eb.record(sb)
    ea.record(sa)
    torch.cuda.synchronize()
    return e0.elapsed_time(eb) * 1e3, e0.elapsed_time(ea) * 1e3


print("B's completion (us) after the release; rows = stream of A (long dispatch), columns = stream of B")
print("%6s" % "" + "".join("%8s" % n for n, _ in streams) + "   A alone")
for na, sa in streams:
    line = "%6s" % na
    ta = 0
    for nb, sb in streams:
        if sa is sb:
            line += "%8s" % "-"
            continue
        held(sa, sb)
        tb, ta = held(sa, sb)
        line += "%8.0f" % tb
    print(line + "   %6.0f" % ta)
