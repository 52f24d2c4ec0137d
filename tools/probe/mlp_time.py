#!/usr/bin/env python3
"""The fused MLP kernels alone (svae_mlp_fwd_prep / svae_mlp_bwd) on random data at several row counts: is a
launch a latency chain (time flat in R up to one workgroup per CU) or work-bound?   python tools/probe/mlp_time.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from svae_cf_amd import _lib
dev = "cuda"
lib = _lib.load()
nfrag = lib.svae_mlp_frag_elems()
f32 = dict(device=dev, dtype=torch.float32)
w1, wl, w2 = torch.randn(160, 224, **f32) * .05, torch.randn(128, 160, **f32) * .05, torch.randn(160, 64, **f32) * .05
frags = torch.zeros(3 * nfrag, device=dev, dtype=torch.int16)
st = torch.cuda.current_stream().cuda_stream
_lib.call("svae_mlp_prep", w1.data_ptr(), wl.data_ptr(), w2.data_ptr(), frags.data_ptr(), st)
b1, bl, b2 = torch.zeros(160, **f32), torch.zeros(128, **f32), torch.zeros(160, **f32)
for R in (128, 1024, 4096, 8192, 16384, 21760, 32768, 65536):
    Rp = (R + 63) // 64 * 64
    hs, eps = torch.randn(R, 224, **f32), torch.randn(R, 64, **f32)
    enc, tmp, z, d1 = (torch.empty(R + 1, n, **f32) for n in (160, 128, 64, 160))
    kl = torch.empty(R + 1, **f32)
    h2b, h2s = (torch.empty(Rp, 160, device=dev, dtype=torch.int16) for _ in range(2))
    dd1, dtmp, denc, dhs = torch.randn(R, 160, **f32), torch.empty(R, 128, **f32), torch.empty(R, 160, **f32), torch.empty(R, 224, **f32)
    wkl = torch.ones(R, **f32)
    def fwd():
        _lib.call("svae_mlp_fwd_prep", hs.data_ptr(), eps.data_ptr(), frags.data_ptr(), b1.data_ptr(), bl.data_ptr(),
                  b2.data_ptr(), enc.data_ptr(), tmp.data_ptr(), z.data_ptr(), d1.data_ptr(), kl.data_ptr(), R, Rp, 0, Rp,
                  h2b.data_ptr(), h2s.data_ptr(), None, st)
    def bwd():
        _lib.call("svae_mlp_bwd", dd1.data_ptr(), tmp.data_ptr(), eps.data_ptr(), enc.data_ptr(), wkl.data_ptr(),
                  frags.data_ptr(), 0.2, dtmp.data_ptr(), denc.data_ptr(), dhs.data_ptr(), R, st)
    out = []
    for fn in (fwd, bwd):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record()
        e1.synchronize()
        out.append(e0.elapsed_time(e1) / 50 * 1e3)
    if hasattr(lib, "svae_debug_mlp"):
        import ctypes
        fwd(); torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 8)()
        lib.svae_debug_mlp(buf)
        d = [buf[i] - buf[0] for i in range(8)]
        print("   fwd stamps (cycles from kernel entry, workgroup 0 wave 0): loads issued %d, staging issued+waited %d, barrier %d, "
              "layer1 %d, layer2+z %d, layer3+prep issued %d, stores drained %d" % tuple(d[1:]))
    print("R %6d  workgroups %4d  fwd+prep %6.1f us  bwd %6.1f us" % (R, (R + 127) // 128, out[0], out[1]), flush=True)
