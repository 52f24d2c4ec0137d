#!/usr/bin/env python3
"""The GRU's two weight-gradient GEMMs alone on random data (R rows): microseconds per launch.
    [SVAE_HIP_LIB=...] python tools/probe/wgrad_time.py [R]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from svae_cf_amd import _lib
R = int(sys.argv[1]) if len(sys.argv) > 1 else 21814
G, E, HP = 672, 256, 224
dev = "cuda"
dg = (torch.randn(R + 1, 896, device=dev) * 0.1).to(torch.bfloat16)
emb16 = torch.randn(R + 1, E, device=dev).to(torch.bfloat16)
emb = torch.randn(R + 1, E, device=dev)
hprev = torch.randn(R + 1, HP, device=dev)
ns = 32
slabs = torch.empty(ns * (G * E + G), device=dev)
st = torch.cuda.current_stream().cuda_stream
def run(fn, *a):
    for _ in range(5):
        _lib.call(fn, *a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        _lib.call(fn, *a)
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / 50 * 1e3
sp = slabs.data_ptr()
print("dW_ih  bf16 dY, bf16 X : %.1f us" % run("svae_wgrad_cols_bf16dy_bf16x", dg.data_ptr(), emb16.data_ptr(), sp, sp + 4 * G * E, R, G, E, 896, E, ns, G * E + G, 672, 0, st))
print("dW_ih  bf16 dY, fp32 X : %.1f us" % run("svae_wgrad_cols_bf16dy", dg.data_ptr(), emb.data_ptr(), sp, sp + 4 * G * E, R, G, E, 896, E, ns, G * E + G, 672, 0, st))
print("dW_hh  bf16 dY, fp32 X : %.1f us" % run("svae_wgrad_cols_bf16dy", dg.data_ptr(), hprev.data_ptr(), sp, sp + 4 * G * HP, R, G, HP, 896, HP, ns, G * HP + G, 448, 224, st))
