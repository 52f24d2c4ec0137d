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

hprev16 = hprev.to(torch.bfloat16)
s_hh = torch.empty(48 * (G * HP + G), device=dev)
s_ih = torch.empty(48 * (G * E + G), device=dev)
a_hh, a_ih = s_hh.data_ptr(), s_ih.data_ptr()
print("pair, 128 tiles, 32 splits : %.1f us" % run("svae_wgrad_pair_bf16dy", dg.data_ptr(), R, G, 896, 32, hprev.data_ptr(), HP, HP, a_hh, a_hh + 4 * G * HP, G * HP + G, 448, 224,
      emb16.data_ptr(), 1, E, E, a_ih, a_ih + 4 * G * E, G * E + G, 672, 0, st))
for nsb in (24, 32, 40, 48):
    print("pair, 256 tiles, %d splits : %.1f us" % (nsb, run("svae_wgrad_pair_big", dg.data_ptr(), R, G, 896, nsb, hprev16.data_ptr(), HP, HP, a_hh, a_hh + 4 * G * HP, G * HP + G, 448, 224,
          emb16.data_ptr(), E, E, a_ih, a_ih + 4 * G * E, G * E + G, 672, 0, st)))
print("fp32 -> bf16 copy of hprev : %.1f us" % run("svae_f32_to_bf16", hprev.data_ptr(), hprev16.data_ptr(), R * HP, st))
print("reduce pair (40 splits)    : %.1f us" % run("svae_reduce_slabs_pair", a_hh, s_hh.data_ptr(), G * HP + G, G * HP + G, a_ih, s_ih.data_ptr(), G * E + G, G * E + G, 40, st))
print("pair, 128 tiles, all bf16, hand-counted waits : %.1f us" % run("svae_wgrad_pair_bf16", dg.data_ptr(), R, G, 896, 32, hprev16.data_ptr(), HP, HP, a_hh, a_hh + 4 * G * HP, G * HP + G, 448, 224,
      emb16.data_ptr(), E, E, a_ih, a_ih + 4 * G * E, G * E + G, 672, 0, st))
