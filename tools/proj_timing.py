"""Phase timing of the fused projection kernels (needs csrc/exp/lib_PT.so built with -DSV_PROJ_TIMING:
   make -C svae_cf_amd/csrc exp/lib_PT.so).  Prints shader cycles per 64-row tile for waves 0 and 4 of block 0."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SVAE_HIP_LIB"] = os.path.join(ROOT, "svae_cf_amd/csrc/exp/lib_%s.so" % (sys.argv[1] if len(sys.argv) > 1 else "PT"))
import svae_cf_amd as S
from svae_cf_amd import synth, _lib
I, B = 20000, 256
_, _, mu, sigma, lmax = synth.SHAPES["ml-20m"]
users = synth.make_users(4096, I, mu, sigma, lmax, seed=98765)
rng = np.random.default_rng(98765)
batches = synth.bucketed_batches([len(u) for u in users], B, rng)
eng = S.SVAEEngine(I, device="cuda", prec=S.PREC_BF16)
idx = batches[int(sys.argv[2]) if len(sys.argv) > 2 else 3]
plan = S.build_plan([users[i] for i in idx], 4, user_keys=idx, device="cuda", total_items=I)
for _ in range(3):
    eng.train_step(plan, 0.01, seed=1)
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 48)()
lib = _lib.load()
lib.svae_debug_proj.argtypes = [ctypes.c_void_p]
rc = lib.svae_debug_proj(ctypes.byref(out))
v = np.array(list(out), dtype=np.float64).reshape(3, 2, 8)
print("rows", plan.R, "rc", rc)
for mode, name in ((1, "FDH"), (2, "DW")):
    for w in range(2):
        nt = max(v[mode, w, 7], 1)
        ph = v[mode, w, :6] / nt
        print("%s wave %d tiles %d: cycles per tile [gemm1, exp+pack, dma issue, gemm2, vmcnt wait, barrier] = %s  sum %.0f"
              % (name, 4 * w, nt, np.round(ph, 0), ph.sum()))

blk = (ctypes.c_ulonglong * (3 * 4096 * 4))()
lib.svae_debug_proj_blocks.argtypes = [ctypes.c_void_p]
lib.svae_debug_proj_blocks(ctypes.byref(blk))
b = np.array(list(blk), dtype=np.float64).reshape(3, 4096, 4)
blk2 = (ctypes.c_ulonglong * (3 * 4096 * 2))()
lib.svae_debug_proj_blocks2.argtypes = [ctypes.c_void_p]
lib.svae_debug_proj_blocks2(ctypes.byref(blk2))
b2 = np.array(list(blk2), dtype=np.float64).reshape(3, 4096, 2)
for mode, name in ((1, "FDH"), (2, "DW")):
    m = b[mode]
    n = int((m[:, 3] > 0).sum())
    m = m[:n]
    e = b2[mode][:n]
    pro, epi = (m[:, 0] - e[:, 0]) / 100.0, (e[:, 1] - m[:, 1]) / 100.0
    print("%s: prologue us (entry -> loop start) median %.1f max %.1f; epilogue us (loop end -> stores landed) median %.1f max %.1f; "
          "loop us median %.1f; tiles per block median %.0f" % (name, np.median(pro), pro.max(), np.median(epi), epi.max(),
                                                                 np.median(m[:, 1] - m[:, 0]) / 100.0, np.median(m[:, 3])))
    first = e[:, 0].min()
    print("   whole kernel (first entry -> last store landed): %.1f us; sum of loops / (CUs x span) = %.2f"
          % ((e[:, 1].max() - first) / 100.0, (m[:, 1] - m[:, 0]).sum() / ((256 if mode == 1 else 256) * (e[:, 1].max() - first))))
    t0 = m[:, 0].min()
    start, end = (m[:, 0] - t0) / 100.0, (m[:, 1] - t0) / 100.0     # us (100 MHz counter)
    cyc_tile = m[:, 2] / np.maximum(m[:, 3], 1)
    print("%s: %d blocks, kernel span %.1f us; loop cycles per tile: min %.0f median %.0f max %.0f; block duration us: median %.1f max %.1f"
          % (name, n, end.max(), cyc_tile.min(), np.median(cyc_tile), cyc_tile.max(), np.median(end - start), (end - start).max()))
    order = np.argsort(start)
    print("   start times (us) deciles:", np.round(np.percentile(start, [0, 10, 25, 50, 75, 90, 100]), 1),
          " end deciles:", np.round(np.percentile(end, [0, 10, 25, 50, 75, 90, 100]), 1))
    clk = m[:, 2] / np.maximum((m[:, 1] - m[:, 0]) * 10.0, 1)          # cycles per ns
    print("   in-kernel clock GHz: median %.2f" % np.median(clk))

wv = (ctypes.c_ulonglong * (3 * 8 * 8 * 8))()
lib.svae_debug_proj_waves.argtypes = [ctypes.c_void_p]
lib.svae_debug_proj_waves(ctypes.byref(wv))
w = np.array(list(wv), dtype=np.float64).reshape(3, 8, 8, 8)
for mode, name in ((1, "FDH"), (2, "DW")):
    m = w[mode]
    if m.max() == 0:
        continue
    t0 = m[:, 0, 0].min()
    print("%s block 0, tiles 8..15, per wave: [A0, B0, A1, B1, end-of-tile sync] cycles (median over tiles); arrival skew at the barrier" % name)
    m = m[:, :, [0, 1, 2, 3, 4, 7]]              # (stamps 5, 6 belonged to the s_barrier form)
    ph = np.diff(m, axis=2)                      # [wave][tile][5]
    for wv_ in range(8):
        print("   wave %d: %s   tile total %.0f" % (wv_, np.round(np.median(ph[wv_], axis=0), 0), np.median(m[wv_, 1:, 0] - m[wv_, :-1, 0])))
    arr = m[:, :, 4]                             # arrival at the barrier [wave][tile]
    print("   barrier arrival spread per tile (max - min over waves):", np.round(arr.max(axis=0) - arr.min(axis=0), 0))
    print("   arrival order example (tile 10, cycles after first):", np.round(arr[:, 2] - arr[:, 2].min(), 0))
