"""Phase timing of the bf16 GRU forward (needs csrc/exp/lib_T.so built with -DSV_GRU_TIMING)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SVAE_HIP_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "svae_cf_amd/csrc/exp/lib_T.so")
import svae_cf_amd as S
from svae_cf_amd import synth, _lib
I, B = 20000, 256
_, _, mu, sigma, lmax = synth.SHAPES["ml-20m"]
users = synth.make_users(4096, I, mu, sigma, lmax, seed=98765)
rng = np.random.default_rng(98765)
batches = synth.bucketed_batches([len(u) for u in users], B, rng)
eng = S.SVAEEngine(I, device="cuda", prec=S.PREC_BF16)
for bi in (3, 0, 2):
    idx = batches[bi]
    plan = S.build_plan([users[i] for i in idx], 4, user_keys=idx, device="cuda", total_items=I)
    for _ in range(3):
        eng.train_step(plan, 0.01, seed=1)
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 8)()
    lib = _lib.load()
    lib.svae_debug_gru.argtypes = [ctypes.c_void_p]
    rc = lib.svae_debug_gru(ctypes.byref(out))
    v = np.array(list(out)[:8], dtype=np.float64)
    print("Tmax", plan.Tmax, "rc", rc, "per-step ticks by phase [top, addr+A, mfma, gi-wait, math, store-issue, drain, barrier]:", np.round(v / plan.Tmax, 1), "sum", round(v.sum() / plan.Tmax, 1))
