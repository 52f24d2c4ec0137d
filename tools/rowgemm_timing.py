"""Kernel-only timing of svae_rowgemm (HIP events) for a few row counts; SVAE_HIP_LIB selects an experimental build."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svae_cf_amd import _lib
from svae_cf_amd.layout import E, G
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
wih = (torch.randn(G, E) * 0.1).cuda()
bias = torch.randn(G).cuda()
out = []
for R in (2957, 21814, 50944):
    for which, (K, N) in enumerate(((E, G), (G, E))):
        x = torch.randn(R, K).cuda()
        y = torch.empty(R, N, device="cuda")
        frags = torch.zeros(lib.svae_rowgemm_frag_elems(which), dtype=torch.bfloat16, device="cuda")
        _lib.call("svae_rowgemm_prep", wih.data_ptr(), which, frags.data_ptr(), st)
        for _ in range(3):
            _lib.call("svae_rowgemm", x.data_ptr(), frags.data_ptr(), None, y.data_ptr(), R, which, st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            _lib.call("svae_rowgemm", x.data_ptr(), frags.data_ptr(), None, y.data_ptr(), R, which, st)
        e1.record()
        e1.synchronize()
        out.append("R=%d which=%d %.1f us" % (R, which, e0.elapsed_time(e1) * 1e3 / 20))
print(os.environ.get("SVAE_HIP_LIB", "default").split("/")[-1], " | ".join(out))
