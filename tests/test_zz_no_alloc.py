"""svae_masked_topk_static (the allocation-free variant: scratch in a static device array of the library) against
svae_masked_topk (caller-owned scratch) through the C-ABI: bit-identical indices and values, on calls in a row that
shrink -- the array keeps the last call's contents, and no call may depend on them.  The cases are the ones in which
the kernel really uses the scratch: a catalogue beyond the LDS bit-set (masked copy of the row in scratch), constant
rows and K above the LDS sort (arg-max rounds over the materialised row); plus the usual case, which does not.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _both(scores, seen_lists, K):
    from svae_cf_amd import _lib
    st = torch.cuda.current_stream().cuda_stream
    n, I = scores.shape
    ld = (I + 3) // 4 * 4
    buf = torch.zeros(n, ld)
    buf[:, :I] = scores
    buf = buf.cuda()
    off = np.zeros(n + 1, dtype=np.int32)
    off[1:] = np.cumsum([len(x) for x in seen_lists])
    seen = torch.from_numpy(np.concatenate([np.asarray(x, dtype=np.int32) for x in seen_lists]
                                           + [np.zeros(1, np.int32)])).cuda()
    off_t = torch.from_numpy(off).cuda()
    out = []
    for static in (False, True):
        idx = torch.full((n * K,), -7, dtype=torch.int32, device="cuda")
        val = torch.full((n * K,), float("nan"), device="cuda")
        if static:
            _lib.call("svae_masked_topk_static", buf.data_ptr(), ld, seen.data_ptr(), off_t.data_ptr(), n, I, K,
                      idx.data_ptr(), val.data_ptr(), st)
        else:
            work = torch.empty_like(buf)
            _lib.call("svae_masked_topk", buf.data_ptr(), ld, seen.data_ptr(), off_t.data_ptr(), n, I, K,
                      idx.data_ptr(), val.data_ptr(), work.data_ptr(), st)
        torch.cuda.synchronize()
        out.append((idx.cpu(), val.cpu()))
    (i0, v0), (i1, v1) = out
    assert torch.equal(i0, i1), "indices differ between svae_masked_topk and svae_masked_topk_static"
    assert torch.equal(v0.view(torch.int32), v1.view(torch.int32)), "values differ"
    # and both equal the stable descending sort of the masked row (the reference's torch.sort, nb:632)
    ref = scores.clone()
    for u, lst in enumerate(seen_lists):
        for it in lst:
            ref[u, it] = ref[u, it] - (ref[u, it] * 1.0).abs() * 1e8
    rv, ri = torch.sort(ref, dim=1, descending=True, stable=True)
    assert torch.equal(i1.view(n, K).long(), ri[:, :K])
    assert torch.equal(v1.view(n, K), rv[:, :K])


def test_static_scratch_topk_matches_on_consecutive_shrinking_calls():
    g = torch.Generator().manual_seed(11)
    # 1. catalogue beyond the LDS bit-set: every row is copied (masked) into the scratch
    big = torch.randn(6, 150000, generator=g)
    _both(big, [[1, 7, 149999], [], list(range(0, 30000, 3)), [0], [5], []], 100)
    # 2. smaller, right behind it: constant rows -> arg-max rounds over the row materialised in the (dirty) scratch
    _both(torch.full((2, 5000), 0.25), [[], [3]], 100)
    # 3. the usual case (scratch untouched), 4. K above the LDS sort (rounds), smaller again
    _both(torch.randn(4, 5000, generator=g), [[1, 2], [], [4999], [0]], 100)
    _both(torch.randn(2, 3000, generator=g), [[], [17]], 2000)
    # 5. large again after small ones
    _both(torch.randn(3, 140000, generator=g), [[], [139999], [1, 2, 3]], 10)


def test_static_scratch_topk_refuses_oversize():
    from svae_cf_amd import _lib
    st = torch.cuda.current_stream().cuda_stream
    d = torch.zeros(8, device="cuda")
    di = torch.zeros(8, dtype=torch.int32, device="cuda")
    with pytest.raises(_lib.SvaeHipError, match="exceeds the static scratch"):
        # 9 users x 1 000 000 floats > 8 Mi floats; refused before anything is launched (pointers are not touched)
        _lib.call("svae_masked_topk_static", d.data_ptr(), 1000000, di.data_ptr(), di.data_ptr(), 9, 1000000, 100,
                  di.data_ptr(), d.data_ptr(), st)
