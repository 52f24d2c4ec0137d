"""The weight-gradient GEMM entry points of the C-ABI on their own (csrc/wgrad.hip): gW[N][K] = dY^T X and gb = column
sums of dY, the autograd of y = x W^T + b inside loss.backward() (nb:1205), in every operand form the step uses --
fp32 operands rounded on the fly, the split (fp32-class) instantiation, bf16 dY (the GRU's merged gate-gradient
array, with the column remap of the recurrent weights) and bf16 dY with bf16 X (swizzled LDS images read with
ds_read_b64_tr_b16).  Checked against a float64 product of the operands as the kernel sees them (bf16-rounded for the
bf16 forms): the only difference left is the fp32 accumulation order, gate 2e-6 relative to the largest entry (the
exact integer case is bit-exact and catches any misplaced row, column or k index).  Ragged row counts (1, a partial
stage, several splits with a partial last split) and widths that are not multiples of the 128-column tiles."""
import numpy as np
import pytest
import torch

from svae_cf_amd import _lib

pytestmark = pytest.mark.gpu


def _run(fn, dy, x, R, N, K, ldy, ldx, ns, remap, prec=None):
    dev = dy.device
    stride = N * K + N
    slabs = torch.full((ns * stride,), float("nan"), device=dev)
    out = torch.empty(stride, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    sp = slabs.data_ptr()
    tail = (remap[0], remap[1], st) if prec is None else (remap[0], remap[1], prec, st)
    _lib.call(fn, dy.data_ptr(), x.data_ptr(), sp, sp + 4 * N * K, R, N, K, ldy, ldx, ns, stride, *tail)
    _lib.call("svae_reduce_slabs", sp, out.data_ptr(), stride, ns, stride, st)
    torch.cuda.synchronize()
    return out[: N * K].view(N, K).cpu().numpy(), out[N * K:].cpu().numpy()


def _ref(dy, x, R, N, K, remap):
    cols = np.arange(N)
    cols = cols + np.where(cols >= remap[0], remap[1], 0)
    d = dy[:R].double().cpu().numpy()[:, cols]
    xx = x[:R, :K].double().cpu().numpy()
    return d.T @ xx, d.sum(0)


CASES = [
    # R, N, K, ldy, ldx, nsplit, remap
    (1, 672, 256, 896, 256, 1, (672, 0)),
    (37, 672, 256, 896, 256, 1, (672, 0)),
    (1000, 672, 224, 896, 224, 5, (448, 224)),
    (4099, 672, 256, 896, 256, 32, (672, 0)),
    (2500, 200, 72, 208, 72, 7, (200, 0)),
]


@pytest.mark.parametrize("form", ["fp32", "split", "bf16dy", "bf16dy_bf16x"])
@pytest.mark.parametrize("case", CASES)
def test_wgrad_forms_against_float64(form, case):
    R, N, K, ldy, ldx, ns, remap = case
    g = torch.Generator(device="cuda").manual_seed(R * 7 + N)
    dy = torch.randn(R + 3, ldy, device="cuda", generator=g) * 0.1
    x = torch.randn(R + 3, ldx, device="cuda", generator=g)
    if form == "fp32":
        got = _run("svae_wgrad_cols", dy, x, R, N, K, ldy, ldx, ns, remap)
        # the bias gradient sums dY as stored (fp32); only the MFMA operands are rounded
        ref = (_ref(dy.bfloat16(), x.bfloat16(), R, N, K, remap)[0], _ref(dy, x, R, N, K, remap)[1])
    elif form == "split":
        got = _run("svae_wgrad_cols_prec", dy, x, R, N, K, ldy, ldx, ns, remap, prec=2)
        ref = _ref(dy, x, R, N, K, remap)
    elif form == "bf16dy":
        d16 = dy.bfloat16()
        got = _run("svae_wgrad_cols_bf16dy", d16, x, R, N, K, ldy, ldx, ns, remap)
        ref = _ref(d16, x.bfloat16(), R, N, K, remap)
    else:
        if ldx % 8 or remap[0] % 8 or remap[1] % 8:
            pytest.skip("the all-bf16 form takes multiples of 8 columns")
        d16, x16 = dy.bfloat16(), x.bfloat16()
        got = _run("svae_wgrad_cols_bf16dy_bf16x", d16, x16, R, N, K, ldy, ldx, ns, remap)
        ref = _ref(d16, x16, R, N, K, remap)
    tol = 2e-6 if form != "split" else 1e-6
    for a, b, name in ((got[0], ref[0], "gW"), (got[1], ref[1], "gb")):
        assert np.isfinite(a).all(), (form, name)
        assert np.abs(a - b).max() <= tol * max(np.abs(b).max(), 1e-30) * max(1.0, np.sqrt(R) / 8), (form, name, np.abs(a - b).max(), np.abs(b).max())


@pytest.mark.parametrize("form", ["fp32", "bf16dy", "bf16dy_bf16x"])
def test_wgrad_small_integers_are_exact(form):
    """Operands are small integers (exact in bf16, sums exact in fp32): any misplaced row, column or k index shows as
    a wrong integer.  Asymmetric data, R not a multiple of the stage."""
    R, N, K, ldy, ldx, ns, remap = 333, 672, 224, 896, 224, 3, (448, 224)
    r = torch.arange(R + 1, device="cuda").view(-1, 1)
    dy = ((r * 3 + torch.arange(ldy, device="cuda") * 5) % 7 - 3).float()
    x = ((r * 5 + torch.arange(ldx, device="cuda") * 11) % 5 - 2).float()
    if form == "fp32":
        got = _run("svae_wgrad_cols", dy, x, R, N, K, ldy, ldx, ns, remap)
    elif form == "bf16dy":
        got = _run("svae_wgrad_cols_bf16dy", dy.bfloat16(), x, R, N, K, ldy, ldx, ns, remap)
    else:
        got = _run("svae_wgrad_cols_bf16dy_bf16x", dy.bfloat16(), x.bfloat16(), R, N, K, ldy, ldx, ns, remap)
    ref = _ref(dy, x, R, N, K, remap)
    assert np.array_equal(got[0], ref[0].astype(np.float32))
    assert np.array_equal(got[1], ref[1].astype(np.float32))


@pytest.mark.parametrize("x1_bf16", [0, 1])
@pytest.mark.parametrize("R,ns", [(333, 3), (4099, 32)])
def test_wgrad_pair_equals_the_two_single_launches(x1_bf16, R, ns):
    """svae_wgrad_pair_bf16dy + svae_reduce_slabs_pair (the GRU's two weight gradients in one launch) against
    svae_wgrad_cols_bf16dy / _bf16x + svae_reduce_slabs: the same workgroups write the same slabs -- bit-identical."""
    N, ldy, K0, K1 = 672, 896, 224, 256
    g = torch.Generator(device="cuda").manual_seed(R)
    dy = (torch.randn(R + 3, ldy, device="cuda", generator=g) * 0.1).bfloat16()
    x0 = torch.randn(R + 3, K0, device="cuda", generator=g)
    x1 = torch.randn(R + 3, K1, device="cuda", generator=g)
    x1u = x1.bfloat16() if x1_bf16 else x1
    r0, r1 = (448, 224), (672, 0)
    a = _run("svae_wgrad_cols_bf16dy", dy, x0, R, N, K0, ldy, K0, ns, r0)
    b = _run("svae_wgrad_cols_bf16dy_bf16x" if x1_bf16 else "svae_wgrad_cols_bf16dy", dy, x1u, R, N, K1, ldy, K1, ns, r1)
    st0, st1 = N * K0 + N, N * K1 + N
    s0 = torch.full((ns * st0,), float("nan"), device="cuda")
    s1 = torch.full((ns * st1,), float("nan"), device="cuda")
    o0, o1 = torch.empty(st0, device="cuda"), torch.empty(st1, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    _lib.call("svae_wgrad_pair_bf16dy", dy.data_ptr(), R, N, ldy, ns,
              x0.data_ptr(), K0, K0, s0.data_ptr(), s0.data_ptr() + 4 * N * K0, st0, r0[0], r0[1],
              x1u.data_ptr(), x1_bf16, K1, K1, s1.data_ptr(), s1.data_ptr() + 4 * N * K1, st1, r1[0], r1[1], st)
    _lib.call("svae_reduce_slabs_pair", s0.data_ptr(), o0.data_ptr(), st0, st0, s1.data_ptr(), o1.data_ptr(), st1, st1, ns, st)
    torch.cuda.synchronize()
    o0, o1 = o0.cpu().numpy(), o1.cpu().numpy()
    assert np.array_equal(o0[: N * K0].reshape(N, K0), a[0]) and np.array_equal(o0[N * K0:], a[1])
    assert np.array_equal(o1[: N * K1].reshape(N, K1), b[0]) and np.array_equal(o1[N * K1:], b[1])


@pytest.mark.parametrize("R,ns", [(333, 3), (4099, 40), (21814, 40)])
def test_wgrad_pair_on_256_tiles_equals_the_128_tile_kernels(R, ns):
    """svae_wgrad_pair_big (both X operands bf16, 256 x 256 tiles) against svae_wgrad_pair_bf16dy on the same operands:
    the same bf16 products, another summation order (tile shape and split count): 2e-6 of the result's scale; and the
    bf16 copy svae_f32_to_bf16 makes is torch's (round to nearest even), bit for bit."""
    N, ldy, K0, K1 = 672, 896, 224, 256
    g = torch.Generator(device="cuda").manual_seed(R)
    dy = (torch.randn(R + 3, ldy, device="cuda", generator=g) * 0.1).bfloat16()
    x0 = torch.randn(R + 3, K0, device="cuda", generator=g)
    x1 = torch.randn(R + 3, K1, device="cuda", generator=g).bfloat16()
    st = torch.cuda.current_stream().cuda_stream
    x0b = torch.empty(R + 3, K0, device="cuda", dtype=torch.bfloat16)
    _lib.call("svae_f32_to_bf16", x0.data_ptr(), x0b.data_ptr(), (R + 3) * K0, st)
    assert torch.equal(x0b, x0.bfloat16())
    r0, r1 = (448, 224), (672, 0)
    st0, st1 = N * K0 + N, N * K1 + N

    def run(big, ns_):
        s0 = torch.full((ns_ * st0,), float("nan"), device="cuda")
        s1 = torch.full((ns_ * st1,), float("nan"), device="cuda")
        o0, o1 = torch.empty(st0, device="cuda"), torch.empty(st1, device="cuda")
        if big:
            _lib.call("svae_wgrad_pair_big", dy.data_ptr(), R, N, ldy, ns_,
                      x0b.data_ptr(), K0, K0, s0.data_ptr(), s0.data_ptr() + 4 * N * K0, st0, r0[0], r0[1],
                      x1.data_ptr(), K1, K1, s1.data_ptr(), s1.data_ptr() + 4 * N * K1, st1, r1[0], r1[1], st)
        else:
            _lib.call("svae_wgrad_pair_bf16dy", dy.data_ptr(), R, N, ldy, ns_,
                      x0.data_ptr(), K0, K0, s0.data_ptr(), s0.data_ptr() + 4 * N * K0, st0, r0[0], r0[1],
                      x1.data_ptr(), 1, K1, K1, s1.data_ptr(), s1.data_ptr() + 4 * N * K1, st1, r1[0], r1[1], st)
        _lib.call("svae_reduce_slabs_pair", s0.data_ptr(), o0.data_ptr(), st0, st0, s1.data_ptr(), o1.data_ptr(), st1, st1, ns_, st)
        torch.cuda.synchronize()
        return o0.double(), o1.double()
    a0, a1 = run(False, min(ns, 32))
    b0, b1 = run(True, ns)
    assert torch.isfinite(b0).all() and torch.isfinite(b1).all()
    for a, b in ((a0, b0), (a1, b1)):
        assert float((a - b).abs().max()) <= 2e-6 * float(a.abs().max()), float((a - b).abs().max() / a.abs().max())
    # and against fp64 on the bf16-rounded operands (what both compute)
    dyf = dy[:R].double()
    cols = torch.cat([torch.arange(448), torch.arange(448, 672) + 224]).cuda()
    ref0 = dyf[:, cols].t() @ x0b[:R].double()
    assert float((b0[: N * K0].reshape(N, K0) - ref0).abs().max()) <= 1e-5 * float(ref0.abs().max())


@pytest.mark.parametrize("R,ns", [(333, 3), (4099, 32), (21814, 32)])
def test_hand_waited_bf16_pair_equals_the_builtin_read_kernels(R, ns):
    """svae_wgrad_pair_bf16 (inline-asm LDS reads, hand-counted waits) against svae_wgrad_cols_bf16dy_bf16x on the same
    bf16 operands: the same tiles, stages and MFMA order -- weight gradients BIT-IDENTICAL; the bias gradients (column
    sums, summed in another order) to 2e-6 of their scale."""
    N, ldy, K0, K1 = 672, 896, 224, 256
    g = torch.Generator(device="cuda").manual_seed(R + 1)
    dy = (torch.randn(R + 3, ldy, device="cuda", generator=g) * 0.1).bfloat16()
    x0 = torch.randn(R + 3, K0, device="cuda", generator=g).bfloat16()
    x1 = torch.randn(R + 3, K1, device="cuda", generator=g).bfloat16()
    r0, r1 = (448, 224), (672, 0)
    a = _run("svae_wgrad_cols_bf16dy_bf16x", dy, x0, R, N, K0, ldy, K0, ns, r0)
    b = _run("svae_wgrad_cols_bf16dy_bf16x", dy, x1, R, N, K1, ldy, K1, ns, r1)
    st0, st1 = N * K0 + N, N * K1 + N
    s0 = torch.full((ns * st0,), float("nan"), device="cuda")
    s1 = torch.full((ns * st1,), float("nan"), device="cuda")
    o0, o1 = torch.empty(st0, device="cuda"), torch.empty(st1, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    _lib.call("svae_wgrad_pair_bf16", dy.data_ptr(), R, N, ldy, ns,
              x0.data_ptr(), K0, K0, s0.data_ptr(), s0.data_ptr() + 4 * N * K0, st0, r0[0], r0[1],
              x1.data_ptr(), K1, K1, s1.data_ptr(), s1.data_ptr() + 4 * N * K1, st1, r1[0], r1[1], st)
    _lib.call("svae_reduce_slabs_pair", s0.data_ptr(), o0.data_ptr(), st0, st0, s1.data_ptr(), o1.data_ptr(), st1, st1, ns, st)
    torch.cuda.synchronize()
    o0, o1 = o0.cpu().numpy(), o1.cpu().numpy()
    assert np.array_equal(o0[: N * K0].reshape(N, K0), a[0]) and np.array_equal(o1[: N * K1].reshape(N, K1), b[0])
    for got, ref in ((o0[N * K0:], a[1]), (o1[N * K1:], b[1])):
        assert np.all(np.isfinite(got)) and float(np.abs(got - ref).max()) <= 2e-6 * float(np.abs(ref).max())
