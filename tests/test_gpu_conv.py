"""HIP hypercomplex convolution (forward / data-grad / weight-grad) through the C ABI against
the oracle (fp64) and the reference fixtures.  Tolerance: 1e-4 relative to max|ref| (fp32 MFMA is an
exact-fp32 fma chain; north_star asks 1e-3)."""
import numpy as np
import pytest
import torch

from oracle import seld_oracle as O
from tests.golden.cases import OP_CASES, op_cotangent, op_inputs

pytestmark = pytest.mark.gpu
REL = 1e-4


def _close(got, ref, rel=REL):
    got = got.detach().cpu().double().numpy() if torch.is_tensor(got) else np.asarray(got, np.float64)
    ref = ref.detach().cpu().double().numpy() if torch.is_tensor(ref) else np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    scale = max(float(np.abs(ref).max()), 1e-6)
    err = float(np.abs(got - ref).max())
    assert err <= rel * scale, f"max err {err:.3e} vs scale {scale:.3e}"


CONV_CASES = [c for c in OP_CASES if "conv" in c["kind"]]


@pytest.mark.parametrize("case", CONV_CASES, ids=[c["name"] for c in CONV_CASES])
def test_conv_matches_fixture_and_oracle(case, golden):
    import seld_amd
    H = seld_amd.hip_ops
    g = golden("ops")
    x, ws, bias = op_inputs(case, torch.float32)
    dev = torch.device("cuda:0")
    xd = x.to(dev).requires_grad_(True)
    wd = [w.to(dev).requires_grad_(True) for w in ws]
    bd = bias.to(dev).requires_grad_(True) if bias is not None else None
    y = H.hyper_conv(xd, wd, bd, case["stride"], case["padding"], case["dilation"])
    (y * op_cotangent(y.shape).to(dev)).sum().backward()
    n = case["name"]
    _close(y, g[n + ".y"])
    _close(xd.grad, g[n + ".dx"])
    for i, w in enumerate(wd):
        _close(w.grad, g[f"{n}.dw{i}"])
    if bd is not None:
        _close(bd.grad, g[n + ".dbias"])


@pytest.mark.parametrize("algebra,shape,cout,k,pad,dil", [
    (8, (3, 192, 96), 384, 3, 5, 5),        # TCN dilated conv at config-3 widths (zero-quadrant skip active)
    (8, (2, 384, 100), 192, 1, 0, 1),       # skip / residual 1x1
    (8, (1, 8, 16, 52), 192, (3, 3), 1, 1),   # first CNN layer shape (Cin/8 = 1, K = 72)
    (8, (1, 192, 4, 40), 192, (3, 3), 1, 1),
    (4, (2, 64, 77), 128, 3, 2, 2),         # odd length -> scalar epilogue path
    (1, (2, 24, 50), 40, 3, 1, 1),          # real-valued model
    (1, (2, 8, 9, 33), 16, (3, 3), 1, 1),
    (8, (4, 8, 32, 512), 192, (3, 3), 1, 1),    # persistent short-reduction kernel (first CNN layer), 12 channel tiles
    (4, (2, 8, 64, 512), 64, (3, 3), 1, 1),     # same kernel, quaternion model, 4 channel tiles
    (8, (1, 16, 40, 1024), 128, (1, 3), (0, 2), (1, 2)),   # 1x3 dilated, 16-channel input, 8 channel tiles
    (8, (2, 16, 32, 512), 192, (3, 3), 1, 1),   # config 4 / 5 first layer: 16-channel mag+phase input, K = 144 -> the
                                                # generic branch of the short-reduction kernel
])
def test_conv_random_vs_oracle(algebra, shape, cout, k, pad, dil):
    import seld_amd
    H = seld_amd.hip_ops
    gen = torch.Generator().manual_seed(1234)
    x = torch.randn(shape, generator=gen)
    kk = (k,) if isinstance(k, int) else k
    wshape = (cout // algebra, shape[1] // algebra) + tuple(kk)
    ws = [torch.randn(wshape, generator=gen) * 0.2 for _ in range(algebra)]
    bias = torch.randn(cout, generator=gen)
    dev = torch.device("cuda:0")
    xd = x.to(dev).requires_grad_(True)
    wd = [w.to(dev).requires_grad_(True) for w in ws]
    bd = bias.to(dev).requires_grad_(True)
    y = H.hyper_conv(xd, wd, bd, 1, pad, dil)
    cot = torch.randn(y.shape, generator=gen)
    (y * cot.to(dev)).sum().backward()

    x64 = x.double().requires_grad_(True)
    w64 = [w.double().requires_grad_(True) for w in ws]
    b64 = bias.double().requires_grad_(True)
    yr = O.hypercomplex_conv(x64, w64, b64, 1, pad, 1, dil, mode="explicit")
    (yr * cot.double()).sum().backward()
    _close(y, yr)
    _close(xd.grad, x64.grad)
    for a, b in zip(wd, w64):
        _close(a.grad, b.grad)
    _close(bd.grad, b64.grad)


# Loop-invariant-staging kernel (hc_conv_vec.hip): tile forced with SELD_CONV_CFG so that small problems take it.
# Cases cover every tap set, all three algebras, the dual-quaternion zero quadrant with a workgroup straddling the
# primal/dual boundary (12 tiles over 192 channels) and without (6 tiles), edge workgroups (every tile of a 1-image
# problem) next to interior ones (N >= 3), images shorter than a tile, a ragged last tile, channel counts that do
# not fill the tile, halo wider than a quad (dilation 5 / 27) and the fused epilogues.
VEC_CASES = [
    # algebra, x shape, cout, k, pad, dil, cfg
    (8, (4, 192, 128), 384, 3, 1, 1, "12,1"),
    (8, (4, 192, 128), 192, 3, 5, 5, "12,1"),
    (8, (2, 192, 64), 192, 3, 1, 1, "6,1"),          # an all-primal and an all-dual workgroup
    (8, (3, 384, 64), 192, 1, 0, 1, "6,1"),
    (8, (5, 192, 40), 192, 3, 27, 27, "6,1"),        # images shorter than the 64-position tile, ragged tail
    (8, (3, 192, 8, 32), 192, (3, 3), 1, 1, "12,1"),
    (8, (2, 96, 6, 16), 32, (3, 3), 1, 1, "6,1"),      # 32 of 96 tile channels used
    (4, (3, 96, 100), 64, 3, 2, 2, "6,1"),
    (4, (2, 16, 12, 24), 48, (3, 3), 1, 1, "12,1"),
    (1, (3, 24, 72), 48, 1, 0, 1, "6,1"),
    (1, (2, 8, 9, 36), 16, (3, 3), 1, 1, "6,1"),
    (1, (4, 8, 256), 24, 3, 3, 3, "12,1"),
]


@pytest.mark.parametrize("algebra,shape,cout,k,pad,dil,cfg", VEC_CASES)
def test_conv_vec_kernel_vs_oracle(algebra, shape, cout, k, pad, dil, cfg, seld_env):
    import seld_amd
    H = seld_amd.hip_ops
    seld_env.set("SELD_CONV_NO_HCQ", "1")        # this test is about the 16/48-product kernel generation
    seld_env.set("SELD_CONV_CFG", cfg)
    kk = (k,) if isinstance(k, int) else k
    desc = H.make_conv_desc(tuple(shape), cout, algebra, kk, 1, pad, dil)
    want = "hc_conv_vec_kernel<" + cfg.replace(",", ", ")
    assert H._label(desc, 0).startswith(want), H._label(desc, 0)
    assert H._label(desc, 1).startswith(want), H._label(desc, 1)
    gen = torch.Generator().manual_seed(4321)
    x = torch.randn(shape, generator=gen)
    wshape = (cout // algebra, shape[1] // algebra) + tuple(kk)
    ws = [torch.randn(wshape, generator=gen) * 0.2 for _ in range(algebra)]
    bias = torch.randn(cout, generator=gen)
    dev = torch.device("cuda:0")
    xd = x.to(dev).requires_grad_(True)
    wd = [w.to(dev).requires_grad_(True) for w in ws]
    bd = bias.to(dev).requires_grad_(True)
    y = H.hyper_conv(xd, wd, bd, 1, pad, dil)
    cot = torch.randn(y.shape, generator=gen)
    (y * cot.to(dev)).sum().backward()
    x64 = x.double().requires_grad_(True)
    w64 = [w.double() for w in ws]
    yr = O.hypercomplex_conv(x64, w64, bias.double(), 1, pad, 1, dil, mode="explicit")
    (yr * cot.double()).sum().backward()
    _close(y, yr)
    _close(xd.grad, x64.grad)
    # fused epilogues on the same kernel: + addend, batch statistics
    L = seld_amd._lib
    addend = torch.randn(y.shape, generator=gen).to(dev)
    stats_rep = H.new_stats(cout, dev)
    y2 = H.conv_fwd(desc, xd.detach(), [w.detach() for w in wd], bd.detach(),
                    epilogue=L.SELD_EPI_ADD | L.SELD_EPI_STATS, addend=addend, stats=stats_rep)
    torch.cuda.synchronize()
    ref2 = yr.detach() + addend.double().cpu()
    _close(y2, ref2)
    stats = stats_rep.view(H.STATS_REPLICAS, 2 * cout).sum(0).double().cpu()
    red = tuple(i for i in range(ref2.dim()) if i != 1)
    assert torch.allclose(stats[:cout], ref2.sum(dim=red), rtol=1e-4, atol=2e-3)
    assert torch.allclose(stats[cout:], (ref2 * ref2).sum(dim=red), rtol=1e-4, atol=2e-3)


PAIR_CASES = [
    # algebra, x shape, cout, k, pad, dil, addends
    (8, (4, 192, 128), 384, 3, 5, 5, False),        # conv1_filter | conv1_gate
    (8, (3, 384, 64), 192, 1, 0, 1, True),          # conv2_skip (+ running sum) | conv2_residual (+ x)
    (4, (2, 96, 96), 64, 3, 2, 2, True),
    (8, (2, 192, 8, 32), 192, (3, 3), 1, 1, False),
]


@pytest.mark.parametrize("algebra,shape,cout,k,pad,dil,adds", PAIR_CASES)
def test_conv_pair_matches_two_single_calls(algebra, shape, cout, k, pad, dil, adds, seld_env):
    """seld_hc_conv_pair_* (one launch for two convolutions of the same input) against the single entry points
    and, through them, the oracle: outputs, the summed data gradient and both sets of weight gradients."""
    import ctypes
    import seld_amd
    H, L = seld_amd.hip_ops, seld_amd._lib
    seld_env.set("SELD_CONV_NO_HCQ", "1")        # the 16/48-product generation (test_hcq_* covers the other one)
    kk = (k,) if isinstance(k, int) else k
    desc = H.make_conv_desc(tuple(shape), cout, algebra, kk, 1, pad, dil)
    assert H._pair_ok(desc, 0) and H._pair_ok(desc, 2)
    assert H._pair_ok(desc, 1) == (len(shape) == 3)      # summed data gradient: 1-D layers; 2-D falls back to two calls
    gen = torch.Generator().manual_seed(99)
    dev = torch.device("cuda:0")
    wshape = (cout // algebra, shape[1] // algebra) + tuple(kk)
    x = torch.randn(shape, generator=gen).to(dev)
    sets = []
    for _ in range(2):
        ws = [(torch.randn(wshape, generator=gen) * 0.2).to(dev) for _ in range(algebra)]
        sets.append((ws, torch.randn(cout, generator=gen).to(dev)))
    y_shape = H.conv_fwd(desc, x, sets[0][0]).shape
    addends = [torch.randn(y_shape, generator=gen).to(dev) if adds else None for _ in range(2)]
    cots = [torch.randn(y_shape, generator=gen).to(dev) for _ in range(2)]

    def run(pair):
        xs = x.clone().requires_grad_(True)
        wl = [[w.clone().requires_grad_(True) for w in ws] for ws, _ in sets]
        bl = [b.clone().requires_grad_(True) for _, b in sets]
        al = [a.clone().requires_grad_(True) if a is not None else None for a in addends]
        if pair:
            ya, yb = H.hyper_conv_pair(xs, wl[0], bl[0], wl[1], bl[1], 1, pad, dil, al[0], al[1])
        else:
            one = lambda i: (H.hyper_conv(xs, wl[i], bl[i], 1, pad, dil) if al[i] is None
                             else H.hyper_conv_add(xs, wl[i], bl[i], al[i], 1, pad, dil))
            ya, yb = one(0), one(1)
        ((ya * cots[0]).sum() + (yb * cots[1]).sum()).backward()
        return (ya, yb, xs.grad, [w.grad for w in wl[0]], [w.grad for w in wl[1]], bl[0].grad, bl[1].grad,
                [a.grad if a is not None else None for a in al])

    got, ref = run(True), run(False)
    _close(got[0], ref[0]); _close(got[1], ref[1]); _close(got[2], ref[2])
    for a, b in zip(got[3] + got[4], ref[3] + ref[4]):
        _close(a, b)
    _close(got[5], ref[5]); _close(got[6], ref[6])
    for a, b in zip(got[7], ref[7]):
        if b is not None:
            _close(a, b)
    # the pair weight-gradient entry point itself (the autograd path above only takes it with FlatAdam's slots)
    dwa = [torch.zeros(wshape, device=dev) for _ in range(algebra)]
    dwb = [torch.zeros(wshape, device=dev) for _ in range(algebra)]
    dba, dbb = torch.zeros(cout, device=dev), torch.zeros(cout, device=dev)
    L.check(L.lib().seld_hc_conv_pair_bwd_weight_acc(ctypes.byref(desc), L.ptr(x), L.ptr(cots[0]), L.ptr(cots[1]),
                                                     L.ptr_array8(dwa), L.ptr_array8(dwb), L.ptr(dba), L.ptr(dbb),
                                                     L.current_stream()), "pair wgrad")
    torch.cuda.synchronize()
    for a, b in zip(dwa + dwb, ref[3] + ref[4]):
        _close(a, b)
    _close(dba, ref[5]); _close(dbb, ref[6])


def test_conv_epilogues():
    import seld_amd
    H = seld_amd.hip_ops
    L = seld_amd._lib
    gen = torch.Generator().manual_seed(7)
    dev = torch.device("cuda:0")
    x = torch.randn(2, 32, 64, generator=gen).to(dev)
    ws = [(torch.randn(8, 4, 1, generator=gen) * 0.3).to(dev) for _ in range(8)]
    desc = H.make_conv_desc(tuple(x.shape), 64, 8, (1,), 1, 0, 1)
    base = H.conv_fwd(desc, x, ws)
    addend = torch.randn(2, 64, 64, generator=gen).to(dev)
    stats_rep = H.new_stats(64, dev)
    y = H.conv_fwd(desc, x, ws, epilogue=L.SELD_EPI_ADD | L.SELD_EPI_STATS, addend=addend, stats=stats_rep)
    torch.cuda.synchronize()
    stats = stats_rep.view(H.STATS_REPLICAS, 128).sum(0)
    assert torch.allclose(y, base + addend, atol=1e-5)
    ref = (base + addend).double()
    assert torch.allclose(stats[:64].double().cpu(), ref.sum(dim=(0, 2)).cpu(), rtol=1e-4, atol=1e-3)
    assert torch.allclose(stats[64:].double().cpu(), (ref * ref).sum(dim=(0, 2)).cpu(), rtol=1e-4, atol=1e-3)
    acc = addend.clone()
    H.conv_fwd(desc, x, ws, out=acc, epilogue=L.SELD_EPI_ACCUMULATE)
    assert torch.allclose(acc, base + addend, atol=1e-5)


# Full-size layers of the benchmark workload (config 3, batch 32): too big for the CPU oracle, so parity rests on
# size-independent identities that tie the three kernels of a layer to each other and to linearity:
#   <conv_W(x), dy> = <x, dgrad_W(dy)> = <W, wgrad(x, dy)>      (adjointness; the Hamilton structure cancels out)
#   conv_W(a*x1 + b*x2) = a*conv_W(x1) + b*conv_W(x2)
FULL_LAYERS = [
    ("cnn0", (32, 8, 128, 512), 192, (3, 3), 1, 1),
    ("cnn1", (32, 192, 16, 512), 192, (3, 3), 1, 1),
    ("tcn_k3_d8", (32, 192, 512), 384, (3,), 8, 8),
    ("tcn_k3_d89", (32, 192, 512), 384, (3,), 89, 89),
    ("tcn_k1", (32, 384, 512), 192, (1,), 0, 1),
]


@pytest.mark.parametrize("name,shape,cout,k,pad,dil", FULL_LAYERS, ids=[c[0] for c in FULL_LAYERS])
def test_full_size_adjoint_and_linearity(name, shape, cout, k, pad, dil):
    import seld_amd
    H = seld_amd.hip_ops
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(77)
    A = 8
    x = torch.randn(shape, device=dev, generator=gen, requires_grad=True)
    ws = [(torch.randn((cout // A, shape[1] // A) + tuple(k), device=dev, generator=gen) * 0.1).requires_grad_(True)
          for _ in range(A)]
    y = H.hyper_conv(x, ws, None, 1, pad, dil)
    dy = torch.randn(y.shape, device=dev, generator=gen)
    lhs = (y.detach().double() * dy.double()).sum()
    (y * dy).sum().backward()
    mid = (x.detach().double() * x.grad.double()).sum()
    rhs = sum((w.detach().double() * w.grad.double()).sum() for w in ws)
    scale = float(y.detach().double().norm() * dy.double().norm())
    assert abs(float(lhs - mid)) < 2e-5 * scale, (float(lhs), float(mid))
    assert abs(float(lhs - rhs)) < 2e-5 * scale, (float(lhs), float(rhs))
    with torch.no_grad():
        x2 = torch.randn(shape, device=dev, generator=gen)
        wd = [w.detach() for w in ws]
        y12 = H.hyper_conv(0.75 * x.detach() - 1.5 * x2, wd, None, 1, pad, dil)
        y2 = H.hyper_conv(x2, wd, None, 1, pad, dil)
        err = float((y12 - (0.75 * y.detach() - 1.5 * y2)).abs().max())
        assert err < 1e-4 * float(y12.abs().max()), err


def test_first_layer_kernel_matches_generic_kernel_and_oracle(seld_env):
    """The persistent short-reduction kernel in its first-layer shape (8 -> 192 channels, 3x3, 64-position tiles in one
    output row, bias folded into the accumulators, BatchNorm statistics in the epilogue) against the generic kernel on
    the same full-size input (bit-level agreement is not expected: different summation order) and against the oracle on
    one sample."""
    import seld_amd
    H, L = seld_amd.hip_ops, seld_amd._lib
    DEV = "cuda:0"
    g = torch.Generator().manual_seed(7)
    x = torch.randn(2, 8, 128, 512, generator=g)
    ws = [torch.randn(24, 1, 3, 3, generator=g) * 0.2 for _ in range(8)]
    bias = torch.randn(192, generator=g)
    xd, wd, bd = x.to(DEV), [w.to(DEV) for w in ws], bias.to(DEV)
    desc = H.make_conv_desc(tuple(x.shape), 192, 8, (3, 3), 1, 1, 1)
    stats = H.new_stats(192, torch.device(DEV))
    y = H.conv_fwd(desc, xd, wd, bias=bd, epilogue=L.SELD_EPI_STATS, stats=stats)
    seld_env.set("SELD_CONV_NO_SMALLK", "1")
    stats_ref = H.new_stats(192, torch.device(DEV))
    y_ref = H.conv_fwd(desc, xd, wd, bias=bd, epilogue=L.SELD_EPI_STATS, stats=stats_ref)
    seld_env.unset("SELD_CONV_NO_SMALLK")
    scale = float(y_ref.abs().max())
    assert float((y - y_ref).abs().max()) <= 1e-5 * scale
    s, sr = stats.view(-1, 2, 192).sum(0), stats_ref.view(-1, 2, 192).sum(0)
    assert torch.allclose(s, sr, rtol=1e-4, atol=1e-2)
    ref = O.dual_quaternion_conv(x[:1].double(), *[w.double() for w in ws], bias.double(), 1, 1, 1, 1)
    assert float((y[:1].cpu().double() - ref).abs().max()) <= 1e-4 * float(ref.abs().max())


# ---- 8-multiplication Hamilton product kernels (csrc/hcq_conv.hip) ------------------------------------------------------
# Every instantiation family: dual quaternion with 16-aligned block channels, with 24 (mixed 8 + 8 tile in workgroups
# of its own when there are few position tiles, three tiles per workgroup when there are many), quaternion with one and
# two tiles; 1x1 / 1x3 / 3x3 taps; every K-chunk size (8 and 4 for 1x3 by halo width, 16 / 24 / 8 for 1x1, 4 for 3x3);
# halos across image rows and batch items; dilation larger than the tile.
HCQ_CASES = [
    # algebra, x shape, cout, k, pad, dil
    (8, (2, 192, 128), 384, 3, 5, 5),            # TCN filter / gate; data gradient has 24 block channels, few tiles -> split
    (8, (3, 192, 64), 384, 3, 55, 55),           # halo wider than the tile, K chunk 4
    (8, (2, 384, 128), 192, 1, 0, 1),            # skip / residual: chunk 16 forward, 24 in the data gradient
    (8, (2, 192, 128), 128, 1, 0, 1),            # chunk 8 (24 block channels of input, 16 of output)
    (8, (1, 192, 4, 128), 192, (3, 3), 1, 1),    # 3x3, halo rows, 24 block channels split
    (8, (9, 192, 8, 512), 192, (3, 3), 1, 1),    # >= 512 position tiles: three tiles per workgroup
    (8, (2, 128, 3, 64), 128, (3, 3), 1, 1),
    (4, (2, 64, 128), 128, 3, 2, 2),             # quaternion, two tiles
    (4, (2, 128, 64), 64, 1, 0, 1),              # one tile
    (4, (2, 64, 5, 64), 64, (3, 3), 1, 1),
    (4, (3, 64, 192), 64, 3, 34, 34),
]


@pytest.mark.parametrize("algebra,shape,cout,k,pad,dil", HCQ_CASES)
def test_hcq_conv_vs_oracle(algebra, shape, cout, k, pad, dil):
    """Forward, data gradient (through autograd) and the fused epilogues of the fast-product kernels against the
    fp64 oracle (the reference's block-matrix algorithm), tolerance as every other convolution test."""
    import seld_amd
    H, L = seld_amd.hip_ops, seld_amd._lib
    kk = (k,) if isinstance(k, int) else k
    desc = H.make_conv_desc(tuple(shape), cout, algebra, kk, 1, pad, dil)
    assert H.hcq_label(desc, 0).startswith("hcq_conv_kernel<") and H.hcq_label(desc, 1).startswith("hcq_conv_kernel<")
    gen = torch.Generator().manual_seed(77)
    x = torch.randn(shape, generator=gen)
    wshape = (cout // algebra, shape[1] // algebra) + tuple(kk)
    ws = [torch.randn(wshape, generator=gen) * 0.2 for _ in range(algebra)]
    bias = torch.randn(cout, generator=gen)
    dev = torch.device("cuda:0")
    xd = x.to(dev).requires_grad_(True)
    wd = [w.to(dev).requires_grad_(True) for w in ws]
    bd = bias.to(dev).requires_grad_(True)
    y = H.hyper_conv(xd, wd, bd, 1, pad, dil)
    cot = torch.randn(y.shape, generator=gen)
    (y * cot.to(dev)).sum().backward()
    x64 = x.double().requires_grad_(True)
    w64 = [w.double().requires_grad_(True) for w in ws]
    yr = O.hypercomplex_conv(x64, w64, bias.double(), 1, pad, 1, dil, mode="explicit")
    (yr * cot.double()).sum().backward()
    _close(y, yr)
    _close(xd.grad, x64.grad)
    for a, b in zip(wd, w64):                    # weight gradients (their own kernels) on the same shapes
        _close(a.grad, b.grad)
    addend = torch.randn(y.shape, generator=gen).to(dev)
    stats_rep = H.new_stats(cout, dev)
    y2 = H.conv_fwd(desc, xd.detach(), [w.detach() for w in wd], bd.detach(),
                    epilogue=L.SELD_EPI_ADD | L.SELD_EPI_STATS, addend=addend, stats=stats_rep)
    ref2 = yr.detach() + addend.double().cpu()
    _close(y2, ref2)
    stats = stats_rep.view(H.STATS_REPLICAS, 2 * cout).sum(0).double().cpu()
    red = tuple(i for i in range(ref2.dim()) if i != 1)
    assert torch.allclose(stats[:cout], ref2.sum(dim=red), rtol=1e-4, atol=2e-3)
    assert torch.allclose(stats[cout:], (ref2 ** 2).sum(dim=red), rtol=1e-4, atol=2e-3)
    # y += conv(x)
    y3 = y2.clone()
    H.conv_fwd(desc, xd.detach(), [w.detach() for w in wd], None, out=y3, epilogue=L.SELD_EPI_ACCUMULATE)
    _close(y3, ref2 + (yr.detach() - bias.double().view(1, -1, *([1] * (yr.dim() - 2)))))


@pytest.mark.parametrize("algebra,shape,cout", [
    (8, (2, 8, 6, 128), 192),        # the 8-channel first layer: 1 block channel x 9 taps = 9 of 12 k-slots (at the
                                     # benchmark's size this layer stays on hc_conv_smallk_kernel: hcq_plan)
    (8, (2, 16, 5, 64), 192),        # the 16-channel (mag + phase) first layer: K = 144 -> 18 of 20
    (4, (2, 8, 4, 64), 64),          # quaternion first layer (config 2): 2 block channels
    (8, (1, 48, 3, 64), 128),        # 6 block channels: three chunks of two
    (4, (3, 12, 2, 128), 128),       # 3 block channels: three chunks of one
    (8, (2, 8, 16, 128), 192),       # image height a multiple of 8: the row-walking kernel (hcq_first_kernel), 3 tiles
    (8, (1, 16, 8, 192), 192),       # ... two block channels
    (8, (2, 8, 8, 64), 128),         # ... two tiles
    (4, (2, 8, 24, 64), 64),         # ... quaternion, one tile
    (4, (1, 4, 8, 128), 128),        # ... quaternion, one block channel, two tiles
])
def test_hcq_forward_with_padded_k_groups(algebra, shape, cout):
    """3x3 layers whose input has fewer than 4 block channels per K chunk (the networks' first layers): the last k-group
    of a chunk is padded (zero packed weights, csrc/hcq_conv.hip).  Forward + BatchNorm statistics against the oracle."""
    import seld_amd
    H, L = seld_amd.hip_ops, seld_amd._lib
    desc = H.make_conv_desc(tuple(shape), cout, algebra, (3, 3), 1, 1, 1)
    assert H.hcq_label(desc, 0).startswith(("hcq_conv_kernel<3, 3, ", "hcq_first_kernel<"))
    gen = torch.Generator().manual_seed(78)
    x = torch.randn(shape, generator=gen)
    wshape = (cout // algebra, shape[1] // algebra, 3, 3)
    ws = [torch.randn(wshape, generator=gen) * 0.2 for _ in range(algebra)]
    bias = torch.randn(cout, generator=gen)
    dev = torch.device("cuda:0")
    wd = [w.to(dev) for w in ws]
    stats_rep = H.new_stats(cout, dev)
    y = H.conv_fwd(desc, x.to(dev), wd, bias.to(dev), epilogue=L.SELD_EPI_STATS, stats=stats_rep)
    yr = O.hypercomplex_conv(x.double(), [w.double() for w in ws], bias.double(), 1, 1, 1, 1, mode="explicit")
    _close(y, yr)
    stats = stats_rep.view(H.STATS_REPLICAS, 2 * cout).sum(0).double().cpu()
    assert torch.allclose(stats[:cout], yr.sum(dim=(0, 2, 3)), rtol=1e-4, atol=2e-3)
    assert torch.allclose(stats[cout:], (yr ** 2).sum(dim=(0, 2, 3)), rtol=1e-4, atol=2e-3)
    y0 = H.conv_fwd(desc, x.to(dev), wd)                       # plain epilogue
    _close(y0, yr - bias.double().view(1, -1, 1, 1))


@pytest.mark.parametrize("algebra,shape,cout,k,pad,dil", [c for c in HCQ_CASES if c[0] == 8][:5] + [HCQ_CASES[7], HCQ_CASES[8]])
def test_hcq_pair_matches_two_single_calls(algebra, shape, cout, k, pad, dil):
    """Two convolutions of one input in one launch (forward) and the sum of their data gradients in one launch, on the
    fast-product kernels, against the two single calls; replaying the step after an in-place weight edit must pick up
    the new weights (packed-form cache)."""
    import seld_amd
    H = seld_amd.hip_ops
    kk = (k,) if isinstance(k, int) else k
    desc = H.make_conv_desc(tuple(shape), cout, algebra, kk, 1, pad, dil)
    assert H._hcq_ok(desc, 0, 2) and H._hcq_ok(desc, 1, 2)
    gen = torch.Generator().manual_seed(5)
    dev = torch.device("cuda:0")
    wshape = (cout // algebra, shape[1] // algebra) + tuple(kk)
    x = torch.randn(shape, generator=gen).to(dev)
    wl = [[(torch.randn(wshape, generator=gen) * 0.2).to(dev).requires_grad_(True) for _ in range(algebra)] for _ in range(2)]
    y_shape = H.conv_fwd(desc, x, [w.detach() for w in wl[0]]).shape
    adds = [torch.randn(y_shape, generator=gen).to(dev), None]
    cots = [torch.randn(y_shape, generator=gen).to(dev) for _ in range(2)]
    stats = [H.new_stats(cout, dev), H.new_stats(cout, dev)]

    def run(pair):
        xs = x.clone().requires_grad_(True)
        if pair:
            ya, yb = H.hyper_conv_pair(xs, wl[0], None, wl[1], None, 1, pad, dil, adds[0], adds[1], stats[0], stats[1])
        else:
            ya = H.hyper_conv_add(xs, wl[0], None, adds[0], 1, pad, dil)
            yb = H.hyper_conv(xs, wl[1], None, 1, pad, dil)
        ((ya * cots[0]).sum() + (yb * cots[1]).sum()).backward()
        return ya.detach(), yb.detach(), xs.grad

    for attempt in range(2):
        for s_ in stats:
            s_.zero_()
        got, ref = run(True), run(False)
        for a, b in zip(got, ref):
            _close(a, b)
        red = tuple(i for i in range(ref[0].dim()) if i != 1)
        for s_, r in zip(stats, ref[:2]):
            tot = s_.view(H.STATS_REPLICAS, 2 * cout).sum(0)
            assert torch.allclose(tot[:cout], r.sum(dim=red), rtol=1e-4, atol=2e-3)
        with torch.no_grad():                       # edit the weights in place: the packed forms must follow
            for w in wl[0] + wl[1]:
                w.mul_(-0.5).add_(0.01)
                w.grad = None


@pytest.mark.parametrize("algebra,shape,cout,k,pad,dil", [HCQ_CASES[0], HCQ_CASES[2], HCQ_CASES[4], HCQ_CASES[6],
                                                          HCQ_CASES[7], HCQ_CASES[8], HCQ_CASES[9]])
def test_hcq_wgrad_matches_block_matrix_kernels(algebra, shape, cout, k, pad, dil, seld_env):
    """Fast-product weight gradient (hcq_wgrad.hip: regular row tiles, the mixed 8 + 8 row tile, quaternion; single
    and pair launches) against the 16/48-product kernels, which the oracle tests pin."""
    import seld_amd
    H = seld_amd.hip_ops
    seld_env.set("SELD_HCQ_WGRAD_DQ", "1")
    kk = (k,) if isinstance(k, int) else k
    desc = H.make_conv_desc(tuple(shape), cout, algebra, kk, 1, pad, dil)
    if not H._hcq_wgrad_ok(desc):
        pytest.skip("shape not taken by the fast-product weight-gradient kernel (LDS budget)")
    gen = torch.Generator().manual_seed(11)
    dev = torch.device("cuda:0")
    x = torch.randn(shape, generator=gen).to(dev)
    wshape = (cout // algebra, shape[1] // algebra) + tuple(kk)
    yshape = (shape[0], cout) + tuple(shape[2:])
    dyA, dyB = torch.randn(yshape, generator=gen).to(dev), torch.randn(yshape, generator=gen).to(dev)
    new = [[torch.zeros(wshape, device=dev) for _ in range(algebra)] for _ in range(3)]
    H.hcq_wgrad_acc(desc, x, dyA, new[0])
    if H._hcq_wgrad_ok(desc, 2):
        H.hcq_wgrad_acc(desc, x, dyA, new[1], dyB, new[2])
    seld_env.set("SELD_CONV_NO_HCQ", "1")
    old = [[torch.zeros(wshape, device=dev) for _ in range(algebra)] for _ in range(2)]
    H.conv_bwd_weight(desc, x, dyA, wshape, False, into=old[0])
    H.conv_bwd_weight(desc, x, dyB, wshape, False, into=old[1])
    for a, b in zip(new[0], old[0]):
        _close(a, b)
    if H._hcq_wgrad_ok(desc, 2) or True:
        seld_env.unset("SELD_CONV_NO_HCQ")
        if H._hcq_wgrad_ok(desc, 2):
            for a, b in zip(new[1] + new[2], old[0] + old[1]):
                _close(a, b)


@pytest.mark.parametrize("shape,cout,k,pad,dil", [
    ((8, 192, 512), 384, 3, 5, 5),               # TCN filter / gate: 48 output block channels, 72 columns (one group of 96)
    ((8, 384, 512), 192, 1, 0, 1),               # skip / residual: 24 block channels, 48 columns (one group of 64)
    ((2, 192, 8, 512), 192, (3, 3), 1, 1),       # cnn.1 / cnn.2: 216 columns = four groups of 64
    ((1, 192, 8, 512), 384, (3, 3), 1, 1),       # 48 block channels x 216 columns
    ((9, 192, 512), 384, 3, 55, 55),             # dilation wider than a step: every step gathers at the row ends; odd split
])
def test_hcq_wgrad_row_matches_block_matrix_kernels(shape, cout, k, pad, dil, seld_env):
    """The 24-product dual-quaternion weight gradient (hcq_wgrad_row.hip: forms staged into the row-chunk GEMM, two tile
    families + fold, single and pair launches, scratch handed back zeroed) against the 48-product kernels, which the
    oracle tests pin; the first shape also against the oracle directly."""
    import seld_amd
    H = seld_amd.hip_ops
    seld_env.set("SELD_HCQ_WGRAD_ROW", "1")              # opt-in: correct, not yet faster than the 48-product kernels
    kk = (k,) if isinstance(k, int) else k
    desc = H.make_conv_desc(tuple(shape), cout, 8, kk, 1, pad, dil)
    assert H._hcq_wgrad_row_bytes(desc) > 0 and H._hcq_wgrad_row_bytes(desc, 2) > 0
    assert H._hcq_wgrad_row_label(desc).startswith("hcq_wgrad_row_kernel<")
    gen = torch.Generator().manual_seed(13)
    dev = torch.device("cuda:0")
    x = torch.randn(shape, generator=gen).to(dev)
    wshape = (cout // 8, shape[1] // 8) + tuple(kk)
    yshape = (shape[0], cout) + tuple(shape[2:])
    dyA, dyB = torch.randn(yshape, generator=gen).to(dev), torch.randn(yshape, generator=gen).to(dev)
    new = [[torch.zeros(wshape, device=dev) for _ in range(8)] for _ in range(3)]
    H.hcq_wgrad_acc(desc, x, dyA, new[0])
    H.hcq_wgrad_acc(desc, x, dyA, new[1], dyB, new[2])
    H.hcq_wgrad_acc(desc, x, dyA, new[0])                       # accumulates; the scratch came back zeroed
    for ws in H._wgrad_row_scratch.values():
        assert float(ws.abs().max()) == 0.0
    seld_env.set("SELD_CONV_NO_HCQ", "1")
    assert H._hcq_wgrad_row_bytes(desc) == 0
    old = [[torch.zeros(wshape, device=dev) for _ in range(8)] for _ in range(2)]
    H.conv_bwd_weight(desc, x, dyA, wshape, False, into=old[0])
    H.conv_bwd_weight(desc, x, dyB, wshape, False, into=old[1])
    for a, b in zip(new[0], old[0]):
        _close(a, 2.0 * b)
    for a, b in zip(new[1] + new[2], old[0] + old[1]):
        _close(a, b)
    if shape == (8, 192, 512):
        x64 = x.cpu().double()
        w64 = [torch.zeros(wshape, dtype=torch.float64, requires_grad=True) for _ in range(8)]
        yr = O.hypercomplex_conv(x64, w64, None, 1, pad, 1, dil, mode="explicit")
        (yr * dyB.cpu().double()).sum().backward()
        for a, b in zip(new[2], w64):
            _close(a, b.grad)


# ---------------------------------------------------------------------------------------------------------------------
# grouped persistent weight gradient (csrc/hcq_wgrad_grp.hip)
# ---------------------------------------------------------------------------------------------------------------------
def _grp_job(H, shape, cout, k, pad, dil, gen, dev):
    kk = (k,) if isinstance(k, int) else k
    desc = H.make_conv_desc(tuple(shape), cout, 8, kk, 1, pad, dil)
    x = torch.randn(shape, generator=gen).to(dev)
    yshape = (shape[0], cout) + tuple(shape[2:])
    dy = torch.randn(yshape, generator=gen).to(dev)
    wshape = (cout // 8, shape[1] // 8) + tuple(kk)
    return desc, x, dy, wshape


GRP_LISTS = {
    # the residual blocks' dilated pairs (dilations straddling the 16-position step and the 4-column alignment) + tcn.conv1-like
    "tcn_1x3": [((4, 192, 512), 384, 3, d, d) for d in (1, 2, 3, 5, 8, 13, 21, 34, 55)] + [((4, 192, 512), 384, 3, 1, 1)],
    # skip / residual 1x1
    "tcn_1x1": [((4, 384, 512), 192, 1, 0, 1)] * 5,
    # cnn.1 / cnn.2: three kernel rows each, different heights in one launch
    "cnn_3x3": [((2, 192, 16, 512), 192, (3, 3), 1, 1), ((2, 192, 2, 512), 192, (3, 3), 1, 1)],
    # tcn.conv2: 384 -> 384 1x3 as three single-tap sub-jobs
    "wide_1x3": [((4, 384, 128), 384, 3, 1, 1)],
    # everything at once: four families in one call, shorter rows
    "mixed": [((2, 192, 256), 384, 3, 2, 2), ((2, 384, 256), 192, 1, 0, 1), ((2, 384, 256), 384, 3, 1, 1)],
}


@pytest.mark.parametrize("name", sorted(GRP_LISTS))
def test_wgrad_group_matches_block_matrix_kernels_and_oracle(name, seld_env):
    """seld_hcq_wgrad_group on a list of layers (one persistent launch per shape family: work split across layer
    boundaries, sub-jobs per kernel row / tap, combined and split accumulator layouts) against the 48-product per-layer
    kernels, which the oracle tests pin; the first job of every list also against the fp64 oracle; accumulation into
    non-zero gradient slots; bit-identical results from two calls (no atomics, fixed summation order)."""
    import seld_amd
    H = seld_amd.hip_ops
    gen = torch.Generator().manual_seed(17)
    dev = torch.device("cuda:0")
    specs = GRP_LISTS[name]
    jobs, refs = [], []
    for shape, cout, k, pad, dil in specs:
        desc, x, dy, wshape = _grp_job(H, shape, cout, k, pad, dil, gen, dev)
        dws = [torch.full(wshape, 0.25 * (c + 1), device=dev) for c in range(8)]           # non-zero slots: the call ADDS
        jobs.append((desc, x, dy, dws))
    assert H.wgrad_group(jobs), "a job of the list was not taken by the grouped kernels"
    torch.cuda.synchronize()
    first = [[w.clone() for w in j[3]] for j in jobs]
    seld_env.set("SELD_CONV_NO_HCQ", "1")                 # the 48-product per-layer kernels
    for (desc, x, dy, dws), got in zip(jobs, first):
        old = [torch.full_like(w, 0.25 * (c + 1)) for c, w in enumerate(dws)]
        H.conv_bwd_weight(desc, x, dy, tuple(dws[0].shape), False, into=old)
        scale = max(float(o.abs().max()) for o in old)
        for a, b in zip(got, old):
            assert float((a - b).abs().max()) <= REL * scale, (name, float((a - b).abs().max()), scale)
    seld_env.unset("SELD_CONV_NO_HCQ")
    # oracle on the first job
    desc, x, dy, dws = jobs[0]
    shape, cout, k, pad, dil = specs[0]
    kk = (k,) if isinstance(k, int) else k
    w64 = [torch.zeros(dws[0].shape, dtype=torch.float64, requires_grad=True) for _ in range(8)]
    yr = O.hypercomplex_conv(x.cpu().double(), w64, None, 1, pad, 1, dil, mode="explicit")
    (yr * dy.cpu().double()).sum().backward()
    scale = max(float(w.grad.abs().max()) for w in w64)
    for c, (a, b) in enumerate(zip(first[0], w64)):
        assert float((a.cpu().double() - 0.25 * (c + 1) - b.grad).abs().max()) <= REL * scale
    # second call on fresh slots: bit-identical
    for j in jobs:
        for c, w in enumerate(j[3]):
            w.fill_(0.25 * (c + 1))
    assert H.wgrad_group(jobs)
    torch.cuda.synchronize()
    for j, got in zip(jobs, first):
        for a, b in zip(j[3], got):
            assert torch.equal(a, b), name


def test_wgrad_group_refuses_other_shapes():
    import seld_amd
    H = seld_amd.hip_ops
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(1)
    for shape, cout, k, pad, dil in [((2, 16, 64), 32, 3, 1, 1), ((2, 192, 520), 384, 3, 1, 1), ((2, 192, 512), 384, 3, 70, 70)]:
        desc, x, dy, wshape = _grp_job(H, shape, cout, k, pad, dil, gen, dev)
        dws = [torch.zeros(wshape, device=dev) for _ in range(8)]
        assert H.wgrad_group([(desc, x, dy, dws)]) is False
        assert all(float(w.abs().max()) == 0.0 for w in dws)
