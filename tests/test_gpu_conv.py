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
