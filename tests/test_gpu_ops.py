"""Per-kernel GPU checks of the non-convolution ops against the oracle / torch-CPU fp64 and the fixtures."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import seld_oracle as O
from tests.golden.cases import OP_CASES, op_cotangent, op_inputs
from tests.helpers import pkg

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _close(got, ref, rel=1e-4, what=""):
    got = got.detach().cpu().double().numpy() if torch.is_tensor(got) else np.asarray(got, np.float64)
    ref = ref.detach().cpu().double().numpy() if torch.is_tensor(ref) else np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    scale = max(float(np.abs(ref).max()), 1e-6)
    err = float(np.abs(got - ref).max())
    assert err <= rel * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e}"


LIN_CASES = [c for c in OP_CASES if "lin" in c["kind"]]


@pytest.mark.parametrize("case", LIN_CASES, ids=[c["name"] for c in LIN_CASES])
def test_linear_matches_fixture(case, golden):
    H, L = pkg().hip_ops, pkg()._lib
    g = golden("ops")
    x, ws, bias = op_inputs(case)
    xd = x.to(DEV).requires_grad_(True)
    wd = [w.to(DEV).requires_grad_(True) for w in ws]
    bd = bias.to(DEV).requires_grad_(True) if bias is not None else None
    kind = L.SELD_LIN_DUALQ if case["kind"].startswith("dq") else L.SELD_LIN_QUAT
    y = H.hyper_linear(xd, wd, bd, kind)
    (y * op_cotangent(y.shape).to(DEV)).sum().backward()
    n = case["name"]
    _close(y, g[n + ".y"], what="y")
    _close(xd.grad, g[n + ".dx"], what="dx")
    for i, w in enumerate(wd):
        _close(w.grad, g[f"{n}.dw{i}"], rel=5e-4, what=f"dw{i}")   # signed 4/8-block folds of fp32 sums cancel
    if bd is not None:
        _close(bd.grad, g[n + ".dbias"], what="dbias")


@pytest.mark.parametrize("kind,fin,fout,rows", [("dq", 384, 384, 130), ("dq", 384, 768, 64), ("q", 192, 192, 70),
                                                ("q", 384, 126 * 2 + 4, 33)])
def test_hyper_linear_wide_vs_oracle(kind, fin, fout, rows):
    """The classifier heads of the wide models: feature counts whose component blocks are multiples of 48 take the
    48-deep-tile GEMM (csrc/linear.hip) forward and in the data gradient; against the oracle in fp64."""
    H, L = pkg().hip_ops, pkg()._lib
    A = 8 if kind == "dq" else 4
    gen = torch.Generator().manual_seed(31)
    x = torch.randn(rows, fin, generator=gen)
    ws = [torch.randn(fin // A, fout // A, generator=gen) * 0.1 for _ in range(A)]
    bias = torch.randn(fout, generator=gen)
    xd = x.to(DEV).requires_grad_(True)
    wd = [w.to(DEV).requires_grad_(True) for w in ws]
    bd = bias.to(DEV).requires_grad_(True)
    y = H.hyper_linear(xd, wd, bd, L.SELD_LIN_DUALQ if kind == "dq" else L.SELD_LIN_QUAT)
    cot = torch.randn(rows, fout, generator=gen)
    (y * cot.to(DEV)).sum().backward()
    x64 = x.double().requires_grad_(True)
    w64 = [w.double().requires_grad_(True) for w in ws]
    b64 = bias.double().requires_grad_(True)
    yr = O.dual_quaternion_linear(x64, w64, b64) if kind == "dq" else O.quaternion_linear(x64, *w64, b64)
    (yr * cot.double()).sum().backward()
    _close(y, yr, what="y")
    _close(xd.grad, x64.grad, what="dx")
    for i, (a, b) in enumerate(zip(wd, w64)):
        _close(a.grad, b.grad, rel=5e-4, what=f"dw{i}")
    _close(bd.grad, b64.grad, what="dbias")


def test_real_linear_large():
    hnn = pkg().hip_nn
    torch.manual_seed(0)
    lin = hnn.Linear(384, 126).to(DEV)
    x = torch.randn(3, 70, 384, device=DEV, requires_grad=True)
    y = lin(x)
    cot = torch.randn_like(y)
    (y * cot).sum().backward()
    x64 = x.detach().cpu().double().requires_grad_(True)
    w64 = lin.weight.detach().cpu().double().requires_grad_(True)
    b64 = lin.bias.detach().cpu().double().requires_grad_(True)
    yr = F.linear(x64, w64, b64)
    (yr * cot.cpu().double()).sum().backward()
    _close(y, yr)
    _close(x.grad, x64.grad)
    _close(lin.weight.grad, w64.grad)
    _close(lin.bias.grad, b64.grad)


@pytest.mark.parametrize("shape,act", [((4, 24, 100), 2), ((2, 16, 12, 64), 1), ((3, 8, 77), 0), ((2, 8, 5, 9), 3),
                                       ((32, 12, 512), 2), ((48, 8, 512), 1), ((24, 8, 2048), 2)])
@pytest.mark.parametrize("training,two_pass", [(True, False), (True, True), (False, False)])
def test_bn_act(shape, act, training, two_pass, monkeypatch):
    """Training mode runs the one-workgroup-per-channel backward where a channel fits (N*S <= 32768, S % 4 == 0:
    1, 2, 4 and 8 register groups per thread are all here) and the reduce + apply pair otherwise / when forced."""
    if two_pass:
        monkeypatch.setenv("SELD_BN_TWO_PASS", "1")
    H, hnn = pkg().hip_ops, pkg().hip_nn
    torch.manual_seed(1)
    C = shape[1]
    bn = (hnn.BatchNorm1d if len(shape) == 3 else hnn.BatchNorm2d)(C).to(DEV)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.5, 0.5)
        bn.running_mean.uniform_(-0.2, 0.2)
        bn.running_var.uniform_(0.5, 1.5)
    ref = (torch.nn.BatchNorm1d if len(shape) == 3 else torch.nn.BatchNorm2d)(C).double()
    ref.load_state_dict({k: v.detach().cpu().double() if v.is_floating_point() else v.cpu() for k, v in bn.state_dict().items()})
    bn.train(training)
    ref.train(training)
    x = (torch.randn(shape) * 1.3 + 0.4)
    xd = x.to(DEV).requires_grad_(True)
    y = H.bn_act(xd, bn, act)
    cot = torch.randn(shape)
    (y * cot.to(DEV)).sum().backward()
    x64 = x.double().requires_grad_(True)
    z = ref(x64)
    yr = [z, torch.relu(z), torch.tanh(z), torch.sigmoid(z)][act]
    (yr * cot.double()).sum().backward()
    _close(y, yr, what="y")
    _close(xd.grad, x64.grad, rel=2e-4, what="dx")
    _close(bn.weight.grad, ref.weight.grad, rel=2e-4, what="dgamma")
    _close(bn.bias.grad, ref.bias.grad, rel=2e-4, what="dbeta")
    _close(bn.running_mean, ref.running_mean, what="running_mean")
    _close(bn.running_var, ref.running_var, what="running_var")
    assert int(bn.num_batches_tracked) == int(ref.num_batches_tracked)


@pytest.mark.parametrize("shape,two_pass", [((6, 8, 96), False), ((6, 8, 96), True), ((5, 8, 33), False)])
def test_bn_act_twin_outputs_sum_their_gradients(shape, two_pass, monkeypatch):
    """ResBlock's x_hat has two consumers (model.py:116-132): bn_act(twin=True) hands out two handles and adds the two
    incoming gradients inside its backward kernel.  Also with one handle unused."""
    H, hnn = pkg().hip_ops, pkg().hip_nn
    if two_pass:
        monkeypatch.setenv("SELD_BN_TWO_PASS", "1")
    torch.manual_seed(5)
    C = shape[1]
    bn = hnn.BatchNorm1d(C).to(DEV).train()
    ref = torch.nn.BatchNorm1d(C).double().train()
    x = torch.randn(shape) * 0.7 + 0.2
    c1, c2 = torch.randn(shape), torch.randn(shape)
    for use_second in (True, False):
        bn.zero_grad(); ref.zero_grad()
        xd = x.to(DEV).requires_grad_(True)
        a, b = H.bn_act(xd, bn, 2, twin=True)
        assert a.data_ptr() == b.data_ptr()
        loss = (a * c1.to(DEV)).sum() + ((b * c2.to(DEV)).sum() if use_second else 0.0)
        loss.backward()
        x64 = x.double().requires_grad_(True)
        yr = torch.tanh(ref(x64))
        ((yr * c1.double()).sum() + ((yr * c2.double()).sum() if use_second else 0.0)).backward()
        _close(xd.grad, x64.grad, rel=2e-4, what="dx")
        _close(bn.weight.grad, ref.weight.grad, rel=2e-4, what="dgamma")
        _close(bn.bias.grad, ref.bias.grad, rel=2e-4, what="dbeta")


@pytest.mark.parametrize("training,two_pass,nct", [(True, False, (3, 16, 52)), (True, True, (3, 16, 52)),
                                                   (False, False, (3, 16, 52)), (True, False, (32, 8, 512)),
                                                   (True, False, (12, 8, 512)), (True, False, (40, 8, 512))])
def test_gate(training, two_pass, nct, monkeypatch):
    H, hnn = pkg().hip_ops, pkg().hip_nn
    if two_pass:
        monkeypatch.setenv("SELD_BN_TWO_PASS", "1")
    torch.manual_seed(2)
    N, C, T = nct
    bf, bg = hnn.BatchNorm1d(C).to(DEV), hnn.BatchNorm1d(C).to(DEV)
    for b in (bf, bg):
        with torch.no_grad():
            b.weight.uniform_(0.5, 1.5); b.bias.uniform_(-0.5, 0.5)
            b.running_mean.uniform_(-0.2, 0.2); b.running_var.uniform_(0.5, 1.5)
        b.train(training)
    rf, rg = torch.nn.BatchNorm1d(C).double(), torch.nn.BatchNorm1d(C).double()
    for r, b in ((rf, bf), (rg, bg)):
        r.load_state_dict({k: v.detach().cpu().double() if v.is_floating_point() else v.cpu() for k, v in b.state_dict().items()})
        r.train(training)
    yf, yg = torch.randn(N, C, T), torch.randn(N, C, T)
    mask = (torch.rand(N * C) > 0.5).float() * 2.0
    a, b_ = yf.to(DEV).requires_grad_(True), yg.to(DEV).requires_grad_(True)
    y = H.gate(a, b_, bf, bg, mask.to(DEV))
    cot = torch.randn(N, C, T)
    (y * cot.to(DEV)).sum().backward()
    a64, b64 = yf.double().requires_grad_(True), yg.double().requires_grad_(True)
    yr = torch.tanh(rf(a64)) * torch.sigmoid(rg(b64)) * mask.double().view(N, C, 1)
    (yr * cot.double()).sum().backward()
    _close(y, yr, what="y")
    _close(a.grad, a64.grad, rel=2e-4, what="dyf")
    _close(b_.grad, b64.grad, rel=2e-4, what="dyg")
    _close(bf.weight.grad, rf.weight.grad, rel=2e-4)
    _close(bf.bias.grad, rf.bias.grad, rel=2e-4)
    _close(bg.weight.grad, rg.weight.grad, rel=2e-4)
    _close(bg.bias.grad, rg.bias.grad, rel=2e-4)
    _close(bf.running_var, rf.running_var)
    y2 = H.gate_plain(a.detach(), b_.detach())
    _close(y2, torch.tanh(yf.double()) * torch.sigmoid(yg.double()), what="plain")


@pytest.mark.parametrize("shape,ph,pw", [((2, 6, 16, 40), 8, 1), ((2, 6, 4, 40), 2, 1), ((3, 5, 64), 1, 2), ((2, 3, 9, 10), 2, 3)])
def test_maxpool(shape, ph, pw):
    H = pkg().hip_ops
    torch.manual_seed(3)
    x = torch.randn(shape)
    xd = x.to(DEV).requires_grad_(True)
    y = H.maxpool(xd, ph, pw)
    cot = torch.randn(y.shape)
    (y * cot.to(DEV)).sum().backward()
    x64 = x.double().requires_grad_(True)
    yr = F.max_pool1d(x64, pw) if len(shape) == 3 else F.max_pool2d(x64, (ph, pw))
    (yr * cot.double()).sum().backward()
    _close(y, yr, rel=0)
    _close(xd.grad, x64.grad, rel=0)


def test_mha_module_matches_fixture(golden):
    g = golden("mha")
    M = pkg().model
    E, T, N = 48, 20, 2
    mha = M.MultiHeadAttention(E, 8)
    O.closed_form_fill_(list(mha.state_dict().items()), amp=0.6)
    mha = mha.to(DEV)
    x = O.closed_form_input((N, T, E)).to(DEV).requires_grad_(True)
    y = mha(x, x, x)
    (y * O.closed_form_input(tuple(y.shape)).flip(1).to(DEV)).sum().backward()
    _close(y, g["y"], rel=2e-4, what="y")
    _close(x.grad, g["dx"], rel=5e-4, what="dx")
    _close(mha.queries.weight.grad, g["dwq"], rel=5e-4, what="dwq")
    _close(mha.keys.weight.grad, g["dwk"], rel=5e-4, what="dwk")
    _close(mha.values.weight.grad, g["dwv"], rel=5e-4, what="dwv")
    _close(mha.fc_out.weight.grad, g["dwo"], rel=5e-4, what="dwo")
    _close(mha.fc_out.bias.grad, g["dbo"], rel=5e-4, what="dbo")


@pytest.mark.parametrize("N,H_,hd,T", [(2, 8, 48, 256), (1, 8, 16, 100), (1, 4, 64, 130), (2, 8, 2, 33), (1, 2, 48, 600),
                                       (1, 2, 32, 144), (1, 4, 64, 128), (2, 3, 16, 48), (1, 2, 48, 608)])
def test_mha_core_vs_sdpa(N, H_, hd, T):
    """Flash kernels vs torch's scaled_dot_product_attention in fp64 on the CPU, incl. a spiky row that forces
    the online-softmax rescale across key tiles (guide rule 26).  Head dims 16/32/48/64 with T % 16 == 0 run on the
    fp32-MFMA kernels (mha_mfma.hip), everything else on the VALU kernels (mha.hip)."""
    H = pkg().hip_ops
    gen = torch.Generator().manual_seed(5)
    E = H_ * hd
    q, k, v = (torch.randn(N, E, T, generator=gen) for _ in range(3))
    k[:, :, T // 2 + 7] *= 6.0          # a key far above the rest, in a later tile
    qd, kd, vd = (t.to(DEV).requires_grad_(True) for t in (q, k, v))
    out = H.mha_core(qd, kd, vd, H_)
    cot = torch.randn(N, E, T, generator=gen)
    (out * cot.to(DEV)).sum().backward()

    def heads(t):
        return t.double().view(N, H_, hd, T).permute(0, 1, 3, 2)    # (N, H, T, hd)
    q64, k64, v64 = (t.double().requires_grad_(True) for t in (q, k, v))
    o = F.scaled_dot_product_attention(heads(q64), heads(k64), heads(v64))
    o = o.permute(0, 1, 3, 2).reshape(N, E, T)
    (o * cot.double()).sum().backward()
    _close(out, o, rel=2e-4, what="out")
    _close(qd.grad, q64.grad, rel=5e-4, what="dq")
    _close(kd.grad, k64.grad, rel=5e-4, what="dk")
    _close(vd.grad, v64.grad, rel=5e-4, what="dv")


def test_loss_kernel():
    H = pkg().hip_ops
    gen = torch.Generator().manual_seed(6)
    sed = torch.rand(4, 8, 42, generator=gen).clamp(1e-4, 1 - 1e-4)
    sed[0, 0, 0], sed[0, 0, 1] = 0.0, 1.0        # exercises the log clamp at -100
    doa = torch.rand(4, 8, 126, generator=gen) * 2 - 1
    tgt = torch.cat(((torch.rand(4, 8, 42, generator=gen) < 0.1).float(), torch.rand(4, 8, 126, generator=gen) * 2 - 1), 2)
    a, b = sed.to(DEV).requires_grad_(True), doa.to(DEV).requires_grad_(True)
    loss = H.seld_loss(a, b, tgt.to(DEV), 1.0, 5.0)
    loss.backward()
    s64, d64 = sed.double().requires_grad_(True), doa.double().requires_grad_(True)
    ref = O.seld_loss(s64, d64, tgt.double(), 42, 1.0, 5.0)
    ref.backward()
    assert abs(loss.item() - ref.item()) < 1e-5 * max(1.0, abs(ref.item()))
    gs, gr = a.grad.cpu().double(), s64.grad
    finite = torch.isfinite(gr) & (gr.abs() < 1e6)
    _close(gs[finite], gr[finite], rel=1e-4)
    _close(b.grad, d64.grad, rel=1e-4)


def test_adam_flat_matches_oracle():
    H = pkg().hip_ops
    gen = torch.Generator().manual_seed(7)
    n = 10007
    p = torch.randn(n, generator=gen)
    m = torch.zeros(n)
    v = torch.zeros(n)
    pd, md, vd = p.to(DEV), m.to(DEV), v.to(DEV)
    p64, m64, v64 = p.double(), m.double(), v.double()
    for step in range(1, 4):
        g = torch.randn(n, generator=gen) * (10.0 ** torch.randint(-6, 1, (n,), generator=gen).float())
        H.adam_flat_step(pd, g.to(DEV), md, vd, step, lr=1e-4)
        p64, m64, v64 = O.adam_step(p64, g.double(), m64, v64, step)
    _close(pd, p64, rel=1e-6)
    _close(md, m64, rel=1e-4)
    _close(vd, v64, rel=1e-4)


def test_transpose_and_act():
    H, L = pkg().hip_ops, pkg()._lib
    x = torch.randn(3, 37, 50)
    y = H.transpose12(x.to(DEV))
    assert torch.equal(y.cpu(), x.permute(0, 2, 1).contiguous())
    for act, fn in ((L.SELD_ACT_RELU, torch.relu), (L.SELD_ACT_TANH, torch.tanh), (L.SELD_ACT_SIGMOID, torch.sigmoid)):
        xd = x.to(DEV).requires_grad_(True)
        yy = H.act(xd, act)
        yy.sum().backward()
        x64 = x.double().requires_grad_(True)
        r = fn(x64)
        r.sum().backward()
        _close(yy, r)
        _close(xd.grad, x64.grad)


def test_stft_matches_fixture_and_oracle(golden):
    H = pkg().hip_ops
    g = golden("stft")
    n = np.arange(6400)
    x = np.stack([np.sin(2 * np.pi * (100 + 37 * c) * n / 32000) + 0.1 * np.sin(0.013 * n * (c + 1)) for c in range(8)])
    xd = torch.from_numpy(x).float().to(DEV)
    for key, nov, ph in (("magphase_112", 112, True), ("mag_112", 112, False), ("magphase_128", 128, True)):
        out = H.stft_magphase(xd, 512, nov, ph).cpu().double().numpy()
        ref = g[key]
        assert out.shape == ref.shape
        C = 8
        assert np.abs(out[:C] - ref[:C]).max() < 2e-6 * max(1.0, np.abs(ref[:C]).max()) + 2e-7
        if ph:
            mask = ref[:C] > 1e-4
            dphi = np.angle(np.exp(1j * (out[C:] - ref[C:])))
            assert np.abs(dphi[mask]).max() < 2e-3
    # a longer clip against the closed-form oracle (frames not a multiple of the workgroup's 16)
    rng = np.random.RandomState(0)
    xl = rng.randn(3, 40000)
    out = H.stft_magphase(torch.from_numpy(xl).float().to(DEV), 512, 112, True).cpu().double().numpy()
    ref = O.spectrum_fast(xl.astype(np.float32).astype(np.float64), 512, 112, output_phase=True)
    assert out.shape == ref.shape
    assert np.abs(out[:3] - ref[:3]).max() < 1e-5


def _check_spectrum(out, ref, C, phase, what):
    """Magnitude: 2e-6 of the largest magnitude.  Phase: 1e-3 rad (north-star's fp32 tolerance) wherever the bin
    carries signal, |Z| > 1e-3 * max|Z| -- below that the angle of an fp32 transform is rounding noise."""
    assert out.shape == ref.shape, (what, out.shape, ref.shape)
    top = float(np.abs(ref[:C]).max())
    assert np.abs(out[:C] - ref[:C]).max() < 2e-6 * max(1.0, top) + 2e-7, what
    if phase:
        mask = ref[:C] > 1e-3 * top
        assert mask.mean() > 0.05, what
        dphi = np.angle(np.exp(1j * (out[C:] - ref[C:])))
        assert np.abs(dphi[mask]).max() < 1e-3, (what, float(np.abs(dphi[mask]).max()))


def test_spectrum_fast_drop_in_matches_reference(golden):
    """`utility_functions.spectrum_fast` -- the reference's name, arguments and result layout
    (utility_functions.py:129-155) -- against the reference's own outputs: defaults, magnitude only, and every flag
    off its default (DC bin kept, last frame kept, another segment length)."""
    UF = pkg().utility_functions
    g = golden("stft")
    n = np.arange(6400)
    x = np.stack([np.sin(2 * np.pi * (100 + 37 * c) * n / 32000) + 0.1 * np.sin(0.013 * n * (c + 1)) for c in range(8)])
    out = UF.spectrum_fast(x, nperseg=512, noverlap=112)
    assert isinstance(out, np.ndarray) and out.dtype == np.float64 == np.dtype(str(g["dtype_f64"]))   # float64 in -> float64 out
    # float32 in -> float32 out, as scipy's stft gives the reference (the usual case: librosa loads float32)
    out32 = UF.spectrum_fast(x.astype(np.float32), nperseg=512, noverlap=112)
    assert out32.dtype == np.float32 == np.dtype(str(g["dtype_f32"]))
    _check_spectrum(out32.astype(np.float64), g["magphase_112_f32"].astype(np.float64), 8, True, "float32 input")
    # batched (2, 3, samples): the reference's axis handling, literally (phase on axis -3, cuts on axes 1 and 2)
    xb = np.stack((x[:3], x[3:6]))
    ob = UF.spectrum_fast(xb, nperseg=512, noverlap=112)
    rb = g["batched_112"]
    assert ob.shape == rb.shape == (2, 5, 256, 17)
    for n_ in range(2):        # planes 0-1 are magnitudes of channels 1-2, planes 2-4 the three phases
        _check_spectrum(np.concatenate((ob[n_, :2], ob[n_, 3:5])), np.concatenate((rb[n_, :2], rb[n_, 3:5])), 2, True, "batched")
    om = UF.spectrum_fast(xb, nperseg=512, noverlap=112, cut_dc=False, output_phase=False)
    assert om.shape == g["batched_mag_nodc"].shape
    _check_spectrum(om.reshape(6, 256, 17), g["batched_mag_nodc"].reshape(6, 256, 17), 6, False, "batched magnitude")
    _check_spectrum(out, g["magphase_112"], 8, True, "defaults, noverlap 112")
    _check_spectrum(UF.spectrum_fast(x, 512, 112, output_phase=False), g["mag_112"], 8, False, "magnitude only")
    _check_spectrum(UF.spectrum_fast(x, 512, 128), g["magphase_128"], 8, True, "noverlap 128")
    _check_spectrum(UF.spectrum_fast(x, nperseg=512, noverlap=128, cut_dc=False, output_phase=True,
                                     cut_last_timeframe=False), g["magphase_128_dc_last"], 8, True, "dc + last frame kept")
    _check_spectrum(UF.spectrum_fast(x[:2], nperseg=256, noverlap=56, cut_dc=False, output_phase=False),
                    g["mag_256_56_dc"], 2, False, "nperseg 256, dc kept")
    # tensor in -> float32 device tensor out, same values
    t = UF.spectrum_fast(torch.from_numpy(x).float().to(DEV), 512, 112)
    assert torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32
    assert np.array_equal(t.cpu().numpy().astype(np.float64), out)
    # an explicit window array = scipy's get_window('hamming') reproduces the built-in one
    w = 0.54 - 0.46 * np.cos(2 * np.pi * np.arange(512) / 512)
    _check_spectrum(UF.spectrum_fast(x, 512, 112, window=w), g["magphase_112"], 8, True, "explicit window")
    with pytest.raises(ValueError):
        UF.spectrum_fast(x, 512, 512)
    with pytest.raises(ValueError):
        UF.spectrum_fast(x[0], 512, 112)


def test_spectrum_fast_full_clip():
    """The shape of the reference's smoke entry (model.py:555-562): 8 channels x 60 s at 32 kHz -> (16, 256, 4800),
    checked on a slice of frames against the closed-form oracle, and timed (offline stage; bytes = input + output)."""
    import time
    UF = pkg().utility_functions
    rng = np.random.RandomState(5)
    L_ = 32000 * 60
    x = (rng.randn(8, L_) * (0.2 + np.abs(np.sin(np.arange(L_) * 1e-5)))).astype(np.float32)
    xd = torch.from_numpy(x).to(DEV)
    out = UF.spectrum_fast(xd, nperseg=512, noverlap=112)
    assert tuple(out.shape) == (16, 256, 4800)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        out = UF.spectrum_fast(xd, nperseg=512, noverlap=112)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    nbytes = xd.numel() * 4 + out.numel() * 4
    print(f"spectrum_fast (8, 1920000) -> (16, 256, 4800): {dt * 1e6:.0f} us, {nbytes / dt / 1e9:.0f} GB/s "
          f"({nbytes / dt / 8e12 * 100:.1f} % of 8 TB/s)")
    # frames 1000..1063 and the last 40 (which run into the zero padding at the end of the signal) against the oracle
    # on the matching stretch: a stretch that starts on a hop boundary 2 frames early has its frame j at global frame
    # f0 - 2 + j, and from j = 1 on no frame touches the stretch's own left zero boundary
    got = out.cpu().double().numpy()
    for f0, nf in ((1000, 64), (4760, 40)):
        s0 = 400 * (f0 - 2)
        s1 = min(L_, s0 + 400 * (nf + 4) + 512)
        ref = O.spectrum_fast(x[:, s0:s1].astype(np.float64), 512, 112, output_phase=True, cut_last_timeframe=False)
        _check_spectrum(np.concatenate((got[:8, :, f0:f0 + nf], got[8:, :, f0:f0 + nf])),
                        np.concatenate((ref[:8, :, 2:2 + nf], ref[8:, :, 2:2 + nf])), 8, True, f"frames {f0}..{f0 + nf - 1}")


@pytest.mark.parametrize("algebra,cin,cout,hw,ph,training", [(8, 8, 192, (16, 64), 8, True), (4, 8, 64, (8, 32), 2, True),
                                                             (8, 8, 64, (16, 96), 8, False),
                                                             (8, 16, 192, (16, 64), 8, True),    # K = 144: 64 x 160 tiles
                                                             (4, 8, 64, (16, 64), 8, True),      # quaternion, pooling convolution
                                                             (4, 8, 128, (8, 128), 8, False)])   # ... two tiles, eval statistics
def test_first_stage_fused_backward(algebra, cin, cout, hw, ph, training, monkeypatch):
    """conv -> BatchNorm2d -> ReLU -> MaxPool(ph, 1) on an input that needs no gradient: the fused backward
    (seld_bn_relu_pool_bwd_coef + seld_hc_conv_bwd_weight_bnpool_acc, dy never written) against the unfused path
    (seld_bn_relu_pool_bwd + seld_hc_conv_bwd_weight_acc), and that one against torch autograd on the CPU in fp64."""
    P = pkg()
    H, T = P.hip_ops, P.train
    gen = torch.Generator().manual_seed(21)
    x = torch.randn(3, cin, *hw, generator=gen)
    ws0 = [torch.randn(cout // algebra, cin // algebra, 3, 3, generator=gen) * 0.3 for _ in range(algebra)]
    b0 = torch.randn(cout, generator=gen) * 0.1
    g0, be0 = torch.rand(cout, generator=gen) + 0.5, torch.randn(cout, generator=gen) * 0.2
    # negative and zero BatchNorm weights: the pooling convolution kernel (hcq_first_pool_kernel) picks a window's element by
    # the sign of gamma -- largest conv output, smallest, or (gamma == 0: everything ties) the first row
    g0[1::3] *= -1.0
    g0[5] = 0.0
    cot = torch.randn(3, cout, hw[0] // ph, hw[1], generator=gen)

    def run(fused):
        if fused:
            monkeypatch.delenv("SELD_NO_FUSED_STAGE0", raising=False)
        else:
            monkeypatch.setenv("SELD_NO_FUSED_STAGE0", "1")
        ws = [torch.nn.Parameter(w.clone().to(DEV)) for w in ws0]
        bias = torch.nn.Parameter(b0.clone().to(DEV))
        bn = P.hip_nn.BatchNorm2d(cout).to(DEV)
        with torch.no_grad():
            bn.weight.copy_(g0.to(DEV)); bn.bias.copy_(be0.to(DEV))
        bn.train(training)
        opt = T.FlatAdam(ws + [bias] + list(bn.parameters()), lr=1e-3)
        opt.zero_grad()
        y = H.conv_bn_relu_pool(x.to(DEV), ws, bias, bn, ph, 1, 1, 1, 1)
        (y * cot.to(DEV)).sum().backward()
        torch.cuda.synchronize()
        return y.detach().cpu(), [w.grad.detach().cpu().clone() for w in ws], bias.grad.cpu().clone(), \
            bn.weight.grad.cpu().clone(), bn.bias.grad.cpu().clone(), bn.running_var.cpu().clone()

    got, ref = run(True), run(False)
    _close(got[0], ref[0], rel=1e-6, what="pooled")
    for a, b in zip(got[1], ref[1]):
        _close(a, b, rel=2e-4, what="dw fused vs unfused")
    _close(got[3], ref[3], rel=1e-5, what="dgamma"); _close(got[4], ref[4], rel=1e-5, what="dbeta")
    _close(got[5], ref[5], rel=1e-6, what="running_var")
    # fp64 reference through the oracle's convolution and torch's batch_norm / max_pool2d
    w64 = [w.double().requires_grad_(True) for w in ws0]
    b64, g64, be64 = b0.double().requires_grad_(True), g0.double().requires_grad_(True), be0.double().requires_grad_(True)
    yr = O.hypercomplex_conv(x.double(), w64, b64, 1, 1, 1, 1, mode="explicit")
    rm, rv = torch.zeros(cout, dtype=torch.float64), torch.ones(cout, dtype=torch.float64)
    z = F.batch_norm(yr, rm, rv, g64, be64, training=training, momentum=0.1, eps=1e-5)
    pr = F.max_pool2d(F.relu(z), (ph, 1))
    (pr * cot.double()).sum().backward()
    _close(got[0], pr, rel=2e-5, what="pooled vs fp64")
    for a, b in zip(got[1], w64):
        _close(a, b.grad, rel=5e-4, what="dw vs fp64")
    _close(got[3], g64.grad, rel=5e-4, what="dgamma vs fp64")
    _close(got[4], be64.grad, rel=5e-4, what="dbeta vs fp64")
    scale = max(float(w64[0].grad.abs().max()), 1e-6)
    assert float((got[2].double() - b64.grad).abs().max()) < 1e-3 * scale * cot.numel() ** 0.5     # ~0 under batch statistics


@pytest.mark.parametrize("cin,hw", [(8, (16, 64)), (16, (8, 128)), (8, (16, 96))])
def test_first_stage_dropout_in_the_pooled_pass(cin, hw):
    """The stage's Dropout fused into the first stage (seld_bn_pool_finish on the pooling-convolution path, a dropout
    launch inside the function otherwise) against conv_bn_relu_pool followed by a separate dropout at the same Philox
    offset: same mask, same outputs, same gradients."""
    P = pkg()
    H, T = P.hip_ops, P.train
    cout, ph = 192, 8
    gen = torch.Generator().manual_seed(23)
    x = torch.randn(2, cin, *hw, generator=gen)
    ws0 = [torch.randn(cout // 8, cin // 8, 3, 3, generator=gen) * 0.3 for _ in range(8)]
    g0 = torch.rand(cout, generator=gen) - 0.4
    g0[7] = 0.0            # xhat of this channel cannot come back from the stage's output: the raw window value is kept for it
    cot = torch.randn(2, cout, hw[0] // ph, hw[1], generator=gen)

    def run(fused_dropout):
        ws = [torch.nn.Parameter(w.clone().to(DEV)) for w in ws0]
        bn = P.hip_nn.BatchNorm2d(cout).to(DEV).train()
        with torch.no_grad():
            bn.weight.copy_(g0.to(DEV))
        opt = T.FlatAdam(ws + list(bn.parameters()), lr=1e-3)
        opt.zero_grad()
        H.philox.set_offset(1000)
        if fused_dropout:
            y = H.conv_bn_relu_pool(x.to(DEV), ws, None, bn, ph, 1, 1, 1, 1, drop_p=0.3)
        else:
            y = H.dropout(H.conv_bn_relu_pool(x.to(DEV), ws, None, bn, ph, 1, 1, 1, 1), 0.3, True)
        (y * cot.to(DEV)).sum().backward()
        torch.cuda.synchronize()
        return y.detach().cpu(), [w.grad.detach().cpu().clone() for w in ws], bn.weight.grad.cpu().clone(), H.philox.offset

    got, ref = run(True), run(False)
    assert got[3] == ref[3]                                   # the same number of draws
    zero = (ref[0] == 0)
    assert 0.2 < float(zero.float().mean()) < 0.9             # dropout really happened (ReLU zeros on top of p = 0.3)
    assert torch.equal(got[0] == 0, zero)
    _close(got[0], ref[0], rel=1e-6, what="dropped output")
    for a, b in zip(got[1], ref[1]):
        _close(a, b, rel=2e-4, what="dw")
    _close(got[2], ref[2], rel=1e-5, what="dgamma")


# ------------------------------------------------------------------------------------------
# dataset normalisation (train.py:242-408)
# ------------------------------------------------------------------------------------------
def _norm_args(mode, n_mics, domain, phase):
    import types
    return types.SimpleNamespace(dataset_normalization=mode, n_mics=n_mics, domain=domain, phase=phase)


def test_dataset_normalisation_matches_reference_fixture(golden):
    """The three predictor arrays normalised on the device by train.normalize_dataset against what the reference's
    train.main produced for the same inputs (norm.npz).  The unit norm repeats the reference's float32 operations
    one by one: bit-exact for float32 sources.  Standardisation: the moments are accumulated in double instead of
    numpy's float32 pairwise sums, tolerance 1e-5 of the array's scale.  float64 sources (normalised in float64 by
    the reference, cast afterwards; here cast first): 1e-5 as well."""
    from tests.golden.cases import NORM_CASES, norm_input
    T = pkg().train
    g = golden("norm")
    for name, shape, dtype, mode, n_mics, domain, phase in NORM_CASES:
        dev = [torch.from_numpy(norm_input(shape, dtype, k)).float().to(DEV) for k in range(3)]
        T.normalize_dataset(_norm_args(mode, n_mics, domain, phase), *dev)
        for k in range(3):
            got, ref = dev[k].cpu().numpy(), g[f"{name}.{k}"]
            if mode != "True" and dtype == np.float32:
                # every operation of the kernel is correctly rounded: bit-exact against the IEEE evaluation of the
                # reference's expression tree; against the reference itself the q channels are within 1 ulp (its
                # torch-CPU sqrt is not correctly rounded, see oracle.dq_unit_norm_ieee), the rest identical
                unit = mode in ("UnitNorm", "DQ_Normalization") and domain in ("DQ", "dq")
                ieee = O.dq_unit_norm_ieee(norm_input(shape, dtype, k)) if unit else ref
                assert np.array_equal(got, ieee), (name, k, np.abs(got - ieee).max())
                assert np.array_equal(got[:, 4:], ref[:, 4:])
                assert np.abs(got[:, :4] - ref[:, :4]).max() <= 2 * np.spacing(np.float32(1.0))
            else:
                _close(got, ref, rel=1e-5, what=f"{name}.{k}")
    with pytest.raises(ValueError):
        T.normalize_dataset(_norm_args("UnitNorm", 2, "DQ", True), torch.zeros(2, 16, 4, 4, device=DEV))


def test_dq_unit_norm_edge_cases():
    """Zero q -> NaN exactly where the reference's expressions give NaN; empty array; scalar path (hw % 4 != 0);
    misaligned view rejected; CPU tensors rejected."""
    H, L = pkg().hip_ops, pkg()._lib
    g = torch.Generator().manual_seed(3)
    x = torch.randn(3, 8, 5, 7, generator=g)
    x[1, :4, 2, 3] = 0.0                       # |q| = 0
    x[2, :, 0, 0] = 0.0
    ref = O.dq_unit_norm(x)
    ieee = O.dq_unit_norm_ieee(x.numpy())
    got = H.dq_unit_norm_(x.to(DEV)).cpu()
    assert torch.equal(torch.isnan(got), torch.isnan(ref)) and torch.isnan(ref).any()
    assert np.array_equal(got.numpy(), ieee, equal_nan=True)
    assert np.nanmax(np.abs(got.numpy() - ref.numpy()) / np.maximum(np.abs(ref.numpy()), 1e-30)) <= 2.4e-7   # 1 ulp of sqrt -> up to 2 ulp of q
    H.dq_unit_norm_(torch.zeros(0, 8, 4, 4, device=DEV))
    with pytest.raises(L.SeldHipError):
        H.dq_unit_norm_(torch.zeros(2, 4, 4, 4, device=DEV))
    with pytest.raises(L.SeldHipError):
        H.dq_unit_norm_(torch.zeros(2, 8, 4, 4))
    with pytest.raises(L.SeldHipError):
        H.dq_unit_norm_(torch.zeros(2, 8, 4, 8, device=DEV)[..., ::2])
    with pytest.raises(L.SeldHipError):
        H.group_standardize_(torch.zeros(0, 8, 4, 4, device=DEV), 0, 8)


def test_dataset_normalisation_full_size_properties():
    """48 full clips (8, 256, 4800) (spectrum_fast output of a 60 s recording): 4.7e8 values, too many
    for the oracle in the default suite; check the properties the operation defines instead:
    |q| = 1 and q.p = 0 after the unit norm (idempotence follows), mean 0 / std 1 after standardisation, and the
    returned (mean, std) against torch's float64 reduction."""
    H = pkg().hip_ops
    g = torch.Generator(device=DEV).manual_seed(11)
    x = torch.rand(48, 8, 256, 4800, device=DEV, generator=g) + 0.05
    y = x.clone()
    H.dq_unit_norm_(y)
    q, p = y[:, :4], y[:, 4:]
    assert float(((q * q).sum(1) - 1).abs().max()) < 1e-6
    assert float((q * p).sum(1).abs().max()) < 2e-6
    z = y.clone()
    H.dq_unit_norm_(z)
    assert float((z - y).abs().max()) < 2e-6
    w = x.clone()
    ms = H.group_standardize_(w, 0, 4).cpu()
    ref_m, ref_s = float(x[:, :4].double().mean()), float(x[:, :4].double().std(unbiased=False))
    assert abs(float(ms[0]) - ref_m) < 1e-6 and abs(float(ms[1]) - ref_s) < 1e-6
    assert abs(float(w[:, :4].double().mean())) < 1e-5 and abs(float(w[:, :4].double().std(unbiased=False)) - 1) < 1e-5
    assert torch.equal(w[:, 4:], x[:, 4:])
    # the reference's host expressions on a bounded slice of the same array, timed beside the device passes
    import time
    host = x[:4].cpu()
    t0 = time.perf_counter()
    ref = O.dq_unit_norm(host)
    t_unit = time.perf_counter() - t0
    assert np.array_equal(O.dq_unit_norm_ieee(host.numpy()), y[:4].cpu().numpy())     # bit-exact at full clip size too
    assert float((ref - y[:4].cpu()).abs().max()) <= 2.4e-7
    t0 = time.perf_counter()
    O.group_standardize(host.numpy(), 0, 8)
    t_std = time.perf_counter() - t0
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    ev[0].record()
    H.dq_unit_norm_(y)
    ev[1].record()
    H.group_standardize_(w, 0, 8)
    ev[2].record()
    torch.cuda.synchronize()
    print(f"N1 48 clips: unit norm {ev[0].elapsed_time(ev[1]):.3f} ms, standardise {ev[1].elapsed_time(ev[2]):.3f} ms on the "
          f"device; host expressions {t_unit * 12e3:.0f} ms / {t_std * 12e3:.0f} ms (scaled from 4 clips, "
          f"{torch.get_num_threads()} threads)")


def test_first_stage_full_size_without_conv_output(monkeypatch):
    """The benchmark's first stage, (32, 8, 128, 512) -> 192 channels, training mode with Dropout: the path that never
    writes the convolution output (csrc/first_stage.hip: statistics from the input's second moments, window value + row
    from the pooling convolution, backward from the pooled-size tensors) against the path that writes y
    (SELD_FIRST_STAGE_STORE_Y=1) -- outputs, running statistics, BatchNorm and convolution weight gradients -- and samples
    0 and 31 of the output against the fp64 oracle with the batch statistics of the no-output path.  All 32-bit indexing of
    the full-size tensors is in play here (1.6 GB of y on the comparison path).  Also: two runs of the no-output path give
    bit-identical gradients (no atomics)."""
    P = pkg()
    H, T = P.hip_ops, P.train
    gen = torch.Generator().manual_seed(29)
    N, cout, hw, ph = 32, 192, (128, 512), 8
    x = torch.randn(N, 8, *hw, generator=gen)
    ws0 = [torch.randn(cout // 8, 1, 3, 3, generator=gen) * 0.3 for _ in range(8)]
    g0, be0 = torch.rand(cout, generator=gen) + 0.5, torch.randn(cout, generator=gen) * 0.2
    g0[1::5] *= -1.0
    cot = torch.randn(N, cout, hw[0] // ph, hw[1], generator=gen)
    xd, cotd = x.to(DEV), cot.to(DEV)

    def run(store_y):
        if store_y:
            monkeypatch.setenv("SELD_FIRST_STAGE_STORE_Y", "1")
        else:
            monkeypatch.delenv("SELD_FIRST_STAGE_STORE_Y", raising=False)
        ws = [torch.nn.Parameter(w.clone().to(DEV)) for w in ws0]
        bn = P.hip_nn.BatchNorm2d(cout).to(DEV).train()
        with torch.no_grad():
            bn.weight.copy_(g0.to(DEV)); bn.bias.copy_(be0.to(DEV))
        opt = T.FlatAdam(ws + list(bn.parameters()), lr=1e-3)
        opt.zero_grad()
        H.philox.set_offset(4000)
        y = H.conv_bn_relu_pool(xd, ws, None, bn, ph, 1, 1, 1, 1, drop_p=0.3)
        (y * cotd).sum().backward()
        torch.cuda.synchronize()
        return (y.detach(), [w.grad.detach().clone() for w in ws], bn.weight.grad.clone(), bn.bias.grad.clone(),
                bn.running_mean.clone(), bn.running_var.clone())

    new, new2, old = run(False), run(False), run(True)
    for a, b in zip(new[1] + [new[2], new[3]], new2[1] + [new2[2], new2[3]]):
        assert torch.equal(a, b), "the no-output path is not reproducible"
    assert torch.equal(new[0] == 0, old[0] == 0) or float(((new[0] == 0) != (old[0] == 0)).float().mean()) < 1e-5
    _close(new[0].cpu(), old[0].cpu(), rel=1e-5, what="stage output")
    _close(new[4].cpu(), old[4].cpu(), rel=1e-5, what="running_mean")
    assert torch.allclose(new[5], old[5], rtol=1e-5), "running_var"
    for a, b in zip(new[1], old[1]):
        _close(a.cpu(), b.cpu(), rel=3e-4, what="conv weight gradient")
    _close(new[2].cpu(), old[2].cpu(), rel=1e-4, what="dgamma")
    _close(new[3].cpu(), old[3].cpu(), rel=1e-4, what="dbeta")
    # fp64 oracle on samples 0 and 31: y = conv(x), z = relu(a y + b) with the batch statistics implied by the running buffers
    mom = 0.1
    mean = (new[4].cpu().double() - 0.0 * (1 - mom)) / mom                     # running_mean started at 0
    cnt = N * hw[0] * hw[1]
    var = ((new[5].cpu().double() - (1 - mom)) / mom) * (cnt - 1) / cnt        # running_var started at 1
    a = g0.double() / torch.sqrt(var + 1e-5)
    b = be0.double() - mean * a
    for n in (0, 31):
        yr = O.hypercomplex_conv(x[n:n + 1].double(), [w.double() for w in ws0], None, 1, 1, 1, 1, mode="explicit")
        zr = F.max_pool2d(F.relu(yr * a.view(1, -1, 1, 1) + b.view(1, -1, 1, 1)), (ph, 1))
        got = new[0][n:n + 1].cpu().double()
        kept = got != 0
        assert 0.3 < float(kept.float().mean()) < 0.8
        err = ((got * 0.7 - zr) * kept).abs().max()                            # kept elements carry z / (1 - p)
        assert float(err) <= 2e-4 * float(zr.abs().max()), (n, float(err))


# ------------------------------------------------------------------------------------------
# round-3 launch diet: linear backward without zero-fills / atomics, loss without a pre-zeroed scalar, the later CNN
# stages' Dropout inside the BN+ReLU+MaxPool passes
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kind,fin,fout,rows", [(1, 384, 42, 5), (1, 70, 126, 130), (1, 384, 384, 2049), (8, 384, 384, 16384),
                                                (4, 64, 128, 1000)])
def test_linear_backward_row_splits(kind, fin, fout, rows):
    """seld_hc_linear_bwd: x^T dy and the column sums of dy in up to 8 row splits, plain stores + an ordered fold
    (csrc/linear.hip) -- against fp64 on row counts below one split, ragged, and far beyond 8 x 128; run twice on a dirty
    workspace: the same bits (nothing depends on the workspace's previous contents, nothing is accumulated atomically)."""
    P = pkg()
    H, L = P.hip_ops, P._lib
    import ctypes
    gen = torch.Generator().manual_seed(17)
    x = torch.randn(rows, fin, generator=gen)
    dy = torch.randn(rows, fout, generator=gen)
    if kind == 1:
        ws = [torch.randn(fout, fin, generator=gen) * 0.1]
    else:
        ws = [torch.randn(fin // kind, fout // kind, generator=gen) * 0.1 for _ in range(kind)]
    xd, dyd, wd = x.to(DEV), dy.to(DEV), [w.to(DEV) for w in ws]
    lib = L.lib()
    lib.seld_hc_linear_bwd_workspace.restype = ctypes.c_size_t
    nbytes = lib.seld_hc_linear_bwd_workspace(kind, fin, fout)

    def run(fill, with_w=True):
        wsb = torch.full(((nbytes + 3) // 4,), fill, device=DEV)
        dws = [torch.full_like(w, 7.0) for w in wd]
        db = torch.full((fout,), 7.0, device=DEV)
        L.check(lib.seld_hc_linear_bwd(kind, rows, fin, fout, L.ptr(xd), L.ptr(dyd), L.ptr_array8(wd), None,
                                       L.ptr_array8(dws) if with_w else None, L.ptr(db), L.ptr(wsb), ctypes.c_size_t(nbytes),
                                       L.current_stream()), "seld_hc_linear_bwd")
        torch.cuda.synchronize()
        return dws, db

    (dw1, db1), (dw2, db2) = run(0.0), run(float("nan"))
    for a, b in zip(dw1, dw2):
        assert torch.equal(a, b)
    assert torch.equal(db1, db2)
    _, db3 = run(3.0, with_w=False)                       # bias gradient alone
    assert torch.equal(db3, db1)
    x64 = x.double()
    w64 = [w.double().requires_grad_(True) for w in ws]
    b64 = torch.zeros(fout, dtype=torch.float64, requires_grad=True)
    if kind == 1:
        yr = F.linear(x64, w64[0], b64)
    else:
        yr = O.dual_quaternion_linear(x64, w64, b64) if kind == 8 else O.quaternion_linear(x64, *w64, b64)
    (yr * dy.double()).sum().backward()
    for i, (a, b) in enumerate(zip(dw1, w64)):
        _close(a, b.grad, rel=1e-4, what=f"dw{i}")
    _close(db1, b64.grad, rel=1e-4, what="dbias")


def test_loss_needs_no_zeroed_scalar_and_repeats_bit_for_bit():
    """seld_loss_fwd_bwd WRITES the loss (ticketed ordered reduction, csrc/nn_ops.hip): a poisoned output buffer, many
    evaluations back to back, sizes from one workgroup to the 256-workgroup cap."""
    P = pkg()
    L = P._lib
    import ctypes
    lib = L.lib()
    for rows in (3, 700, 40000):
        gen = torch.Generator().manual_seed(rows)
        sed = torch.rand(rows, 42, generator=gen).clamp(1e-4, 1 - 1e-4)
        doa = torch.rand(rows, 126, generator=gen) * 2 - 1
        tgt = torch.cat(((torch.rand(rows, 42, generator=gen) < 0.1).float(), torch.rand(rows, 126, generator=gen) * 2 - 1), 1)
        a, b, t = sed.to(DEV), doa.to(DEV), tgt.to(DEV)
        outs = []
        for k in range(6):
            loss = torch.full((1,), float("nan") if k % 2 else 1e9, device=DEV)
            L.check(lib.seld_loss_fwd_bwd(L.ptr(a), L.ptr(b), L.ptr(t), ctypes.c_int64(rows), 42, 126, ctypes.c_float(1.0),
                                          ctypes.c_float(5.0), L.ptr(loss), None, None, L.current_stream()), "seld_loss_fwd_bwd")
            outs.append(loss)
        torch.cuda.synchronize()
        vals = [float(o.item()) for o in outs]
        assert all(v == vals[0] for v in vals), vals
        ref = O.seld_loss(sed.double()[None], doa.double()[None], tgt.double()[None], 42, 1.0, 5.0)
        assert abs(vals[0] - ref.item()) < 2e-5 * max(1.0, abs(ref.item()))


@pytest.mark.parametrize("shape,ph", [((2, 192, 16, 64), 8), ((3, 16, 4, 40), 2), ((2, 8, 12, 32), 3)])
def test_bn_relu_pool_with_the_stage_dropout_inside(shape, ph):
    """bn_relu_pool(..., drop_p) (seld_bn_relu_pool_fwd_drop / _bwd_drop: CNN stages 2 and 3, model.py:278-282) against
    bn_relu_pool followed by a separate dropout at the same Philox offset: same mask, same output, same gradients, the
    same number of draws."""
    P = pkg()
    H = P.hip_ops
    gen = torch.Generator().manual_seed(41)
    y0 = torch.randn(*shape, generator=gen)
    C = shape[1]
    g0, b0 = torch.rand(C, generator=gen) + 0.5, torch.randn(C, generator=gen) * 0.1
    cot = torch.randn(shape[0], C, shape[2] // ph, shape[3], generator=gen)

    def run(inside):
        bn = P.hip_nn.BatchNorm2d(C).to(DEV).train()
        with torch.no_grad():
            bn.weight.copy_(g0.to(DEV)); bn.bias.copy_(b0.to(DEV))
        y = y0.to(DEV).requires_grad_(True)
        H.philox.set_offset(5000)
        if inside:
            out = H.bn_relu_pool(y, bn, ph, 1, None, 0.3)
        else:
            out = H.dropout(H.bn_relu_pool(y, bn, ph, 1, None), 0.3, True)
        (out * cot.to(DEV)).sum().backward()
        torch.cuda.synchronize()
        return out.detach().cpu(), y.grad.cpu(), bn.weight.grad.cpu(), bn.bias.grad.cpu(), H.philox.offset

    got, ref = run(True), run(False)
    assert got[4] == ref[4]
    zero = ref[0] == 0
    assert 0.3 < float(zero.float().mean()) < 0.95
    assert torch.equal(got[0], ref[0])
    _close(got[1], ref[1], rel=1e-5, what="dy")
    _close(got[2], ref[2], rel=1e-4, what="dgamma")
    _close(got[3], ref[3], rel=1e-4, what="dbeta")
    # eval mode: no dropout, nothing drawn
    bn = P.hip_nn.BatchNorm2d(C).to(DEV).eval()
    H.philox.set_offset(0)
    out = H.bn_relu_pool(y0.to(DEV), bn, ph, 1, None, 0.3)
    assert H.philox.offset == 0 and out.shape == cot.shape


def test_attention_with_stacked_projections_matches_the_separate_path():
    """MultiHeadAttention.forward_nct with FlatAdam's parameter layout: values / keys / queries are consecutive blocks of
    the flat buffer, so the three 1x1 projections run as ONE convolution (hip_ops.stacked_conv_weight), the attention core
    reads the packed (N, 3E, T) tensor (seld_mha_fwd_packed / _bwd_packed) and the three weight gradients are one launch
    into the adjacent gradient slots.  Against the module WITHOUT the layout (separate projections, model.py:28-48) and
    against torch scaled_dot_product_attention in fp64."""
    P = pkg()
    M, T_, H = P.model, P.train, P.hip_ops
    E, heads, T, N = 384, 8, 256, 2
    torch.manual_seed(12)
    ref = M.MultiHeadAttention(E, heads).to(DEV)
    fused = M.MultiHeadAttention(E, heads).to(DEV)
    fused.load_state_dict(ref.state_dict())
    opt = T_.FlatAdam(fused.parameters(), lr=1e-3)          # re-homes the parameters: now adjacent, with gradient slots
    opt.zero_grad()
    assert H.stacked_conv_weight((fused.values.weight, fused.keys.weight, fused.queries.weight)) is not None
    assert H.stacked_conv_weight((ref.values.weight, ref.keys.weight, ref.queries.weight)) is None       # separate allocations
    x0 = torch.randn(N, E, T, device=DEV) * 0.5
    cot = torch.randn(N, E, T, device=DEV)
    xa = x0.clone().requires_grad_(True)
    ya = fused.forward_nct(xa)
    (ya * cot).sum().backward()
    H.join_side_stream()
    xb = x0.clone().requires_grad_(True)
    yb = ref._attend(xb, xb, xb)
    (yb * cot).sum().backward()
    H.join_side_stream()
    torch.cuda.synchronize()
    _close(ya, yb, rel=1e-5, what="y")
    _close(xa.grad, xb.grad, rel=1e-4, what="dx")
    for name in ("values", "keys", "queries"):
        _close(getattr(fused, name).weight.grad, getattr(ref, name).weight.grad, rel=1e-4, what="dw " + name)
    _close(fused.fc_out.weight.grad, ref.fc_out.weight.grad, rel=1e-4, what="dwo")
    _close(fused.fc_out.bias.grad, ref.fc_out.bias.grad, rel=1e-4, what="dbo")
    # fp64
    sd = {k: v.detach().cpu().double() for k, v in ref.state_dict().items()}
    x64 = x0.cpu().double().requires_grad_(True)
    proj = lambda w: torch.einsum("oi,nit->not", w.squeeze(-1), x64)
    q, k, v = proj(sd["queries.weight"]), proj(sd["keys.weight"]), proj(sd["values.weight"])
    hd = E // heads
    sp = lambda t: t.reshape(N, heads, hd, T).transpose(2, 3)
    o = F.scaled_dot_product_attention(sp(q), sp(k), sp(v)).transpose(2, 3).reshape(N, E, T)
    y64 = torch.einsum("oi,nit->not", sd["fc_out.weight"], o) + sd["fc_out.bias"][None, :, None]
    (y64 * cot.cpu().double()).sum().backward()
    _close(ya, y64, rel=2e-4, what="y vs fp64")
    _close(xa.grad, x64.grad, rel=5e-4, what="dx vs fp64")
    # without gradients (eval / no_grad): the stacked path needs the parameter layout only
    with torch.no_grad():
        yc = fused.forward_nct(x0)
    _close(yc, yb, rel=1e-5, what="no_grad")
