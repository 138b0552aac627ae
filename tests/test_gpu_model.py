"""Whole-model parity of the HIP path (through the C ABI) against the reference fixtures and the oracle.
Tolerance: north_star asks 1e-3 absolute on (sed, doa); we also require 1e-3 * max|ref| on every
intermediate tap and on the pre-activation logits (SURVEY App. C caveat)."""
import numpy as np
import pytest
import torch

from oracle import seld_oracle as O
from tests.golden.cases import MODEL_CASES, train_target
from tests.helpers import build_model, fill_weights, pkg

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _np(t):
    return t.detach().cpu().double().numpy()


def _close(got, ref, rel=1e-3, what="", floor=1e-6):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    scale = max(float(np.abs(ref).max()), floor)
    err = float(np.abs(got - ref).max())
    assert err <= rel * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e}"


def _prepared(case):
    m = build_model(case)
    fill_weights(m.state_dict().items(), case)
    return m.to(DEV)


@pytest.mark.parametrize("case", MODEL_CASES, ids=[c["name"] for c in MODEL_CASES])
def test_eval_forward_matches_reference(case, golden):
    g = golden("model_" + case["name"])
    m = _prepared(case).eval()
    x = O.closed_form_input((case["B"], case["input_channels"], case["freq_dim"], case["time_dim"])).to(DEV)
    taps = {}
    hooks = []
    M = pkg().model

    def mk(nm):
        def hook(mod, inp, out):
            taps[nm] = out
        return hook
    for nm, mod in m.named_modules():
        if nm.endswith("tcn.conv1") or (".cnn." in nm and nm.count(".") == 2):
            hooks.append(mod.register_forward_hook(mk(nm)))
    hooks.append(m.sed[-2].register_forward_hook(mk("sed_logits")))
    hooks.append(m.doa[-2].register_forward_hook(mk("doa_logits")))
    with torch.no_grad():
        sed, doa = m(x)
    torch.cuda.synchronize()
    assert np.abs(_np(sed) - g["sed"]).max() < 1e-3
    assert np.abs(_np(doa) - g["doa"]).max() < 1e-3
    for k in g:
        if k.startswith("tap.") and k[4:] in taps:
            _close(_np(taps[k[4:]]), g[k], what=k)


@pytest.mark.parametrize("case", [c for c in MODEL_CASES if c.get("train") and c.get("taps", True)], ids=lambda c: c["name"])
def test_resblock_taps_match_reference(case, golden):
    """Per-block (residual, skip) outputs through the public ResBlock.forward."""
    g = golden("model_" + case["name"])
    m = _prepared(case).eval()
    prefixes = ["branch_A", "branch_B"] if "branch_A.cnn.0" in "".join(g.keys()) else ["seld_block"]
    for pre in prefixes:
        blk = getattr(m, pre)
        cnn_out = torch.from_numpy(g[f"tap.{pre}.cnn.2"]).to(DEV)
        B, C, Fp, T = cnn_out.shape
        x = cnn_out.reshape(B, C * Fp, T).contiguous()
        with torch.no_grad():
            for i, rb in enumerate(blk.tcn.ResBlocks):
                x, skip = rb(x)
                _close(_np(x), g[f"tap.{pre}.tcn.ResBlocks.{i}.residual"], what=f"res{i}")
                _close(_np(skip), g[f"tap.{pre}.tcn.ResBlocks.{i}.skip"], what=f"skip{i}")
        att_in = torch.from_numpy(g[f"tap.{pre}.tcn.conv1"]).to(DEV)       # (N, E, T)
        with torch.no_grad():
            a = blk.tcn.attention(att_in.permute(0, 2, 1).contiguous(), att_in.permute(0, 2, 1).contiguous(),
                                  att_in.permute(0, 2, 1).contiguous())
        _close(_np(a), g[f"tap.{pre}.tcn.attention"], what="attention")


# The wide cases run twice: with the tile the library picks for the (small) test shape, and with the 192-channel x
# 64-position tile forced -- the kernel instantiations the benchmark shapes run (hc_conv_vec_kernel<12, ...>).
_TRAIN_RUNS = [(c, None) for c in MODEL_CASES if c.get("train")] + \
              [(c, "12,1") for c in MODEL_CASES if c.get("train") and not c.get("taps", True)]


@pytest.mark.parametrize("case,cfg", _TRAIN_RUNS, ids=[c["name"] + ("" if f is None else "-tile" + f) for c, f in _TRAIN_RUNS])
def test_train_step_matches_reference(case, cfg, golden, seld_env):
    """forward (batch statistics) -> loss -> backward -> Adam: loss, every gradient checksum, selected full
    gradients, parameter deltas and BatchNorm running statistics against the reference's step."""
    if cfg is not None:
        seld_env.set("SELD_CONV_CFG", cfg)
    g = golden("model_" + case["name"])
    T = pkg().train
    m = _prepared(case).train()
    opt = T.FlatAdam(m.parameters(), lr=1e-4)
    x = O.closed_form_input((case["B"], case["input_channels"], case["freq_dim"], case["time_dim"])).to(DEV)
    target = train_target(case).to(DEV)
    n_sed = int(case["output_classes"] * 3)
    opt.zero_grad()
    sed, doa = m(x)
    loss = T.seld_loss_fn(sed, doa, target, n_sed, 1.0, 5.0)
    loss.backward()
    torch.cuda.synchronize()
    assert np.abs(_np(sed) - g["train.sed"]).max() < 1e-3
    assert np.abs(_np(doa) - g["train.doa"]).max() < 1e-3
    assert abs(loss.item() - float(g["train.loss"][0])) <= 1e-4 * max(1.0, abs(float(g["train.loss"][0])))
    names = str(g["train.param_names"]).split("\n")
    params = dict(m.named_parameters())
    assert list(params.keys()) == names
    cks = g["train.grad_checksums"]
    before = {n: p.detach().clone() for n, p in params.items()}
    numel = np.array([params[n].numel() for n in names], dtype=np.float64)
    ref_rms = np.sqrt(np.nan_to_num(cks[:, 1]) / numel)
    floor = 1e-4 * ref_rms.max()          # gradients below 1e-4 of the largest RMS are fp32 noise in both stacks
    gtol = case.get("grad_tol", 1e-3) / 1e-3      # config-width cases: see tests/golden/cases.py
    for i, n in enumerate(names):
        gr = params[n].grad
        if np.isnan(cks[i, 0]):
            assert gr is None or float(gr.abs().max()) == 0.0, n
            continue
        got = np.array([gr.double().sum().item(), (gr.double() ** 2).sum().item()])
        rms = max(ref_rms[i], floor)
        assert abs(got[1] - cks[i, 1]) <= gtol * 4e-3 * rms * rms * numel[i], (n, got, cks[i])
        assert abs(got[0] - cks[i, 0]) <= gtol * 2e-3 * rms * numel[i], (n, got, cks[i])
    for k in g:
        if k.startswith("train.grad."):
            _close(_np(params[k[len("train.grad."):]].grad), g[k], rel=gtol * 1e-3, what=k, floor=10 * floor)   # exact zeros of the reference (conv bias under BatchNorm) are fp32 cancellation noise here
    opt.step()
    torch.cuda.synchronize()
    dck = g["train.delta_checksums"]
    for i, n in enumerate(names):
        if ref_rms[i] < 10 * floor:
            continue      # gradient is (numerically) zero in the fp64 reference: Adam turns fp32 noise into +-lr steps
        d = (params[n].detach() - before[n]).double()
        got = np.array([d.sum().item(), (d ** 2).sum().item()])
        # Adam's first step is lr * g / (|g| + 1e-8): elements whose gradient is at the fp32 noise level (|g| ~ 1e-8)
        # move by up to lr in either stack, so allow three such elements on top of the relative tolerance;
        # the kernel itself is checked exactly in test_gpu_ops.py::test_adam_flat_matches_oracle
        assert abs(got[1] - dck[i, 1]) <= gtol * 5e-3 * dck[i, 1] + 3 * (1e-4) ** 2, (n, got, dck[i])
    sd = m.state_dict()
    rnames = str(g["train.running_names"]).split("\n")
    rck = g["train.running_checksums"]
    for i, n in enumerate(rnames):
        v = sd[n].double()
        got = np.array([v.sum().item(), (v ** 2).sum().item()])
        # the sum of a running mean cancels: absolute tolerance from the vector's norm
        assert np.allclose(got, rck[i], rtol=1e-4, atol=1e-6 + 1e-4 * np.sqrt(rck[i][1])), (n, got, rck[i])


def test_dropout_statistics_and_reuse():
    """Train-mode dropout cannot match the CPU RNG; check keep-rate, scaling and that backward reuses the mask."""
    H = pkg().hip_ops
    torch.manual_seed(3)
    x = torch.ones(1 << 20, device=DEV, requires_grad=True)
    y = H.dropout(x, 0.3, True)
    y.sum().backward()
    keep = (y > 0).float().mean().item()
    assert abs(keep - 0.7) < 5e-3
    assert torch.allclose(y[y > 0], torch.full_like(y[y > 0], 1 / 0.7))
    assert torch.equal(x.grad > 0, y > 0)
    mask = H.channel_dropout_mask(64, 128, 0.5, torch.device(DEV))
    assert abs((mask > 0).float().mean().item() - 0.5) < 0.03
    assert set(mask.unique().tolist()) <= {0.0, 2.0}


def test_smoke_entry():
    import __graft_entry__
    __graft_entry__.smoke()


def test_full_clip_inference_matches_oracle():
    """SURVEY 8(f) N2 -- the inference shape of `evaluate_test` (train.py:84-104): one whole clip, B = 1, T = 4800 frames
    at the config-3 widths, eval mode.  The attention then runs at Tq = 2400 (its T x T energy tensor, 184 MB per
    sample in the reference, is never built) and every T-dependent kernel sees a length it is not tuned for.  The oracle
    (the reference's algorithm on the CPU) runs the same clip; tolerance as for every other model test."""
    import time
    from tests.golden.cases import model_kwargs
    case = dict(next(c for c in MODEL_CASES if c["name"] == "c3_F128"), time_dim=4800, B=1)
    m = _prepared(case).eval()
    x = O.closed_form_input((1, case["input_channels"], case["freq_dim"], case["time_dim"]))
    with torch.no_grad():
        m(x.to(DEV))                                   # warm-up (module load)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sed, doa = m(x.to(DEV))
        torch.cuda.synchronize()
        gpu_s = time.perf_counter() - t0
        sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
        t0 = time.perf_counter()
        rsed, rdoa = O.seld_forward(sd, O.SeldConfig(**model_kwargs(case)), x, train=False)
        cpu_s = time.perf_counter() - t0
    print(f"full clip (1, 8, 128, 4800): HIP {gpu_s * 1e3:.1f} ms, oracle {cpu_s:.1f} s")
    assert tuple(sed.shape) == (1, 600, 42) and tuple(doa.shape) == (1, 600, 126)
    assert np.abs(_np(sed) - _np(rsed)).max() < 1e-3
    assert np.abs(_np(doa) - _np(rdoa)).max() < 1e-3


def test_six_step_trajectory_matches_oracle():
    """Six consecutive training steps (zero_grad -> forward -> loss -> backward -> Adam) of the tiny DQ model against
    the oracle driven by torch.optim.Adam in fp64, two alternating batches: every step's loss within 2e-4 relative and
    the parameters after the last step close to the oracle's (bounds calibrated against torch's own fp32, see below;
    measured here: worst element 0.33 lr, mean 0.0037 lr).  Exercises what a single step cannot: the gradient slots claimed / re-zeroed
    across steps, the pooled BatchNorm statistics buffers, Adam's bias correction and the weight re-layouts that are
    issued ahead on the side stream while the previous step's Adam may still be in flight."""
    from tests.golden.cases import model_kwargs
    T = pkg().train
    case = next(c for c in MODEL_CASES if c["name"] == "tiny_DQ")
    m = _prepared(case).train()
    sd64 = {k: v.detach().cpu().double().clone() for k, v in m.state_dict().items()}
    names = [n for n, _ in m.named_parameters()]
    leaves = [sd64[n].requires_grad_(True) for n in names]
    cfg = O.SeldConfig(**model_kwargs(case))
    lr = 1e-3
    opt = T.FlatAdam(m.parameters(), lr=lr)
    ropt = torch.optim.Adam(leaves, lr=lr)
    n_sed = int(case["output_classes"] * 3)
    shape = (case["B"], case["input_channels"], case["freq_dim"], case["time_dim"])
    xs = [O.closed_form_input(shape), O.closed_form_input(shape).flip(3) * 0.7]
    tg = [train_target(case), train_target(case).flip(1)]
    for step in range(6):
        x, t = xs[step % 2], tg[step % 2]
        opt.zero_grad()
        sed, doa = m(x.to(DEV))
        loss = T.seld_loss_fn(sed, doa, t.to(DEV), n_sed, 1.0, 5.0)
        loss.backward()
        opt.step()
        ropt.zero_grad()
        rsed, rdoa = O.seld_forward(sd64, cfg, x.double(), train=True, mode="explicit")
        rloss = O.seld_loss(rsed, rdoa, t.double(), n_sed)
        rloss.backward()
        ropt.step()
        assert abs(loss.item() - rloss.item()) <= 2e-4 * abs(rloss.item()), (step, loss.item(), rloss.item())
    # Parameters: Adam normalises every step, so rounding differences in small gradients are amplified from step to step.
    # Calibration (run once on the host): the SAME oracle in torch-CPU fp32 ends 1.87 lr (worst element) / 0.0186 lr (mean
    # over the elements whose gradient is above rounding noise) away from its fp64 run, and its loss 3.4e-4.  The HIP path
    # is held to a tighter mean and to the same order for the worst element.
    params = dict(m.named_parameters())
    worst, total, count = 0.0, 0.0, 0
    for i, n in enumerate(names):
        st = ropt.state.get(leaves[i])
        if st is None:
            continue
        solid = st["exp_avg_sq"].sqrt() > 1e-6               # gradient above rounding noise
        if not bool(solid.any()):
            continue
        d = (params[n].detach().cpu().double() - sd64[n].detach()).abs()[solid]
        worst = max(worst, d.max().item())
        total += d.sum().item()
        count += d.numel()
    print(f"six-step trajectory: final loss {loss.item():.6f} (oracle {rloss.item():.6f}), parameter deviation from the fp64 "
          f"oracle: worst {worst / lr:.3f} lr, mean {total / count / lr:.4f} lr over {count} elements")
    assert worst < 3.0 * lr and total / count < 0.01 * lr     # the worst element is one sign flip of a noise-level gradient


def test_side_stream_lag_does_not_corrupt_gradients(monkeypatch):
    """The accumulating weight-gradient kernels run on a side stream and read `dy`; the same `dy` is handed to
    autograd as the gradient of the residual addend, which the engine would add into IN PLACE on the main stream when
    it holds the last reference (hip_ops._on_side_stream keeps one until the join).  Make the side stream lag far behind
    the main stream (a long sleep queued on it before the backward pass) and require the same gradients as with
    everything on one stream."""
    H, T = pkg().hip_ops, pkg().train
    case = next(c for c in MODEL_CASES if c["name"] == "tiny_DQ")
    x = O.closed_form_input((case["B"], case["input_channels"], case["freq_dim"], case["time_dim"])).to(DEV)
    target = train_target(case).to(DEV)
    n_sed = int(case["output_classes"] * 3)

    def grads(side, lag):
        monkeypatch.setenv("SELD_WGRAD_SIDE_STREAM", "1" if side else "0")
        m = _prepared(case).train()
        opt = T.FlatAdam(m.parameters(), lr=1e-4)
        opt.zero_grad()
        sed, doa = m(x)
        loss = T.seld_loss_fn(sed, doa, target, n_sed, 1.0, 5.0)
        if lag:
            if H._side["stream"] is None:
                H._side["stream"] = torch.cuda.Stream()
            with torch.cuda.stream(H._side["stream"]):
                torch.cuda._sleep(400_000_000)          # ~0.2 s: the main stream finishes its whole backward first
        loss.backward()
        H.join_side_stream()
        torch.cuda.synchronize()
        return opt.flat_grad.detach().clone()

    ref = grads(False, False)
    got = grads(True, True)
    scale = float(ref.abs().max())
    assert scale > 0
    err = float((got - ref).abs().max())
    assert err <= 1e-4 * scale, f"side-stream lag changed the gradients: {err:.3e} of {scale:.3e}"


@pytest.mark.parametrize("name", ["tiny_2stream", "tiny_DQ"])
def test_two_stream_branches_on_two_queues_match_one_queue(name, monkeypatch):
    """The two ConvTC blocks of the two-stream model, and the two classifier heads of every model, run on two HIP streams
    (hip_ops.run_branches).  Make the second queue
    lag far behind the first (a long sleep queued on it before the step) and require the same outputs, loss and
    gradients as with both branches on one queue, for two consecutive steps (the second reuses pooled statistics
    buffers and the packed weight forms across the streams)."""
    H, T = pkg().hip_ops, pkg().train
    case = next(c for c in MODEL_CASES if c["name"] == name)
    x = O.closed_form_input((case["B"], case["input_channels"], case["freq_dim"], case["time_dim"])).to(DEV)
    target = train_target(case).to(DEV)
    n_sed = int(case["output_classes"] * 3)

    def run(two_queues):
        monkeypatch.setenv("SELD_BRANCH_STREAMS", "1" if two_queues else "0")
        torch.manual_seed(3)
        H.philox.set_offset(0)
        m = _prepared(case).train()
        opt = T.FlatAdam(m.parameters(), lr=1e-3)
        out = []
        for step in range(2):
            opt.zero_grad()
            if two_queues and step == 0:
                if H._branch["stream"] is None:
                    H._branch["stream"] = torch.cuda.Stream()
                with torch.cuda.stream(H._branch["stream"]):
                    torch.cuda._sleep(200_000_000)       # ~0.1 s: whatever the main queue does not wait for, it now overtakes
            sed, doa = m(x)
            loss = T.seld_loss_fn(sed, doa, target, n_sed, 1.0, 5.0)
            loss.backward()
            H.join_side_stream()
            torch.cuda.synchronize()
            out.append((sed.detach().clone(), doa.detach().clone(), float(loss.detach()), opt.flat_grad.detach().clone()))
            opt.step()
        return out

    ref, got = run(False), run(True)
    # step 0 agrees to float-atomic ordering noise; step 1 starts from parameters that Adam moved by up to lr wherever a
    # gradient is such noise (see tests/test_gpu_dp.py), so it is held to what a missed dependency would break, not more
    for step, ((s0, d0, l0, g0), (s1, d1, l1, g1)) in enumerate(zip(ref, got)):
        tol = 1e-5 if step == 0 else 2e-3
        assert float((s0 - s1).abs().max()) <= tol and float((d0 - d1).abs().max()) <= tol, step
        assert abs(l0 - l1) <= tol * max(1.0, abs(l0)), step
        scale = float(g0.abs().max())
        assert float((g0 - g1).abs().max()) <= (2e-4 if step == 0 else 5e-2) * scale, step
