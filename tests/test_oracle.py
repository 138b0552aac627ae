"""The oracle (oracle/seld_oracle.py) against the fixtures captured from the reference
(tests/golden/make_golden.py).  CPU only; this is what pins the oracle."""
import numpy as np
import pytest
import torch

from oracle import seld_oracle as O
from tests.golden.cases import MODEL_CASES, OP_CASES, model_kwargs, op_cotangent, op_inputs, train_target

DT = torch.float64
TOL = 2e-6  # fixtures are stored as float32


def _close(a, b, tol=TOL):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    scale = max(1.0, float(np.abs(b).max()))
    err = float(np.abs(a - b).max())
    assert err <= tol * scale, f"max err {err} (scale {scale})"


def run_op(case, mode, dtype=DT):
    x, ws, bias = op_inputs(case, dtype)
    x.requires_grad_(True)
    [w.requires_grad_(True) for w in ws]
    if bias is not None:
        bias.requires_grad_(True)
    kind = case["kind"]
    if kind in ("qconv", "dqconv"):
        y = O.hypercomplex_conv(x, ws, bias, case["stride"], case["padding"], 1, case["dilation"], mode)
    elif kind in ("qlinear", "qlinear_fn"):
        y = O.quaternion_linear(x, *ws, bias=bias, mode=mode)
    else:
        y = O.dual_quaternion_linear(x, tuple(ws), bias, mode)
    (y * op_cotangent(y.shape, dtype)).sum().backward()
    return y, x, ws, bias


@pytest.mark.parametrize("mode", ["assembled", "explicit"])
@pytest.mark.parametrize("case", OP_CASES, ids=[c["name"] for c in OP_CASES])
def test_ops_match_reference(case, mode, golden):
    g = golden("ops")
    y, x, ws, bias = run_op(case, mode)
    n = case["name"]
    _close(y.detach(), g[n + ".y"])
    _close(x.grad, g[n + ".dx"])
    for i, w in enumerate(ws):
        _close(w.grad, g[f"{n}.dw{i}"])
    if bias is not None:
        _close(bias.grad, g[n + ".dbias"])


def test_mha_matches_reference(golden):
    g = golden("mha")
    E, T, N = 48, 20, 2
    sd = {"queries.weight": torch.empty(E, E, 1, dtype=DT), "keys.weight": torch.empty(E, E, 1, dtype=DT),
          "values.weight": torch.empty(E, E, 1, dtype=DT), "fc_out.weight": torch.empty(E, E, dtype=DT),
          "fc_out.bias": torch.empty(E, dtype=DT)}
    # reference state-dict order: values, keys, queries, fc_out (model.py:20-23)
    ordered = [(k, sd[k]) for k in ("values.weight", "keys.weight", "queries.weight", "fc_out.weight", "fc_out.bias")]
    O.closed_form_fill_(ordered, amp=0.6)
    for v in sd.values():
        v.requires_grad_(True)
    x = O.closed_form_input((N, T, E), DT).requires_grad_(True)
    y = O.multi_head_attention(x, sd["queries.weight"], sd["keys.weight"], sd["values.weight"],
                               sd["fc_out.weight"], sd["fc_out.bias"])
    (y * O.closed_form_input(tuple(y.shape), DT).flip(1)).sum().backward()
    _close(y.detach(), g["y"])
    _close(x.grad, g["dx"])
    _close(sd["queries.weight"].grad, g["dwq"])
    _close(sd["keys.weight"].grad, g["dwk"])
    _close(sd["values.weight"].grad, g["dwv"])
    _close(sd["fc_out.weight"].grad, g["dwo"])
    _close(sd["fc_out.bias"].grad, g["dbo"])


def build_state(case, dtype=DT):
    """State dict with the reference's key names/shapes/order, built by the host-side mirror."""
    from tests.helpers import reference_layout_state
    return reference_layout_state(case, dtype)        # weights filled per case (tests.helpers.fill_weights)


@pytest.mark.parametrize("mode", ["assembled", "explicit"])
@pytest.mark.parametrize("case", MODEL_CASES, ids=[c["name"] for c in MODEL_CASES])
def test_model_eval_matches_reference(case, mode, golden):
    if mode == "explicit" and not case.get("train", False):
        pytest.skip("large widths are checked in assembled mode only (time)")
    g = golden("model_" + case["name"])
    cfg = O.SeldConfig(**model_kwargs(case))
    sd = build_state(case)
    x = O.closed_form_input((case["B"], case["input_channels"], case["freq_dim"], case["time_dim"]), DT)
    taps = {}
    with torch.no_grad():
        sed, doa = O.seld_forward(sd, cfg, x, train=False, mode=mode, taps=taps)
    _close(sed, g["sed"])
    _close(doa, g["doa"])
    for k in g:
        if not k.startswith("tap."):
            continue
        name = k[4:]
        if name.endswith(".attention"):
            ref = g[k]                      # reference hook sees (N, T, E)
            got = taps[name].permute(0, 2, 1)
        elif ".cnn." in name:
            got, ref = taps[name], g[k]
        else:
            got, ref = taps[name], g[k]
        _close(got, ref)


@pytest.mark.parametrize("case", [c for c in MODEL_CASES if c.get("train")], ids=lambda c: c["name"])
def test_model_train_step_matches_reference(case, golden):
    g = golden("model_" + case["name"])
    cfg = O.SeldConfig(**model_kwargs(case))
    sd = build_state(case)
    names = str(g["train.param_names"]).split("\n")
    params = {n: sd[n].requires_grad_(True) for n in names}
    x = O.closed_form_input((case["B"], case["input_channels"], case["freq_dim"], case["time_dim"]), DT)
    stats = {}
    sed, doa = O.seld_forward(sd, cfg, x, train=True, mode="assembled", stats_out=stats)
    target = train_target(case, DT)
    n_sed = int(case["output_classes"] * 3)
    loss = O.seld_loss(sed, doa, target, n_sed)
    loss.backward()
    _close(sed.detach(), g["train.sed"])
    _close(doa.detach(), g["train.doa"])
    _close([loss.item()], g["train.loss"])
    cks = g["train.grad_checksums"]
    dck = g["train.delta_checksums"]
    for i, n in enumerate(names):
        p = params[n]
        if np.isnan(cks[i, 0]):
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, n
            continue
        got = np.array([p.grad.sum().item(), (p.grad ** 2).sum().item()])
        assert np.allclose(got, cks[i], rtol=2e-5, atol=1e-7), (n, got, cks[i])
        newp, _, _ = O.adam_step(p.detach(), p.grad, torch.zeros_like(p), torch.zeros_like(p), 1)
        d = newp - p.detach()
        got = np.array([d.sum().item(), (d ** 2).sum().item()])
        assert np.allclose(got, dck[i], rtol=2e-4, atol=1e-9), (n, got, dck[i])
    for k in g:
        if k.startswith("train.grad."):
            _close(params[k[len("train.grad."):]].grad, g[k], tol=1e-5)
    rnames = str(g["train.running_names"]).split("\n")
    rck = g["train.running_checksums"]
    for i, n in enumerate(rnames):
        if n not in stats:      # batch_gate1.* are never used by the forward (model.py:90)
            val = sd[n]
        else:
            val = stats[n]
        got = np.array([val.sum().item(), (val ** 2).sum().item()])
        assert np.allclose(got, rck[i], rtol=2e-5, atol=1e-7), (n, got, rck[i])


def test_stft_matches_reference(golden):
    g = golden("stft")
    n = np.arange(6400)
    x = np.stack([np.sin(2 * np.pi * (100 + 37 * c) * n / 32000) + 0.1 * np.sin(0.013 * n * (c + 1)) for c in range(8)])
    a = O.spectrum_fast(x, 512, 112, output_phase=True)
    assert a.shape == g["magphase_112"].shape
    nb = a.shape[0] // 2
    assert np.abs(a[:nb] - g["magphase_112"][:nb]).max() < 1e-12
    # phase: compare on the unit circle where the magnitude is not numerically zero
    mag = g["magphase_112"][:nb]
    mask = mag > 1e-9
    dphi = np.angle(np.exp(1j * (a[nb:] - g["magphase_112"][nb:])))
    assert np.abs(dphi[mask]).max() < 1e-6
    b = O.spectrum_fast(x, 512, 112, output_phase=False)
    assert np.abs(b - g["mag_112"]).max() < 1e-12
    c = O.spectrum_fast(x, 512, 128, output_phase=True)
    assert c.shape == g["magphase_128"].shape
    assert np.abs(c[:nb] - g["magphase_128"][:nb]).max() < 1e-12


def test_dataset_normalisation_matches_reference(golden):
    """norm.npz = the arrays the reference's train.main hands to its TensorDatasets (make_golden.gen_norm);
    the oracle restatement must reproduce them bit for bit (same torch / numpy primitives, same dtype)."""
    from tests.golden.cases import NORM_CASES, norm_input
    g = golden("norm")
    for name, shape, dtype, mode, n_mics, domain, phase in NORM_CASES:
        for k in range(3):
            got = O.normalize_dataset(norm_input(shape, dtype, k), mode, n_mics, domain, phase)
            ref = g[f"{name}.{k}"]
            assert got.dtype == np.float32 and got.shape == ref.shape
            assert np.array_equal(got, ref), (name, k, np.abs(got - ref).max())
    with pytest.raises(ValueError):
        O.normalize_dataset(norm_input((2, 16, 4, 4), np.float32, 0), "UnitNorm", 2, "DQ", True)
    # the IEEE evaluation the HIP kernel is held to: p channels identical to the reference, q within 1 ulp
    x = norm_input((5, 8, 6, 10), np.float32, 0)
    ieee, ref = O.dq_unit_norm_ieee(x), g["unit_f32.0"]
    assert np.array_equal(ieee[:, 4:], ref[:, 4:])
    assert np.abs(ieee[:, :4] - ref[:, :4]).max() <= np.spacing(np.float32(1.0))


def test_test_metrics_match_reference(golden):
    """metrics.npz = the reference's evaluate_test results and the counters of its metric classes
    (make_golden.gen_metrics).  Counters exact, floating-point results to 1e-12."""
    from tests.golden.cases import METRIC_CASES, metric_inputs
    g = golden("metrics")
    for name, clips, frames, seed, variant in METRIC_CASES:
        sed, doa, target = metric_inputs(clips, frames, seed, variant)
        results, counts, total_de = O.evaluate_clips(sed, doa, target, num_frames=frames, epoch=7)
        assert [counts[k] for k in O.METRIC_COUNTERS] == g[name + ".counters"].tolist(), name
        assert abs(total_de - g[name + ".total_DE"][0]) <= 1e-12 * max(1.0, total_de), name
        assert np.allclose(np.array(results, dtype=np.float64), g[name + ".results"], rtol=1e-12, atol=1e-12), name
    # no reference event at all: the reference divides by Nref = 0 (train.py:136)
    sed, doa, target = metric_inputs(1, 20, 3, "mixed")
    target[:] = 0
    with pytest.raises(ZeroDivisionError):
        O.evaluate_clips(sed, doa, target, num_frames=20)
