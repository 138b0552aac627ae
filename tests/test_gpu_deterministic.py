"""SELD_DETERMINISTIC=1: training is run-to-run reproducible -- the reference's CPU path is
(/root/reference/train.py:214-221 fixes every seed) -- so the self-comparison tests that had to be loosened to the size of
float-atomic ordering noise run here at rounding level: two eager runs bit for bit, recorded (HIP graph) against eager,
two queues against one, and the default mode against the deterministic one (same mathematics, summation order only)."""
import numpy as np
import pytest
import torch

from oracle import seld_oracle as O
from tests.golden.cases import MODEL_CASES, train_target
from tests.helpers import build_model, pkg

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _case(name, dropout):
    case = next(c for c in MODEL_CASES if c["name"] == name)
    return dict(case, dropout_perc=0.3, spatial_dropout_rate=0.5) if dropout else case


def _run(name, steps, mode="eager", dropout=True, lr=1e-3):
    T, H = pkg().train, pkg().hip_ops
    case = _case(name, dropout)
    torch.manual_seed(5)
    H.philox.set_offset(0)
    H.hcq_weights.reset()
    m = build_model(case)
    if case.get("fill", "closed_form") != "init":
        O.closed_form_fill_(list(m.state_dict().items()))
    m = m.to(DEV).train()
    opt = T.FlatAdam(m.parameters(), lr=lr)
    x = O.closed_form_input((case["B"], case["input_channels"], case["freq_dim"], case["time_dim"])).to(DEV)
    target = train_target(case).to(DEV)
    n_sed = int(case["output_classes"] * 3)
    losses = []
    if mode == "graph":
        runner = T.GraphedTrainStep(m, opt, x, target, n_sed, 1.0, 5.0, warmup=1)
        for _ in range(steps - 1):
            losses.append(float(runner().item()))
    else:
        for _ in range(steps):
            opt.zero_grad()
            sed, doa = m(x)
            loss = T.seld_loss_fn(sed, doa, target, n_sed, 1.0, 5.0)
            loss.backward()
            opt.step()
            losses.append(float(loss.item()))
    torch.cuda.synchronize()
    return losses, opt.flat_param.detach().clone(), {k: v.clone() for k, v in m.state_dict().items() if "running" in k}


@pytest.mark.parametrize("name", ["tiny_DQ", "tiny_Q", "tiny_R", "tiny_2stream", "c3w_train"])
def test_two_runs_are_bit_identical(name, seld_env):
    """Three training steps with Dropout on, twice: losses, every parameter and every running statistic bit for bit."""
    seld_env.set("SELD_DETERMINISTIC", "1")
    la, pa, ra = _run(name, 3)
    lb, pb, rb = _run(name, 3)
    assert la == lb, (la, lb)
    assert torch.equal(pa, pb), float((pa - pb).abs().max())
    for k in ra:
        assert torch.equal(ra[k], rb[k]), k


@pytest.mark.parametrize("name", ["tiny_DQ", "c3w_train"])
def test_default_mode_differs_by_rounding_only(name, seld_env):
    """One step in the default mode (split reductions, float atomics, side stream) against the deterministic one: the same
    mathematics, so loss and parameters agree to summation-order rounding (Adam's first step moves every parameter by
    ~lr whatever its gradient: compare the step DIRECTION on parameters whose gradient is not noise, i.e. the loss and
    the bulk of the parameters)."""
    seld_env.set("SELD_DETERMINISTIC", "1")
    ld, pd, _ = _run(name, 1, dropout=False)
    seld_env.unset("SELD_DETERMINISTIC")
    lf, pf, _ = _run(name, 1, dropout=False)
    assert abs(ld[0] - lf[0]) <= 2e-6 * abs(lf[0]), (ld, lf)
    d = (pd - pf).abs()
    assert float((d > 1e-4).float().mean()) < 2e-2, float((d > 1e-4).float().mean())     # lr = 1e-3: a flipped noise-level element moves 2e-3


@pytest.mark.parametrize("name", ["tiny_DQ", "c3w_train"])
def test_recorded_step_matches_eager_at_rounding_level(name, seld_env):
    """Five steps recorded-and-replayed against five eager steps in deterministic mode, Dropout off (a replay draws its
    masks from the device-resident counter: same masks only by construction of the counters, tested elsewhere): the
    launches are the same launches, so the parameters agree to 1e-6 relative ('rounding-level tolerances')."""
    seld_env.set("SELD_DETERMINISTIC", "1")
    le, pe, re_ = _run(name, 5, "eager", dropout=False)
    lg, pg, rg = _run(name, 5, "graph", dropout=False)
    assert np.allclose(lg, le[1:], rtol=1e-6, atol=0), (lg, le)
    scale = float(pe.abs().max())
    assert float((pg - pe).abs().max()) <= 1e-6 * scale, float((pg - pe).abs().max())
    for k in re_:
        assert torch.allclose(rg[k], re_[k], rtol=1e-6, atol=1e-7 * float(re_[k].abs().max()) + 1e-12), k


def test_two_queues_match_one_queue_bitwise(seld_env, monkeypatch):
    """The two-stream model's branches and the classifier heads on two HIP queues against one queue, deterministic mode:
    the same kernels on the same data in another interleaving -- bit for bit."""
    seld_env.set("SELD_DETERMINISTIC", "1")
    monkeypatch.setenv("SELD_BRANCH_STREAMS", "1")
    l2, p2, r2 = _run("tiny_2stream", 3)
    monkeypatch.setenv("SELD_BRANCH_STREAMS", "0")
    l1, p1, r1 = _run("tiny_2stream", 3)
    assert l1 == l2
    assert torch.equal(p1, p2), float((p1 - p2).abs().max())
