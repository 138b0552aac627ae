"""Host-side state that sits between the reference-shaped Python surface and the kernels: the packed weight
forms of the fast-product convolutions after in-place weight edits, after a replayed (HIP-graph) Adam step, and when new
shapes are registered between replays; the device-resident step state after host-side changes.

All at the config-3 widths (192 / 384 channels: the layers that run on csrc/hcq_conv.hip; the 16-wide fixtures never
reach those kernels).  Reference semantics at stake: `model.load_state_dict` + `model(x)` (train.py:47-81, 84-104) must
see the loaded weights in EVERY layer."""
import numpy as np
import pytest
import torch

from oracle import seld_oracle as O
from tests.golden.cases import MODEL_CASES, model_kwargs, train_target
from tests.helpers import build_model, pkg

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
CASE = next(c for c in MODEL_CASES if c["name"] == "c3w_train")


def _setup(lr=1e-3, dropout=False):
    T, H = pkg().train, pkg().hip_ops
    case = dict(CASE, dropout_perc=0.3, spatial_dropout_rate=0.5) if dropout else CASE
    torch.manual_seed(3)
    H.philox.set_offset(0)
    H.hcq_weights.reset()
    m = build_model(case).to(DEV).train()
    opt = T.FlatAdam(m.parameters(), lr=lr)
    x = O.closed_form_input((case["B"], case["input_channels"], case["freq_dim"], case["time_dim"])).to(DEV)
    target = train_target(case).to(DEV)
    return m, opt, x, target, int(case["output_classes"] * 3)


def _eager_step(m, opt, x, target, n_sed):
    T = pkg().train
    m.train()
    opt.zero_grad()
    sed, doa = m(x)
    loss = T.seld_loss_fn(sed, doa, target, n_sed, 1.0, 5.0)
    loss.backward()
    opt.step()
    return loss


def _eval(m, x):
    m.eval()
    with torch.no_grad():
        sed, doa = m(x)
    torch.cuda.synchronize()
    m.train()
    return torch.cat((sed.flatten(), doa.flatten())).clone()


def _eval_fresh_cache(m, x):
    """The same forward with every packed form rebuilt from the weights as they are now."""
    pkg().hip_ops.hcq_weights.reset()
    return _eval(m, x)


def test_hcq_layers_are_in_play():
    H = pkg().hip_ops
    m, opt, x, target, n_sed = _setup()
    _eval(m, x)
    live = [e for e in H.hcq_weights.entries.values() if e is not None]
    assert len(live) >= 20, "the config-width model no longer runs on the fast-product kernels: these tests would be vacuous"


def test_packed_forms_follow_load_state_dict():
    """train step -> eval at shape A -> eval at shape B -> train step -> eval at A (bulk re-pack of A's AND B's entries) ->
    load_state_dict(other weights) -> eval at B: B's entries were bulk-packed, never requested since, then edited in
    place -- the ADVICE r2 hole (test leg after a non-improving epoch).  Also against the fp64 oracle."""
    m, opt, x, target, n_sed = _setup()
    xa, xb = x, x[:1].contiguous()
    _eager_step(m, opt, x, target, n_sed)
    _eval(m, xa)
    _eval(m, xb)
    _eager_step(m, opt, x, target, n_sed)
    _eval(m, xa)
    other = {k: (v.detach().clone() * (1.0 + 0.05 * np.sin(i)) if v.is_floating_point() and "running" not in k else v.detach().clone())
             for i, (k, v) in enumerate(m.state_dict().items())}
    m.load_state_dict(other)
    got = _eval(m, xb)
    ref = _eval_fresh_cache(m, xb)
    assert torch.equal(got, ref), float((got - ref).abs().max())
    # and data-gradient (mode 1) entries: a training step right after the load against the same step with a fresh cache
    sd0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
    _eager_step(m, opt, x, target, n_sed)
    g_after_load = opt.flat_grad.detach().clone()
    m.load_state_dict(sd0)
    pkg().hip_ops.hcq_weights.reset()
    _eager_step(m, opt, x, target, n_sed)
    g_fresh = opt.flat_grad.detach().clone()
    assert float((g_after_load - g_fresh).abs().max()) <= 2e-5 * float(g_fresh.abs().max())     # float-atomic order only
    # oracle: the loaded weights, eval mode
    m.load_state_dict(other)
    got = _eval(m, xb)
    cfg = O.SeldConfig(**model_kwargs(CASE))
    sd64 = {k: v.detach().cpu().double() for k, v in other.items()}
    with torch.no_grad():
        sed_r, doa_r = O.seld_forward(sd64, cfg, xb.cpu().double(), train=False, mode="assembled")
    ref = torch.cat((sed_r.flatten(), doa_r.flatten()))
    assert float((got.cpu().double() - ref).abs().max()) < 1e-3


def test_save_load_model_roundtrip_refreshes_forms(tmp_path):
    """train.load_model (train.py:47-81) after the live weights moved on: eval must be the checkpoint's."""
    T = pkg().train
    m, opt, x, target, n_sed = _setup()
    _eager_step(m, opt, x, target, n_sed)
    at_save = _eval(m, x)
    T.save_model(m, opt, {"step": 1}, str(tmp_path / "ck"))
    for _ in range(2):
        _eager_step(m, opt, x, target, n_sed)
    moved = _eval(m, x)
    assert float((moved - at_save).abs().max()) > 1e-5
    T.load_model(m, opt, str(tmp_path / "ck"), True, torch.device(DEV))
    back = _eval(m, x)
    assert torch.equal(back, at_save), float((back - at_save).abs().max())


def test_replay_then_eval_sees_the_replayed_adam_step():
    """replay, eval, replay, eval: an eager forward after a replay must use forms of the weights the replay produced
    (GraphedTrainStep.__call__ tells the cache), and a replay after a cache reset still finds its recorded table."""
    T, H = pkg().train, pkg().hip_ops
    m, opt, x, target, n_sed = _setup()
    runner = T.GraphedTrainStep(m, opt, x, target, n_sed, 1.0, 5.0, warmup=1)
    for _ in range(2):
        runner()
        got = _eval(m, x)
        # level the host cache with its epoch, as a validation pass between replays does
        again = _eval(m, x)
        assert torch.equal(got, again)
        ref = _eval_fresh_cache(m, x)
        assert torch.equal(got, ref), float((got - ref).abs().max())
    runner()            # the recorded pack launch still points at the (pinned) table it was recorded with
    torch.cuda.synchronize()
    assert np.isfinite(float(runner.loss.item()))


def test_first_replay_gradient_matches_eager_step_from_the_same_state():
    """The gradient a replay leaves in the flat buffer against an eager step from the identical parameter / optimiser /
    running-statistics state: only float-atomic ordering may differ (1e-4 of max|g|; forms stale by one Adam step of
    lr = 1e-3 would be ~1e-2 off)."""
    T, H = pkg().train, pkg().hip_ops
    m, opt, x, target, n_sed = _setup()
    runner = T.GraphedTrainStep(m, opt, x, target, n_sed, 1.0, 5.0, warmup=1)
    snap = dict(p=opt.flat_param.clone(), m1=opt.exp_avg.clone(), m2=opt.exp_avg_sq.clone(), step=opt.step_count,
                buf={k: v.clone() for k, v in m.state_dict().items() if "running" in k or "num_batches" in k})
    runner()
    torch.cuda.synchronize()
    g_replay = opt.flat_grad.clone()
    p_replay = opt.flat_param.clone()
    with torch.no_grad():
        opt.flat_param.copy_(snap["p"]); opt.exp_avg.copy_(snap["m1"]); opt.exp_avg_sq.copy_(snap["m2"])
        sd = m.state_dict()
        for k, v in snap["buf"].items():
            sd[k].copy_(v)
    opt.step_count = snap["step"]
    H.hcq_weights.weights_changed()
    _eager_step(m, opt, x, target, n_sed)
    torch.cuda.synchronize()
    scale = float(opt.flat_grad.abs().max())
    assert float((g_replay - opt.flat_grad).abs().max()) <= 1e-4 * scale
    # Adam: same step number, same learning rate from the device state; elements with noise-level gradients may part by ~lr
    d = (p_replay - opt.flat_param).abs()
    assert float(d.mean()) <= 1e-2 * 1e-3 and float((d > 0.5e-3).float().mean()) <= 1e-3


def test_new_shape_registered_between_replays():
    """A validation pass at ANOTHER batch size between replays adds cache entries, which rebuilds the cache's table: the
    recorded launch must keep reading the table it was recorded with (pinned), not freed memory.  The trajectory with the
    interleaved evals equals the one without."""
    T, H = pkg().train, pkg().hip_ops

    def run(interleave):
        m, opt, x, target, n_sed = _setup()
        runner = T.GraphedTrainStep(m, opt, x, target, n_sed, 1.0, 5.0, warmup=1)
        losses = []
        for i in range(4):
            losses.append(float(runner().item()))
            if interleave:
                _eval(m, x[:1].contiguous() if i % 2 == 0 else torch.cat((x, x[:1]), 0))     # batch 1, then batch 3
                # churn the allocator so that a freed table would be recycled
                junk = [torch.full((1 << 16,), float(i), device=DEV) for _ in range(8)]
                del junk
        torch.cuda.synchronize()
        return losses, opt.flat_param.clone()
    la, pa = run(False)
    lb, pb = run(True)
    assert np.allclose(la, lb, rtol=2e-3), (la, lb)
    d = (pa - pb).abs()
    assert float(d.mean()) <= 0.1 * 1e-3, float(d.mean())


def test_step_state_follows_host_side_changes():
    """Eager steps and a restored optimiser between replays: the device step counter (Adam bias correction) and the Philox
    base are brought up to date before the next replay; an eager draw after a replay lies beyond that replay's range."""
    T, H = pkg().train, pkg().hip_ops
    m, opt, x, target, n_sed = _setup(dropout=True)
    runner = T.GraphedTrainStep(m, opt, x, target, n_sed, 1.0, 5.0, warmup=1)
    st = lambda: H.philox.state(torch.device(DEV)).cpu()
    per_step, base0 = int(st()[3]), int(st()[0])
    assert per_step > 0 and base0 == per_step          # one eager warm-up step drew `per_step` groups
    runner(); runner()
    assert int(st()[0]) == base0 + 2 * per_step and H.philox.offset == 0      # the base already points past replay 2's draws
    assert int(st()[1]) == opt.step_count == 3
    _eager_step(m, opt, x, target, n_sed)               # draws per_step groups on top of the base, step 4 on the host
    assert H.philox.offset == per_step and opt.step_count == 4
    runner()
    assert int(st()[1]) == opt.step_count == 5
    assert int(st()[0]) == base0 + 4 * per_step and H.philox.offset == 0
    assert H.philox.get_offset() == int(st()[0])
    sd = opt.state_dict()
    for v in sd["state"].values():
        v["step"] = torch.tensor(40.0)
    opt.load_state_dict(sd)
    runner()
    assert int(st()[1]) == opt.step_count == 41


def test_plain_backward_on_a_cut_model_is_refused():
    """dp.BackwardCut left installed + a plain loss.backward(): the front end would silently get no gradient;
    the next forward refuses instead, and the context-manager form removes the cut."""
    T, DP = pkg().train, pkg().dp
    m, opt, x, target, n_sed = _setup()
    cut = DP.BackwardCut(m)
    opt.zero_grad()
    sed, doa = m(x)
    T.seld_loss_fn(sed, doa, target, n_sed, 1.0, 5.0).backward()
    with pytest.raises(RuntimeError, match="BackwardCut"):
        m(x)
    cut.finish()
    cut.remove()
    with cut:
        opt.zero_grad()
        sed, doa = m(x)
        T.seld_loss_fn(sed, doa, target, n_sed, 1.0, 5.0).backward()
        cut.finish()
    front = dict(m.named_parameters())["seld_block.cnn.0.0.r_weight"].grad
    torch.cuda.synchronize()
    assert float(front.abs().max()) > 0
    assert not hasattr(m.seld_block, "_backward_cuts")


@pytest.mark.parametrize("name", ["c3w_train", "c4w_train", "c5w_train", "c2w_train"])
def test_every_config_width_records_and_replays(name):
    """train.GraphedTrainStep at the widths of configs 2-5 in the DEFAULT mode (side stream, two-queue branches, the first
    stage's second moments gathered beside the weight-form pack): the capture must close and the replays must follow the
    eager trajectory.  (The two-stream model once crashed `capture_end` when its branches forked the side stream from a
    branch stream -- no value test saw it, only the config-5 bench line.)"""
    T, H = pkg().train, pkg().hip_ops
    case = dict(next(c for c in MODEL_CASES if c["name"] == name), dropout_perc=0.0, spatial_dropout_rate=0.0)

    def run(graph):
        torch.manual_seed(3)
        H.philox.set_offset(0)
        H.hcq_weights.reset()
        m = build_model(case).to(DEV).train()
        opt = T.FlatAdam(m.parameters(), lr=1e-3)
        x = O.closed_form_input((case["B"], case["input_channels"], case["freq_dim"], case["time_dim"])).to(DEV)
        target = train_target(case).to(DEV)
        n_sed = int(case["output_classes"] * 3)
        losses = []
        if graph:
            runner = T.GraphedTrainStep(m, opt, x, target, n_sed, 1.0, 5.0, warmup=1)
            for _ in range(3):
                losses.append(float(runner().item()))
        else:
            for _ in range(4):
                losses.append(float(_eager_step(m, opt, x, target, n_sed).item()))
            losses = losses[1:]                  # the recorded runner's warm-up step is the eager run's first
        torch.cuda.synchronize()
        return losses
    lg, le = run(True), run(False)
    assert all(np.isfinite(v) for v in lg), lg
    assert np.allclose(lg, le, rtol=5e-3), (lg, le)
