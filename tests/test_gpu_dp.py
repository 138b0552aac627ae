"""Data-parallel training of the REAL model on the HIP path with world_size 2 (SURVEY section 4 tier 5): two fresh
child processes on the one GPU of the box (gloo; RCCL refuses two ranks on one device), each running FlatAdam +
dp.BucketedGradSync + dp_train_step on its half of the batch.  Also the single-process recorded step
(train.GraphedTrainStep) against the eager step."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import seld_oracle as O
from tests.golden.cases import MODEL_CASES, model_kwargs, train_target
from tests.helpers import build_model, pkg

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_ranks(tmp_path, case, mode, steps, env_extra=None):
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE="2",
                   LOCAL_RANK=str(rank))
        env.update(env_extra or {})
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dp_worker.py"), str(tmp_path), case,
                                       mode, str(steps)], env=env, cwd=ROOT))
    try:
        for p in procs:
            assert p.wait(timeout=420) == 0
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return [torch.load(tmp_path / f"rank{r}.pt") for r in range(2)]


@pytest.mark.timeout(600)
def test_two_rank_step_matches_oracle_of_local_bn_shards(tmp_path):
    """Replicas bit-identical after the step; the exchanged gradient == sum over the two shards of the oracle's
    gradient with BatchNorm statistics taken per shard (what each rank computes locally)."""
    case = dict(next(c for c in MODEL_CASES if c["name"] == "tiny_DQ"), B=4)
    r0, r1 = _run_ranks(tmp_path, "tiny_DQ", "eager", 2)
    assert torch.equal(r0["param"], r1["param"]), "replicas diverged"
    assert torch.equal(r0["first_grad"], r1["first_grad"])
    assert r0["step_count"] == 2
    # oracle: per-shard forward/backward in fp64, gradients summed
    m = build_model(case)
    O.closed_form_fill_(list(m.state_dict().items()))
    x = O.closed_form_input((4, case["input_channels"], case["freq_dim"], case["time_dim"]))
    target = train_target(case)
    cfg = O.SeldConfig(**model_kwargs(case))
    n_sed = int(case["output_classes"] * 3)
    total = {}
    ref_losses = []
    for lo, hi in ((0, 2), (2, 4)):
        sd64 = {k: v.detach().double().clone() for k, v in m.state_dict().items()}
        for v in sd64.values():
            v.requires_grad_(v.is_floating_point())
        sed, doa = O.seld_forward(sd64, cfg, x[lo:hi].double(), train=True, mode="explicit")
        loss = O.seld_loss(sed, doa, target[lo:hi].double(), n_sed)
        loss.backward()
        ref_losses.append(float(loss.detach()))
        for k, v in sd64.items():
            if v.grad is not None:
                total[k] = total.get(k, 0) + v.grad
    assert abs(r0["losses"][0] - ref_losses[0]) <= 1e-4 * abs(ref_losses[0])
    assert abs(r1["losses"][0] - ref_losses[1]) <= 1e-4 * abs(ref_losses[1])
    flat = r0["first_grad"].double()
    names, offsets = r0["names"], r0["offsets"]
    params = dict(m.named_parameters())
    scale = max(float(g.abs().max()) for g in total.values())
    checked = 0
    for n, off in zip(names, offsets):
        got = flat[off:off + params[n].numel()].view(params[n].shape)
        ref = total.get(n)
        if ref is None:
            assert float(got.abs().max()) == 0.0, n
            continue
        err = float((got - ref).abs().max())
        assert err <= 2e-3 * max(float(ref.abs().max()), 1e-3 * scale), (n, err, float(ref.abs().max()))
        checked += 1
    assert checked > 400
    # the late bucket (front-end convolutions) sits at the front of the flat buffer
    late = [n for n, off in zip(names, offsets) if off < r0["late_numel"]]
    assert late and all(".cnn." in n for n in late)


@pytest.mark.timeout(600)
def test_two_rank_graphed_step_matches_eager(tmp_path):
    """The recorded data-parallel step (three graphs, collectives between them) against the eager one: same losses,
    same parameters after the same number of steps, replicas identical."""
    (tmp_path / "e").mkdir()
    (tmp_path / "g").mkdir()
    e0, e1 = _run_ranks(tmp_path / "e", "tiny_DQ", "eager", 4)
    g0, g1 = _run_ranks(tmp_path / "g", "tiny_DQ", "graph", 3)       # + 1 warm-up step inside the constructor
    assert torch.equal(g0["param"], g1["param"])
    assert g0["step_count"] == 4 and int(g0["state"][1]) == 4
    assert np.allclose(g0["losses"], e0["losses"][1:], rtol=3e-3) and abs(g0["losses"][0] - e0["losses"][1]) <= 2e-4 * e0["losses"][1]
    # Parameters: Adam normalises every element's step to ~lr, so elements whose gradient is float-atomic ordering noise
    # move by up to lr per step in either run (tests/test_gpu_model.py::test_six_step_trajectory has the calibration);
    # the mean deviation is what shows a wrong exchange
    d = (g0["param"] - e0["param"]).abs()
    lr, steps = 1e-3, 4
    assert d.mean().item() <= 0.1 * lr, d.mean().item()
    assert (d > 0.5 * lr).float().mean().item() <= 2e-2, (d > 0.5 * lr).float().mean().item()
    assert d.max().item() <= 2 * steps * lr * 1.01, d.max().item()       # the hard bound: lr per step, either way


@pytest.mark.timeout(600)
def test_two_rank_graphed_step_matches_eager_deterministic(tmp_path):
    """The same comparison with SELD_DETERMINISTIC=1 in both ranks: every reduction in a fixed order, so the recorded
    three-graph step and the eager step agree at rounding level (the loose bounds above exist for the float-atomic
    ordering noise of the default mode only) and two eager runs of the pair of ranks are bit-identical."""
    det = {"SELD_DETERMINISTIC": "1"}
    for d in ("e", "e2", "g"):
        (tmp_path / d).mkdir()
    e0, e1 = _run_ranks(tmp_path / "e", "tiny_DQ", "eager", 4, det)
    f0, _ = _run_ranks(tmp_path / "e2", "tiny_DQ", "eager", 4, det)
    g0, g1 = _run_ranks(tmp_path / "g", "tiny_DQ", "graph", 3, det)
    assert torch.equal(e0["param"], e1["param"]) and torch.equal(g0["param"], g1["param"])
    assert torch.equal(e0["param"], f0["param"]), "two deterministic runs differ"
    assert e0["losses"] == f0["losses"]
    assert np.allclose(g0["losses"], e0["losses"][1:], rtol=1e-6, atol=0), (g0["losses"], e0["losses"])
    d = (g0["param"] - e0["param"]).abs()
    assert float(d.max()) <= 1e-6 * float(e0["param"].abs().max()), float(d.max())


def _one_process_steps(mode, steps, dropout):
    T, H = pkg().train, pkg().hip_ops
    case = next(c for c in MODEL_CASES if c["name"] == "tiny_DQ")
    if dropout:
        case = dict(case, dropout_perc=0.3, spatial_dropout_rate=0.5)
    torch.manual_seed(5)
    H.philox.set_offset(0)
    m = build_model(case)
    O.closed_form_fill_(list(m.state_dict().items()))
    m = m.to(DEV).train()
    opt = T.FlatAdam(m.parameters(), lr=1e-3)
    x = O.closed_form_input((case["B"], case["input_channels"], case["freq_dim"], case["time_dim"])).to(DEV)
    target = train_target(case).to(DEV)
    n_sed = int(case["output_classes"] * 3)
    losses = []
    if mode == "graph":
        runner = T.GraphedTrainStep(m, opt, x, target, n_sed, 1.0, 5.0, warmup=1)
        assert len(runner.graphs) == 1
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
    return losses, opt.flat_param.detach().clone(), {k: v.clone() for k, v in m.state_dict().items()}, opt


def test_graphed_step_matches_eager_step():
    """Five steps recorded-and-replayed against five eager steps (dropout off): losses, parameters, BatchNorm running
    statistics and num_batches_tracked; Adam's bias correction follows the device-resident step counter."""
    le, pe, sde, _ = _one_process_steps("eager", 5, False)
    lg, pg, sdg, opt = _one_process_steps("graph", 5, False)
    # the first steps agree to rounding; later ones can part by ~1e-3 when a float-atomic ordering difference flips
    # a pooling arg-max (two eager runs do the same: tools/diag_r2.py)
    assert np.allclose(lg[:3], le[1:4], rtol=2e-4) and np.allclose(lg, le[1:], rtol=3e-3), (lg, le)
    # Parameters: Adam moves a parameter by up to lr per step whatever the size of its gradient, so one whose gradient is
    # float-atomic noise around zero can end anywhere within 2 * steps * lr of its twin (observed maxima 3e-3..6e-3 from
    # run to run).  Bound the bulk tightly and the stragglers by that hard limit.
    # Two EAGER runs already part the same way (tools/chaos_check.py: mean 3.1e-5, max 6e-3 -- from step 4 on there are two
    # trajectories, 5.1294 / 5.1332 at step 5, depending on which way one noise-level decision falls), so the mean is
    # held to 0.1 lr, not to rounding.
    d = (pg - pe).abs()
    lr, steps = 1e-3, 5
    assert d.mean().item() <= 0.1 * lr, d.mean().item()
    assert (d > 0.5 * lr).float().mean().item() <= 2e-2, (d > 0.5 * lr).float().mean().item()
    assert d.max().item() <= 2 * steps * lr * 1.01, d.max().item()
    assert opt.step_count == 5
    for k in sde:
        if k.endswith("num_batches_tracked"):
            assert int(sdg[k]) == int(sde[k]) == (0 if "batch_gate1" in k else 5), k     # batch_gate1 is never used (model.py:90)
        elif "running_" in k:
            # (small entries of a running mean move by the same absolute amount as large ones between the two trajectories)
            # 3 % of the tensor's scale (the two trajectories part by ~1 % in the deeper blocks): a replay that skipped ONE
            # of the five updates would be off by ~17 % (momentum 0.1: 1 - 0.9^4 against 1 - 0.9^5)
            assert torch.allclose(sdg[k], sde[k], rtol=5e-3, atol=3e-2 * float(sde[k].abs().max()) + 1e-5), k


def test_graphed_step_draws_fresh_dropout_masks():
    """With dropout on, every replay must see new masks (the Philox base lives in device memory): the losses of
    consecutive replays on the SAME batch differ, and the device counter advanced by the draws of one step each time."""
    H = pkg().hip_ops
    lg, _, _, _ = _one_process_steps("graph", 5, True)
    assert len(set(round(v, 7) for v in lg)) == len(lg), lg
    st = H.philox.state(torch.device(DEV)).cpu()
    per_step = int(st[3])
    assert per_step > 0 and int(st[0]) % per_step == 0 and int(st[0]) // per_step >= 4
    assert H.philox.get_offset() == int(st[0]) + H.philox.offset
