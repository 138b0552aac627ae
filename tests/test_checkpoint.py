"""Checkpoint format, resume and file rotation (train.py:26-81, 577-616, 671-684).

tests/golden/ref_checkpoint_tiny_DQ.pt was written by the REFERENCE's save_model; ckpt.npz holds what the reference
computed after loading it with its own load_model (make_golden.gen_ckpt)."""
import os

import numpy as np
import pytest
import torch

from tests.golden.cases import MODEL_CASES, train_target
from tests.helpers import build_model, pkg

HERE = os.path.dirname(os.path.abspath(__file__))
REF_CKPT = os.path.join(HERE, "golden", "ref_checkpoint_tiny_DQ.pt")
CASE = next(c for c in MODEL_CASES if c["name"] == "tiny_DQ")


def _trio(device="cpu", lr=1e-3):
    T = pkg().train
    m = build_model(CASE).to(device)
    opt = T.FlatAdam(m.parameters(), lr=lr)
    sch = T.StepLR(opt, 2, 0.5)
    return m, opt, sch


def test_reference_checkpoint_loads(golden):
    """Model, Adam moments / step / lr, StepLR counters, loop state and RNG states of a reference checkpoint."""
    T = pkg().train
    g = golden("ckpt")
    m, opt, sch = _trio(lr=123.0)
    torch.manual_seed(999)
    state = T.load_model(m, opt, REF_CKPT, False, "cpu", sch)
    raw = torch.load(REF_CKPT, map_location="cpu", weights_only=False)
    assert state == raw["state"] and state["step"] == 2 and state["epochs"] == 2
    for k, v in m.state_dict().items():
        assert torch.equal(v, raw["model_state_dict"][k]), k
    assert opt.step_count == 2
    assert abs(opt.param_groups[0]["lr"] - float(g["lr"][0])) < 1e-15 and abs(float(g["lr"][0]) - 5e-4) < 1e-12
    assert sch.last_epoch == 2 and sch.base_lr == 1e-3 and sch.step_size == 2 and sch.gamma == 0.5
    first = opt.params[0]
    assert torch.equal(opt.exp_avg_sq[:first.numel()].view(first.shape), raw["optimizer_state_dict"]["state"][0]["exp_avg_sq"])
    off = 0
    for i, p in enumerate(opt.params):           # every parameter's moments, including those torch has no entry for
        st = raw["optimizer_state_dict"]["state"].get(i)
        got = opt.exp_avg[off:off + p.numel()].view(p.shape)
        assert torch.equal(got, st["exp_avg"] if st is not None else torch.zeros_like(got)), i
        off += p.numel()
    assert torch.equal(torch.get_rng_state(), raw["random_states"][1])
    assert all(np.array_equal(a, b) for a, b in zip(np.random.get_state()[1:2], raw["random_states"][0][1:2]))
    sch.step()                                   # epoch 3 of a step-2 schedule keeps 5e-4, epoch 4 halves it
    assert abs(opt.param_groups[0]["lr"] - 5e-4) < 1e-15
    sch.step()
    assert abs(opt.param_groups[0]["lr"] - 2.5e-4) < 1e-15


def test_our_checkpoint_is_readable_by_torch_adam_and_steplr(tmp_path):
    """What the reference's load_model does with a file (train.py:56-81): the same keys, and state dicts that
    torch.optim.Adam / StepLR over the reference's parameter list accept."""
    T = pkg().train
    m, opt, sch = _trio()
    T.load_model(m, opt, REF_CKPT, False, "cpu", sch)
    opt.exp_avg.mul_(1.5)                         # make it differ from the file it came from
    path = str(tmp_path / "sub" / "checkpoint")
    state = {"step": 7, "epochs": 3, "best_loss": 0.5, "worse_epochs": 0, "best_epoch": 3}
    T.save_model(m, opt, state, path, sch)
    ours = torch.load(path, map_location="cpu", weights_only=False)
    ref = torch.load(REF_CKPT, map_location="cpu", weights_only=False)
    assert list(ours.keys()) == list(ref.keys())
    assert set(ours["optimizer_state_dict"]["param_groups"][0]) == set(ref["optimizer_state_dict"]["param_groups"][0])
    assert set(ours["scheduler_state_dict"]) >= {"step_size", "gamma", "base_lrs", "last_epoch", "_step_count", "_last_lr"}
    params = [torch.nn.Parameter(p.detach().clone()) for p in m.parameters()]
    topt = torch.optim.Adam(params, lr=1.0)
    tsch = torch.optim.lr_scheduler.StepLR(topt, step_size=99, gamma=0.1)
    topt.load_state_dict(ours["optimizer_state_dict"])
    tsch.load_state_dict(ours["scheduler_state_dict"])
    assert topt.param_groups[0]["lr"] == opt.param_groups[0]["lr"] and tsch.step_size == 2 and tsch.last_epoch == 2
    off = 0
    for p, tp in zip(opt.params, params):
        assert torch.equal(topt.state[tp]["exp_avg"], opt.exp_avg[off:off + p.numel()].view(p.shape))
        assert float(topt.state[tp]["step"]) == 2.0
        off += p.numel()
    for tp in params:                             # and torch can step with it
        tp.grad = torch.ones_like(tp)
    topt.step()
    # round trip through our own loader, and the pre-'state' format (train.py:72-75)
    m2, opt2, sch2 = _trio()
    assert T.load_model(m2, opt2, path, False, "cpu", sch2) == state
    assert torch.equal(opt2.exp_avg, opt.exp_avg) and torch.equal(opt2.exp_avg_sq, opt.exp_avg_sq) and opt2.step_count == 2
    del ours["state"]
    ours["step"] = 11
    ours["model_state_dict"] = {"module." + k: v for k, v in ours["model_state_dict"].items()}   # DataParallel files
    torch.save(ours, path)
    assert T.load_model(m2, None, path, False, "cpu") == {"step": 11}
    with pytest.raises(ValueError):
        bad = dict(ref["optimizer_state_dict"])
        bad["param_groups"] = [dict(bad["param_groups"][0], params=[0, 1, 2])]
        opt2.load_state_dict(bad)


def test_checkpoint_rotation_follows_the_reference_loop(tmp_path):
    """Validation losses 3, 2, 2.5, 1 with checkpoint_step 2, walked through train.py:577-616, 671-684 by hand:
       e1: best (3);  best_of_checkpoint saved too (first finite loss)                         -> holds epoch 1
       e2: best (2);  previous best copied to best_of_checkpoint (epoch 1); periodic copy      -> holds epoch 1
       e3: worse;     2.5 < 3 and not the best -> best_of_checkpoint saved                      -> holds epoch 3
       e4: best (1);  previous best (epoch 2) copied to best_of_checkpoint; periodic copy      -> holds epoch 2"""
    T = pkg().train
    model = torch.nn.Linear(4, 3)
    model.model_name = "toy"
    opt = T.FlatAdam(model.parameters(), lr=1e-3)
    sch = T.StepLR(opt, 2, 0.5)
    model_dir = str(tmp_path / "toy")
    rot = T.CheckpointRotation(model_dir, "toy", checkpoint_step=2)
    state = {"step": 0, "worse_epochs": 0, "epochs": 0, "best_loss": np.inf, "best_epoch": 0, "best_test_epoch": 0}

    def epochs_in(path):
        return torch.load(path, map_location="cpu", weights_only=False)["state"]["epochs"]

    expect_improved = [True, True, False, True]
    expect_boc = [1, 1, 3, 2]
    for epoch, (val, imp, boc) in enumerate(zip([3.0, 2.0, 2.5, 1.0], expect_improved, expect_boc), start=1):
        state["epochs"] += 1
        assert rot.end_of_epoch(model, opt, sch, state, epoch, val) is imp
        assert epochs_in(rot.checkpoint_path) == epoch
        assert epochs_in(rot.best_of_checkpoint_path) == boc, epoch
    assert state["best_epoch"] == 4 and state["best_loss"] == 1.0 and state["worse_epochs"] == 0
    assert epochs_in(rot.best_path) == 4
    assert rot.best_epoch_checkpoint == 2 and rot.best_loss_checkpoint == 2.0
    d2, d4 = model_dir + "checkpoint_epoch_2/", model_dir + "checkpoint_epoch_4/"
    assert sorted(os.listdir(d2)) == ["checkpoint_best_epoch_2", "checkpoint_best_model_checkpoint_epoch_1", "checkpoint_epoch_2"]
    assert sorted(os.listdir(d4)) == ["checkpoint_best_epoch_4", "checkpoint_best_model_checkpoint_epoch_2", "checkpoint_epoch_4"]
    assert epochs_in(d4 + "checkpoint_best_model_checkpoint_epoch_2") == 2
    # worse epochs accumulate; an equal loss counts as worse (train.py:590 uses >=)
    state["epochs"] += 1
    assert rot.end_of_epoch(model, opt, sch, state, 5, 1.0) is False and state["worse_epochs"] == 1


@pytest.mark.gpu
def test_resume_from_reference_checkpoint_matches_reference(golden):
    """Load the reference's checkpoint on the device, compare the eval outputs, take ONE training step with the
    restored Adam / StepLR state and compare loss and parameters with what the reference got from the same file."""
    T = pkg().train
    g = golden("ckpt")
    dev = torch.device("cuda:0")
    m, opt, sch = _trio(dev, lr=77.0)
    T.load_model(m, opt, REF_CKPT, True, dev, sch)
    from oracle.seld_oracle import closed_form_input
    x = closed_form_input((CASE["B"], CASE["input_channels"], CASE["freq_dim"], CASE["time_dim"])).to(dev)
    target = train_target(CASE).to(dev)
    m.eval()
    with torch.no_grad():
        sed, doa = m(x)
    assert np.abs(sed.cpu().numpy() - g["sed"]).max() < 1e-4 and np.abs(doa.cpu().numpy() - g["doa"]).max() < 1e-4
    before = {k: v.detach().clone() for k, v in m.named_parameters()}
    m.train()
    opt.zero_grad()
    sed, doa = m(x)
    loss = T.seld_loss_fn(sed, doa, target, int(CASE["output_classes"] * 3), 1.0, 5.0)
    loss.backward()
    opt.step()
    assert abs(loss.item() - float(g["loss3"][0])) < 1e-4 * abs(float(g["loss3"][0]))
    lr = float(g["lr"][0])
    params = dict(m.named_parameters())
    for key in [k for k in g if k.startswith("after.")]:
        name = key[len("after."):]
        ref_after, ref_before = g[key], g["before." + name]
        assert np.array_equal(before[name].cpu().numpy(), ref_before)
        got = params[name].detach().cpu().numpy()
        moved = np.abs(ref_after - ref_before).max()
        assert moved > 0.1 * lr, (name, moved)                      # the step is visible ...
        assert np.abs(got - ref_after).max() < 0.02 * lr, (name, np.abs(got - ref_after).max(), lr)   # ... and matches
    # every parameter through its checksums.  Adam normalises the step, so an element whose gradient is rounding
    # noise (sqrt(exp_avg_sq) < 1e-6 here: e.g. two channels of cnn.2.1.bias sit at 1e-9) moves by up to lr in a
    # direction that depends on the summation order; such elements get a full lr of slack, the others 2 %
    raw = torch.load(REF_CKPT, map_location="cpu", weights_only=False)["optimizer_state_dict"]["state"]
    names = str(g["param_names"]).split("\n")
    for i, ((s1, s2), name) in enumerate(zip(g["param_checksums"], names)):
        p = params[name].detach().double()
        noisy = int((raw[i]["exp_avg_sq"].sqrt() < 1e-6).sum()) if i in raw else 0
        slack = lr * (noisy + 0.02 * p.numel())
        assert abs(p.sum().item() - s1) <= slack + 1e-6, (name, noisy)
        assert abs((p ** 2).sum().item() - s2) <= 2 * p.abs().max().item() * slack + 1e-6, (name, noisy)


def test_after_test_keeps_best_model_on_test(tmp_path):
    """train.py:655-670 by hand: a Global SELD of 0.8 (<= 1) saves, 0.9 does not, 0.8 again (<=) saves; 'test_best'
    mode reports the best epoch while the 'new best' flag is up and the best-of-checkpoint epoch afterwards."""
    T = pkg().train
    model = torch.nn.Linear(4, 3)
    opt = T.FlatAdam(model.parameters(), lr=1e-3)
    rot = T.CheckpointRotation(str(tmp_path / "toy"), "toy", checkpoint_step=0)
    state = {"step": 0, "worse_epochs": 0, "epochs": 1, "best_loss": np.inf, "best_epoch": 0, "best_test_epoch": 0}
    rot.end_of_epoch(model, opt, None, state, 1, 2.0)
    assert rot.test_checkpoint("test_best") == (rot.best_path, None)
    res = [0.0] * 16
    res[10] = 0.8
    assert rot.after_test(model, opt, None, state, 1, res, "test_best") and state["best_test_epoch"] == 1
    assert os.path.isfile(rot.checkpoint_path + "_best_model_on_Test") and rot.new_best is False
    assert rot.test_checkpoint("test_best") == (rot.best_of_checkpoint_path, 1)
    assert rot.test_checkpoint("test_last") == (None, None)
    res[10] = 0.9
    assert not rot.after_test(model, opt, None, state, 2, res, "test_last")
    res[10] = 0.8
    assert rot.after_test(model, opt, None, state, 3, res, "test_last") and state["best_test_epoch"] == 3


@pytest.mark.gpu
def test_main_runs_the_reference_loop_end_to_end(tmp_path):
    """train.main on pickled arrays: device-side normalisation, two epochs of training, validation, the test leg with the
    device metrics every epoch, the reference's checkpoint files; then a resume from the last checkpoint."""
    import pickle
    T = pkg().train
    rng = np.random.default_rng(5)
    paths = {}
    for split, n in (("training", 4), ("validation", 2), ("test", 2)):
        x = (rng.random((n, 8, 128, 64)) + 0.05).astype(np.float32)
        act = (rng.random((n, 8, 42)) < 0.15).astype(np.float32)
        loc = rng.uniform(-1, 1, (n, 8, 126)).astype(np.float32) * np.repeat(act, 3, axis=2)
        y = np.concatenate([act, loc], axis=2)
        y[:, 0, 0] = 1.0                                  # at least one reference event (train.py:136 divides by Nref)
        for kind, arr in (("predictors", x), ("target", y)):
            paths[f"{split}_{kind}_path"] = str(tmp_path / f"{split}_{kind}.pkl")
            pickle.dump(arr, open(paths[f"{split}_{kind}_path"], "wb"))
    flags = dict(paths, results_path=str(tmp_path / "res"), checkpoint_dir=str(tmp_path / "ck"), use_cuda="True", gpu_id=0,
                 batch_size=2, epochs=2, min_n_epochs=2, patience=1, test_step=1, checkpoint_step=2, test_mode="test_best",
                 num_frames=8, dataset_normalization="UnitNorm", n_mics=2, domain="DQ", domain_classifier="DQ", phase="False",
                 input_channels=8, time_dim=64, freq_dim=128, output_classes=14, class_overlaps=3,
                 cnn_filters="[16, 16, 16]", pool_size="[[8, 2], [8, 2], [2, 2]]", pool_time="TCN", D="[10]",
                 dilation_mode="fibonacci", G=32, U=16, V="[16, 16]", V_kernel_size=3, fc_layers="[16]",
                 fc_activations="linear", fc_dropout="Last", use_bias_conv="False", use_bias_linear="True", batch_norm="BN",
                 dropout_perc=0.0, spatial_dropout_rate=0.0, lr=1e-3, use_lr_scheduler="True", lr_scheduler_step_size=1,
                 lr_scheduler_gamma=0.5, min_lr=1e-6, TextArgs="none")
    args = T.parse_args([f"--{k}={v}" for k, v in flags.items()])
    state = T.main(args)
    assert state["epochs"] == 2 and state["step"] == 4 and np.isfinite(state["best_loss"])
    ck_root = str(tmp_path / "ck")
    model_dirs = [d for d in os.listdir(ck_root) if os.path.isdir(os.path.join(ck_root, d)) and "checkpoint_epoch" not in d]
    assert len(model_dirs) == 1
    md = os.path.join(ck_root, model_dirs[0])
    files = set(os.listdir(md))
    # (checkpoint_best_model_on_Test appears only once a Global SELD score <= 1 is seen, train.py:520, 655; a net trained
    #  for four steps scores above 1)
    assert {"checkpoint", "checkpoint_best_model", "checkpoint_best_model_of_checkpoint"} <= files
    periodic = md + "checkpoint_epoch_2/"
    assert os.path.isdir(periodic) and "checkpoint_epoch_2" in os.listdir(periodic)
    last = torch.load(os.path.join(md, "checkpoint"), map_location="cpu", weights_only=False)
    assert last["state"]["epochs"] == 2 and abs(last["optimizer_state_dict"]["param_groups"][0]["lr"] - 2.5e-4) < 1e-12
    # resume: one more epoch from the stored state
    flags.update(load_model=os.path.join(md, "checkpoint"), epochs=3, min_n_epochs=3)
    state2 = T.main(T.parse_args([f"--{k}={v}" for k, v in flags.items()]))
    assert state2["epochs"] == 3 and state2["step"] == 6
