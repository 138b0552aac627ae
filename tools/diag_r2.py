#!/usr/bin/env python3
"""Round-2 diagnostics (GPU): first-layer gradient at config widths, eager/graph trajectories, side-stream race."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import seld_oracle as O          # noqa: E402
from tests.golden.cases import MODEL_CASES, train_target   # noqa: E402
from tests.helpers import build_model, pkg   # noqa: E402

DEV = "cuda:0"
P = pkg()
H, T, L = P.hip_ops, P.train, P._lib


def prepared(case):
    m = build_model(case)
    O.closed_form_fill_(list(m.state_dict().items()))
    return m.to(DEV)


def first_layer(case_name):
    case = next(c for c in MODEL_CASES if c["name"] == case_name)
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", f"model_{case_name}.npz")))
    x = O.closed_form_input((case["B"], case["input_channels"], case["freq_dim"], case["time_dim"])).to(DEV)
    target = train_target(case).to(DEV)
    names = str(g["train.param_names"]).split("\n")
    cks = g["train.grad_checksums"]
    for fused in (True, False):
        if fused:
            os.environ.pop("SELD_NO_FUSED_STAGE0", None)
        else:
            os.environ["SELD_NO_FUSED_STAGE0"] = "1"
        m = prepared(case).train()
        opt = T.FlatAdam(m.parameters(), lr=1e-4)
        opt.zero_grad()
        sed, doa = m(x)
        loss = T.seld_loss_fn(sed, doa, target, 42, 1.0, 5.0)
        loss.backward()
        H.join_side_stream()
        torch.cuda.synchronize()
        params = dict(m.named_parameters())
        print(f"--- {case_name} fused_stage0={fused} loss {loss.item():.6f} (ref {float(g['train.loss'][0]):.6f})")
        bad = 0
        for i, n in enumerate(names):
            gr = params[n].grad
            if np.isnan(cks[i, 0]):
                continue
            got2 = (gr.double() ** 2).sum().item()
            ratio = got2 / max(cks[i, 1], 1e-300)
            if not (0.9 < ratio < 1.1):
                bad += 1
                if bad <= 12:
                    print(f"   {n:60s} sumsq {got2:.3e} ref {cks[i, 1]:.3e}")
        print("   params with sum-of-squares off by > 10 %:", bad, "of", len(names))
        for k in g:
            if k.startswith("train.grad."):
                ref = g[k].astype(np.float64)
                got = params[k[len("train.grad."):]].grad.detach().cpu().double().numpy()
                print(f"   {k:70s} max|ref| {np.abs(ref).max():.3e} max err {np.abs(got - ref).max():.3e}")
    os.environ.pop("SELD_NO_FUSED_STAGE0", None)


def trajectories():
    case = next(c for c in MODEL_CASES if c["name"] == "tiny_DQ")
    x = O.closed_form_input((case["B"], case["input_channels"], case["freq_dim"], case["time_dim"])).to(DEV)
    target = train_target(case).to(DEV)

    def run(mode, steps=6):
        torch.manual_seed(5)
        m = prepared(case).train()
        opt = T.FlatAdam(m.parameters(), lr=1e-3)
        losses = []
        if mode == "graph":
            r = T.GraphedTrainStep(m, opt, x, target, 42, 1.0, 5.0, warmup=1)
            losses.append(float("nan"))
            for _ in range(steps - 1):
                losses.append(float(r().item()))
        else:
            for _ in range(steps):
                opt.zero_grad()
                sed, doa = m(x)
                loss = T.seld_loss_fn(sed, doa, target, 42, 1.0, 5.0)
                loss.backward()
                opt.step()
                losses.append(float(loss.item()))
        torch.cuda.synchronize()
        return losses
    for mode in ("eager", "eager", "graph", "graph", "eager"):
        print(mode, ["%.6f" % v for v in run(mode)])
    os.environ["SELD_WGRAD_SIDE_STREAM"] = "0"
    for mode in ("eager", "graph"):
        print("side stream off", mode, ["%.6f" % v for v in run(mode)])
    os.environ.pop("SELD_WGRAD_SIDE_STREAM")


def race():
    case = next(c for c in MODEL_CASES if c["name"] == "tiny_DQ")
    x = O.closed_form_input((case["B"], case["input_channels"], case["freq_dim"], case["time_dim"])).to(DEV)
    target = train_target(case).to(DEV)

    class NoKeep(list):
        def append(self, t):
            pass

    def grads(side, lag, keep=True):
        os.environ["SELD_WGRAD_SIDE_STREAM"] = "1" if side else "0"
        saved = H._side["keep"]
        if not keep:
            H._side["keep"] = NoKeep()
        m = prepared(case).train()
        opt = T.FlatAdam(m.parameters(), lr=1e-4)
        opt.zero_grad()
        sed, doa = m(x)
        loss = T.seld_loss_fn(sed, doa, target, 42, 1.0, 5.0)
        if lag:
            if H._side["stream"] is None:
                H._side["stream"] = torch.cuda.Stream()
            with torch.cuda.stream(H._side["stream"]):
                torch.cuda._sleep(400_000_000)
        loss.backward()
        H.join_side_stream()
        torch.cuda.synchronize()
        H._side["keep"] = saved
        return opt.flat_grad.detach().clone()
    ref = grads(False, False)
    for keep in (True, False):
        got = grads(True, True, keep)
        print(f"race: keep={keep}: max err {float((got - ref).abs().max()):.3e} of scale {float(ref.abs().max()):.3e}")
    os.environ.pop("SELD_WGRAD_SIDE_STREAM")


if __name__ == "__main__":
    what = sys.argv[1:] or ["first", "traj", "race"]
    if "first" in what:
        first_layer("c3w_train")
        first_layer("tiny_DQ")
    if "traj" in what:
        trajectories()
    if "race" in what:
        race()
