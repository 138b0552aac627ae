#!/usr/bin/env python3
"""Which host-side ops launch the small torch kernels (copies, fills, adds) of one training step?
   python tools/op_profile.py [--workload c3] [--filter copy_,fill_,add_,zeros]
Prints, per aten op matching the filter, the call count per step and the Python stack that issued it."""
import argparse
import collections
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c3")
    ap.add_argument("--filter", default="aten::copy_,aten::fill_,aten::add_,aten::add,aten::zero_,aten::clone")
    ap.add_argument("--steps", type=int, default=2)
    args = ap.parse_args()
    pkg = importlib.import_module(bench.PKG)
    T, DP = pkg.train, pkg.dp
    dev = torch.device("cuda", 0)
    w = bench.WORKLOADS[args.workload]
    model = pkg.model.SELD_Model(**bench.model_kwargs(w)).to(dev).train()
    opt = T.FlatAdam(model.parameters(), lr=1e-4)
    sync = DP.FlatGradSync(flat_grad=opt.flat_grad)
    x, target = T.synthetic_batch(w["batch"], w["input_channels"], 128, 512, 42, 1234, dev)
    for _ in range(2):
        DP.dp_train_step(model, opt, sync, x, target, 42, T.seld_loss_fn)
    torch.cuda.synchronize()
    names = set(args.filter.split(","))
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU], with_stack=True) as prof:
        for _ in range(args.steps):
            DP.dp_train_step(model, opt, sync, x, target, 42, T.seld_loss_fn)
        torch.cuda.synchronize()
    counts = collections.Counter()
    for ev in prof.events():
        if ev.name in names:
            frames = [f for f in (ev.stack or []) if "sound-event" in f or "seld" in f or "autograd" in f]
            counts[(ev.name, " <- ".join(s.split("/")[-1] for s in frames[:3]))] += 1
    for (name, where), n in counts.most_common(60):
        print(f"{n / args.steps:7.1f}/step  {name:14s} {where}")


if __name__ == "__main__":
    main()
