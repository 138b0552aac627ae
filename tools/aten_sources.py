#!/usr/bin/env python3
"""Which Python lines launch the ATen / runtime kernels still on the training step's hot path (add, copy, fill)?"""
import os, sys, importlib
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
pkg = importlib.import_module(bench.PKG)
T, DP, H = pkg.train, pkg.dp, pkg.hip_ops
dev = torch.device("cuda:0")
w = bench.WORKLOADS["c3"]
import numpy as np
np.random.seed(1); torch.manual_seed(1)
model = pkg.model.SELD_Model(**bench.model_kwargs(w)).to(dev).train()
opt = T.FlatAdam(model.parameters(), lr=1e-4)
sync = DP.BucketedGradSync(opt, model)
x, target = T.synthetic_batch(8, 8, 128, 512, 42, 1234, dev)
step = lambda: DP.dp_train_step(model, opt, sync, x, target, 42, T.seld_loss_fn)
for _ in range(3): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
import collections
agg = collections.Counter()
for ev in prof.events():
    n = ev.name
    if n.startswith("aten::") and any(k in n for k in ("add", "copy_", "fill_", "zero_", "clone", "contiguous", "mul", "cat", "zeros")):
        st = [f for f in (ev.stack or []) if "sound-event" in f or "bench" in f or "train.py" in f]
        agg[(n, (st[0] if st else "<autograd engine / no python frame>") + "  " + str(ev.input_shapes)[:90])] += 1
for (n, where), c in sorted(agg.items(), key=lambda kv: -kv[1])[:40]:
    print(f"{c:4d}  {n:22s} {where}")

print("--- device kernels in one step (name, launches, total us)")
kagg = collections.Counter(); kt = collections.Counter()
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CUDA:
        kagg[ev.name[:90]] += 1; kt[ev.name[:90]] += ev.device_time
tot = sum(kagg.values())
print("total launches", tot)
for n, c in sorted(kagg.items(), key=lambda kv: -kt[kv[0]]):
    print(f"{c:4d} {kt[n]:9.0f}  {n}")
