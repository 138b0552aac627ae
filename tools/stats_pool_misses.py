#!/usr/bin/env python3
"""Where does a training step still allocate (and zero-fill) a BatchNorm statistics buffer instead of reusing a pooled one?"""
import os, sys, importlib, traceback, collections
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
pkg = importlib.import_module(bench.PKG)
T, DP, H = pkg.train, pkg.dp, pkg.hip_ops
dev = torch.device("cuda:0")
w = bench.WORKLOADS["c3"]
model = pkg.model.SELD_Model(**bench.model_kwargs(w)).to(dev).train()
opt = T.FlatAdam(model.parameters(), lr=1e-4)
sync = DP.BucketedGradSync(opt, model)
x, target = T.synthetic_batch(4, 8, 128, 512, 42, 1234, dev)
step = lambda: DP.dp_train_step(model, opt, sync, x, target, 42, T.seld_loss_fn)
for _ in range(3): step()
torch.cuda.synchronize()
orig = torch.zeros
seen = collections.Counter()
def spy(*a, **k):
    seen["".join(traceback.format_stack(limit=6)[:-1])] += 1
    return orig(*a, **k)
torch.zeros = spy
step()
torch.zeros = orig
for st, c in seen.items():
    print("=" * 20, c); print(st)
print({k: len(v) for k, v in H._stats_pool.items()})
