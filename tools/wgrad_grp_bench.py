#!/usr/bin/env python3
"""Stand-alone timing of the grouped persistent weight gradient (csrc/hcq_wgrad_grp.hip) at the benchmark's shapes
(config 3, batch 32) against the per-layer kernels it replaces.  Prints microseconds per call (HIP events, median of 5)."""
import os
import sys
import statistics

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import seld_amd  # noqa: E402

H = seld_amd.hip_ops
dev = torch.device("cuda:0")
B = int(os.environ.get("B", "32"))


def job(shape, cout, k, pad, dil):
    kk = (k,) if isinstance(k, int) else k
    desc = H.make_conv_desc(tuple(shape), cout, 8, kk, 1, pad, dil)
    x = torch.randn(shape, device=dev)
    dy = torch.randn((shape[0], cout) + tuple(shape[2:]), device=dev)
    dws = [torch.zeros((cout // 8, shape[1] // 8) + tuple(kk), device=dev) for _ in range(8)]
    return desc, x, dy, dws


def timeit(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return statistics.median(ts)


def per_layer(jobs):
    for desc, x, dy, dws in jobs:
        H.conv_bwd_weight(desc, x, dy, tuple(dws[0].shape), False, into=dws)


dil = (1, 1, 2, 3, 5, 8, 13, 21, 34, 55)
groups = {
    "tcn_1x3 x20": [job((B, 192, 512), 384, 3, d, d) for d in dil for _ in range(2)],
    "tcn_1x1 x19": [job((B, 384, 512), 192, 1, 0, 1) for _ in range(19)],
    "cnn.1+cnn.2": [job((B, 192, 16, 512), 192, (3, 3), 1, 1), job((B, 192, 2, 512), 192, (3, 3), 1, 1)],
    "tcn.conv1+conv2": [job((B, 192, 256), 384, 3, 1, 1), job((B, 384, 128), 384, 3, 1, 1)],
}
which = sys.argv[1:] or list(groups)
for name in which:
    jobs = groups[name]
    t_new = timeit(lambda: H.wgrad_group(jobs))
    t_old = timeit(lambda: per_layer(jobs))
    fl = sum(H.conv_work(j[0], 2)[0] for j in jobs)
    print(f"{name:18s} grouped {t_new:9.1f} us ({fl / t_new / 1e6:6.1f} TF-eq)   per-layer {t_old:9.1f} us ({fl / t_old / 1e6:6.1f} TF-eq)", flush=True)
