#!/usr/bin/env python3
"""Fixed cost vs per-chunk cost of the fast-product conv kernel: time against the number of K chunks."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import seld_amd
H = seld_amd.hip_ops
dev = torch.device("cuda:0")

def timeit(f, iters=30):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

for N in (32, 64, 128):
    for ib in (8, 24, 48, 96):
        xs = (N, 8 * ib, 512)
        x = torch.randn(xs, device=dev)
        ws = [torch.randn(48, ib, 3, device=dev) * 0.1 for _ in range(8)]
        desc = H.make_conv_desc(xs, 384, 8, (3,), 1, 1, 1)
        wp = H.hcq_pack(desc, 0, ws)
        y = torch.empty((N, 384, 512), device=dev)
        t = timeit(lambda: H.hcq_conv(desc, 0, x, wp, (y,)))
        print(f"N {N:4d} IB {ib:3d} chunks {ib // 8:2d}: {t:7.1f} us  label {H.hcq_label(desc, 0)}", flush=True)
