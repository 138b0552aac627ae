#!/usr/bin/env python3
"""Attention core (csrc/mha_mfma.hip) at the config-3 shape: N = 32, 8 heads x 48, T = 256; forward and backward, us."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import seld_amd  # noqa: E402

H = seld_amd.hip_ops
dev = torch.device("cuda:0")
N, heads, hd, T = 32, 8, 48, 256
E = heads * hd
torch.manual_seed(0)
qkv = (torch.randn(N, 3 * E, T, device=dev) * 0.5).requires_grad_(True)
cot = torch.randn(N, E, T, device=dev)


def timeit(f, iters=30):
    for _ in range(5):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


with torch.no_grad():
    t_f = timeit(lambda: H.mha_core_packed(qkv, heads))


def fb():
    qkv.grad = None
    (H.mha_core_packed(qkv, heads) * cot).sum().backward()


t_fb = timeit(fb)
flops_f = 4.0 * N * heads * T * T * hd
print(f"forward {t_f:.1f} us ({flops_f / t_f / 1e6:.1f} TFLOP/s), forward + backward (incl. mul/sum) {t_fb:.1f} us")
