"""BN + ReLU + max-pool forward at the first CNN stage's size: time and achieved HBM rate (reads y, writes pooled + index)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import seld_amd
H = seld_amd.hip_ops
dev = torch.device("cuda:0")
bn = torch.nn.BatchNorm2d(192).to(dev).eval()
y = torch.randn(32, 192, 128, 512, device=dev)
with torch.no_grad():
    for _ in range(3): H.bn_relu_pool(y, bn, 8, 1)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(20): H.bn_relu_pool(y, bn, 8, 1)
    b.record(); torch.cuda.synchronize()
us = a.elapsed_time(b) / 20 * 1e3
print(f"bn_relu_pool fwd (32,192,128,512) ph=8: {us:.1f} us, {(y.numel()*4*1.078)/us/1e6:.2f} TB/s")
