#!/usr/bin/env python3
"""Weight-gradient kernels at the config-3 layer shapes: the 24-product row kernel against the 48-product one
(default; SELD_HCQ_WGRAD_ROW=1 selects the 24-product one), single and pair launches, HIP events on the launch stream."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import seld_amd
H = seld_amd.hip_ops
dev = torch.device("cuda:0")
SHAPES = {"cnn1": ((32, 192, 16, 512), 192, (3, 3), 1, 1), "cnn2": ((32, 192, 2, 512), 192, (3, 3), 1, 1),
          "tcn_k3": ((32, 192, 512), 384, (3,), 5, 5), "tcn_k1": ((32, 384, 512), 192, (1,), 0, 1)}
only = sys.argv[1].split(",") if len(sys.argv) > 1 else list(SHAPES)
for name, (shape, cout, k, pad, dil) in SHAPES.items():
    if name not in only:
        continue
    desc = H.make_conv_desc(shape, cout, 8, k, 1, pad, dil)
    x = torch.randn(shape, device=dev)
    yshape = (shape[0], cout) + tuple(shape[2:])
    dyA, dyB = torch.randn(yshape, device=dev), torch.randn(yshape, device=dev)
    wshape = (cout // 8, shape[1] // 8) + tuple(k)
    gA = [torch.zeros(wshape, device=dev) for _ in range(8)]
    gB = [torch.zeros(wshape, device=dev) for _ in range(8)]
    for pair in (False, True):
        if pair and len(shape) == 4:
            continue
        fn = (lambda: H.hcq_wgrad_acc(desc, x, dyA, gA, dyB, gB)) if pair else (lambda: H.hcq_wgrad_acc(desc, x, dyA, gA))
        if not H._hcq_wgrad_row_bytes(desc, 2 if pair else 1):
            fn = (lambda: H.conv_bwd_weight(desc, x, dyA, wshape, False, into=gA)) if not pair else None
        if fn is None:
            continue
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        print(f"{name:8s} {'pair' if pair else 'single'}: {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us")
