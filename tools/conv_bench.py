#!/usr/bin/env python3
"""Micro-benchmark of the hypercomplex conv kernels at the config-3 layer shapes (HIP events, same stream).
   python tools/conv_bench.py [--iters 20] [--only tcn_k3] [--algebra 8]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import seld_amd  # noqa: E402

H = seld_amd.hip_ops
SHAPES = {
    "cnn0": dict(x=(32, 8, 128, 512), cout=192, k=(3, 3), pad=1, dil=1),
    "cnn0_16ch": dict(x=(16, 16, 128, 512), cout=192, k=(3, 3), pad=1, dil=1),                 # config 4, 16 samples / GPU
    "cnn0_q": dict(x=(32, 8, 128, 512), cout=64, k=(3, 3), pad=1, dil=1, algebra=4),            # config 2
    "cnn0_b16": dict(x=(16, 8, 128, 512), cout=192, k=(3, 3), pad=1, dil=1),                   # config 5, one stream
    "cnn1": dict(x=(32, 192, 16, 512), cout=192, k=(3, 3), pad=1, dil=1),
    "cnn2": dict(x=(32, 192, 2, 512), cout=192, k=(3, 3), pad=1, dil=1),
    "tcn_k3": dict(x=(32, 192, 512), cout=384, k=(3,), pad=5, dil=5),
    "tcn_k3_d55": dict(x=(32, 192, 512), cout=384, k=(3,), pad=55, dil=55),
    "tcn_k1": dict(x=(32, 384, 512), cout=192, k=(1,), pad=0, dil=1),
    "mha_proj": dict(x=(32, 384, 256), cout=384, k=(1,), pad=0, dil=1, algebra=1),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default="")
    ap.add_argument("--algebra", type=int, default=8)
    ap.add_argument("--which", default="fwd,dgrad,wgrad")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    rows = []
    for name, s in SHAPES.items():
        if args.only and name not in args.only.split(","):
            continue
        A = s.get("algebra", args.algebra)
        x = torch.randn(s["x"], device=dev)
        cin = s["x"][1]
        ws = [torch.randn((s["cout"] // A, cin // A) + s["k"], device=dev) * 0.1 for _ in range(A)]
        desc = H.make_conv_desc(s["x"], s["cout"], A, s["k"], 1, s["pad"], s["dil"])
        y = H.conv_fwd(desc, x, ws)
        dy = torch.randn_like(y)
        gw = [torch.zeros_like(w) for w in ws]
        flops, by = H.conv_work(desc, 0)
        st = H.new_stats(s["cout"], dev)
        fns = {"fwd": lambda: H.conv_fwd(desc, x, ws, out=y),
               "fwd_stats": lambda: H.conv_fwd(desc, x, ws, out=y, epilogue=seld_amd._lib.SELD_EPI_STATS, stats=st),
               "dgrad": lambda: H.conv_bwd_data(desc, dy, ws, tuple(x.shape)),
               "wgrad": lambda: H.conv_bwd_weight(desc, x, dy, tuple(ws[0].shape), False, into=gw)}
        for which in args.which.split(","):
            f = fns[which]
            for _ in range(3):
                f()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(args.iters):
                f()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / args.iters * 1e3
            rows.append(dict(layer=name, op=which, us=round(us, 1), tflops=round(flops / us / 1e6, 1),
                             gbs=round(by / us / 1e3, 1)))
            print(json.dumps(rows[-1]), flush=True)


if __name__ == "__main__":
    main()
