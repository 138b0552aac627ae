#!/usr/bin/env python3
"""Post-processing + test metrics for a resident test set: kernel time and achieved HBM rate.
Algorithmic bytes: 32 n per frame (n = classes * overlaps = 42: 1344 B), read once.

    python tools/metrics_bench.py [--clips 500] [--frames 600]
"""
import argparse
import importlib
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from tests.golden.cases import metric_inputs  # noqa: E402  (closed-form inputs only)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clips", type=int, default=500)
    ap.add_argument("--frames", type=int, default=600)
    ap.add_argument("--reps", type=int, default=50)
    args = ap.parse_args()
    H = importlib.import_module(bench.PKG).hip_ops
    dev = torch.device("cuda", 0)
    sed, doa, target = (torch.from_numpy(a).to(dev) for a in metric_inputs(args.clips, args.frames, 21, "mixed"))
    acc = H.metrics_new(dev)
    for _ in range(5):
        H.metrics_accumulate(acc, sed, doa, target, args.frames)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(args.reps):
        H.metrics_accumulate(acc, sed, doa, target, args.frames)
    b.record()
    torch.cuda.synchronize()
    t = a.elapsed_time(b) * 1e-3 / args.reps
    nbytes = (sed.numel() + doa.numel() + target.numel()) * 4
    print(json.dumps(dict(op="metrics_accumulate", clips=args.clips, frames=args.frames, ms=t * 1e3, bytes=nbytes,
                          achieved_GBps=nbytes / t / 1e9, frac_of_hbm_peak=nbytes / t / 8e12,
                          recordings_per_s=args.clips / t)))


if __name__ == "__main__":
    main()
