#!/usr/bin/env python3
"""Dataset normalisation on a resident predictor array: achieved HBM rate of each kernel.
(The CPU restatement is timed beside it by tests/test_gpu_ops.py::test_dataset_normalisation_full_size_properties;
tools never import oracle/.)

    python tools/norm_bench.py [--items 48]   # clips of (8, 256, 4800)

Algorithmic bytes: unit norm 64 B per position (8 planes read + written); standardisation 12 B per element
(read for the moments, read + write for the apply).  Prints one JSON line per op."""
import argparse
import importlib
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

HBM_PEAK = 8.0e12


def timed(fn, reps):
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e-3 / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--items", type=int, default=48)
    ap.add_argument("--channels", type=int, default=8)
    ap.add_argument("--freq", type=int, default=256)
    ap.add_argument("--time", type=int, default=4800)
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    H = importlib.import_module(bench.PKG).hip_ops
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(5)
    x = torch.rand(args.items, args.channels, args.freq, args.time, device=dev, generator=g) + 0.05
    positions = args.items * args.freq * args.time
    elements = positions * args.channels

    y = x.clone()
    H.dq_unit_norm_(y)
    t = timed(lambda: H.dq_unit_norm_(y), args.reps)
    print(json.dumps(dict(op="dq_unit_norm", shape=list(x.shape), ms=t * 1e3, bytes=64 * positions,
                          achieved_GBps=64 * positions / t / 1e9, frac_of_hbm_peak=64 * positions / t / HBM_PEAK,
                          items_per_s=args.items / t)))

    y = x.clone()
    H.group_standardize_(y, 0, args.channels)
    t = timed(lambda: H.group_standardize_(y, 0, args.channels), args.reps)
    print(json.dumps(dict(op="group_standardize", shape=list(x.shape), ms=t * 1e3, bytes=12 * elements,
                          achieved_GBps=12 * elements / t / 1e9, frac_of_hbm_peak=12 * elements / t / HBM_PEAK,
                          items_per_s=args.items / t)))


if __name__ == "__main__":
    main()
