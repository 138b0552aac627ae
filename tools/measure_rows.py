#!/usr/bin/env python3
"""Measures the rows of SURVEY 8 that bench.py does not carry (offline / widening rows) on the GPU box and prints a
markdown table for profiles/rNN_rows_measured.md:  spectrum_fast on the 60-s clip (utility_functions.py:129-155),
dataset normalisation (train.py:242-408), decode + metrics (train.py:84-166), full-clip inference (train.py:84-104).
HIP events, median of 10 after a warm-up; bytes are the algorithmic ones stated per row."""
import os
import statistics
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import seld_amd  # noqa: E402

H, UF, T, M = seld_amd.hip_ops, seld_amd.utility_functions, seld_amd.train, seld_amd.model
dev = torch.device("cuda:0")


def timeit(fn, reps=10):
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


rows = []
# ---- spectrum_fast, 8 channels x 60 s at 32 kHz -> (16, 256, 4800)
rng = np.random.RandomState(5)
x = torch.from_numpy((rng.randn(8, 32000 * 60)).astype(np.float32)).to(dev)
out = UF.spectrum_fast(x, nperseg=512, noverlap=112)
us = timeit(lambda: UF.spectrum_fast(x, nperseg=512, noverlap=112))
nbytes = x.numel() * 4 + out.numel() * 4
rows.append(("spectrum_fast (8, 1 920 000) -> (16, 256, 4800), `stft_kernel`", us, nbytes, "input + magnitude/phase output once"))
# ---- dataset normalisation, 48 clips of (8, 256, 4800)
xs = torch.rand(48, 8, 256, 4800, device=dev) + 0.1
us = timeit(lambda: H.dq_unit_norm_(xs), 5)
rows.append(("dq unit norm, 48 x (8, 256, 4800), in place", us, xs.numel() * 8, "read + write once"))
us = timeit(lambda: H.group_standardize_(xs, 0, 8), 5)
rows.append(("group standardise (moments + apply), same array", us, xs.numel() * 12, "read twice, write once"))
del xs
# ---- decode + metrics, 500 recordings x 600 frames
g = torch.Generator().manual_seed(3)
sed = torch.rand(500, 600, 42, generator=g).to(dev)
doa = (torch.rand(500, 600, 126, generator=g) * 2 - 1).to(dev)
tgt = torch.cat(((torch.rand(500, 600, 42, generator=g) < 0.08).float(), torch.rand(500, 600, 126, generator=g) * 2 - 1), 2).to(dev)
acc = H.metrics_new(dev)
us = timeit(lambda: H.metrics_accumulate(acc, sed, doa, tgt, 600), 5)
rows.append(("decode + L3DAS21 / DCASE21 counters, 500 x 600 frames", us, (sed.numel() + doa.numel() + tgt.numel()) * 4, "inputs once"))
# ---- full-clip inference, config-3 widths, T = 4800
np.random.seed(1)
torch.manual_seed(1)
kw = dict(time_dim=4800, freq_dim=128, input_channels=8, output_classes=14, domain='DQ', domain_classifier='DQ',
          cnn_filters=[192] * 3, pool_size=[[8, 2], [8, 2], [2, 2]], pool_time='TCN', D=[10], dilation_mode='fibonacci', G=384,
          U=192, V=[384, 384], V_kernel_size=3, fc_layers=[384], fc_activations='linear', fc_dropout='Last', dropout_perc=0.3,
          class_overlaps=3, use_bias_conv=0, use_bias_linear=1, batch_norm='BN')
m = M.SELD_Model(**kw).to(dev).eval()
xc = torch.randn(1, 8, 128, 4800, device=dev)
with torch.no_grad():
    us = timeit(lambda: m(xc), 5)
rows.append(("full-clip inference B = 1, T = 4800 (config-3 widths, F = 128)", us, None, "whole forward"))

print("| row | median us | algorithmic bytes | GB/s | of 8 TB/s | bytes counted |")
print("|---|---|---|---|---|---|")
for name, us, nb, what in rows:
    if nb is None:
        print(f"| {name} | {us:.0f} | - | - | - | {what} |")
    else:
        gbs = nb / us / 1e3
        print(f"| {name} | {us:.0f} | {nb / 1e6:.1f} MB | {gbs:.0f} | {gbs / 8000 * 100:.1f} % | {what} |")
