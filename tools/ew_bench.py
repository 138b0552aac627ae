#!/usr/bin/env python3
"""Stand-alone device timing of the TCN BatchNorm / gate backward kernels at the c3 shape, through the C-ABI:
one-pass (one workgroup per channel) against reduce + apply."""
import os, sys, importlib, ctypes
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
pkg = importlib.import_module(bench.PKG)
L = pkg._lib
dev = torch.device("cuda:0")
N, C, S = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (32, 384, 512)))
torch.manual_seed(0)
r = lambda *s: torch.randn(*s, device=dev)
dy, x, y, yf, yg, dy2 = r(N, C, S), r(N, C, S), torch.tanh(r(N, C, S)), r(N, C, S), r(N, C, S), r(N, C, S)
mean, invstd, gamma, beta = r(C) * 0.1, torch.rand(C, device=dev) + 0.5, torch.rand(C, device=dev) + 0.5, r(C) * 0.1
mask = (torch.rand(N * C, device=dev) > 0.5).float() * 2
red = torch.zeros(4 * C, device=dev); dx = torch.empty_like(x); dyf = torch.empty_like(x); dyg = torch.empty_like(x)
lib, p, st = L.lib(), L.ptr, L.current_stream()
def timed(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
bn = (p(mean), p(invstd), p(gamma), p(beta))
mb = N * C * S * 4 / 1e6
t = timed(lambda: L.check(lib.seld_bn_act_bwd_fused(p(dy), p(x), p(y), N, C, S, p(mean), p(invstd), p(gamma), 2, p(red), p(None), p(dx), st), "f"))
print(f"bn_act one-pass            {t:7.1f} us  {4 * mb / t:.2f} TB/s algorithmic (3 reads + 1 write of {mb:.1f} MB)")
t = timed(lambda: L.check(lib.seld_bn_act_bwd_fused(p(dy), p(x), p(y), N, C, S, p(mean), p(invstd), p(gamma), 2, p(red), p(dy2), p(dx), st), "f"))
print(f"bn_act one-pass + dy2      {t:7.1f} us  {5 * mb / t:.2f} TB/s")
t1 = timed(lambda: L.check(lib.seld_bn_act_bwd_reduce(p(dy), p(x), p(y), N, C, S, *bn, 2, p(red), st), "r"))
t2 = timed(lambda: L.check(lib.seld_bn_act_bwd_apply(p(dy), p(x), p(y), N, C, S, *bn, 2, p(red), 1, p(dx), st), "a"))
print(f"bn_act reduce + apply      {t1:7.1f} + {t2:.1f} us")
g = (p(dy), p(yf), p(yg), N, C, S, *bn, *bn, p(mask))
t = timed(lambda: L.check(lib.seld_gate_bwd_fused(*g, p(red), p(dyf), p(dyg), st), "g"))
print(f"gate one-pass              {t:7.1f} us  {5 * mb / t:.2f} TB/s algorithmic (3 reads + 2 writes)")
t1 = timed(lambda: L.check(lib.seld_gate_bwd_reduce(*g, p(red), st), "r"))
t2 = timed(lambda: L.check(lib.seld_gate_bwd_apply(*g, p(red), 1, p(dyf), p(dyg), st), "a"))
print(f"gate reduce + apply        {t1:7.1f} + {t2:.1f} us")
