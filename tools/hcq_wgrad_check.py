#!/usr/bin/env python3
"""Fast-product weight-gradient kernel (hcq_wgrad.hip) against the 16/48-product kernels: values and time."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import seld_amd  # noqa: E402

H, L = seld_amd.hip_ops, seld_amd._lib
dev = torch.device("cuda:0")
SHAPES = [
    ("tcn_k3_d5", 8, (32, 192, 512), 384, (3,), 5, 5),
    ("tcn_k3_d13", 8, (32, 192, 512), 384, (3,), 13, 13),
    ("tcn_k3_d55", 8, (32, 192, 512), 384, (3,), 55, 55),
    ("tcn_k1", 8, (32, 384, 512), 192, (1,), 0, 1),
    ("cnn1", 8, (32, 192, 16, 512), 192, (3, 3), 1, 1),
    ("cnn2", 8, (32, 192, 2, 512), 192, (3, 3), 1, 1),
    ("tcn_conv2", 8, (32, 384, 128), 384, (3,), 1, 1),
    ("q_tcn_k3", 4, (32, 64, 512), 128, (3,), 2, 2),
    ("q_cnn1", 4, (32, 64, 16, 512), 64, (3, 3), 1, 1),
]


def timeit(f, iters=20):
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def setenv(on):
    os.environ["SELD_HCQ_WGRAD_DQ"] = "1"
    if on:
        os.environ["SELD_CONV_NO_HCQ"] = "1"
    else:
        os.environ.pop("SELD_CONV_NO_HCQ", None)
    L.reload_env()


def main():
    only = sys.argv[1:]
    for name, A, xs, cout, k, pad, dil in SHAPES:
        if only and name not in only:
            continue
        g = torch.Generator().manual_seed(3)
        x = torch.randn(xs, generator=g).to(dev)
        cin = xs[1]
        wshape = (cout // A, cin // A) + k
        desc = H.make_conv_desc(xs, cout, A, k, 1, pad, dil)
        yshape = (xs[0], cout) + tuple(xs[2:])
        dyA = torch.randn(yshape, generator=g).to(dev)
        dyB = torch.randn(yshape, generator=g).to(dev)
        flops, _ = H.conv_work(desc, 2)
        res = {}
        for tag, old in (("old", True), ("new", False)):
            setenv(old)
            gA = [torch.zeros(wshape, device=dev) for _ in range(A)]
            gB = [torch.zeros(wshape, device=dev) for _ in range(A)]
            if not old and not H._hcq_wgrad_ok(desc):
                res[tag] = None
                continue
            H.conv_bwd_weight(desc, x, dyA, wshape, False, into=gA)
            torch.cuda.synchronize()
            one = [t.clone() for t in gA]
            t1 = timeit(lambda: H.conv_bwd_weight(desc, x, dyA, wshape, False, into=gA))
            # pair
            for t in gA + gB:
                t.zero_()
            if old:
                import ctypes
                fn = lambda: L.check(L.lib().seld_hc_conv_pair_bwd_weight_acc(ctypes.byref(desc), L.ptr(x), L.ptr(dyA), L.ptr(dyB),
                                                                              L.ptr_array8(gA), L.ptr_array8(gB), None, None,
                                                                              L.current_stream()), "pair")
                try:
                    fn()
                except L.SeldHipError:
                    fn = None
            else:
                fn = (lambda: H.hcq_wgrad_acc(desc, x, dyA, gA, dyB, gB)) if H._hcq_wgrad_ok(desc, 2) else None
                if fn:
                    fn()
            torch.cuda.synchronize()
            pair = ([t.clone() for t in gA], [t.clone() for t in gB]) if fn else None
            t2 = timeit(fn) if fn else float("nan")
            res[tag] = (one, t1, pair, t2)
        setenv(False)
        old, new = res["old"], res["new"]
        if new is None:
            print(f"{name:12s} unsupported (old {old[1]:.1f} us)")
            continue
        scale = max(float(t.abs().max()) for t in old[0])
        err = max(float((a - b).abs().max()) for a, b in zip(new[0], old[0])) / scale
        line = f"{name:12s} single: err {err:.1e} new {new[1]:7.1f} us ({flops / new[1] / 1e6:6.1f} TF-eq) old {old[1]:7.1f} us"
        if new[2] is not None and old[2] is not None:
            e2 = max(float((a - b).abs().max()) for a, b in zip(new[2][0] + new[2][1], old[2][0] + old[2][1])) / scale
            line += f" | pair: err {e2:.1e} new {new[3]:7.1f} us ({2 * flops / new[3] / 1e6:6.1f} TF-eq) old {old[3]:7.1f} us"
        elif new[2] is not None:
            line += f" | pair: new {new[3]:7.1f} us ({2 * flops / new[3] / 1e6:6.1f} TF-eq)"
        print(line + "  " + H._hcq_wgrad_label(desc), flush=True)


if __name__ == "__main__":
    main()
