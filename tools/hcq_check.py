#!/usr/bin/env python3
"""Fast-Hamilton conv kernels (hcq_conv.hip) against the 16/48-product kernels: values and time."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import seld_amd  # noqa: E402

H, L = seld_amd.hip_ops, seld_amd._lib
dev = torch.device("cuda:0")
SHAPES = [
    ("tcn_k3_d5", 8, (32, 192, 512), 384, (3,), 5, 5),
    ("tcn_k3_d1", 8, (32, 192, 512), 384, (3,), 1, 1),
    ("tcn_k3_d55", 8, (32, 192, 512), 384, (3,), 55, 55),
    ("tcn_k1", 8, (32, 384, 512), 192, (1,), 0, 1),
    ("cnn1", 8, (32, 192, 16, 512), 192, (3, 3), 1, 1),
    ("cnn2", 8, (32, 192, 2, 512), 192, (3, 3), 1, 1),
    ("q_tcn_k3", 4, (32, 64, 512), 128, (3,), 2, 2),
    ("q_tcn_k1", 4, (32, 128, 512), 64, (1,), 0, 1),
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


def main():
    only = sys.argv[1:]
    for name, A, xs, cout, k, pad, dil in SHAPES:
        if only and name not in only:
            continue
        g = torch.Generator().manual_seed(3)
        x = torch.randn(xs, generator=g).to(dev)
        cin = xs[1]
        ws = [(torch.randn((cout // A, cin // A) + k, generator=g) * 0.1).to(dev) for _ in range(A)]
        desc = H.make_conv_desc(xs, cout, A, k, 1, pad, dil)
        y_ref = H.conv_fwd(desc, x, ws)
        dy = torch.randn(y_ref.shape, generator=g).to(dev)
        dx_ref = H.conv_bwd_data(desc, dy, ws, tuple(x.shape))
        flops, _ = H.conv_work(desc, 0)
        line = f"{name:12s}"
        for mode, src, ref in ((0, x, y_ref), (1, dy, dx_ref)):
            wp = H.hcq_pack(desc, mode, ws)
            if wp is None:
                line += f"  mode{mode}: unsupported"
                continue
            out = torch.empty_like(ref)
            H.hcq_conv(desc, mode, src, wp, (out,))
            torch.cuda.synchronize()
            err = float((out - ref).abs().max()) / float(ref.abs().max())
            t_new = timeit(lambda: H.hcq_conv(desc, mode, src, wp, (out,)))
            t_old = timeit((lambda: H.conv_fwd(desc, x, ws, out=y_ref)) if mode == 0 else
                           (lambda: H.conv_bwd_data(desc, dy, ws, tuple(x.shape))))
            t_pack = timeit(lambda: H.hcq_pack(desc, mode, ws, out=wp))
            line += (f"  {'fwd' if mode == 0 else 'dgrad'}: err {err:.1e} new {t_new:7.1f} us ({flops / t_new / 1e6:6.1f} TF-eq) "
                     f"old {t_old:7.1f} us pack {t_pack:5.1f} us |")
        print(line, flush=True)
        if "pair" in only or not only:
            # pairs: two convolutions of one input in one launch; the sum of their data gradients in one launch
            ws2 = [(torch.randn((cout // A, cin // A) + k, generator=g) * 0.1).to(dev) for _ in range(A)]
            y2_ref = H.conv_fwd(desc, x, ws2)
            wp = H.hcq_pack(desc, 0, ws, ws2)
            if wp is not None:
                o1, o2 = torch.empty_like(y_ref), torch.empty_like(y_ref)
                H.hcq_conv(desc, 0, x, wp, (o1, o2))
                e1 = float((o1 - y_ref).abs().max()) / float(y_ref.abs().max())
                e2 = float((o2 - y2_ref).abs().max()) / float(y2_ref.abs().max())
                t = timeit(lambda: H.hcq_conv(desc, 0, x, wp, (o1, o2)))
                dy2 = torch.randn(y_ref.shape, generator=g).to(dev)
                dx2_ref = dx_ref + H.conv_bwd_data(desc, dy2, ws2, tuple(x.shape))
                wpd = H.hcq_pack(desc, 1, ws, ws2)
                od = torch.empty_like(dx_ref)
                H.hcq_conv(desc, 1, dy, wpd, (od,), x2=dy2)
                e3 = float((od - dx2_ref).abs().max()) / float(dx2_ref.abs().max())
                td = timeit(lambda: H.hcq_conv(desc, 1, dy, wpd, (od,), x2=dy2))
                print(f"{'':12s}  pair fwd: err {e1:.1e} {e2:.1e} {t:7.1f} us ({2 * flops / t / 1e6:6.1f} TF-eq) | pair dgrad: err {e3:.1e} "
                      f"{td:7.1f} us ({2 * flops / td / 1e6:6.1f} TF-eq)", flush=True)


if __name__ == "__main__":
    main()
