"""Thin tensor-level wrappers over the C ABI (include/seld_hip.h) + the autograd glue.

Everything here requires CUDA(HIP) tensors: fp32, contiguous.  No eager/CPU fallback exists.
"""
import ctypes

import torch

from . import _lib as L


def _req(t, name):
    if t is None:
        return None
    if not t.is_cuda:
        raise L.SeldHipError(f"{name}: expected a HIP device tensor (this package has no CPU path)")
    if t.dtype != torch.float32:
        raise L.SeldHipError(f"{name}: expected float32, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def _pair(v):
    if isinstance(v, (tuple, list)):
        return (int(v[0]), int(v[1])) if len(v) == 2 else (1, int(v[0]))
    return (int(v), int(v))


def make_conv_desc(x_shape, cout, algebra, kernel, stride, padding, dilation, groups=1):
    """x_shape: (N, C, T) or (N, C, H, W).  kernel/stride/padding/dilation: int or tuple."""
    d = L.ConvDesc()
    nd = len(x_shape) - 2
    if nd not in (1, 2):
        raise Exception("The convolutional input is either 3, 4 or 5 dimensions. input.dim = " + str(len(x_shape)))
    d.algebra, d.ndim, d.N, d.Cin, d.Cout, d.groups = algebra, nd, x_shape[0], x_shape[1], cout, groups

    def two(v):
        if nd == 1:
            v = v[0] if isinstance(v, (tuple, list)) else v
            return (1, int(v))
        return _pair(v)
    if nd == 1:
        d.in_[0], d.in_[1] = 1, x_shape[2]
        k = two(kernel); s = two(stride); p = (0, two(padding)[1]); dl = two(dilation)
    else:
        d.in_[0], d.in_[1] = x_shape[2], x_shape[3]
        k = two(kernel); s = two(stride); p = two(padding); dl = two(dilation)
    for i in range(2):
        d.k[i], d.stride[i], d.pad[i], d.dil[i] = k[i], s[i], p[i], dl[i]
    return d


def conv_out_shape(desc):
    out = (ctypes.c_int32 * 2)()
    L.check(L.lib().seld_hc_conv_out_shape(ctypes.byref(desc), out), "seld_hc_conv_out_shape")
    return out[0], out[1]


def _y_shape(desc, o):
    return (desc.N, desc.Cout, o[1]) if desc.ndim == 1 else (desc.N, desc.Cout, o[0], o[1])


def conv_fwd(desc, x, ws, bias=None, out=None, epilogue=0, addend=None, stats=None):
    x = _req(x, "x")
    ws = [_req(w, "w") for w in ws]
    bias = _req(bias, "bias")
    o = conv_out_shape(desc)
    y = out if out is not None else torch.empty(_y_shape(desc, o), device=x.device, dtype=torch.float32)
    L.check(L.lib().seld_hc_conv_fwd_ex(ctypes.byref(desc), L.ptr(x), L.ptr_array8(ws), L.ptr(bias), L.ptr(y),
                                        ctypes.c_int32(epilogue), L.ptr(_req(addend, "addend")), L.ptr(stats),
                                        L.current_stream()), "seld_hc_conv_fwd")
    return y


def conv_bwd_data(desc, dy, ws, x_shape):
    dy = _req(dy, "dy")
    ws = [_req(w, "w") for w in ws]
    dx = torch.empty(x_shape, device=dy.device, dtype=torch.float32)
    L.check(L.lib().seld_hc_conv_bwd_data(ctypes.byref(desc), L.ptr(dy), L.ptr_array8(ws), L.ptr(dx),
                                          L.current_stream()), "seld_hc_conv_bwd_data")
    return dx


def conv_bwd_weight(desc, x, dy, w_shape, want_bias):
    x = _req(x, "x")
    dy = _req(dy, "dy")
    nbytes = L.lib().seld_hc_conv_bwd_weight_workspace(ctypes.byref(desc))
    ws = torch.empty((nbytes + 3) // 4, device=x.device, dtype=torch.float32)
    dws = [torch.empty(w_shape, device=x.device, dtype=torch.float32) for _ in range(desc.algebra)]
    dbias = torch.empty(desc.Cout, device=x.device, dtype=torch.float32) if want_bias else None
    L.check(L.lib().seld_hc_conv_bwd_weight(ctypes.byref(desc), L.ptr(x), L.ptr(dy), L.ptr_array8(dws), L.ptr(dbias),
                                            L.ptr(ws), ctypes.c_size_t(nbytes), L.current_stream()),
            "seld_hc_conv_bwd_weight")
    return dws, dbias


class HyperConvFn(torch.autograd.Function):
    """y = W (x) x  for algebra 1/4/8; replaces quaternion_conv / dual_quaternion_conv / F.convNd."""

    @staticmethod
    def forward(ctx, x, bias, stride, padding, dilation, *ws):
        algebra = len(ws)
        k = tuple(ws[0].shape[2:])
        desc = make_conv_desc(tuple(x.shape), ws[0].shape[0] * algebra, algebra, k, stride, padding, dilation)
        x = _req(x, "x")
        y = conv_fwd(desc, x, ws, bias)
        ctx.desc = desc
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x, *ws)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, *ws = ctx.saved_tensors
        dy = _req(dy, "dy")
        dx = None
        if ctx.needs_input_grad[0]:
            dx = conv_bwd_data(ctx.desc, dy, ws, tuple(x.shape))
        dws = [None] * len(ws)
        dbias = None
        if any(ctx.needs_input_grad[5:]) or (ctx.has_bias and ctx.needs_input_grad[1]):
            dws, dbias = conv_bwd_weight(ctx.desc, x, dy, tuple(ws[0].shape), ctx.has_bias)
        return (dx, dbias, None, None, None, *dws)


def hyper_conv(x, ws, bias, stride, padding, dilation):
    return HyperConvFn.apply(x, bias, stride, padding, dilation, *ws)
