"""Thin tensor-level wrappers over the C ABI (include/seld_hip.h) + the autograd glue.

Everything here requires CUDA(HIP) tensors: fp32, contiguous.  No eager/CPU fallback exists.
"""
import ctypes
import os

import torch

from . import _lib as L


class KernelTimer:
    """HIP-event timing of individual conv launches on the stream they are enqueued on (bench.py's roofline
    leg).  Inactive by default: then the wrappers below add nothing to the launch path."""

    def __init__(self):
        self.active = False
        self.only = None       # None: every conv launch; else the set of kernel labels to time (the others run bare)
        self.records = []      # (label, start_event, end_event, flops, bytes)

    def reset(self):
        self.records = []

    def summary(self):
        """label -> dict(calls, ms, flops, bytes); call after torch.cuda.synchronize()."""
        out = {}
        for label, e0, e1, fl, by in self.records:
            d = out.setdefault(label, dict(calls=0, ms=0.0, flops=0.0, bytes=0.0))
            d["calls"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["flops"] += fl
            d["bytes"] += by
        return out


kernel_timer = KernelTimer()


def conv_work(desc, which):
    """Algorithmic flops / bytes of one conv call: x and y once, COMPONENT weights once, structured
    flops (48 of 64 blocks for the dual quaternion, all 16 for the quaternion).  which: 0 fwd, 1 dgrad, 2 wgrad."""
    A = desc.algebra
    o = conv_out_shape(desc)
    s_in = desc.in_[0] * desc.in_[1]
    s_out = o[0] * o[1]
    K = desc.k[0] * desc.k[1]
    nb = {1: 1, 4: 16, 8: 48}[A]
    blk = (desc.Cout // A) * (desc.Cin // A)
    flops = 2.0 * desc.N * s_out * K * nb * blk
    by = 4.0 * (desc.N * desc.Cin * s_in + desc.N * desc.Cout * s_out + A * blk * K)
    return flops, by


def _label(desc, which):
    buf = ctypes.create_string_buffer(64)
    L.check(L.lib().seld_hc_conv_kernel_label(ctypes.byref(desc), which, buf, 64), "seld_hc_conv_kernel_label")
    return buf.value.decode()


_label_cache = {}


class _Timed:
    def __init__(self, desc, which, mult=1, variant=False, label=None):
        """variant: the launch runs the kernel's other instantiation (pair data gradient, fused first-stage weight
        gradient): its symbol ends in ', 1>' instead of ', 0>'.  label: the kernel symbol when the caller knows it
        (the fast-product kernels)."""
        self.mult = mult
        self.variant = variant or (mult == 2 and which == 1)
        self.label = label
        self.on = kernel_timer.active
        if self.on and kernel_timer.only is not None:
            lab = label
            if lab is None:
                key = (bytes(desc), which)
                lab = _label_cache.get(key)
                if lab is None:
                    lab = _label_cache[key] = _label(desc, which)
                if self.variant and lab.endswith(", 0>"):
                    lab = lab[:-4] + ", 1>"
            self.on = lab in kernel_timer.only
        if self.on:
            self.desc, self.which = desc, which
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)

    def __enter__(self):
        if self.on:
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if self.on:
            self.e1.record()
            fl, by = conv_work(self.desc, self.which)
            lab = self.label
            if lab is None:
                lab = _label(self.desc, self.which)
                if self.variant and lab.endswith(", 0>"):
                    lab = lab[:-4] + ", 1>"
            kernel_timer.records.append((lab, self.e0, self.e1, fl * self.mult, by * self.mult))
        return False


def deterministic():
    """SELD_DETERMINISTIC=1: run-to-run reproducible training (include/seld_hip.h).  The library reads the switch itself
    (reductions in one ordered chain); here: BatchNorm statistics by seld_channel_stats instead of the convolution
    epilogues' atomics, weight gradients by the grouped kernels (no atomics) or seld_hc_conv_bwd_weight_det, no side
    stream.  Set it before the first library call (or call _lib.reload_env())."""
    return os.environ.get("SELD_DETERMINISTIC", "0") not in ("", "0")


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
    if stats is not None and deterministic():
        # statistics by one ordered reduction per channel instead of the epilogue's float atomics
        y = conv_fwd(desc, x, ws, bias, out=y, epilogue=epilogue & ~L.SELD_EPI_STATS, addend=addend, stats=None)
        channel_stats(y, out=stats)
        return y
    wp = hcq_weights.get(desc, 0, ws) if desc.algebra > 1 else None
    if wp is not None:                      # 8-multiplication Hamilton product (csrc/hcq_conv.hip)
        with _Timed(desc, 0, label=_hcq_label_cached(desc, 0, 1) if kernel_timer.active else None):
            hcq_conv(desc, 0, x, wp, (y,), (bias,), (epilogue,), (_req(addend, "addend"),), (stats,))
        return y
    with _Timed(desc, 0):
        L.check(L.lib().seld_hc_conv_fwd_ex(ctypes.byref(desc), L.ptr(x), L.ptr_array8(ws), L.ptr(bias), L.ptr(y),
                                            ctypes.c_int32(epilogue), L.ptr(_req(addend, "addend")), L.ptr(stats),
                                            L.current_stream()), "seld_hc_conv_fwd")
    return y


def conv_bwd_data(desc, dy, ws, x_shape, ahead=None):
    """`ahead`: (workspace, event) from _transpose_ahead -- the weights are already re-laid out."""
    dy = _req(dy, "dy")
    dx = torch.empty(x_shape, device=dy.device, dtype=torch.float32)
    lib = L.lib()
    wp = hcq_weights.get(desc, 1, ws) if desc.algebra > 1 else None
    if wp is not None:
        with _Timed(desc, 1, label=_hcq_label_cached(desc, 1, 1) if kernel_timer.active else None):
            hcq_conv(desc, 1, dy, wp, (dx,))
        return dx
    if ahead is not None:
        wt, ev = ahead
        torch.cuda.current_stream().wait_event(ev)
        with _Timed(desc, 1):
            L.check(lib.seld_hc_conv_bwd_data_wt(ctypes.byref(desc), L.ptr(dy), L.ptr(wt), L.ptr(dx), L.current_stream()),
                    "seld_hc_conv_bwd_data_wt")
        return dx
    ws = [_req(w, "w") for w in ws]
    lib.seld_hc_conv_bwd_data_workspace.restype = ctypes.c_size_t
    nbytes = lib.seld_hc_conv_bwd_data_workspace(ctypes.byref(desc))
    wsb = torch.empty((nbytes + 3) // 4, device=dy.device, dtype=torch.float32)
    with _Timed(desc, 1):
        L.check(lib.seld_hc_conv_bwd_data_ex(ctypes.byref(desc), L.ptr(dy), L.ptr_array8(ws), L.ptr(dx), L.ptr(wsb),
                                             ctypes.c_size_t(nbytes), L.current_stream()), "seld_hc_conv_bwd_data")
    return dx


def _hcq_wgrad_ok(desc, npair=1):
    if deterministic():
        return False
    key = (bytes(desc), npair, "wgrad")
    v = _hcq_labels.get(key)
    if v is None:
        v = _hcq_labels[key] = desc.algebra > 1 and (_hcq_wgrad_row_bytes(desc, npair) > 0 or
                                                     bool(L.lib().seld_hcq_wgrad_supported(ctypes.byref(desc), int(npair))))
    return v


def _hcq_wgrad_label(desc, npair=1):
    key = (bytes(desc), npair, "wgrad_label")
    v = _hcq_labels.get(key)
    if v is None:
        buf = ctypes.create_string_buffer(96)
        L.check(L.lib().seld_hcq_wgrad_label(ctypes.byref(desc), int(npair), buf, 96), "seld_hcq_wgrad_label")
        v = _hcq_labels[key] = buf.value.decode()
    return v


def _hcq_wgrad_row_bytes(desc, npair=1):
    """Scratch bytes of the 24-product dual-quaternion weight gradient (csrc/hcq_wgrad_row.hip), 0 = shape not taken."""
    key = (bytes(desc), npair, "wgrad_row")
    v = _hcq_labels.get(key)
    if v is None:
        lib = L.lib()
        lib.seld_hcq_wgrad_row_workspace.restype = ctypes.c_size_t
        v = _hcq_labels[key] = int(lib.seld_hcq_wgrad_row_workspace(ctypes.byref(desc), int(npair))) if desc.algebra == 8 else 0
    return v


def _hcq_wgrad_row_label(desc, npair=1):
    key = (bytes(desc), npair, "wgrad_row_label")
    v = _hcq_labels.get(key)
    if v is None:
        buf = ctypes.create_string_buffer(96)
        L.check(L.lib().seld_hcq_wgrad_row_label(ctypes.byref(desc), int(npair), buf, 96), "seld_hcq_wgrad_row_label")
        v = _hcq_labels[key] = buf.value.decode()
    return v


_wgrad_row_scratch = {}      # (device, stream) -> zeroed fp32 scratch; every call hands it back zeroed


def _wgrad_row_ws(nbytes, device):
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    t = _wgrad_row_scratch.get(key)
    if t is None or t.numel() * 4 < nbytes:
        t = _wgrad_row_scratch[key] = torch.zeros((nbytes + 3) // 4, device=device, dtype=torch.float32)
    return t


def hcq_wgrad_acc(desc, x, dyA, dwA, dyB=None, dwB=None):
    """dwA[c] += wgrad(x, dyA) [, dwB[c] += wgrad(x, dyB)] on the fast-product kernels: the row-chunk GEMM on forms for
    the dual quaternion (seld_hcq_wgrad_row_acc), else seld_hcq_wgrad_acc."""
    npair = 2 if dyB is not None else 1
    nbytes = _hcq_wgrad_row_bytes(desc, npair)
    if nbytes:
        ws = _wgrad_row_ws(nbytes, x.device)
        with _Timed(desc, 2, npair, label=_hcq_wgrad_row_label(desc, npair) if kernel_timer.active else None):
            L.check(L.lib().seld_hcq_wgrad_row_acc(ctypes.byref(desc), npair, L.ptr(x), L.ptr(dyA), L.ptr(dyB),
                                                   L.ptr_array8(dwA), L.ptr_array8(dwB) if dwB is not None else None,
                                                   L.ptr(ws), ctypes.c_size_t(ws.numel() * 4), L.current_stream()),
                    "seld_hcq_wgrad_row_acc")
        return
    with _Timed(desc, 2, npair, label=_hcq_wgrad_label(desc, npair) if kernel_timer.active else None):
        L.check(L.lib().seld_hcq_wgrad_acc(ctypes.byref(desc), npair, L.ptr(x), L.ptr(dyA), L.ptr(dyB), L.ptr_array8(dwA),
                                           L.ptr_array8(dwB) if dwB is not None else None, L.current_stream()),
                "seld_hcq_wgrad_acc")


# ---- grouped weight gradients (csrc/hcq_wgrad_grp.hip) -----------------------------------------------------------------
_wgrad_group_scratch = {}


def wgrad_group_jobs(jobs):
    """ctypes array of seld_wgrad_job from [(desc, x, dy, [8 gradient tensors]), ...]."""
    arr = (L.WgradJob * len(jobs))()
    for a, (desc, x, dy, dws) in zip(arr, jobs):
        ctypes.memmove(ctypes.byref(a.desc), ctypes.byref(desc), ctypes.sizeof(L.ConvDesc))
        a.x, a.dy = x.data_ptr(), dy.data_ptr()
        for i in range(8):
            a.dw[i] = dws[i].data_ptr()
    return arr


def wgrad_group_bytes(jobs_arr):
    """Scratch bytes of a grouped call, 0 when one of the jobs is not a shape the grouped kernels take."""
    return int(L.lib().seld_hcq_wgrad_group_workspace(jobs_arr, len(jobs_arr)))


def wgrad_group(jobs):
    """dw[c] += weight gradient for every (desc, x, dy, dws) of `jobs` -- dual-quaternion convolutions of the shape
    families of seld_hcq_wgrad_group -- in one persistent launch per family.  Returns False (nothing launched) when a job
    is not taken.  Deterministic: no atomics, fixed summation order."""
    arr = wgrad_group_jobs(jobs)
    nbytes = wgrad_group_bytes(arr)
    if nbytes == 0:
        return False
    dev = jobs[0][1].device
    if torch.cuda.is_current_stream_capturing():
        # a recorded step: the scratch comes from (and stays in) the graph's own memory pool
        ws = torch.empty(nbytes, device=dev, dtype=torch.uint8)
    else:
        key = (dev, torch.cuda.current_stream(dev).cuda_stream)
        ws = _wgrad_group_scratch.get(key)
        if ws is None or ws.numel() < nbytes:
            ws = _wgrad_group_scratch[key] = torch.empty(nbytes, device=dev, dtype=torch.uint8)
    L.check(L.lib().seld_hcq_wgrad_group(arr, len(arr), L.ptr(ws), ctypes.c_size_t(ws.numel()), L.current_stream()),
            "seld_hcq_wgrad_group")
    return True


def wgrad_group_family(desc):
    """Shape family (0..3) of `desc` in the grouped kernels, -1 = not taken (cached)."""
    key = (bytes(desc), "grp_family")
    v = _hcq_labels.get(key)
    if v is None:
        v = _hcq_labels[key] = int(L.lib().seld_hcq_wgrad_group_family(ctypes.byref(desc)))
    return v


class _DeferredWgrads:
    """Weight gradients the backward pass does NOT launch where autograd reaches them: the dual-quaternion layers of the
    TCN and of the 3x3 stages are collected -- (desc, x, dy, gradient slots), the tensors kept alive -- and issued as ONE
    grouped call (seld_hcq_wgrad_group: one persistent launch per shape family) when the backward pass ends, i.e. before
    anything reads the gradient buffer (data-parallel exchange, Adam).  Why: a workgroup that keeps a layer's whole output
    tile in registers needs hundreds of positions to amortise it, and one layer spread over 256 CUs has 64
    (csrc/hcq_wgrad_grp.hip).  SELD_WGRAD_GROUP=0 restores the per-layer launches on the side stream.

    Families deferred: 0 (192 -> 384 1x3), 1 (384 -> 192 1x1), 2 (192 -> 192 3x3); family 3 (384 -> 384 1x3, tcn.conv2) is
    three steps per workgroup at its size and stays on the per-layer kernels."""
    FAMILIES = (0, 1, 2)

    def __init__(self):
        self.jobs = []
        self.armed = False

    @staticmethod
    def enabled():
        return os.environ.get("SELD_WGRAD_GROUP", "1") != "0"

    def takes(self, desc):
        if desc.algebra != 8 or not self.enabled():
            return False
        fam = wgrad_group_family(desc)
        return fam in self.FAMILIES or (fam == 3 and deterministic())        # the grouped kernels have no atomics

    def add(self, desc, x, dy, dws):
        # the stream this backward node runs on produced dy (branch B of the two-stream model runs on its own queue)
        self.jobs.append((desc, x, dy, list(dws), torch.cuda.current_stream(x.device)))
        if not self.armed:
            self.armed = True
            try:
                torch.autograd.Variable._execution_engine.queue_callback(self.flush)
            except RuntimeError:            # not inside a backward pass: issue at once
                self.flush()

    def flush(self):
        self.armed = False
        jobs, self.jobs = self.jobs, []
        if not jobs:
            return
        here = torch.cuda.current_stream(jobs[0][1].device)
        for st in {j[4] for j in jobs}:
            if st != here:
                here.wait_stream(st)
                for _, x, dy, _, st_ in jobs:
                    if st_ == st:
                        x.record_stream(here)
                        dy.record_stream(here)
        jobs = [j[:4] for j in jobs]
        timed = kernel_timer.active and (kernel_timer.only is None or "hcq_wgrad_grp_kernel" in kernel_timer.only)
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        if not wgrad_group(jobs):            # cannot happen for jobs `takes` accepted; never lose a gradient to it
            for desc, x, dy, dws in jobs:
                conv_bwd_weight(desc, x, dy, tuple(dws[0].shape), False, into=dws)
        elif timed:
            e1.record()
            fl = by = 0.0
            for desc, _, _, _ in jobs:
                f_, b_ = conv_work(desc, 2)
                fl, by = fl + f_, by + b_
            kernel_timer.records.append(("hcq_wgrad_grp_kernel", e0, e1, fl, by))

    def discard(self):
        self.jobs, self.armed = [], False


deferred_wgrads = _DeferredWgrads()


def conv_bwd_weight(desc, x, dy, w_shape, want_bias, into=None, bias_into=None):
    """Component weight gradients.  `into` (list of A tensors, e.g. views of FlatAdam.flat_grad) selects the
    accumulating entry point: the kernel adds straight into them and nothing is returned for autograd."""
    x = _req(x, "x")
    dy = _req(dy, "dy")
    if deterministic():
        lib = L.lib()
        lib.seld_hc_conv_bwd_weight_det_workspace.restype = ctypes.c_size_t
        nbytes = int(lib.seld_hc_conv_bwd_weight_det_workspace(ctypes.byref(desc)))
        wsb = torch.empty(max(nbytes, 16), device=x.device, dtype=torch.uint8)
        dws = into if into is not None else [torch.zeros(w_shape, device=x.device, dtype=torch.float32) for _ in range(desc.algebra)]
        dbias = bias_into if bias_into is not None else (torch.zeros(desc.Cout, device=x.device, dtype=torch.float32) if want_bias else None)
        with _Timed(desc, 2):
            L.check(lib.seld_hc_conv_bwd_weight_det(ctypes.byref(desc), L.ptr(x), L.ptr(dy), L.ptr_array8(dws), L.ptr(dbias),
                                                    L.ptr(wsb), ctypes.c_size_t(wsb.numel()), L.current_stream()),
                    "seld_hc_conv_bwd_weight_det")
        return (None, None) if into is not None else (dws, dbias)
    if not want_bias and bias_into is None and _hcq_wgrad_ok(desc):
        if into is not None:
            hcq_wgrad_acc(desc, x, dy, into)
            return None, None
        dws = [torch.zeros(w_shape, device=x.device, dtype=torch.float32) for _ in range(desc.algebra)]
        hcq_wgrad_acc(desc, x, dy, dws)
        return dws, None
    if into is not None:
        with _Timed(desc, 2):
            L.check(L.lib().seld_hc_conv_bwd_weight_acc(ctypes.byref(desc), L.ptr(x), L.ptr(dy), L.ptr_array8(into),
                                                        L.ptr(bias_into), L.current_stream()),
                    "seld_hc_conv_bwd_weight_acc")
        return None, None
    dws = [torch.empty(w_shape, device=x.device, dtype=torch.float32) for _ in range(desc.algebra)]
    dbias = torch.empty(desc.Cout, device=x.device, dtype=torch.float32) if want_bias else None
    with _Timed(desc, 2):
        L.check(L.lib().seld_hc_conv_bwd_weight(ctypes.byref(desc), L.ptr(x), L.ptr(dy), L.ptr_array8(dws), L.ptr(dbias),
                                                None, ctypes.c_size_t(0), L.current_stream()),
                "seld_hc_conv_bwd_weight")
    return dws, dbias


_hcq_labels = {}


def _hcq_label_cached(desc, mode, npair):
    key = (bytes(desc), mode, npair)
    lab = _hcq_labels.get(key)
    if lab is None:
        lab = _hcq_labels[key] = hcq_label(desc, mode, npair)
    return lab


def _hcq_ok(desc, mode, npair=1):
    """Does the fast-product kernel take this (shape, direction)?  (cached; no weights needed)"""
    key = (bytes(desc), mode, npair, "ok")
    v = _hcq_labels.get(key)
    if v is None:
        v = _hcq_labels[key] = desc.algebra > 1 and hcq_pack_floats(desc, mode, npair) > 0
    return v


def _direct_targets(params, bias):
    """Gradient slots to accumulate into directly, or None.  A parameter opts in when its owner (FlatAdam)
    pre-attached `.grad` as a view of a flat buffer and set `_seld_direct_grad`."""
    ts = [getattr(t, "_seld_base_param", t) for t in params] + ([bias] if bias is not None else [])
    if all(getattr(t, "_seld_direct_grad", False) and t.grad is not None for t in ts):
        def slot(b, t):
            if b is t:
                return b.grad
            g = getattr(t, "_seld_grad", None)         # stacked_conv_weight: a slot that spans several parameters
            return g if g is not None else b.grad.view(t.shape)
        return [slot(b, t) for b, t in zip(ts, params)], (bias.grad if bias is not None else None)
    return None


def stacked_conv_weight(params):
    """ONE convolution weight (sum of the Cout's, Cin, k...) over parameters that lie back to back in memory -- FlatAdam
    re-homes a model's parameters into one flat buffer in registration order, so the attention's values / keys / queries
    (model.py:18-20) are three consecutive row blocks of it -- or None.  With gradients enabled the gradient slots must be
    adjacent in the same order too: the stacked weight is a fresh leaf over the same storage whose gradient the backward
    kernels write straight into those slots (`_direct_targets`), so autograd never sees the member parameters."""
    p0 = params[0]
    need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in params)
    at = p0.data_ptr()
    for p in params:
        if p.data_ptr() != at or not p.is_contiguous() or p.shape[1:] != p0.shape[1:] or p.dtype != torch.float32:
            return None
        at += 4 * p.numel()
    shape = (sum(int(p.shape[0]) for p in params),) + tuple(p0.shape[1:])
    same_buffer = lambda ts: all(t.untyped_storage().data_ptr() == ts[0].untyped_storage().data_ptr() for t in ts)
    if not same_buffer(params):                     # neighbours by accident of the allocator: not one tensor's memory
        return None
    gview = None
    if need_grad:
        if not all(getattr(p, "_seld_direct_grad", False) and p.grad is not None and p.requires_grad for p in params):
            return None
        if not same_buffer([p.grad for p in params]):
            return None
        gat = p0.grad.data_ptr()
        for p in params:
            if p.grad.data_ptr() != gat or not p.grad.is_contiguous():
                return None
            gat += 4 * p.numel()
        gview = p0.grad.as_strided(shape, p0.grad.stride())
    w = p0.detach().as_strided(shape, p0.stride())
    if need_grad:
        w.requires_grad_(True)
        w._seld_base_param = p0
        w._seld_grad = gview
    return w


def as_conv_weight(param, shape):
    """A contiguous reshape of a parameter used as a convolution weight (the attention's Linear applied as a 1x1
    convolution, model.py:46): the view remembers its parameter, so the backward kernels write that parameter's gradient
    slot directly instead of returning a tensor for autograd to add to it."""
    w = param.view(shape)
    w._seld_base_param = param
    return w


# ---- weight gradients on a second HIP stream -------------------------------------------------------------
# A layer's weight gradient and its data gradient both start from dy and are independent; the accumulating weight-
# gradient kernels write only FlatAdam's gradient slots.  Issued on a side stream they overlap the data gradient and
# the element-wise kernels that follow it on the main stream: a 70 us kernel on this GPU spends ~13 us ramping up and
# draining, which another queue fills (measured: two independent 1x3 convolutions 142 -> 121 us).  The main stream
# joins the side stream when the backward pass ends (autograd engine callback) and in FlatAdam.step().
_side = {"stream": None, "dirty": False, "keep": []}


def _side_enabled():
    return os.environ.get("SELD_WGRAD_SIDE_STREAM", "1") != "0" and not deterministic()


def join_side_stream():
    """Make the current stream wait for everything issued on the side stream."""
    if _side["dirty"]:
        ev = torch.cuda.Event()
        ev.record(_side["stream"])
        torch.cuda.current_stream().wait_event(ev)
        _side["dirty"] = False
    _side["keep"].clear()


def _on_side_stream(fn, *tensors):
    """Run `fn` (kernel launches only) on the side stream, ordered after everything already on the current stream.
    `tensors` are read there: the caching allocator must not recycle them before the side stream is done, and
    nothing on the main stream may overwrite them before the join.  The second point is about autograd: a backward
    that hands `dy` on as the gradient of an addend (HyperConvAddFn / HyperConvPairFn) gives the engine a tensor it
    accumulates into IN PLACE when it holds the only reference -- while the side stream may still be reading it.
    Holding a reference here until the join makes the engine accumulate out of place instead."""
    if _side["stream"] is None:
        _side["stream"] = torch.cuda.Stream()
    st = _side["stream"]
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream())
    st.wait_event(ev)
    with torch.cuda.stream(st):
        fn()
    for t in tensors:
        if t is not None:
            t.record_stream(st)
            _side["keep"].append(t)
    if not _side["dirty"]:
        _side["dirty"] = True
        try:
            torch.autograd.Variable._execution_engine.queue_callback(join_side_stream)
        except RuntimeError:            # not inside a backward pass: join at once
            join_side_stream()


def _transpose_ahead(desc, ws):
    """The data gradient's weight re-layout (seld_hc_conv_transpose_weights), issued NOW on the side stream -- i.e. during
    the forward pass, where it overlaps the convolution -- instead of in front of the data-gradient kernel on the critical
    path of the backward pass (47 launches of ~5 us per step).  Returns (workspace tensor, event) or None."""
    if not _side_enabled() or not torch.is_grad_enabled() or _hcq_ok(desc, 1):
        return None
    lib = L.lib()
    lib.seld_hc_conv_bwd_data_workspace.restype = ctypes.c_size_t
    nbytes = lib.seld_hc_conv_bwd_data_workspace(ctypes.byref(desc))
    wt = torch.empty((nbytes + 3) // 4, device=ws[0].device, dtype=torch.float32)
    if _side["stream"] is None:
        _side["stream"] = torch.cuda.Stream()
    st = _side["stream"]
    ev0 = torch.cuda.Event()
    ev0.record(torch.cuda.current_stream())
    st.wait_event(ev0)                                   # the weights may still be in flight (Adam of the last step)
    with torch.cuda.stream(st):
        L.check(lib.seld_hc_conv_transpose_weights(ctypes.byref(desc), L.ptr_array8([_req(w, "w") for w in ws]), L.ptr(wt),
                                                   ctypes.c_size_t(nbytes), L.current_stream()),
                "seld_hc_conv_transpose_weights")
        ev = torch.cuda.Event()
        ev.record(st)
    wt.record_stream(st)
    return wt, ev


def _conv_backward(ctx, dy, first_w):
    x = ctx.saved_tensors[0]
    ws = ctx.w_params
    dy = _req(dy, "dy")
    dws, dbias = [None] * len(ws), None
    need_w = any(ctx.needs_input_grad[first_w:]) or (ctx.has_bias and ctx.needs_input_grad[1])
    direct = _direct_targets(ws, ctx.bias_param) if need_w else None
    if direct is not None and direct[1] is None and deferred_wgrads.takes(ctx.desc):
        deferred_wgrads.add(ctx.desc, x, dy, direct[0])          # issued with the other layers' when the backward pass ends
        need_w = False
    elif direct is not None and _side_enabled() and ctx.needs_input_grad[0]:
        _on_side_stream(lambda: conv_bwd_weight(ctx.desc, x, dy, tuple(ws[0].shape), ctx.has_bias, into=direct[0],
                                                bias_into=direct[1]), x, dy)
        need_w = False
    dx = conv_bwd_data(ctx.desc, dy, ws, tuple(x.shape), getattr(ctx, "wt_ahead", None)) if ctx.needs_input_grad[0] else None
    if need_w:
        if direct is not None:
            conv_bwd_weight(ctx.desc, x, dy, tuple(ws[0].shape), ctx.has_bias, into=direct[0], bias_into=direct[1])
        else:
            dws, dbias = conv_bwd_weight(ctx.desc, x, dy, tuple(ws[0].shape), ctx.has_bias)
    return dx, dbias, dws


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
        ctx.w_params, ctx.bias_param = ws, bias
        ctx.wt_ahead = _transpose_ahead(desc, ws) if ctx.needs_input_grad[0] else None
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        dx, dbias, dws = _conv_backward(ctx, dy, 5)
        return (dx, dbias, None, None, None, *dws)


def hyper_conv(x, ws, bias, stride, padding, dilation):
    return HyperConvFn.apply(x, bias, stride, padding, dilation, *ws)


# ======================================================================================
# 8-multiplication Hamilton product kernels (csrc/hcq_conv.hip)
# ======================================================================================
def hcq_pack_floats(desc, mode, nsets=1):
    """Floats of the packed weight-form buffer, 0 if the fast-product kernel does not take this shape."""
    lib = L.lib()
    lib.seld_hcq_pack_floats.restype = ctypes.c_size_t
    return int(lib.seld_hcq_pack_floats(ctypes.byref(desc), int(mode), int(nsets)))


class _HcqWeights:
    """Packed weight forms of every (layer, direction) that runs on the fast-product kernels.

    The forms depend on the weights only, so they are rebuilt when the weights change, not per call:
      * `weights_changed()` (FlatAdam.step, anything that rewrites parameters behind torch's back) starts a new epoch;
        the first request of an epoch re-packs EVERY registered entry in one launch (seld_hcq_pack_table);
      * an in-place edit torch knows about (load_state_dict, an eager optimiser) shows in the tensors' version counters
        and re-packs just the entry that is asked for;
      * parameters that moved (Module.to, FlatAdam re-homing them into its flat buffer) are noticed by their pointers.
    Entries hold weak references: a deleted model drops out at the next full re-pack."""

    def __init__(self):
        self.entries = {}
        self.epoch = 0
        self.packed_epoch = -1
        self.table = None
        self.table_dirty = True
        self.max_floats = 0
        self._esize = None

    def weights_changed(self):
        self.epoch += 1

    def reset(self):
        self.entries.clear()
        self.table, self.table_dirty, self.packed_epoch = None, True, -1

    class _Entry:
        __slots__ = ("desc", "mode", "refs", "npair", "nA", "buf", "ptrs", "vers", "epoch", "host")

    @staticmethod
    def _alive(e):
        ws = [r() for r in e.refs]
        return None if any(w is None for w in ws) else ws

    def _fill(self, e, ws):
        """(Re)build the table entry of `e` from the weights' current addresses."""
        lib = L.lib()
        if self._esize is None:
            lib.seld_hcq_pack_entry_bytes.restype = ctypes.c_size_t
            self._esize = int(lib.seld_hcq_pack_entry_bytes())
        host = ctypes.create_string_buffer(self._esize)
        wsA, wsB = ws[:e.nA], (ws[e.nA:] if e.npair == 2 else None)
        L.check(lib.seld_hcq_pack_entry(ctypes.byref(e.desc), e.mode, e.npair, L.ptr_array8(wsA),
                                        L.ptr_array8(wsB) if wsB is not None else None, L.ptr(e.buf), host),
                "seld_hcq_pack_entry")
        e.host = host.raw
        e.ptrs = tuple(w.data_ptr() for w in ws)
        e.vers = None
        e.epoch = -1
        self.table_dirty = True

    def refresh_table(self):
        """Bring the device-side table up to date with the registered entries WITHOUT packing (host work + two small
        host-to-device copies).  train.GraphedTrainStep calls it right before recording: entries registered during the
        warm-up step left the table dirty, and the copies are not allowed inside a stream capture."""
        return self._collect()[0]

    def _collect(self):
        dead = []
        versions = {}
        for key, e in self.entries.items():
            if e is None:
                continue
            ws = self._alive(e)
            if ws is None or any(not w.is_cuda for w in ws):
                dead.append(key)
                continue
            if tuple(w.data_ptr() for w in ws) != e.ptrs:
                self._fill(e, ws)
            versions[id(e)] = tuple(w._version for w in ws)
        for key in dead:
            del self.entries[key]
            self.table_dirty = True
        live = [e for e in self.entries.values() if e is not None]
        if live and self.table_dirty:
            import numpy as np
            raw = b"".join(e.host for e in live)
            self.table = torch.from_numpy(np.frombuffer(raw, dtype=np.uint8).copy()).to(live[0].buf.device)
            starts = np.zeros(len(live) + 1, dtype=np.int32)
            starts[1:] = np.cumsum([(e.buf.numel() + 255) // 256 for e in live])
            self.starts = torch.from_numpy(starts).to(live[0].buf.device)
            self.total_blocks = int(starts[-1])
            self.table_dirty = False
        return live, versions

    def _pack_all(self):
        live, versions = self._collect()
        if not live:
            self.packed_epoch = self.epoch
            return
        L.check(L.lib().seld_hcq_pack_flat(L.ptr(self.table), L.ptr(self.starts), len(live), self.total_blocks,
                                           L.current_stream()), "seld_hcq_pack_flat")
        # the version counters the forms were built from: an in-place edit torch knows about (load_state_dict) between now
        # and the entry's next request shows as a mismatch there and re-packs that entry (with None recorded
        # here, an entry that was bulk-packed and then edited was served stale)
        for e in live:
            e.epoch, e.vers = self.epoch, versions[id(e)]
        self.packed_epoch = self.epoch

    def pin_for_graph(self):
        """Everything a RECORDED seld_hcq_pack_flat launch points at: the table, the block starts and the entries' form
        buffers as they are now.  `_pack_all` never edits a table in place -- it builds new tensors when an entry is added
        -- so a recorded step that holds these references keeps replaying against valid memory whatever shapes are
        registered later (validation at another batch size, an eval-only pooling entry)."""
        live = [e for e in self.entries.values() if e is not None]
        return (self.table, getattr(self, "starts", None), [e.buf for e in live])

    def get(self, desc, mode, wsA, wsB=None):
        """Packed forms for (desc, mode) of the given component tensors, or None if the shape runs on the 16/48-product
        kernels.  Fresh with respect to the weights as they are when the returned launch order is reached."""
        import weakref
        ws = list(wsA) + (list(wsB) if wsB is not None else [])
        key = (bytes(desc), mode, tuple(id(w) for w in ws))
        e = self.entries.get(key, 0)
        if e is None:
            return None
        if e == 0 or self._alive(e) is None:
            npair = 2 if wsB is not None else 1
            n = hcq_pack_floats(desc, mode, npair) if all(w.is_cuda and w.dtype == torch.float32 and w.is_contiguous()
                                                          for w in ws) else 0
            if n == 0:
                self.entries[key] = None
                return None
            e = self._Entry()
            e.desc = L.ConvDesc.from_buffer_copy(bytes(desc))
            e.mode, e.npair, e.nA = int(mode), npair, len(wsA)
            e.refs = [weakref.ref(w) for w in ws]
            e.buf = torch.empty(n, device=ws[0].device, dtype=torch.float32)
            self._fill(e, ws)
            self.entries[key] = e
        ptrs = tuple(w.data_ptr() for w in ws)
        if ptrs != e.ptrs:
            self._fill(e, ws)
        if self.packed_epoch != self.epoch:
            self._pack_all()
        vers = tuple(w._version for w in ws)
        if e.epoch != self.epoch or (e.vers is not None and e.vers != vers):
            wsA_, wsB_ = ws[:e.nA], (ws[e.nA:] if e.npair == 2 else None)
            hcq_pack(desc, mode, wsA_, wsB_, out=e.buf)
            e.epoch = self.epoch
        e.vers = vers
        return e.buf


hcq_weights = _HcqWeights()


# ---- the two branches of the two-stream model on two queues --------------------------------------------------------
_branch = {"stream": None}


def two_queue_branches():
    """SELD_BRANCH_STREAMS=0 turns the second queue off (both branches then run one after the other on the caller's)."""
    return os.environ.get("SELD_BRANCH_STREAMS", "1") != "0"


def run_branches(fa, xa, fb, xb):
    """(fa(xa), fb(xb)) with fb on a second HIP stream.  Used for the two ConvTC blocks of the two-stream model
    (model.py:463-471: independent until their outputs are concatenated, and at 16 samples per GPU neither fills the
    device by itself) and for the SED / DOA classifier heads (model.py:473-480: two chains of small-grid kernels).
    Autograd replays each branch's backward on the stream its forward ran on and orders the streams at the fork and
    the join; the weight forms are packed BEFORE the fork (they are packed once per step, by whoever asks first)."""
    if not (xa.is_cuda and two_queue_branches()):
        return fa(xa), fb(xb)
    if hcq_weights.packed_epoch != hcq_weights.epoch:
        hcq_weights._pack_all()
    main = torch.cuda.current_stream()
    if _branch["stream"] is None:
        _branch["stream"] = torch.cuda.Stream()
    sb = _branch["stream"]
    sb.wait_stream(main)
    xb.record_stream(sb)
    ya = fa(xa)                      # host order A, B as on one queue: the dropout counters are drawn in the same order
    with torch.cuda.stream(sb):
        yb = fb(xb)
    main.wait_stream(sb)
    yb.record_stream(main)
    return ya, yb


class FanOut2Fn(torch.autograd.Function):
    """x -> (x, x) for a tensor with two consumers (the SED and DOA heads, model.py:473-480): the sum of the two
    gradients is this library's add kernel on the consumer's stream instead of the autograd engine's ATen add."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x), x.view_as(x)

    @staticmethod
    def backward(ctx, ga, gb):
        if ga is None or gb is None:
            return ga if gb is None else gb
        ga, gb = _req(ga, "ga"), _req(gb, "gb")
        out = torch.empty_like(ga)
        L.check(L.lib().seld_add(L.ptr(ga), L.ptr(gb), ctypes.c_int64(ga.numel()), L.ptr(out), L.current_stream()), "seld_add")
        return out


def fan_out2(x):
    if not (x.is_cuda and x.requires_grad and torch.is_grad_enabled()):
        return x, x
    return FanOut2Fn.apply(x)


def _drop_kernel_choice_caches():
    hcq_weights.reset()
    _fs_cache.clear()
    _hcq_labels.clear()
    _label_cache.clear()
    _pair_ok_cache.clear()


L._reload_hooks.append(_drop_kernel_choice_caches)


def hcq_label(desc, mode, npair=1):
    buf = ctypes.create_string_buffer(96)
    L.check(L.lib().seld_hcq_kernel_label(ctypes.byref(desc), int(mode), int(npair), buf, 96), "seld_hcq_kernel_label")
    return buf.value.decode()


def hcq_pack(desc, mode, wsA, wsB=None, out=None):
    """Weight forms F_m(W) in MFMA fragment order (seld_hcq_pack): mode 0 forward, 1 data gradient."""
    nsets = 2 if wsB is not None else 1
    n = hcq_pack_floats(desc, mode, nsets)
    if n == 0:
        return None
    if out is None:
        out = torch.empty(n, device=wsA[0].device, dtype=torch.float32)
    L.check(L.lib().seld_hcq_pack(ctypes.byref(desc), int(mode), nsets, L.ptr_array8([_req(w, "w") for w in wsA]),
                                  L.ptr_array8([_req(w, "w") for w in wsB]) if wsB is not None else None, L.ptr(out),
                                  L.current_stream()), "seld_hcq_pack")
    return out


def _ptr2(a, b=None):
    arr = (ctypes.c_void_p * 2)()
    arr[0] = a.data_ptr() if a is not None else 0
    arr[1] = b.data_ptr() if b is not None else 0
    return arr


def hcq_conv(desc, mode, x, wpack, outs, biases=(None, None), epilogues=(0, 0), addends=(None, None), stats=(None, None),
             x2=None):
    """Convolution (mode 0; one or two weight sets -> outs) or data gradient (mode 1; with x2: the sum of the data
    gradients of two convolutions of the same input) from packed weight forms."""
    npair = len(outs) if mode == 0 else (2 if x2 is not None else 1)
    epi = (ctypes.c_int32 * 2)(int(epilogues[0]), int(epilogues[1]) if len(outs) > 1 else 0)
    pad = lambda v: (tuple(v) + (None, None))[:2]
    L.check(L.lib().seld_hcq_conv(ctypes.byref(desc), int(mode), npair, L.ptr(_req(x, "x")), L.ptr(_req(x2, "x2")),
                                  L.ptr(wpack), _ptr2(*pad(outs)), _ptr2(*pad(biases)), epi, _ptr2(*pad(addends)),
                                  _ptr2(*pad(stats)), L.current_stream()), "seld_hcq_conv")
    return outs


# ======================================================================================
# conv with fused epilogue (bias / residual add) as an autograd op
# ======================================================================================
class HyperConvAddFn(torch.autograd.Function):
    """y = W (x) x + addend  -- the `x + conv2_residual(y)` of model.py:132 and the running
    skip-connection sum of model.py:210-212 ride in the conv epilogue (SELD_EPI_ADD)."""

    @staticmethod
    def forward(ctx, x, bias, addend, stride, padding, dilation, *ws):
        algebra = len(ws)
        k = tuple(ws[0].shape[2:])
        desc = make_conv_desc(tuple(x.shape), ws[0].shape[0] * algebra, algebra, k, stride, padding, dilation)
        x = _req(x, "x")
        y = conv_fwd(desc, x, ws, bias, epilogue=L.SELD_EPI_ADD, addend=addend)
        ctx.desc = desc
        ctx.has_bias = bias is not None
        ctx.w_params, ctx.bias_param = ws, bias
        ctx.wt_ahead = _transpose_ahead(desc, ws) if ctx.needs_input_grad[0] else None
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        dx, dbias, dws = _conv_backward(ctx, dy, 6)
        return (dx, dbias, dy if ctx.needs_input_grad[2] else None, None, None, None, *dws)


def hyper_conv_add(x, ws, bias, addend, stride, padding, dilation):
    return HyperConvAddFn.apply(x, bias, addend, stride, padding, dilation, *ws)


# ======================================================================================
# two convolutions of one geometry on the same input in one launch
# ======================================================================================
_pair_ok_cache = {}


def _pair_ok(desc, which):
    key = (bytes(desc), which)
    v = _pair_ok_cache.get(key)
    if v is None:
        v = _pair_ok_cache[key] = bool(L.lib().seld_hc_conv_pair_supported(ctypes.byref(desc), which))
    return v


class HyperConvPairFn(torch.autograd.Function):
    """(yA, yB) = (WA (x) x [+ addA], WB (x) x [+ addB]): conv1_filter | conv1_gate (model.py:121-122) and
    conv2_skip | conv2_residual (model.py:130-132, 210-212) of a residual block, one launch each way
    (seld_hc_conv_pair_*).  Falls back to the single entry points per direction when a shape does not qualify."""

    @staticmethod
    def forward(ctx, x, biasA, biasB, addA, addB, stride, padding, dilation, algebra, statsA, statsB, *ws):
        wsA, wsB = ws[:algebra], ws[algebra:]
        k = tuple(wsA[0].shape[2:])
        desc = make_conv_desc(tuple(x.shape), wsA[0].shape[0] * algebra, algebra, k, stride, padding, dilation)
        x = _req(x, "x")
        o = conv_out_shape(desc)
        yA = torch.empty(_y_shape(desc, o), device=x.device, dtype=torch.float32)
        yB = torch.empty_like(yA)
        det_stats = deterministic() and (statsA is not None or statsB is not None)
        kstA, kstB = (None, None) if det_stats else (statsA, statsB)         # statistics the kernels gather themselves
        epiA = (L.SELD_EPI_ADD if addA is not None else 0) | (L.SELD_EPI_STATS if kstA is not None else 0)
        epiB = (L.SELD_EPI_ADD if addB is not None else 0) | (L.SELD_EPI_STATS if kstB is not None else 0)
        one_launch = all(v == 1 for v in k)          # 1x1 pairs run the pair instantiation of the kernel
        # (issuing the second of two launches on the side stream was measured: 14.69 vs 14.58 ms per step, not kept)
        wp = hcq_weights.get(desc, 0, wsA, wsB) if algebra > 1 else None
        if wp is not None:                           # both convolutions in one launch of the fast-product kernel
            with _Timed(desc, 0, 2, label=_hcq_label_cached(desc, 0, 2) if kernel_timer.active else None):
                hcq_conv(desc, 0, x, wp, (yA, yB), (_req(biasA, "bias"), _req(biasB, "bias")), (epiA, epiB),
                         (_req(addA, "addend"), _req(addB, "addend")), (kstA, kstB))
        else:
            with _Timed(desc, 0, 2, one_launch):
                rc = L.lib().seld_hc_conv_pair_fwd(
                    ctypes.byref(desc), L.ptr(x), L.ptr_array8([_req(w, "w") for w in wsA]),
                    L.ptr_array8([_req(w, "w") for w in wsB]), L.ptr(_req(biasA, "bias")), L.ptr(_req(biasB, "bias")),
                    L.ptr(yA), L.ptr(yB), ctypes.c_int32(epiA), ctypes.c_int32(epiB), L.ptr(_req(addA, "addend")),
                    L.ptr(_req(addB, "addend")), L.ptr(kstA), L.ptr(kstB), L.current_stream())
            if rc == -4:       # SELD_EUNSUPPORTED: e.g. the two weight sets lie more than 4 GB apart
                conv_fwd(desc, x, wsA, biasA, out=yA, epilogue=epiA, addend=addA, stats=kstA)
                conv_fwd(desc, x, wsB, biasB, out=yB, epilogue=epiB, addend=addB, stats=kstB)
            else:
                L.check(rc, "seld_hc_conv_pair_fwd")
        if det_stats:
            if statsA is not None:
                channel_stats(yA, out=statsA)
            if statsB is not None:
                channel_stats(yB, out=statsB)
        ctx.desc, ctx.algebra = desc, algebra
        ctx.params = (wsA, wsB, biasA, biasB)
        ctx.wt_ahead = None
        if ctx.needs_input_grad[0] and _pair_ok(desc, 1) and not _hcq_ok(desc, 1, 2):
            a_, b_ = _transpose_ahead(desc, wsA), _transpose_ahead(desc, wsB)
            ctx.wt_ahead = (a_, b_) if a_ is not None and b_ is not None else None
        ctx.save_for_backward(x)
        return yA, yB

    @staticmethod
    def backward(ctx, dyA, dyB):
        (x,) = ctx.saved_tensors
        desc, A = ctx.desc, ctx.algebra
        wsA, wsB, biasA, biasB = ctx.params
        dyA, dyB = _req(dyA, "dy"), _req(dyB, "dy")
        lib = L.lib()
        dx = None
        if ctx.needs_input_grad[0]:
            wp = hcq_weights.get(desc, 1, wsA, wsB) if A > 1 else None
            if wp is not None:                       # sum of both data gradients in one launch
                dx = torch.empty(tuple(x.shape), device=x.device, dtype=torch.float32)
                with _Timed(desc, 1, 2, label=_hcq_label_cached(desc, 1, 2) if kernel_timer.active else None):
                    hcq_conv(desc, 1, dyA, wp, (dx,), x2=dyB)
            elif _pair_ok(desc, 1) and ctx.wt_ahead is not None:
                (wtA, evA), (wtB, evB) = ctx.wt_ahead
                dx = torch.empty(tuple(x.shape), device=x.device, dtype=torch.float32)
                torch.cuda.current_stream().wait_event(evA)
                torch.cuda.current_stream().wait_event(evB)
                with _Timed(desc, 1, 2):
                    L.check(lib.seld_hc_conv_pair_bwd_data_wt(ctypes.byref(desc), L.ptr(dyA), L.ptr(dyB), L.ptr(wtA),
                                                              L.ptr(wtB), L.ptr(dx), L.current_stream()),
                            "seld_hc_conv_pair_bwd_data_wt")
            elif _pair_ok(desc, 1):
                dx = torch.empty(tuple(x.shape), device=x.device, dtype=torch.float32)
                lib.seld_hc_conv_bwd_data_workspace.restype = ctypes.c_size_t
                nbytes = 2 * lib.seld_hc_conv_bwd_data_workspace(ctypes.byref(desc))
                wsb = torch.empty((nbytes + 3) // 4, device=x.device, dtype=torch.float32)
                with _Timed(desc, 1, 2):
                    L.check(lib.seld_hc_conv_pair_bwd_data(ctypes.byref(desc), L.ptr(dyA), L.ptr(dyB),
                                                           L.ptr_array8(list(wsA)), L.ptr_array8(list(wsB)), L.ptr(dx),
                                                           L.ptr(wsb), ctypes.c_size_t(nbytes), L.current_stream()),
                            "seld_hc_conv_pair_bwd_data")
            else:
                dx = conv_bwd_data(desc, dyA, wsA, tuple(x.shape))
                dx += conv_bwd_data(desc, dyB, wsB, tuple(x.shape))
        need_w = any(ctx.needs_input_grad[11:]) or ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        dwsA, dwsB, dbA, dbB = [None] * A, [None] * A, None, None
        if need_w:
            dirA, dirB = _direct_targets(wsA, biasA), _direct_targets(wsB, biasB)
            if dirA is not None and dirB is not None and dirA[1] is None and dirB[1] is None and deferred_wgrads.takes(desc):
                deferred_wgrads.add(desc, x, dyA, dirA[0])
                deferred_wgrads.add(desc, x, dyB, dirB[0])
            elif dirA is not None and dirB is not None and dirA[1] is None and dirB[1] is None and _hcq_wgrad_ok(desc, 2):
                def pair_wgrad_fast():
                    hcq_wgrad_acc(desc, x, dyA, dirA[0], dyB, dirB[0])
                if _side_enabled():
                    _on_side_stream(pair_wgrad_fast, x, dyA, dyB)
                else:
                    pair_wgrad_fast()
            elif dirA is not None and dirB is not None and _pair_ok(desc, 2) and not deterministic():
                def pair_wgrad():
                    with _Timed(desc, 2, 2):
                        L.check(lib.seld_hc_conv_pair_bwd_weight_acc(ctypes.byref(desc), L.ptr(x), L.ptr(dyA), L.ptr(dyB),
                                                                     L.ptr_array8(dirA[0]), L.ptr_array8(dirB[0]),
                                                                     L.ptr(dirA[1]), L.ptr(dirB[1]), L.current_stream()),
                                "seld_hc_conv_pair_bwd_weight_acc")
                if _side_enabled():
                    _on_side_stream(pair_wgrad, x, dyA, dyB)
                else:
                    pair_wgrad()
            else:
                for ws_, b_, dy_, tgt in ((wsA, biasA, dyA, "A"), (wsB, biasB, dyB, "B")):
                    d_ = _direct_targets(ws_, b_)
                    if d_ is not None:
                        conv_bwd_weight(desc, x, dy_, tuple(ws_[0].shape), b_ is not None, into=d_[0], bias_into=d_[1])
                    else:
                        g, gb = conv_bwd_weight(desc, x, dy_, tuple(ws_[0].shape), b_ is not None)
                        if tgt == "A":
                            dwsA, dbA = g, gb
                        else:
                            dwsB, dbB = g, gb
        return (dx, dbA, dbB, dyA if ctx.needs_input_grad[3] else None, dyB if ctx.needs_input_grad[4] else None,
                None, None, None, None, None, None, *dwsA, *dwsB)


def hyper_conv_pair(x, wsA, biasA, wsB, biasB, stride, padding, dilation, addA=None, addB=None, statsA=None, statsB=None):
    """Two convolutions of the same input.  One call when both have the same shape (and one launch per direction
    where the kernels support the pair form); otherwise exactly the two single calls.  statsA / statsB: zeroed
    statistics buffers (`new_stats`) to receive the BatchNorm batch statistics of the two results."""
    same = (len(wsA) == len(wsB) and tuple(wsA[0].shape) == tuple(wsB[0].shape) and
            (biasA is None) == (biasB is None) and x.is_cuda)
    if same:
        k = tuple(wsA[0].shape[2:])
        desc = make_conv_desc(tuple(x.shape), wsA[0].shape[0] * len(wsA), len(wsA), k, stride, padding, dilation)
        if _pair_ok(desc, 0):
            return HyperConvPairFn.apply(x, biasA, biasB, addA, addB, stride, padding, dilation, len(wsA), statsA, statsB,
                                         *wsA, *wsB)

    def one(ws, bias, add, stats):
        y = hyper_conv(x, ws, bias, stride, padding, dilation) if add is None else \
            hyper_conv_add(x, ws, bias, add, stride, padding, dilation)
        if stats is not None:
            N, C, S = _ncs(y)
            L.check(L.lib().seld_channel_stats(L.ptr(y), N, C, S, L.ptr(stats), L.current_stream()), "seld_channel_stats")
        return y
    return one(wsA, biasA, addA, statsA), one(wsB, biasB, addB, statsB)


# ======================================================================================
# BatchNorm + activation
# ======================================================================================
def _ncs(x):
    N, C = x.shape[0], x.shape[1]
    S = 1
    for d in x.shape[2:]:
        S *= d
    return N, C, S


STATS_REPLICAS = 64      # SELD_STATS_REPLICAS


_stats_pool = {}     # (C, device, stream) -> zero-filled statistics buffers ready for reuse


def new_stats(C, device):
    """Zeroed BatchNorm statistics buffer: SELD_STATS_REPLICAS rows of [sum(C) | sum of squares(C)].  Buffers are
    pooled: `bn_prepare` hands one back after `seld_bn_finalize_ex` has consumed AND re-zeroed it, so a training step
    does not launch a fill per BatchNorm."""
    free = _stats_pool.get((C, device, torch.cuda.current_stream(device).cuda_stream))
    if free:
        return free.pop()
    t = torch.zeros(STATS_REPLICAS * 2 * C, device=device, dtype=torch.float32)
    t._seld_pooled = True
    return t


def channel_stats(x, out=None):
    N, C, S = _ncs(x)
    stats = out if out is not None else new_stats(C, x.device)
    L.check(L.lib().seld_channel_stats(L.ptr(x), N, C, S, L.ptr(stats), L.current_stream()), "seld_channel_stats")
    return stats


def bn_prepare(x, running_mean, running_var, training, momentum, eps, stats=None, num_batches_tracked=None):
    """mean / invstd used by the normalisation; in training mode also the running-buffer update and the
    num_batches_tracked increment of torch.nn.BatchNorm, all in one launch."""
    N, C, S = _ncs(x)
    mean = torch.empty(C, device=x.device, dtype=torch.float32)
    invstd = torch.empty(C, device=x.device, dtype=torch.float32)
    if training:
        if stats is None:
            stats = channel_stats(x)
        pooled = getattr(stats, "_seld_pooled", False)
        L.check(L.lib().seld_bn_finalize_ex(L.ptr(stats), C, ctypes.c_int64(N * S), ctypes.c_float(eps),
                                            ctypes.c_float(momentum), L.ptr(mean), L.ptr(invstd), L.ptr(running_mean),
                                            L.ptr(running_var), L.ptr(num_batches_tracked), int(pooled),
                                            L.current_stream()), "seld_bn_finalize_ex")
        if pooled:      # per stream: the buffer is re-zeroed by a kernel on THIS stream and may only be reused in order behind it
            _stats_pool.setdefault((C, x.device, torch.cuda.current_stream(x.device).cuda_stream), []).append(stats)
    else:
        L.check(L.lib().seld_bn_eval_stats(L.ptr(running_mean), L.ptr(running_var), C, ctypes.c_float(eps),
                                           L.ptr(mean), L.ptr(invstd), L.current_stream()), "seld_bn_eval_stats")
    return mean, invstd


def _nbt(bn):
    """The module's num_batches_tracked buffer when torch.nn.BatchNorm would increment it, else None."""
    if bn.training and getattr(bn, "track_running_stats", True) and bn.num_batches_tracked is not None:
        return bn.num_batches_tracked
    return None


def _claim_grad_slots(params, adjacent=True):
    """Gradient slots of `params` as reduction targets.  Returns (first_slot, clean):
    first_slot -- `.grad` of the first parameter when every parameter opted in (`_seld_direct_grad`, FlatAdam) and,
                  if `adjacent`, the slots follow one another in the flat gradient buffer in this order; else None;
    clean      -- nothing has been written to the slots since the owner's last zero_grad(): the backward kernels may
                  then use the slots themselves as their (zero-initialised) output / reduction buffers.
    Marks the slots written."""
    ts = list(params)
    if not all(getattr(t, "_seld_direct_grad", False) and t.grad is not None for t in ts):
        return None, False
    if adjacent:
        at = ts[0].grad.data_ptr()
        for t in ts:
            if t.grad.data_ptr() != at:
                return None, False
            at += 4 * t.numel()
    owner = getattr(ts[0], "_seld_owner", None)
    gen = owner.grad_generation if owner is not None else None
    clean = gen is not None and all(getattr(t, "_seld_owner", None) is owner and
                                    getattr(t, "_seld_written", None) != gen for t in ts)
    for t in ts:
        t._seld_written = gen
    return ts[0].grad, clean


def _one_pass_ok(N, S, limit):
    """The one-workgroup-per-channel backward kernels (csrc/nn_ops.hip) hold a channel's N*S values in registers."""
    return S % 4 == 0 and N * S <= limit and not os.environ.get("SELD_BN_TWO_PASS")


class BnActFn(torch.autograd.Function):
    """y = act(BatchNorm(x)); torch.nn.BatchNorm1d/2d + ReLU/Tanh of model.py:114-116, 279-280."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, training, momentum, eps, act, stats, nbt, twin):
        x = _req(x, "x")
        N, C, S = _ncs(x)
        mean, invstd = bn_prepare(x, running_mean, running_var, training, momentum, eps, stats, nbt)
        y = torch.empty_like(x)
        L.check(L.lib().seld_bn_act_fwd(L.ptr(x), N, C, S, L.ptr(mean), L.ptr(invstd), L.ptr(gamma), L.ptr(beta),
                                        act, L.ptr(y), L.current_stream()), "seld_bn_act_fwd")
        ctx.training, ctx.act = training, act
        ctx.bn_params = (gamma, beta)
        ctx.save_for_backward(x, y, mean, invstd)
        ctx.twin = twin
        if twin:
            # the same values twice: each consumer's gradient then arrives separately and backward adds them while
            # it loads them (the autograd engine would launch an add kernel for a tensor used twice)
            ctx.set_materialize_grads(False)
            return y, y.view_as(y)
        return y

    @staticmethod
    def backward(ctx, dy, dy2=None):
        x, y, mean, invstd = ctx.saved_tensors
        gamma, beta = ctx.bn_params
        if dy is None:
            dy, dy2 = dy2, None
        if dy is None:
            return (None,) * 12
        dy = _req(dy, "dy")
        dy2 = _req(dy2, "dy2") if dy2 is not None else None
        none = (None,) * 9
        N, C, S = _ncs(x)
        slot, clean = _claim_grad_slots((gamma, beta))
        st = L.current_stream()
        if ctx.training and _one_pass_ok(N, S, 32768):
            # one workgroup per channel: reads every operand once and ADDS [dgamma | dbeta] to `red`
            red = slot if slot is not None else torch.zeros(2 * C, device=x.device, dtype=torch.float32)
            dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
            L.check(L.lib().seld_bn_act_bwd_fused(L.ptr(dy), L.ptr(x), L.ptr(y), N, C, S, L.ptr(mean), L.ptr(invstd),
                                                  L.ptr(gamma), ctx.act, L.ptr(red), L.ptr(dy2), L.ptr(dx), st),
                    "seld_bn_act_bwd_fused")
            if slot is not None:
                return (dx, None, None) + none
            return (dx, red[:C], red[C:]) + none
        if dy2 is not None:
            dy = dy + dy2
        # [dgamma | dbeta]: reduced straight into the (still zero) flat-gradient slots when possible
        red = slot if clean else torch.zeros(2 * C, device=x.device, dtype=torch.float32)
        L.check(L.lib().seld_bn_act_bwd_reduce(L.ptr(dy), L.ptr(x), L.ptr(y), N, C, S, L.ptr(mean), L.ptr(invstd),
                                               L.ptr(gamma), L.ptr(beta), ctx.act, L.ptr(red), st),
                "seld_bn_act_bwd_reduce")
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            L.check(L.lib().seld_bn_act_bwd_apply(L.ptr(dy), L.ptr(x), L.ptr(y), N, C, S, L.ptr(mean), L.ptr(invstd),
                                                  L.ptr(gamma), L.ptr(beta), ctx.act, L.ptr(red), int(ctx.training),
                                                  L.ptr(dx), st), "seld_bn_act_bwd_apply")
        if slot is not None:
            if not clean:
                axpy_(slot, red, 2 * C)
            return (dx, None, None) + none
        return (dx, red[:C], red[C:]) + none


def bn_act(x, bn, act, stats=None, twin=False):
    """`bn` is a torch.nn.BatchNorm*-shaped module (weight, bias, running_mean, running_var, ...).  twin=True returns
    the result twice (two tensors, one storage) for a result with two consumers: see BnActFn.forward."""
    return BnActFn.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.training,
                         bn.momentum if bn.momentum is not None else 0.1, bn.eps, act, stats, _nbt(bn), twin)


class GateFn(torch.autograd.Function):
    """y = tanh(BN_f(yf)) * sigmoid(BN_g(yg)) * channel_mask  (model.py:121-128)."""

    @staticmethod
    def forward(ctx, yf, yg, gf, bf, rmf, rvf, gg, bg, rmg, rvg, training, momentum, eps, mask, nbt_f, nbt_g,
                stats_f=None, stats_g=None):
        yf, yg = _req(yf, "yf"), _req(yg, "yg")
        N, C, S = _ncs(yf)
        if (training and stats_f is not None and stats_g is not None and getattr(stats_f, "_seld_pooled", False)
                and getattr(stats_g, "_seld_pooled", False)):
            # both layers' statistics are ready (the pair convolution gathered them): one finalize launch for the two
            mf, isf, mg, isg = (torch.empty(C, device=yf.device, dtype=torch.float32) for _ in range(4))
            L.check(L.lib().seld_bn_finalize2_ex(L.ptr(stats_f), L.ptr(stats_g), C, ctypes.c_int64(N * S), ctypes.c_float(eps),
                                                 ctypes.c_float(momentum), L.ptr(mf), L.ptr(isf), L.ptr(rmf), L.ptr(rvf),
                                                 L.ptr(nbt_f), L.ptr(mg), L.ptr(isg), L.ptr(rmg), L.ptr(rvg), L.ptr(nbt_g),
                                                 1, L.current_stream()), "seld_bn_finalize2_ex")
            key = (C, yf.device, torch.cuda.current_stream(yf.device).cuda_stream)
            _stats_pool.setdefault(key, []).extend((stats_f, stats_g))
        else:
            mf, isf = bn_prepare(yf, rmf, rvf, training, momentum, eps, stats_f, nbt_f)
            mg, isg = bn_prepare(yg, rmg, rvg, training, momentum, eps, stats_g, nbt_g)
        y = torch.empty_like(yf)
        L.check(L.lib().seld_gate_fwd(L.ptr(yf), L.ptr(yg), N, C, S, L.ptr(mf), L.ptr(isf), L.ptr(gf), L.ptr(bf),
                                      L.ptr(mg), L.ptr(isg), L.ptr(gg), L.ptr(bg), L.ptr(mask), L.ptr(y),
                                      L.current_stream()), "seld_gate_fwd")
        ctx.training = training
        ctx.has_mask = mask is not None
        ctx.bn_params = (gf, bf, gg, bg)
        ctx.save_for_backward(yf, yg, mf, isf, mg, isg, *([mask] if mask is not None else []))
        return y

    @staticmethod
    def backward(ctx, dy):
        yf, yg, mf, isf, mg, isg, *rest = ctx.saved_tensors
        gf, bf, gg, bg = ctx.bn_params
        mask = rest[0] if ctx.has_mask else None
        dy = _req(dy, "dy")
        N, C, S = _ncs(yf)
        slot, clean = _claim_grad_slots((gf, bf, gg, bg))
        st = L.current_stream()
        if ctx.training and _one_pass_ok(N, S, 16384):
            red = slot if slot is not None else torch.zeros(4 * C, device=yf.device, dtype=torch.float32)
            dyf, dyg = torch.empty_like(yf), torch.empty_like(yg)
            L.check(L.lib().seld_gate_bwd_fused(L.ptr(dy), L.ptr(yf), L.ptr(yg), N, C, S, L.ptr(mf), L.ptr(isf),
                                                L.ptr(gf), L.ptr(bf), L.ptr(mg), L.ptr(isg), L.ptr(gg), L.ptr(bg),
                                                L.ptr(mask), L.ptr(red), L.ptr(dyf), L.ptr(dyg), st),
                    "seld_gate_bwd_fused")
            if slot is not None:
                return (dyf, dyg) + (None,) * 16
            return (dyf, dyg, red[:C], red[C:2 * C], None, None, red[2 * C:3 * C], red[3 * C:], None, None,
                    None, None, None, None, None, None, None, None)
        # [dgamma_f | dbeta_f | dgamma_g | dbeta_g]
        red = slot if clean else torch.zeros(4 * C, device=yf.device, dtype=torch.float32)
        args = (L.ptr(dy), L.ptr(yf), L.ptr(yg), N, C, S, L.ptr(mf), L.ptr(isf), L.ptr(gf), L.ptr(bf),
                L.ptr(mg), L.ptr(isg), L.ptr(gg), L.ptr(bg), L.ptr(mask))
        L.check(L.lib().seld_gate_bwd_reduce(*args, L.ptr(red), st), "seld_gate_bwd_reduce")
        dyf, dyg = torch.empty_like(yf), torch.empty_like(yg)
        L.check(L.lib().seld_gate_bwd_apply(*args, L.ptr(red), int(ctx.training), L.ptr(dyf), L.ptr(dyg), st),
                "seld_gate_bwd_apply")
        if slot is not None:
            if not clean:
                axpy_(slot, red, 4 * C)
            return (dyf, dyg) + (None,) * 16
        return (dyf, dyg, red[:C], red[C:2 * C], None, None, red[2 * C:3 * C], red[3 * C:], None, None,
                None, None, None, None, None, None, None, None)


def gate(yf, yg, bn_f, bn_g, mask=None, stats_f=None, stats_g=None):
    return GateFn.apply(yf, yg, bn_f.weight, bn_f.bias, bn_f.running_mean, bn_f.running_var,
                        bn_g.weight, bn_g.bias, bn_g.running_mean, bn_g.running_var, bn_f.training,
                        bn_f.momentum if bn_f.momentum is not None else 0.1, bn_f.eps, mask, _nbt(bn_f), _nbt(bn_g),
                        stats_f, stats_g)


# ======================================================================================
# activations, pooling, dropout, transposes
# ======================================================================================
class ActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, act):
        x = _req(x, "x")
        y = torch.empty_like(x)
        L.check(L.lib().seld_act_fwd(L.ptr(x), ctypes.c_int64(x.numel()), act, L.ptr(y), L.current_stream()), "seld_act_fwd")
        ctx.act = act
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = _req(dy, "dy")
        dx = torch.empty_like(y)
        L.check(L.lib().seld_act_bwd(L.ptr(dy), L.ptr(y), ctypes.c_int64(y.numel()), ctx.act, L.ptr(dx),
                                     L.current_stream()), "seld_act_bwd")
        return dx, None


def act(x, kind):
    return ActFn.apply(x, kind)


class MaxPoolFn(torch.autograd.Function):
    """torch.nn.MaxPool1d / MaxPool2d with stride == window (model.py:178,192,202,281)."""

    @staticmethod
    def forward(ctx, x, ph, pw):
        x = _req(x, "x")
        if x.dim() == 3:
            N, C, H, W = x.shape[0], x.shape[1], 1, x.shape[2]
            oshape = (N, C, W // pw)
        else:
            N, C, H, W = x.shape
            oshape = (N, C, H // ph, W // pw)
        y = torch.empty(oshape, device=x.device, dtype=torch.float32)
        idx = torch.empty(oshape, device=x.device, dtype=torch.uint8)
        L.check(L.lib().seld_maxpool_fwd(L.ptr(x), ctypes.c_int64(N * C), H, W, ph, pw, L.ptr(y), L.ptr(idx),
                                         L.current_stream()), "seld_maxpool_fwd")
        ctx.geom = (N * C, H, W, ph, pw, tuple(x.shape))
        ctx.save_for_backward(idx)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        NC, H, W, ph, pw, xshape = ctx.geom
        dy = _req(dy, "dy")
        dx = torch.empty(xshape, device=dy.device, dtype=torch.float32)
        L.check(L.lib().seld_maxpool_bwd(L.ptr(dy), L.ptr(idx), ctypes.c_int64(NC), H, W, ph, pw, L.ptr(dx),
                                         L.current_stream()), "seld_maxpool_bwd")
        return dx, None, None


def maxpool(x, ph, pw):
    if ph == 1 and pw == 1:
        return x
    return MaxPoolFn.apply(x, int(ph), int(pw))


class _Philox:
    """Counter-based RNG bookkeeping for the dropout kernels.  The key is torch's seed (torch.manual_seed controls it)
    mixed with `stream_id` -- the data-parallel rank, so that ranks draw different masks for their different shards;
    the counter of a draw is  host offset (advances by the number of 128-bit draws) + device base.

    The device base is word 0 of the per-device STEP STATE (4 x uint64, see seld_step_begin in include/seld_hip.h):
    it is zero in eager mode; a step recorded as a HIP graph (train.GraphedTrainStep) is captured with host offsets that
    start at zero and replays with the base advanced by the draws of one step, so every replay sees fresh masks."""

    def __init__(self):
        self.offset = 0
        self.stream_id = 0
        self._state = {}

    def seed(self):
        return (torch.initial_seed() + 0x9E3779B97F4A7C15 * int(self.stream_id)) & 0xFFFFFFFFFFFFFFFF

    def state(self, device):
        """The step state tensor of `device` (int64[4], uint64 semantics): [philox base, step, lr bits, draws/step]."""
        device = torch.device(device)
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        t = self._state.get(device)
        if t is None:
            t = self._state[device] = torch.zeros(4, device=device, dtype=torch.int64)
        return t

    def draw(self, n_groups, device):
        """(seed, offset, step-state tensor) for a kernel that consumes `n_groups` 128-bit draws."""
        off = self.offset
        self.offset += int(n_groups)
        return self.seed(), off, self.state(device)

    def get_offset(self):
        """Absolute position in the stream (checkpointed by train.save_model): host offset + device base."""
        base = sum(int(t[0].item()) for t in self._state.values())
        return self.offset + base

    def set_offset(self, offset):
        self.offset = int(offset)
        for t in self._state.values():
            t[0] = 0


philox = _Philox()


class DropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p):
        x = _req(x, "x")
        n = x.numel()
        seed, off, state = philox.draw((n + 3) // 4, x.device)
        y = torch.empty_like(x)
        L.check(L.lib().seld_dropout_fwd(L.ptr(x), ctypes.c_int64(n), ctypes.c_float(p), ctypes.c_uint64(seed),
                                         ctypes.c_uint64(off), L.ptr(state), L.ptr(y), L.current_stream()),
                "seld_dropout_fwd")
        ctx.rng = (p, seed, off, state)
        return y

    @staticmethod
    def backward(ctx, dy):
        p, seed, off, state = ctx.rng
        dy = _req(dy, "dy")
        dx = torch.empty_like(dy)
        L.check(L.lib().seld_dropout_fwd(L.ptr(dy), ctypes.c_int64(dy.numel()), ctypes.c_float(p),
                                         ctypes.c_uint64(seed), ctypes.c_uint64(off), L.ptr(state), L.ptr(dx),
                                         L.current_stream()), "seld_dropout_fwd")
        return dx, None


def dropout(x, p, training):
    if not training or p == 0.0:
        return x
    return DropoutFn.apply(x, float(p))


def channel_dropout_mask(N, C, p, device):
    """Dropout1d decision per (n, c) row, already scaled by 1/(1-p) (model.py:96-97,127-128)."""
    rows = N * C
    seed, off, state = philox.draw((rows + 3) // 4, device)
    mask = torch.empty(rows, device=device, dtype=torch.float32)
    L.check(L.lib().seld_dropout_mask_rows(ctypes.c_int64(rows), ctypes.c_float(p), ctypes.c_uint64(seed),
                                           ctypes.c_uint64(off), L.ptr(state), L.ptr(mask), L.current_stream()),
            "seld_dropout_mask_rows")
    return mask


class RowScaleFn(torch.autograd.Function):
    """y[n, c, :] = x[n, c, :] * mask[n*C + c]  (stand-alone Dropout1d; inside ResBlock the mask rides in the gate kernel)."""

    @staticmethod
    def forward(ctx, x, mask):
        ctx.save_for_backward(mask)
        return _rowscale(x, mask)

    @staticmethod
    def backward(ctx, dy):
        (mask,) = ctx.saved_tensors
        return _rowscale(_req(dy, "dy"), mask), None


def _rowscale(x, mask):
    # gate kernel with neutral BN constants would be overkill; use bn_act_fwd with gamma = mask per (n,c):
    # treat (N, C, S) as (1, N*C, S) with mean 0, invstd 1, gamma = mask, beta = 0
    x = _req(x, "x")
    N, C, S = _ncs(x)
    zeros = torch.zeros(N * C, device=x.device, dtype=torch.float32)
    ones = torch.ones(N * C, device=x.device, dtype=torch.float32)
    y = torch.empty_like(x)
    L.check(L.lib().seld_bn_act_fwd(L.ptr(x), 1, N * C, S, L.ptr(zeros), L.ptr(ones), L.ptr(mask), L.ptr(zeros),
                                    L.SELD_ACT_NONE, L.ptr(y), L.current_stream()), "seld_bn_act_fwd")
    return y


class TransposeFn(torch.autograd.Function):
    """(N, A, B) -> (N, B, A) contiguous."""

    @staticmethod
    def forward(ctx, x):
        x = _req(x, "x")
        N, A, B = x.shape
        y = torch.empty((N, B, A), device=x.device, dtype=torch.float32)
        L.check(L.lib().seld_transpose_nct_ntc(L.ptr(x), N, A, B, L.ptr(y), L.current_stream()), "seld_transpose")
        return y

    @staticmethod
    def backward(ctx, dy):
        return TransposeFn.apply(dy)


def transpose12(x):
    return TransposeFn.apply(x)


# ======================================================================================
# linear layers
# ======================================================================================
class HyperLinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bias, kind, *ws):
        x2 = _req(x.reshape(-1, x.shape[-1]), "x")
        rows, in_f = x2.shape
        if kind == L.SELD_LIN_REAL:
            out_f = ws[0].shape[0]
        else:
            out_f = ws[0].shape[1] * kind
        ctx.params = (bias, tuple(ws))
        ws = [_req(w, "w") for w in ws]
        y = torch.empty((rows, out_f), device=x.device, dtype=torch.float32)
        L.check(L.lib().seld_hc_linear_fwd(kind, rows, in_f, out_f, L.ptr(x2), L.ptr_array8(ws), L.ptr(_req(bias, "bias")),
                                           L.ptr(y), L.current_stream()), "seld_hc_linear_fwd")
        ctx.meta = (kind, rows, in_f, out_f, tuple(x.shape), bias is not None)
        ctx.save_for_backward(x2, *ws)
        return y.reshape(*x.shape[:-1], out_f)

    @staticmethod
    def backward(ctx, dy):
        kind, rows, in_f, out_f, xshape, has_bias = ctx.meta
        x2, *ws = ctx.saved_tensors
        bias_p, ws_p = ctx.params
        dy2 = _req(dy.reshape(rows, out_f), "dy")
        dev = dy2.device
        dx = torch.empty((rows, in_f), device=dev, dtype=torch.float32) if ctx.needs_input_grad[0] else None
        need_w = any(ctx.needs_input_grad[3:])
        need_b = has_bias and ctx.needs_input_grad[1]
        # The kernel WRITES the parameter gradients: it may write them straight into flat-gradient slots that are
        # still zero (first and only use of the layer since zero_grad); otherwise autograd accumulates.
        direct = False
        if need_w and (need_b or not has_bias) and all(w.is_contiguous() for w in ws_p):
            slot, clean = _claim_grad_slots(list(ws_p) + ([bias_p] if has_bias else []), adjacent=False)
            direct = slot is not None and clean
        if direct:
            dws, dbias = [w.grad for w in ws_p], (bias_p.grad if has_bias else None)
        else:
            dws = [torch.empty_like(w) for w in ws] if need_w else None
            dbias = torch.empty(out_f, device=dev, dtype=torch.float32) if need_b else None
        nbytes = L.lib().seld_hc_linear_bwd_workspace(kind, in_f, out_f)
        wsb = torch.empty((nbytes + 3) // 4, device=dev, dtype=torch.float32)
        L.check(L.lib().seld_hc_linear_bwd(kind, rows, in_f, out_f, L.ptr(x2), L.ptr(dy2), L.ptr_array8(ws), L.ptr(dx),
                                           L.ptr_array8(dws) if dws is not None else None, L.ptr(dbias), L.ptr(wsb),
                                           ctypes.c_size_t(nbytes), L.current_stream()), "seld_hc_linear_bwd")
        dxr = dx.reshape(xshape) if dx is not None else None
        if direct:
            return (dxr, None, None, *([None] * len(ws)))
        return (dxr, dbias, None, *(dws if dws is not None else [None] * len(ws)))


def hyper_linear(x, ws, bias, kind):
    return HyperLinearFn.apply(x, bias, kind, *ws)


# ======================================================================================
# attention core, loss, Adam, STFT
# ======================================================================================
class MhaCoreFn(torch.autograd.Function):
    """softmax(q k^T / sqrt(hd)) v on (N, E, T) tensors (model.py:39-48)."""

    @staticmethod
    def forward(ctx, q, k, v, heads):
        q, k, v = _req(q, "q"), _req(k, "k"), _req(v, "v")
        N, E, T = q.shape
        hd = E // heads
        out = torch.empty_like(q)
        lse = torch.empty((N, heads, T), device=q.device, dtype=torch.float32)
        L.check(L.lib().seld_mha_fwd(L.ptr(q), L.ptr(k), L.ptr(v), N, T, heads, hd, L.ptr(out), L.ptr(lse),
                                     L.current_stream()), "seld_mha_fwd")
        ctx.geom = (N, T, heads, hd)
        ctx.save_for_backward(q, k, v, out, lse)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, out, lse = ctx.saved_tensors
        N, T, H, hd = ctx.geom
        dout = _req(dout, "dout")
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        lib = L.lib()
        lib.seld_mha_bwd_workspace.restype = ctypes.c_size_t
        nbytes = lib.seld_mha_bwd_workspace(N, T, H)
        wsb = torch.empty((nbytes + 3) // 4, device=q.device, dtype=torch.float32)
        L.check(lib.seld_mha_bwd(L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(out), L.ptr(dout), L.ptr(lse), N, T, H, hd,
                                 L.ptr(dq), L.ptr(dk), L.ptr(dv), L.ptr(wsb), ctypes.c_size_t(nbytes),
                                 L.current_stream()), "seld_mha_bwd")
        return dq, dk, dv, None


def mha_core(q, k, v, heads):
    return MhaCoreFn.apply(q, k, v, heads)


class MhaPackedFn(torch.autograd.Function):
    """The same attention on ONE projected tensor qkv (N, 3E, T) = [values | keys | queries] (seld_mha_fwd_packed): the
    three projections of model.py:31-33 are one convolution, and so are their data and weight gradients."""

    @staticmethod
    def forward(ctx, qkv, heads):
        qkv = _req(qkv, "qkv")
        N, E3, T = qkv.shape
        E = E3 // 3
        hd = E // heads
        out = torch.empty((N, E, T), device=qkv.device, dtype=torch.float32)
        lse = torch.empty((N, heads, T), device=qkv.device, dtype=torch.float32)
        L.check(L.lib().seld_mha_fwd_packed(L.ptr(qkv), N, T, heads, hd, L.ptr(out), L.ptr(lse), L.current_stream()),
                "seld_mha_fwd_packed")
        ctx.geom = (N, T, heads, hd)
        ctx.save_for_backward(qkv, out, lse)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, out, lse = ctx.saved_tensors
        N, T, H, hd = ctx.geom
        dout = _req(dout, "dout")
        dqkv = torch.empty_like(qkv)
        lib = L.lib()
        lib.seld_mha_bwd_workspace.restype = ctypes.c_size_t
        nbytes = lib.seld_mha_bwd_workspace(N, T, H)
        wsb = torch.empty((nbytes + 3) // 4, device=qkv.device, dtype=torch.float32)
        L.check(lib.seld_mha_bwd_packed(L.ptr(qkv), L.ptr(out), L.ptr(dout), L.ptr(lse), N, T, H, hd, L.ptr(dqkv),
                                        L.ptr(wsb), ctypes.c_size_t(nbytes), L.current_stream()), "seld_mha_bwd_packed")
        return dqkv, None


def mha_packed_ok(T, head_dim):
    return bool(L.lib().seld_mha_packed_ok(int(T), int(head_dim)))


def mha_core_packed(qkv, heads):
    return MhaPackedFn.apply(qkv, heads)


class SeldLossFn(torch.autograd.Function):
    """BCELoss(sed, t_sed) * w_sed + MSELoss(doa, t_doa) * w_doa  (train.py:186-204)."""

    @staticmethod
    def forward(ctx, sed, doa, target, w_sed, w_doa):
        sed2 = _req(sed.reshape(-1, sed.shape[-1]), "sed")
        doa2 = _req(doa.reshape(-1, doa.shape[-1]), "doa")
        tgt = _req(target.reshape(-1, target.shape[-1]), "target")
        rows, n_sed = sed2.shape
        n_doa = doa2.shape[1]
        loss = torch.empty(1, device=sed.device, dtype=torch.float32)       # written, not accumulated (ticketed reduction)
        dsed, ddoa = torch.empty_like(sed2), torch.empty_like(doa2)
        L.check(L.lib().seld_loss_fwd_bwd(L.ptr(sed2), L.ptr(doa2), L.ptr(tgt), ctypes.c_int64(rows), n_sed, n_doa,
                                          ctypes.c_float(w_sed), ctypes.c_float(w_doa), L.ptr(loss), L.ptr(dsed),
                                          L.ptr(ddoa), L.current_stream()), "seld_loss_fwd_bwd")
        ctx.shapes = (tuple(sed.shape), tuple(doa.shape))
        ctx.save_for_backward(dsed, ddoa)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        dsed, ddoa = ctx.saved_tensors
        s1, s2 = ctx.shapes
        if g.data_ptr() == unit_gradient(g.device).data_ptr():       # backward_from_loss(): d loss / d loss = 1
            return dsed.reshape(s1), ddoa.reshape(s2), None, None, None
        return (dsed * g).reshape(s1), (ddoa * g).reshape(s2), None, None, None


_unit_gradients = {}


def unit_gradient(device):
    """The constant 1.0 the training step seeds the backward pass with: one cached tensor per device, so neither the
    autograd engine (ones_like -> fill) nor SeldLossFn.backward (gradient * 1.0) launches a kernel for it."""
    device = torch.device(device)
    t = _unit_gradients.get(device)
    if t is None:
        t = _unit_gradients[device] = torch.ones((), device=device, dtype=torch.float32)
    return t


def backward_from_loss(loss):
    """loss.backward() seeded with the cached unit gradient."""
    loss.backward(unit_gradient(loss.device))


def seld_loss(sed, doa, target, w_sed=1.0, w_doa=5.0):
    return SeldLossFn.apply(sed, doa, target, float(w_sed), float(w_doa))


def adam_flat_step(param, grad, exp_avg, exp_avg_sq, step, lr=1e-4, beta1=0.9, beta2=0.999, eps=1e-8,
                   weight_decay=0.0, grad_scale=1.0):
    L.check(L.lib().seld_adam_flat(L.ptr(param), L.ptr(grad), L.ptr(exp_avg), L.ptr(exp_avg_sq),
                                   ctypes.c_int64(param.numel()), ctypes.c_float(lr), ctypes.c_float(beta1),
                                   ctypes.c_float(beta2), ctypes.c_float(eps), ctypes.c_float(weight_decay), int(step),
                                   ctypes.c_float(grad_scale), L.current_stream()), "seld_adam_flat")


def step_begin(flat_grad, state=None):
    """Zero the flat gradient buffer and (state given) advance the device-resident step state (seld_step_begin)."""
    L.check(L.lib().seld_step_begin(L.ptr(flat_grad), ctypes.c_int64(flat_grad.numel()), L.ptr(state), L.current_stream()),
            "seld_step_begin")


def adam_flat_step_state(param, grad, exp_avg, exp_avg_sq, state, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0,
                         grad_scale=1.0):
    """seld_adam_flat with the step number and learning rate taken from the device-resident step state."""
    L.check(L.lib().seld_adam_flat_state(L.ptr(param), L.ptr(grad), L.ptr(exp_avg), L.ptr(exp_avg_sq),
                                         ctypes.c_int64(param.numel()), ctypes.c_float(beta1), ctypes.c_float(beta2),
                                         ctypes.c_float(eps), ctypes.c_float(weight_decay), ctypes.c_float(grad_scale),
                                         L.ptr(state), L.current_stream()), "seld_adam_flat_state")


def stft_magphase(x, nperseg=512, noverlap=128, output_phase=True):
    """x: (C, L) float32 device tensor -> (C or 2C, nperseg/2, frames) (utility_functions.py:129-155)."""
    x = _req(x, "x")
    C, Ln = x.shape
    frames = L.lib().seld_stft_frames(Ln, nperseg, noverlap)
    if frames <= 0:
        raise L.SeldHipError("seld_stft_frames: invalid segment parameters")
    out = torch.empty(((2 if output_phase else 1) * C, nperseg // 2, frames), device=x.device, dtype=torch.float32)
    L.check(L.lib().seld_stft_magphase(L.ptr(x), C, Ln, nperseg, noverlap, int(bool(output_phase)), L.ptr(out),
                                       L.current_stream()), "seld_stft_magphase")
    return out



def _req_inplace(x, name, min_channels=1):
    if not x.is_cuda:
        raise L.SeldHipError(f"{name}: expected a HIP device tensor (this package has no CPU path)")
    if x.dtype != torch.float32 or not x.is_contiguous() or x.dim() < 2:
        raise L.SeldHipError(f"{name}: expected a contiguous float32 (items, channels, ...) tensor, got {x.dtype} "
                             f"{tuple(x.shape)} contiguous={x.is_contiguous()}")
    if x.shape[1] < min_channels:
        raise L.SeldHipError(f"{name}: needs at least {min_channels} channels, got {x.shape[1]}")
    items, channels = x.shape[0], x.shape[1]
    hw = x.numel() // max(1, items * channels)
    return items, channels, hw


def dq_unit_norm_(x):
    """In place: channels 0..7 of every (item, f, t) position become a unit dual quaternion
    (train.py:257-275).  x: (items, >=8, F, T) float32 on the device."""
    items, channels, hw = _req_inplace(x, "dq_unit_norm_", 8)
    L.check(L.lib().seld_dq_unit_norm(L.ptr(x), ctypes.c_int64(items), channels, ctypes.c_int64(hw), L.current_stream()),
            "seld_dq_unit_norm")
    return x


def group_standardize_(x, c0, c1):
    """In place: x[:, c0:c1] <- (x[:, c0:c1] - mean) / std with one scalar mean / population std over the
    whole group (train.py:345-349).  Returns a 2-element device tensor (mean, std) as applied."""
    items, channels, hw = _req_inplace(x, "group_standardize_")
    c0, c1 = int(c0), min(int(c1), channels)        # a numpy slice clips at the channel count
    work = torch.empty(3, device=x.device, dtype=torch.float64)
    mean_std = torch.empty(2, device=x.device, dtype=torch.float32)
    L.check(L.lib().seld_group_standardize(L.ptr(x), ctypes.c_int64(items), channels, c0, c1, ctypes.c_int64(hw), L.ptr(work),
                                           L.ptr(mean_std), L.current_stream()), "seld_group_standardize")
    return mean_std



METRIC_COUNTERS = ("TP", "FP", "FN", "dc_TP", "dc_FP", "dc_FN", "dc_S", "dc_D", "dc_I", "dc_Nref", "dc_DE_TP", "dc_DE_FP",
                   "dc_DE_FN")


def metrics_new(device):
    """Zeroed accumulators for `metrics_accumulate`: (13 int64 counters, 1 double)."""
    return (torch.zeros(len(METRIC_COUNTERS), device=device, dtype=torch.int64),
            torch.zeros(1, device=device, dtype=torch.float64))


def metrics_accumulate(acc, sed, doa, target, num_frames, num_classes=14, max_overlaps=3, max_loc_value=2.0,
                       spatial_threshold=2.0, doa_threshold=20, frames_per_block=10):
    """Decode + L3DAS21 / DCASE21 counters of a batch of recordings (train.py:100-126), added to `acc`."""
    sed, doa, target = _req(sed, "sed"), _req(doa, "doa"), _req(target, "target")
    n = num_classes * max_overlaps
    if sed.dim() == 2:
        sed, doa, target = sed[None], doa[None], target[None]
    clips, frames = sed.shape[0], sed.shape[1]
    if tuple(sed.shape) != (clips, frames, n) or tuple(doa.shape) != (clips, frames, 3 * n) or \
            tuple(target.shape) != (clips, frames, 4 * n):
        raise L.SeldHipError(f"metrics_accumulate: shapes {tuple(sed.shape)} / {tuple(doa.shape)} / {tuple(target.shape)} do not "
                             f"match (clips, frames, {n}) / (.., {3 * n}) / (.., {4 * n})")
    counters, total_de = acc
    L.check(L.lib().seld_metrics_accumulate(L.ptr(sed), L.ptr(doa), L.ptr(target), clips, frames, int(num_frames),
                                            int(num_classes), int(max_overlaps), ctypes.c_float(max_loc_value),
                                            ctypes.c_double(spatial_threshold), ctypes.c_double(doa_threshold),
                                            int(frames_per_block), L.ptr(counters), L.ptr(total_de), L.current_stream()),
            "seld_metrics_accumulate")
    return acc


_identity_cache = {}


def gate_plain(yf, yg, mask=None):
    """tanh(yf) * sigmoid(yg) * mask for batch_norm='noBN' models: the gate kernel with identity
    normalisation constants (mean 0, invstd 1, gamma 1, beta 0) and eval-mode backward."""
    C = yf.shape[1]
    key = (yf.device, C)
    if key not in _identity_cache:
        _identity_cache[key] = (torch.zeros(C, device=yf.device), torch.ones(C, device=yf.device))
    zero, one = _identity_cache[key]

    return GateFn.apply(yf, yg, one, zero, zero, one, one, zero, zero, one, False, 0.1, 0.0, mask, None, None, None, None)


# ======================================================================================
# fused CNN stage: conv (+ BatchNorm statistics in its epilogue) -> BN -> ReLU -> MaxPool
# ======================================================================================
class HyperConvStatsFn(torch.autograd.Function):
    """y = W (x) x and, from the same kernel's epilogue, the per-channel sum / sum of squares of y
    (SELD_EPI_STATS) that the BatchNorm which follows needs: saves one full read of y."""

    @staticmethod
    def forward(ctx, x, bias, stride, padding, dilation, *ws):
        algebra = len(ws)
        k = tuple(ws[0].shape[2:])
        desc = make_conv_desc(tuple(x.shape), ws[0].shape[0] * algebra, algebra, k, stride, padding, dilation)
        x = _req(x, "x")
        stats = new_stats(desc.Cout, x.device)
        y = conv_fwd(desc, x, ws, bias, epilogue=L.SELD_EPI_STATS, stats=stats)
        ctx.desc = desc
        ctx.has_bias = bias is not None
        ctx.w_params, ctx.bias_param = ws, bias
        ctx.wt_ahead = _transpose_ahead(desc, ws) if ctx.needs_input_grad[0] else None
        ctx.save_for_backward(x)
        ctx.mark_non_differentiable(stats)
        ctx.set_materialize_grads(False)        # no zero-filled "gradient" of the statistics buffer per backward pass
        return y, stats

    @staticmethod
    def backward(ctx, dy, _dstats):
        if dy is None:
            return (None,) * (5 + len(ctx.w_params))
        dx, dbias, dws = _conv_backward(ctx, dy, 5)
        return (dx, dbias, None, None, None, *dws)


def hyper_conv_stats(x, ws, bias, stride, padding, dilation):
    return HyperConvStatsFn.apply(x, bias, stride, padding, dilation, *ws)


class BnReluPoolFn(torch.autograd.Function):
    """[Dropout(drop_p)](MaxPool2d(ph, pw)(ReLU(BatchNorm2d(y)))) in one pass each way (model.py:278-282)."""

    @staticmethod
    def forward(ctx, y, gamma, beta, running_mean, running_var, training, momentum, eps, ph, pw, stats, nbt, drop_p):
        y = _req(y, "y")
        N, C, Hh, Ww = y.shape
        mean, invstd = bn_prepare(y, running_mean, running_var, training, momentum, eps, stats, nbt)
        pooled = torch.empty((N, C, Hh // ph, Ww // pw), device=y.device, dtype=torch.float32)
        idx = torch.empty(pooled.shape, device=y.device, dtype=torch.uint8)
        ctx.rng = None
        out = None
        p_, seed, off, state = 0.0, 0, 0, None
        if drop_p > 0.0:           # same draw as a DropoutFn on `pooled` at this point
            p_ = float(drop_p)
            seed, off, state = philox.draw((pooled.numel() + 3) // 4, y.device)
            out = torch.empty_like(pooled)
            ctx.rng = (p_, seed, off, state)
        L.check(L.lib().seld_bn_relu_pool_fwd_drop(L.ptr(y), N, C, Hh, Ww, ph, pw, L.ptr(mean), L.ptr(invstd), L.ptr(gamma),
                                                   L.ptr(beta), L.ptr(pooled), L.ptr(idx), ctypes.c_float(p_),
                                                   ctypes.c_uint64(seed), ctypes.c_uint64(off), L.ptr(state), L.ptr(out),
                                                   L.current_stream()), "seld_bn_relu_pool_fwd_drop")
        ctx.geom = (N, C, Hh, Ww, ph, pw, training)
        ctx.bn_params = (gamma, beta)
        ctx.save_for_backward(y, pooled, idx, mean, invstd)
        return pooled if out is None else out

    @staticmethod
    def backward(ctx, dpooled):
        y, pooled, idx, mean, invstd = ctx.saved_tensors
        gamma, beta = ctx.bn_params
        N, C, Hh, Ww, ph, pw, training = ctx.geom
        dpooled = _req(dpooled, "dpooled")
        slot, clean = _claim_grad_slots((gamma, beta))
        red = slot if clean else torch.zeros(2 * C, device=y.device, dtype=torch.float32)
        dy = torch.empty_like(y)
        p_, seed, off, state = ctx.rng if ctx.rng is not None else (0.0, 0, 0, None)
        L.check(L.lib().seld_bn_relu_pool_bwd_drop(L.ptr(dpooled), L.ptr(pooled), L.ptr(idx), L.ptr(y), N, C, Hh, Ww, ph, pw,
                                                   L.ptr(mean), L.ptr(invstd), L.ptr(gamma), L.ptr(beta), int(training),
                                                   L.ptr(red), L.ptr(dy), ctypes.c_float(p_), ctypes.c_uint64(seed),
                                                   ctypes.c_uint64(off), L.ptr(state), L.current_stream()),
                "seld_bn_relu_pool_bwd_drop")
        if slot is not None:
            if not clean:
                axpy_(slot, red, 2 * C)     # one add into the flat gradient slice [dgamma | dbeta]
            return (dy,) + (None,) * 12
        return (dy, red[:C], red[C:]) + (None,) * 10


def axpy_(dst_first, src, n):
    """dst[0:n] += src[0:n] where dst_first is the first of several tensors that are adjacent in one flat buffer."""
    L.check(L.lib().seld_accumulate(L.ptr(dst_first), L.ptr(src), ctypes.c_int64(n), L.current_stream()), "seld_accumulate")


def bn_relu_pool(y, bn, ph, pw, stats=None, drop_p=0.0):
    """drop_p > 0 (training): the stage's Dropout rides in the same kernels when the shape allows, else it follows."""
    drop_p = float(drop_p) if bn.training else 0.0
    fuse = drop_p > 0.0 and bool(L.lib().seld_bn_relu_pool_drop_ok(int(y.shape[2]), int(y.shape[3]), int(ph), int(pw)))
    out = BnReluPoolFn.apply(y, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.training,
                             bn.momentum if bn.momentum is not None else 0.1, bn.eps, int(ph), int(pw), stats, _nbt(bn),
                             drop_p if fuse else 0.0)
    return out if fuse or drop_p == 0.0 else dropout(out, drop_p, True)


class ConvBnReluPoolFn(torch.autograd.Function):
    """pooled = MaxPool2d(ph, 1)(ReLU(BatchNorm2d(W (x) x))) for a convolution whose INPUT needs no gradient -- the first
    CNN stage (model.py:269-283 on the network input).  Forward is the same three kernels as hyper_conv_stats +
    bn_relu_pool.  Backward never writes the gradient w.r.t. the conv output (1.6 GB at batch 32): the per-channel
    reductions come from pooled-size tensors (seld_bn_relu_pool_bwd_coef) and the weight-gradient kernel forms
    dy = y*c1 + dz*a + c0 while it stages its operand (seld_hc_conv_bwd_weight_bnpool_acc).  Needs FlatAdam's gradient
    slots (the kernel accumulates); `conv_bn_relu_pool` falls back to the two separate functions otherwise."""

    @staticmethod
    def forward(ctx, x, bias, gamma, beta, running_mean, running_var, training, momentum, eps, ph, nbt, stride, padding,
                dilation, drop_p, *ws):
        algebra = len(ws)
        ctx.rng = None
        k = tuple(ws[0].shape[2:])
        desc = make_conv_desc(tuple(x.shape), ws[0].shape[0] * algebra, algebra, k, stride, padding, dilation)
        x = _req(x, "x")
        # The step's first request for packed weight forms re-packs every registered layer (one launch, ~50 us).  The
        # input's second moments do not need them: when the no-output path is going to be taken they are gathered on the side
        # stream WHILE the main stream packs, and joined before BatchNorm is evaluated from them.
        gws = early = None
        if (ph == 8 and algebra > 1 and training and _side_enabled() and not kernel_timer.active and
                hcq_weights.packed_epoch != hcq_weights.epoch and _first_stage_nostore(desc) and hcq_pack_floats(desc, 2) > 0):
            # (only when a re-pack is pending -- the two-stream model packs before it forks its branches; timed steps keep the
            #  stage's launches on one stream, in one bracket)
            if _side["stream"] is None:
                _side["stream"] = torch.cuda.Stream()
            st = _side["stream"]
            gws = torch.empty(_fs_bytes(desc, "gram"), device=x.device, dtype=torch.uint8)
            fork = torch.cuda.Event()
            fork.record(torch.cuda.current_stream())
            st.wait_event(fork)
            with torch.cuda.stream(st):
                L.check(L.lib().seld_first_stage_gram(ctypes.byref(desc), L.ptr(x), L.ptr(gws), ctypes.c_size_t(gws.numel()),
                                                      L.current_stream()), "seld_first_stage_gram")
                early = torch.cuda.Event()
                early.record(st)
            gws.record_stream(st)
            x.record_stream(st)
        wp = hcq_weights.get(desc, 2, ws) if (ph == 8 and algebra > 1) else None
        if early is not None:
            torch.cuda.current_stream().wait_event(early)
        nostore = wp is not None and training and _first_stage_nostore(desc)
        stats = new_stats(desc.Cout, x.device) if training and not nostore else None
        ctx.gram = None
        if nostore:
            # no convolution output at all (csrc/first_stage.hip): BatchNorm's statistics from the input's second moments,
            # the pooling convolution writes the window value + row only, the backward pass works from those and x
            lib = L.lib()
            o = conv_out_shape(desc)
            N, C, Hh, Ww = _y_shape(desc, o)
            stage_timer = _Timed(desc, 0, label="first_stage_fwd(gram+bn+finishing_pool_conv)" if kernel_timer.active else None)
            stage_timer.__enter__()
            if early is None:
                gws = torch.empty(_fs_bytes(desc, "gram"), device=x.device, dtype=torch.uint8)
                L.check(lib.seld_first_stage_gram(ctypes.byref(desc), L.ptr(x), L.ptr(gws), ctypes.c_size_t(gws.numel()),
                                                  L.current_stream()), "seld_first_stage_gram")
            mean = torch.empty(C, device=x.device, dtype=torch.float32)
            invstd = torch.empty(C, device=x.device, dtype=torch.float32)
            wg = torch.empty((C, 72), device=x.device, dtype=torch.float32)
            L.check(lib.seld_first_stage_bn(ctypes.byref(desc), L.ptr_array8([_req(w, "w") for w in ws]), L.ptr(_req(bias, "bias")),
                                            L.ptr(gws), ctypes.c_float(eps), ctypes.c_float(momentum), L.ptr(mean),
                                            L.ptr(invstd), L.ptr(running_mean), L.ptr(running_var), L.ptr(nbt), L.ptr(wg),
                                            L.current_stream()), "seld_first_stage_bn")
            # raw: written (and read by the backward pass) only for channels with gamma == 0; untouched memory otherwise
            raw = torch.empty((N, C, Hh // ph, Ww), device=x.device, dtype=torch.float32)
            idx = torch.empty(raw.shape, device=x.device, dtype=torch.uint8)
            result = torch.empty_like(raw)
            p_, seed, off, state = 0.0, 0, 0, None
            if drop_p > 0.0:
                p_ = float(drop_p)
                seed, off, state = philox.draw((raw.numel() + 3) // 4, x.device)
                ctx.rng = (p_, seed, off, state)
            L.check(lib.seld_hcq_first_pool_bn(ctypes.byref(desc), L.ptr(x), L.ptr(wp), L.ptr(_req(bias, "bias")), L.ptr(gamma),
                                               L.ptr(beta), L.ptr(mean), L.ptr(invstd), ctypes.c_float(p_), ctypes.c_uint64(seed),
                                               ctypes.c_uint64(off), L.ptr(state), L.ptr(raw), L.ptr(idx), L.ptr(result),
                                               L.current_stream()), "seld_hcq_first_pool_bn")
            stage_timer.__exit__(None, None, None)
            ctx.desc, ctx.geom = desc, (N, C, Hh, Ww, ph, training)
            ctx.params = (ws, bias, gamma, beta)
            ctx.gram = (gws, wg)
            ctx.save_for_backward(x, raw, idx, mean, invstd, result)      # the output's zeros replay ReLU + Dropout backward
            return result
        if wp is not None:
            # the convolution picks every pooling window's element itself (by the sign of gamma): y is written for the
            # backward pass but never read back in the forward pass (csrc/hcq_conv.hip hcq_first_pool_kernel)
            o = conv_out_shape(desc)
            y = torch.empty(_y_shape(desc, o), device=x.device, dtype=torch.float32)
            N, C, Hh, Ww = y.shape
            raw = torch.empty((N, C, Hh // ph, Ww), device=x.device, dtype=torch.float32)
            idx = torch.empty(raw.shape, device=x.device, dtype=torch.uint8)
            with _Timed(desc, 0, label=_hcq_label_cached(desc, 2, 1) if kernel_timer.active else None):
                L.check(L.lib().seld_hcq_first_pool(ctypes.byref(desc), L.ptr(x), L.ptr(wp), L.ptr(_req(bias, "bias")),
                                                    L.ptr(gamma), int(training), L.ptr(y), L.ptr(stats), L.ptr(raw),
                                                    L.ptr(idx), L.current_stream()), "seld_hcq_first_pool")
            mean, invstd = bn_prepare(y, running_mean, running_var, training, momentum, eps, stats, nbt)
            pooled = torch.empty_like(raw)
            out = None
            p_, seed, off, state = 0.0, 0, 0, None
            if drop_p > 0.0:           # the stage's Dropout in the same pass (same mask as a DropoutFn at this point would draw)
                p_ = float(drop_p)
                seed, off, state = philox.draw((pooled.numel() + 3) // 4, x.device)
                out = torch.empty_like(raw)
                ctx.rng = (p_, seed, off, state)
            L.check(L.lib().seld_bn_pool_finish(L.ptr(raw), N, C, (Hh // ph) * Ww, L.ptr(mean), L.ptr(invstd), L.ptr(gamma),
                                                L.ptr(beta), L.ptr(pooled), ctypes.c_float(p_), ctypes.c_uint64(seed),
                                                ctypes.c_uint64(off), L.ptr(state), L.ptr(out), L.current_stream()),
                    "seld_bn_pool_finish")
        else:
            y = conv_fwd(desc, x, ws, bias, epilogue=L.SELD_EPI_STATS if training else 0, stats=stats)
            N, C, Hh, Ww = y.shape
            mean, invstd = bn_prepare(y, running_mean, running_var, training, momentum, eps, stats, nbt)
            pooled = torch.empty((N, C, Hh // ph, Ww), device=y.device, dtype=torch.float32)
            idx = torch.empty(pooled.shape, device=y.device, dtype=torch.uint8)
            L.check(L.lib().seld_bn_relu_pool_fwd(L.ptr(y), N, C, Hh, Ww, ph, 1, L.ptr(mean), L.ptr(invstd), L.ptr(gamma),
                                                  L.ptr(beta), L.ptr(pooled), L.ptr(idx), L.current_stream()),
                    "seld_bn_relu_pool_fwd")
            out = None
            if drop_p > 0.0:
                seed, off, state = philox.draw((pooled.numel() + 3) // 4, x.device)
                out = torch.empty_like(pooled)
                L.check(L.lib().seld_dropout_fwd(L.ptr(pooled), ctypes.c_int64(pooled.numel()), ctypes.c_float(drop_p),
                                                 ctypes.c_uint64(seed), ctypes.c_uint64(off), L.ptr(state), L.ptr(out),
                                                 L.current_stream()), "seld_dropout_fwd")
                ctx.rng = (float(drop_p), seed, off, state)
        ctx.desc, ctx.geom = desc, (N, C, Hh, Ww, ph, training)
        ctx.params = (ws, bias, gamma, beta)
        ctx.save_for_backward(x, y, pooled, idx, mean, invstd)
        return pooled if out is None else out

    @staticmethod
    def backward(ctx, dpooled):
        if ctx.gram is not None:
            return ConvBnReluPoolFn._backward_nostore(ctx, dpooled)
        x, y, pooled, idx, mean, invstd = ctx.saved_tensors
        ws, bias, gamma, beta = ctx.params
        N, C, Hh, Ww, ph, training = ctx.geom
        dpooled = _req(dpooled, "dpooled")
        # ctx.rng: `dpooled` is the gradient BEHIND the stage's Dropout; both consumers replay its mask while they load it
        p_, seed, off, state = ctx.rng if ctx.rng is not None else (0.0, 0, 0, None)
        drop = (ctypes.c_float(p_), ctypes.c_uint64(seed), ctypes.c_uint64(off), L.ptr(state))
        direct = _direct_targets(ws, bias)
        if direct is None:
            raise L.SeldHipError("ConvBnReluPoolFn needs gradient slots (FlatAdam); use hyper_conv_stats + bn_relu_pool")
        slot, clean = _claim_grad_slots((gamma, beta))
        red = slot if clean else torch.zeros(2 * C, device=y.device, dtype=torch.float32)
        coef = torch.empty(3 * C, device=y.device, dtype=torch.float32)
        st = L.current_stream()
        L.check(L.lib().seld_bn_relu_pool_bwd_coef_drop(L.ptr(dpooled), L.ptr(pooled), L.ptr(idx), L.ptr(y), N, C, Hh, Ww, ph,
                                                        1, L.ptr(mean), L.ptr(invstd), L.ptr(gamma), L.ptr(beta),
                                                        int(training), L.ptr(red), L.ptr(coef), L.ptr(direct[1]), *drop, st),
                "seld_bn_relu_pool_bwd_coef_drop")
        with _Timed(ctx.desc, 2, 1, True):
            L.check(L.lib().seld_hc_conv_bwd_weight_bnpool_drop_acc(ctypes.byref(ctx.desc), L.ptr(x), L.ptr(y),
                                                                    L.ptr(pooled), L.ptr(dpooled), L.ptr(idx), ph,
                                                                    L.ptr(coef), L.ptr_array8(direct[0]), *drop, st),
                    "seld_hc_conv_bwd_weight_bnpool_drop_acc")
        dg = db = None
        if slot is None:
            dg, db = red[:C], red[C:]
        elif not clean:
            axpy_(slot, red, 2 * C)
        return (None, None, dg, db) + (None,) * (11 + len(ws))


    @staticmethod
    def _backward_nostore(ctx, dout):
        x, raw, idx, mean, invstd, out = ctx.saved_tensors
        ws, bias, gamma, beta = ctx.params
        gws, wg = ctx.gram
        N, C, Hh, Ww, ph, training = ctx.geom
        dout = _req(dout, "dout")
        p_, seed, off, state = ctx.rng if ctx.rng is not None else (0.0, 0, 0, None)
        direct = _direct_targets(ws, bias)
        if direct is None:
            raise L.SeldHipError("ConvBnReluPoolFn needs gradient slots (FlatAdam); use hyper_conv_stats + bn_relu_pool")
        slot, _ = _claim_grad_slots((gamma, beta))
        red = slot if slot is not None else torch.zeros(2 * C, device=x.device, dtype=torch.float32)
        nbytes = _fs_bytes(ctx.desc, "bwd")
        wsb = torch.empty(nbytes, device=x.device, dtype=torch.uint8)
        with _Timed(ctx.desc, 2, 1, label="fs_wgrad_kernel" if kernel_timer.active else None):
            L.check(L.lib().seld_first_stage_bwd(ctypes.byref(ctx.desc), L.ptr(x), L.ptr(dout), L.ptr(out), L.ptr(raw), L.ptr(idx),
                                                 L.ptr(mean), L.ptr(invstd), L.ptr(gamma), L.ptr(beta), L.ptr(bias), L.ptr(gws),
                                                 L.ptr(wg), L.ptr(red), ctypes.c_void_p(red.data_ptr() + 4 * C),
                                                 L.ptr_array8(direct[0]), ctypes.c_float(p_), L.ptr(wsb),
                                                 ctypes.c_size_t(nbytes), L.current_stream()),
                    "seld_first_stage_bwd")
        dg = db = None
        if slot is None:
            dg, db = red[:C], red[C:]
        return (None, None, dg, db) + (None,) * (11 + len(ws))


_fs_cache = {}


def _fs_bytes(desc, which):
    """Scratch bytes of the no-output first stage (csrc/first_stage.hip) for `desc`: which = 'gram' | 'bwd'; 0 = not taken."""
    key = (bytes(desc), which)
    v = _fs_cache.get(key)
    if v is None:
        lib = L.lib()
        fn = lib.seld_first_stage_gram_workspace if which == "gram" else lib.seld_first_stage_bwd_workspace
        fn.restype = ctypes.c_size_t
        v = _fs_cache[key] = int(fn(ctypes.byref(desc)))
    return v


def _first_stage_nostore(desc):
    """The first stage without its convolution output: 8 real input channels, shapes first_stage.hip takes (forward AND
    backward), SELD_FIRST_STAGE_STORE_Y=1 restores the path that writes y."""
    return (desc.Cin == 8 and not os.environ.get("SELD_FIRST_STAGE_STORE_Y") and _fs_bytes(desc, "gram") > 0 and
            _fs_bytes(desc, "bwd") > 0)


def conv_bn_relu_pool(x, ws, bias, bn, ph, pw, stride, padding, dilation, drop_p=0.0):
    """conv -> BatchNorm2d -> ReLU -> MaxPool2d(ph, pw) [-> Dropout(drop_p), training mode].  The first stage of the
    network (x needs no gradient) takes the fused forms above when the shape qualifies; everything else is
    hyper_conv[_stats] + bn_relu_pool + dropout."""
    drop_p = float(drop_p) if bn.training else 0.0
    k = tuple(ws[0].shape[2:])
    one = lambda v: v == 1 or tuple(v) == (1, 1) if isinstance(v, (tuple, list)) else v == 1
    fused = (not x.requires_grad and torch.is_grad_enabled() and x.dim() == 4 and k == (3, 3) and int(pw) == 1 and
             one(stride) and one(dilation) and x.shape[2] % int(ph) == 0 and x.shape[3] % 32 == 0 and
             _direct_targets(ws, bias) is not None and not os.environ.get("SELD_NO_FUSED_STAGE0"))
    if fused:
        pad = padding if isinstance(padding, (tuple, list)) else (padding, padding)
        fused = tuple(pad) == (1, 1)          # 'same' 3x3: the output has the input's height and width
    if fused and deterministic():
        # only the path without the convolution output is free of atomics (Gram statistics, partials + ordered folds)
        desc_ = make_conv_desc(tuple(x.shape), ws[0].shape[0] * len(ws), len(ws), k, stride, padding, dilation)
        fused = (len(ws) > 1 and int(ph) == 8 and bn.training and _first_stage_nostore(desc_) and
                 hcq_weights.get(desc_, 2, ws) is not None)
    if fused:
        return ConvBnReluPoolFn.apply(x, bias, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.training,
                                      bn.momentum if bn.momentum is not None else 0.1, bn.eps, int(ph), _nbt(bn),
                                      stride, padding, dilation, drop_p, *ws)
    if bn.training:
        y, stats = hyper_conv_stats(x, ws, bias, stride, padding, dilation)
    else:
        y, stats = hyper_conv(x, ws, bias, stride, padding, dilation), None
    return bn_relu_pool(y, bn, ph, pw, stats, drop_p)
