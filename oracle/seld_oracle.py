"""CPU oracle for the DualQ-SELD-TCN hot path.  TEST INFRASTRUCTURE ONLY.

This file is a torch-CPU (fp32 or fp64) restatement of the reference algorithms
named in SURVEY.md section 8(a).  It is *not* part of the product: only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it.
The product path (the HIP library behind include/seld_hip.h) never calls it.

Parity status: PINNED.  `tests/golden/make_golden.py` imports the reference
(/root/reference, this container only) and stores its outputs; `tests/test_oracle.py`
checks every function here against those fixtures.

Two independent restatements of every hypercomplex product are provided:

* ``mode='assembled'`` follows the reference literally: build the real block
  matrix of the Hamilton product and run ONE real convolution / matmul
  (quaternion_ops.py:125-147, dual_quaternion_ops.py:111-153).  This is the
  variant timed as the CPU baseline.
* ``mode='explicit'`` evaluates the algebra of SURVEY App. A component by
  component (16 / 48 small real convolutions).  It shares no code with the
  assembled variant and is what the HIP kernels are derived from.

All functions are differentiable (torch autograd), so gradient parity of the HIP
backward kernels is checked against autograd through this file.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------
# Hamilton tables
# --------------------------------------------------------------------------------------
# QUAT_TABLE[p][q] = (component index, sign): block (p, q) of the real matrix of the
# left Hamilton product y = W (x) x, rows = output component, cols = input component.
# quaternion_ops.py:131-134 (the four torch.cat rows).
R_, I_, J_, K_ = 0, 1, 2, 3
QUAT_TABLE = (
    ((R_, +1), (I_, -1), (J_, -1), (K_, -1)),
    ((I_, +1), (R_, +1), (K_, -1), (J_, +1)),
    ((J_, +1), (K_, +1), (R_, +1), (I_, -1)),
    ((K_, +1), (J_, -1), (I_, +1), (R_, +1)),
)


def block_table(algebra: int):
    """(comp, sign) or None for every (p, q) block.  algebra 1 = real, 4 = quaternion,
    8 = dual quaternion [[Q, 0], [Q2, Q]] (dual_quaternion_ops.py:134-140)."""
    if algebra == 1:
        return (((0, +1),),)
    if algebra == 4:
        return QUAT_TABLE
    if algebra == 8:
        rows = []
        for p in range(8):
            row = []
            for q in range(8):
                pp, qq = p % 4, q % 4
                comp, sign = QUAT_TABLE[pp][qq]
                if p < 4 and q < 4:
                    row.append((comp, sign))          # Q
                elif p < 4 and q >= 4:
                    row.append(None)                  # structural zero
                elif p >= 4 and q < 4:
                    row.append((comp + 4, sign))      # Q2 (the *_weight_2 tensors)
                else:
                    row.append((comp, sign))          # Q again
            rows.append(tuple(row))
        return tuple(rows)
    raise ValueError("algebra must be 1, 4 or 8")


def assemble_conv_weight(ws: Sequence[torch.Tensor]) -> torch.Tensor:
    """Real block matrix (Cout, Cin, *k) from the component tensors (Cout/A, Cin/A, *k)."""
    A = len(ws)
    table = block_table(A)
    rows = []
    for p in range(A):
        blocks = []
        for q in range(A):
            e = table[p][q]
            if e is None:
                blocks.append(torch.zeros_like(ws[0]))
            else:
                blocks.append(ws[e[0]] if e[1] > 0 else -ws[e[0]])
        rows.append(torch.cat(blocks, dim=1))
    return torch.cat(rows, dim=0)


def _convnd(x, w, bias, stride, padding, dilation, groups):
    if x.dim() == 3:
        return F.conv1d(x, w, bias, stride, padding, dilation, groups)
    if x.dim() == 4:
        return F.conv2d(x, w, bias, stride, padding, dilation, groups)
    if x.dim() == 5:
        return F.conv3d(x, w, bias, stride, padding, dilation, groups)
    raise Exception("The convolutional input is either 3, 4 or 5 dimensions. input.dim = " + str(x.dim()))


def hypercomplex_conv(x, ws, bias=None, stride=1, padding=0, groups=1, dilatation=1, mode="assembled"):
    """quaternion_conv (quaternion_ops.py:125-147) when len(ws)==4, dual_quaternion_conv
    (dual_quaternion_ops.py:111-153) when len(ws)==8, plain conv when len(ws)==1."""
    A = len(ws)
    if mode == "assembled":
        return _convnd(x, assemble_conv_weight(ws), bias, stride, padding, dilatation, groups)
    # explicit algebra: y_p = sum_q sign * (W_comp * x_q)
    table = block_table(A)
    ci = x.shape[1] // A
    xs = [x[:, q * ci:(q + 1) * ci] for q in range(A)]
    outs = []
    for p in range(A):
        acc = None
        for q in range(A):
            e = table[p][q]
            if e is None:
                continue
            t = _convnd(xs[q], ws[e[0]], None, stride, padding, dilatation, groups)
            t = t if e[1] > 0 else -t
            acc = t if acc is None else acc + t
        outs.append(acc)
    y = torch.cat(outs, dim=1)
    if bias is not None:
        y = y + bias.view(1, -1, *([1] * (x.dim() - 2)))
    return y


def quaternion_conv(x, r, i, j, k, bias, stride, padding, groups, dilatation, mode="assembled"):
    return hypercomplex_conv(x, (r, i, j, k), bias, stride, padding, groups, dilatation, mode)


def dual_quaternion_conv(x, r, i, j, k, r2, i2, j2, k2, bias, stride, padding, groups, dilatation,
                         mode="assembled"):
    return hypercomplex_conv(x, (r, i, j, k, r2, i2, j2, k2), bias, stride, padding, groups, dilatation, mode)


# --------------------------------------------------------------------------------------
# Linears
# --------------------------------------------------------------------------------------
def assemble_quaternion_linear_weight(r, i, j, k):
    """(4*in, 4*out) matrix of quaternion_ops.py:310-314: cat on dim 0 then dim 1, i.e.
    block (q_in, p_out) = QUAT_TABLE[p][q]."""
    ws = (r, i, j, k)
    cols = []
    for p in range(4):
        blocks = []
        for q in range(4):
            comp, sign = QUAT_TABLE[p][q]
            blocks.append(ws[comp] if sign > 0 else -ws[comp])
        cols.append(torch.cat(blocks, dim=0))
    return torch.cat(cols, dim=1)


def quaternion_linear(x, r, i, j, k, bias=None, mode="assembled"):
    """quaternion_ops.py:299-327 / QuaternionLinearFunction.forward :395-414."""
    if mode == "assembled":
        y = x @ assemble_quaternion_linear_weight(r, i, j, k)
    else:
        ws = (r, i, j, k)
        n = r.shape[0]
        xs = [x[..., q * n:(q + 1) * n] for q in range(4)]
        outs = []
        for p in range(4):
            acc = 0
            for q in range(4):
                comp, sign = QUAT_TABLE[p][q]
                acc = acc + sign * (xs[q] @ ws[comp])
            outs.append(acc)
        y = torch.cat(outs, dim=-1)
    return y if bias is None else y + bias


def assemble_dual_quaternion_linear_weight(ws):
    """(8*in, 8*out) matrix of dual_quaternion_ops.py:170-188.  NOTE the reference cats the
    rows of the 4x4 table on dim=1 and stacks them on dim=0 for (in, out) shaped tensors, so
    the matrix that multiplies from the right is the *conv-style arrangement applied to
    (in,out) blocks*: block (row-block a, col-block b) = table[a][b] and y = x @ M.
    Hence y_b = sum_a x_a @ M[a][b] (conjugate/transposed structure)."""
    table = block_table(8)
    rows = []
    for a in range(8):
        blocks = []
        for b in range(8):
            e = table[a][b]
            if e is None:
                blocks.append(torch.zeros_like(ws[0]))
            else:
                blocks.append(ws[e[0]] if e[1] > 0 else -ws[e[0]])
        rows.append(torch.cat(blocks, dim=1))
    return torch.cat(rows, dim=0)


def dual_quaternion_linear(x, ws, bias=None, mode="assembled"):
    """dual_quaternion_ops.py:156-203; 3-D inputs are flattened by the layer
    (dual_quaternion_layers.py:183-189) which is a no-op for matmul semantics."""
    if mode == "assembled":
        y = x @ assemble_dual_quaternion_linear_weight(ws)
    else:
        table = block_table(8)
        n = ws[0].shape[0]
        xs = [x[..., a * n:(a + 1) * n] for a in range(8)]
        outs = []
        for b in range(8):
            acc = 0
            for a in range(8):
                e = table[a][b]
                if e is None:
                    continue
                acc = acc + e[1] * (xs[a] @ ws[e[0]])
            outs.append(acc)
        y = torch.cat(outs, dim=-1)
    return y if bias is None else y + bias


# --------------------------------------------------------------------------------------
# BatchNorm / MHA / blocks
# --------------------------------------------------------------------------------------
def batch_norm(x, weight, bias, running_mean, running_var, train: bool, eps=1e-5, momentum=0.1,
               stats_out: Optional[dict] = None, name: str = ""):
    """torch.nn.BatchNorm{1,2}d semantics (model.py:88-92, :279): biased variance for the
    normalisation, unbiased for the running estimate."""
    dims = [0] + list(range(2, x.dim()))
    shape = [1, -1] + [1] * (x.dim() - 2)
    if train:
        mean = x.mean(dim=dims)
        var = x.var(dim=dims, unbiased=False)
        if stats_out is not None:
            n = x.numel() // x.shape[1]
            stats_out[name + ".running_mean"] = (1 - momentum) * running_mean + momentum * mean.detach()
            stats_out[name + ".running_var"] = (1 - momentum) * running_var + momentum * var.detach() * n / max(n - 1, 1)
    else:
        mean, var = running_mean, running_var
    return (x - mean.view(shape)) / torch.sqrt(var.view(shape) + eps) * weight.view(shape) + bias.view(shape)


def multi_head_attention(x, wq, wk, wv, wo, bo, num_heads=8):
    """model.py:12-51 with v=k=q=x of shape (N, T, E); 1x1 conv weights (E,E,1), no bias;
    heads split channel e = h*hd + d; softmax(q k^T / sqrt(hd)); fc_out with bias."""
    N, T, E = x.shape
    hd = E // num_heads
    q = (x @ wq.view(E, E).t()).view(N, T, num_heads, hd)
    k = (x @ wk.view(E, E).t()).view(N, T, num_heads, hd)
    v = (x @ wv.view(E, E).t()).view(N, T, num_heads, hd)
    energy = torch.einsum("nqhd,nkhd->nhqk", q, k)
    att = torch.softmax(energy / (hd ** 0.5), dim=3)
    out = torch.einsum("nhql,nlhd->nqhd", att, v).reshape(N, T, E)
    return out @ wo.t() + bo


@dataclass
class SeldConfig:
    """Mirror of SELD_Model.__init__ (model.py:325-332)."""
    time_dim: int = 512
    freq_dim: int = 256
    input_channels: int = 4
    output_classes: int = 14
    domain: str = "DQ"
    domain_classifier: str = "same"
    cnn_filters: List[int] = field(default_factory=lambda: [64, 64, 64])
    kernel_size_cnn_blocks: int = 3
    pool_size: List[List[int]] = field(default_factory=lambda: [[8, 2], [8, 2], [2, 2]])
    pool_time: str = "TCN"
    D: List = field(default_factory=lambda: [10])
    dilation_mode: str = "fibonacci"
    G: int = 128
    U: int = 128
    kernel_size_dilated_conv: int = 3
    spatial_dropout_rate: float = 0.5
    V: List[int] = field(default_factory=lambda: [128, 128])
    V_kernel_size: int = 3
    fc_layers: List[int] = field(default_factory=lambda: [128])
    fc_activations: str = "Linear"
    fc_dropout: str = "all"
    dropout_perc: float = 0.3
    class_overlaps: float = 3.0
    use_bias_conv: bool = False
    use_bias_linear: bool = True
    batch_norm: str = "BN"
    parallel_ConvTC_block: str = "False"
    parallel_magphase: bool = False

    @property
    def algebra(self) -> int:
        return {"Q": 4, "DQ": 8}.get(self.domain, 1)

    @property
    def classifier(self) -> str:
        return self.domain if self.domain_classifier == "same" else self.domain_classifier

    @property
    def two_stream(self) -> bool:
        return self.parallel_ConvTC_block in {"2Parallel", "2BParallel", "2ParallelBranches", "2PB"}


def dilations(cfg: SeldConfig) -> List[int]:
    """model.py:146-174."""
    out = []
    for n_resblock in cfg.D:
        dilation, prec_1, prec_2 = 1, 1, 0
        if isinstance(n_resblock, list):
            out.extend(n_resblock)
            continue
        for d in range(n_resblock):
            if cfg.dilation_mode == "fibonacci":
                if d == 0:
                    dilation = 1
                else:
                    dilation = prec_1 + prec_2
                    prec_2, prec_1 = prec_1, dilation
            else:
                dilation = 2 ** d
            out.append(dilation)
    return out


_WNAMES4 = ("r_weight", "i_weight", "j_weight", "k_weight")
_WNAMES8 = _WNAMES4 + ("r_weight_2", "i_weight_2", "j_weight_2", "k_weight_2")


def _conv_layer(sd, prefix, algebra, x, stride, padding, dilation, mode):
    bias = sd.get(prefix + ".bias")
    if algebra == 1:
        ws = (sd[prefix + ".weight"],)
    else:
        ws = tuple(sd[prefix + "." + n] for n in (_WNAMES4 if algebra == 4 else _WNAMES8))
    return hypercomplex_conv(x, ws, bias, stride, padding, 1, dilation, mode)


def _bn(sd, prefix, x, train, stats_out):
    return batch_norm(x, sd[prefix + ".weight"], sd[prefix + ".bias"], sd[prefix + ".running_mean"],
                      sd[prefix + ".running_var"], train, stats_out=stats_out, name=prefix)


def res_block(sd, prefix, cfg: SeldConfig, x, dilation, train, mode, stats_out=None, dropout=False):
    """ResBlock.forward, model.py:109-132.  `dropout` (CPU-baseline timing only: RNG parity with the GPU is
    impossible, so every parity check runs with it off) enables the Dropout1d of :127-128."""
    use_bn = cfg.batch_norm in {"BN", "BN_on_TCN", "BNonTCN"}
    k = cfg.kernel_size_dilated_conv
    pad = int(((k - 1) * dilation) / 2)
    if use_bn:
        x = torch.tanh(_bn(sd, prefix + ".batch_filter1", x, train, stats_out))
    yf = _conv_layer(sd, prefix + ".conv1_filter", cfg.algebra, x, 1, pad, dilation, mode)
    yg = _conv_layer(sd, prefix + ".conv1_gate", cfg.algebra, x, 1, pad, dilation, mode)
    if use_bn:
        yf = _bn(sd, prefix + ".batch_filter2", yf, train, stats_out)
        yg = _bn(sd, prefix + ".batch_gate2", yg, train, stats_out)
    y = torch.tanh(yf) * torch.sigmoid(yg)
    if dropout and cfg.spatial_dropout_rate:
        y = F.dropout1d(y, cfg.spatial_dropout_rate, training=True)
    skip = _conv_layer(sd, prefix + ".conv2_skip", cfg.algebra, y, 1, 0, 1, mode)
    res = _conv_layer(sd, prefix + ".conv2_residual", cfg.algebra, y, 1, 0, 1, mode)
    return x + res, skip


def tc_block(sd, prefix, cfg: SeldConfig, x, train, mode, taps=None, stats_out=None, dropout=False):
    """TC_Block.forward, model.py:204-232."""
    skip_sum = None
    for bi, d in enumerate(dilations(cfg)):
        x, skip = res_block(sd, f"{prefix}.ResBlocks.{bi}", cfg, x, d, train, mode, stats_out, dropout)
        skip_sum = skip if skip_sum is None else skip_sum + skip
        if taps is not None:
            taps[f"{prefix}.ResBlocks.{bi}.residual"] = x
            taps[f"{prefix}.ResBlocks.{bi}.skip"] = skip
    out = torch.relu(skip_sum)
    tcn_pool = cfg.pool_time == "TCN"
    if tcn_pool:
        out = F.max_pool1d(out, cfg.pool_size[0][1])
    out = _conv_layer(sd, prefix + ".conv1", cfg.algebra, out, 1, 1, 1, mode)
    if taps is not None:
        taps[prefix + ".conv1"] = out
    a = prefix + ".attention"
    out = multi_head_attention(out.permute(0, 2, 1), sd[a + ".queries.weight"], sd[a + ".keys.weight"],
                               sd[a + ".values.weight"], sd[a + ".fc_out.weight"], sd[a + ".fc_out.bias"],
                               num_heads=8).permute(0, 2, 1)
    if taps is not None:
        taps[prefix + ".attention"] = out
    out = torch.relu(out)
    if tcn_pool:
        out = F.max_pool1d(out, cfg.pool_size[1][1])
    out = _conv_layer(sd, prefix + ".conv2", cfg.algebra, out, 1, 1, 1, mode)
    out = torch.tanh(out)
    if tcn_pool:
        out = F.max_pool1d(out, cfg.pool_size[2][1])
    return out


def conv_tc_block(sd, prefix, cfg: SeldConfig, x, train, mode, taps=None, stats_out=None, dropout=False):
    """ConvTC_Block.forward, model.py:297-322."""
    use_bn = cfg.batch_norm in {"BN", "BN_on_CNN", "BNonCNN"}
    for i, p in enumerate(cfg.pool_size[:len(cfg.cnn_filters)]):
        x = _conv_layer(sd, f"{prefix}.cnn.{i}.0", cfg.algebra, x, 1, 1, 1, mode)
        if use_bn:
            x = _bn(sd, f"{prefix}.cnn.{i}.1", x, train, stats_out)
        x = torch.relu(x)
        pool = [p[0], p[1]] if cfg.pool_time == "CNN" else [p[0], 1]
        x = F.max_pool2d(x, pool)
        if dropout and cfg.dropout_perc:
            x = F.dropout(x, cfg.dropout_perc, training=True)
        if taps is not None:
            taps[f"{prefix}.cnn.{i}"] = x
    B = x.shape[0]
    x = x.permute(0, 3, 1, 2).reshape(B, x.shape[3], -1).permute(0, 2, 1)
    x = tc_block(sd, prefix + ".tcn", cfg, x, train, mode, taps, stats_out, dropout)
    return x.permute(0, 2, 1)


def _head(sd, name, cfg: SeldConfig, x, mode, dropout=False):
    """model.py:430-459."""
    idx = 0
    for _ in cfg.fc_layers:
        p = f"{name}.{idx}"
        if cfg.classifier == "Q":
            x = quaternion_linear(x, *(sd[p + "." + n] for n in _WNAMES4), bias=sd.get(p + ".bias"), mode=mode)
        elif cfg.classifier == "DQ":
            x = dual_quaternion_linear(x, tuple(sd[p + "." + n] for n in _WNAMES8), sd.get(p + ".bias"), mode)
        else:
            x = F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))
        idx += 1
        if cfg.fc_activations in {"relu", "ReLU", "RELU"}:
            x = torch.relu(x)
            idx += 1
        if cfg.fc_dropout in {"all", "ALL", "True"}:
            if dropout and cfg.dropout_perc:
                x = F.dropout(x, cfg.dropout_perc, training=True)
            idx += 1
    if cfg.fc_dropout in {"last", "Last", "LAST"}:
        if dropout and cfg.dropout_perc:
            x = F.dropout(x, cfg.dropout_perc, training=True)
        idx += 1
    p = f"{name}.{idx}"
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def seld_forward(sd: Dict[str, torch.Tensor], cfg: SeldConfig, x, train=False, mode="assembled",
                 taps: Optional[dict] = None, stats_out: Optional[dict] = None, dropout: bool = False):
    """SELD_Model.forward, model.py:461-480.  `sd` uses the reference's state-dict key names
    (SURVEY App. B).  `train=True` selects batch statistics in BatchNorm; dropout is always off."""
    if cfg.two_stream:
        if cfg.parallel_magphase:
            xa = torch.cat((x[:, :4], x[:, 8:12]), 1)
            xb = torch.cat((x[:, 4:8], x[:, 12:]), 1)
        else:
            h = cfg.input_channels // 2
            xa, xb = x[:, :h], x[:, h:]
        a = conv_tc_block(sd, "branch_A", cfg, xa, train, mode, taps, stats_out, dropout)
        b = conv_tc_block(sd, "branch_B", cfg, xb, train, mode, taps, stats_out, dropout)
        feat = torch.cat((a, b), 2)
    else:
        feat = conv_tc_block(sd, "seld_block", cfg, x, train, mode, taps, stats_out, dropout)
    sed_logits = _head(sd, "sed", cfg, feat, mode, dropout)
    doa_logits = _head(sd, "doa", cfg, feat, mode, dropout)
    if taps is not None:
        taps["feat"] = feat
        taps["sed_logits"] = sed_logits
        taps["doa_logits"] = doa_logits
    return torch.sigmoid(sed_logits), torch.tanh(doa_logits)


def seld_loss(sed, doa, target, n_sed, sed_weight=1.0, doa_weight=5.0):
    """train.py:186-204 with nn.BCELoss / nn.MSELoss (mean reductions, :498-499)."""
    t_sed = torch.flatten(target[:, :, :n_sed], 1)
    t_doa = torch.flatten(target[:, :, n_sed:], 1)
    sed = torch.flatten(sed, 1)
    doa = torch.flatten(doa, 1)
    bce = F.binary_cross_entropy(sed, t_sed)
    mse = F.mse_loss(doa, t_doa)
    return bce * sed_weight + mse * doa_weight


def adam_step(p, g, m, v, step, lr=1e-4, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam defaults (train.py:502): returns (p, m, v) after one update; step is 1-based."""
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    mhat = m / (1 - b1 ** step)
    vhat = v / (1 - b2 ** step)
    return p - lr * mhat / (vhat.sqrt() + eps), m, v


# --------------------------------------------------------------------------------------
# STFT magnitude / phase  (utility_functions.py:129-155, scipy.signal.stft defaults)
# --------------------------------------------------------------------------------------
def spectrum_fast(x: np.ndarray, nperseg=512, noverlap=128, cut_dc=True, output_phase=True,
                  cut_last_timeframe=True) -> np.ndarray:
    """Closed-form restatement: periodic Hamming window, zero boundary
    extension of nperseg/2 on both sides, zero tail padding to a whole number of hops,
    rfft of the windowed frames divided by sum(window)."""
    x = np.asarray(x, dtype=np.float64)
    N = nperseg
    hop = N - noverlap
    n = np.arange(N)
    w = 0.54 - 0.46 * np.cos(2 * np.pi * n / N)
    pad = N // 2
    xp = np.concatenate([np.zeros(x.shape[:-1] + (pad,)), x, np.zeros(x.shape[:-1] + (pad,))], axis=-1)
    L = xp.shape[-1]
    nadd = (-(L - N) % hop) % N
    xp = np.concatenate([xp, np.zeros(x.shape[:-1] + (nadd,))], axis=-1)
    L = xp.shape[-1]
    nframes = (L - N) // hop + 1
    idx = np.arange(nframes)[:, None] * hop + n[None, :]
    frames = xp[..., idx] * w                      # (..., frames, N)
    Z = np.fft.rfft(frames, axis=-1) / w.sum()     # (..., frames, N/2+1)
    Z = np.swapaxes(Z, -1, -2)                     # (..., freq, frames)
    out = np.abs(Z)
    if output_phase:
        out = np.concatenate((out, np.angle(Z)), axis=-3)
    if cut_dc:
        out = out[:, 1:, :]
    if cut_last_timeframe:
        out = out[:, :, :-1]
    return out


# --------------------------------------------------------------------------------------
# Dataset normalisation  (train.py:242-408)
# --------------------------------------------------------------------------------------
def dq_unit_norm(x: torch.Tensor) -> torch.Tensor:
    """train.py:257-275: channels 0..7 = (q, p) -> (q/|q|, p - (q.p/|q|^2) q), in the array's own dtype, the
    reference's operation order.  Returns a new tensor; further channels are copied."""
    x = x.clone()
    q = [x[:, i:i + 1] for i in range(4)]
    p = [x[:, i:i + 1] for i in range(4, 8)]
    den0 = q[0] ** 2 + q[1] ** 2 + q[2] ** 2 + q[3] ** 2
    den1 = torch.sqrt(den0)
    cross = q[0] * p[0] + q[1] * p[1] + q[2] * p[2] + q[3] * p[3]
    pn = [p[i] - cross / den0 * q[i] for i in range(4)]
    qn = [q[i] / den1 for i in range(4)]
    x[:, :8] = torch.cat(qn + pn, dim=1)
    return x


def dq_unit_norm_ieee(x: np.ndarray) -> np.ndarray:
    """The same expression tree evaluated by numpy, every operation correctly rounded (IEEE 754).  torch's CPU sqrt
    on this build (AVX512 / MKL VML) is NOT correctly rounded - about 0.6 % of float32 values come out 1 ulp off -
    so `dq_unit_norm` (= the reference) and this differ by at most 1 ulp in the q channels; p is identical."""
    x = np.array(x)
    q = [x[:, i] for i in range(4)]
    p = [x[:, i] for i in range(4, 8)]
    with np.errstate(all="ignore"):
        den0 = ((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) + q[3] * q[3]
        den1 = np.sqrt(den0)
        cross = ((q[0] * p[0] + q[1] * p[1]) + q[2] * p[2]) + q[3] * p[3]
        ratio = cross / den0
        out = [q[i] / den1 for i in range(4)] + [p[i] - ratio * q[i] for i in range(4)]
    x[:, :8] = np.stack(out, 1)
    return x


def group_standardize(x: np.ndarray, c0: int, c1: int) -> np.ndarray:
    """train.py:345-349: x[:, c0:c1] -= mean; /= std (numpy, the array's own dtype, population std)."""
    x = np.array(x)
    m = np.mean(x[:, c0:c1, :, :])
    s = np.std(x[:, c0:c1, :, :])
    x[:, c0:c1, :, :] -= m
    x[:, c0:c1, :, :] /= s
    return x


def normalize_dataset(x: np.ndarray, dataset_normalization: str, n_mics: int, domain: str, phase: bool) -> np.ndarray:
    """The branch structure of train.py:242-408 for ONE predictor array (the reference repeats it for the
    training / validation / test arrays); returns float32 as train.py:425 does."""
    mode = str(dataset_normalization)
    if mode not in {'False', 'false', 'None', 'none'}:
        if mode in {'DQ_Normalization', 'UnitNormNormalization', 'UnitNorm'}:
            if n_mics == 2 and domain in ['DQ', 'dq', 'dQ', 'Dual_Quaternion', 'dual_quaternion']:
                if phase:
                    raise ValueError('DATASET NORMALIZATION FOR PHASE DUAL QUATERNION NOT YET IMPLEMENTED')
                x = dq_unit_norm(torch.tensor(x)).numpy()
        elif n_mics in (1, 2):
            mag = 4 * n_mics
            x = group_standardize(x, 0, mag)
            if phase:
                x = group_standardize(x, mag, x.shape[1])
    return torch.tensor(np.array(x)).float().numpy()


# --------------------------------------------------------------------------------------
# Post-processing + test metrics  (train.py:84-166, utility_functions.py:184-210,
# metrics.py:123-208, Dcase21_metrics.py:33-154, 171-221, 239-278)
# --------------------------------------------------------------------------------------
METRIC_COUNTERS = ("TP", "FP", "FN", "dc_TP", "dc_FP", "dc_FN", "dc_S", "dc_D", "dc_I", "dc_Nref", "dc_DE_TP", "dc_DE_FP",
                   "dc_DE_FN")


def decode_events(sed: np.ndarray, doa: np.ndarray, max_loc_value=2.0, num_classes=14, max_overlaps=3):
    """utility_functions.py:184-210 as arrays: active[f, class, event] (np.round != 0, frames whose rounded
    activities sum to 0 dropped) and the float64 coordinates xyz[f, class, event, 3] = float32(doa * max_loc_value)."""
    sed = np.asarray(sed)
    r = np.round(sed)                                           # half to even: 0.5 -> 0
    active = (r != 0) & (np.sum(r, axis=1, keepdims=True) != 0)
    xyz = (np.asarray(doa) * max_loc_value).reshape(sed.shape[0], num_classes, max_overlaps, 3)
    return active.reshape(sed.shape[0], num_classes, max_overlaps), xyz.astype(np.float64)


def lsd_counts(act_p, xyz_p, act_t, xyz_t, n_frames, spatial_threshold=2.0):
    """metrics.py:123-182, literally including its double count: a frame with predictions but no reference adds its
    predictions to FP twice (once in the `len(t) == 0` branch, once in the unconditional tail), a frame with
    references but no prediction adds them to FN twice; otherwise a reference event is matched when ANY prediction of
    its class lies closer than the threshold, FN += n_t - matched, FP += n_p - matched (which can be negative)."""
    if act_p.shape[0] > n_frames or act_t.shape[0] > n_frames:
        raise KeyError("event beyond n_frames")                 # frames[i[0]] in the reference
    TP = FP = FN = 0
    for f in range(act_p.shape[0]):
        n_p, n_t = int(act_p[f].sum()), int(act_t[f].sum())
        if n_t == 0:
            FP += 2 * n_p
            continue
        if n_p == 0:
            FN += 2 * n_t
            continue
        matched = 0
        for c, e in zip(*np.nonzero(act_t[f])):
            cand = np.nonzero(act_p[f, c])[0]
            if any(np.linalg.norm(xyz_t[f, c, e] - xyz_p[f, c, k]) < spatial_threshold for k in cand):
                matched += 1
        TP += matched
        FN += n_t - matched
        FP += n_p - matched
    return TP, FP, FN


def _angular_distance_deg(a, b):
    """Dcase21_metrics.py:171-188 for one pair of Cartesian vectors (float64)."""
    n1 = np.sqrt(a[0] ** 2 + a[1] ** 2 + a[2] ** 2 + 1e-10)
    n2 = np.sqrt(b[0] ** 2 + b[1] ** 2 + b[2] ** 2 + 1e-10)
    d = (a[0] / n1) * (b[0] / n2) + (a[1] / n1) * (b[1] / n2) + (a[2] / n1) * (b[2] / n2)
    return np.arccos(np.clip(d, -1, 1)) * 180 / np.pi


def dcase_counts(act_p, xyz_p, act_t, xyz_t, max_frames, doa_threshold=20, frames_per_block=10):
    """segment_labels (Dcase21_metrics.py:239-278) + SELDMetrics.update_seld_scores (:51-154) for one recording,
    on the arrays of `decode_events`.  Returns the increments of the class's counters and of `_total_DE`."""
    from scipy.optimize import linear_sum_assignment
    n_cls = act_p.shape[1]
    c = dict(TP=0, FP=0, FN=0, S=0, D=0, I=0, Nref=0, DE_TP=0, DE_FP=0, DE_FN=0)
    total_de = 0.0
    for start in range(0, max_frames, frames_per_block):
        frames = [f for f in range(start, start + frames_per_block) if f < act_p.shape[0]]
        loc_fn = loc_fp = 0
        for k in range(n_cls):
            g_cnt = [int(act_t[f, k].sum()) for f in frames]
            p_cnt = [int(act_p[f, k].sum()) for f in frames]
            nb_gt, nb_pred = max(g_cnt, default=0), max(p_cnt, default=0)
            c["Nref"] += nb_gt
            if nb_gt and nb_pred:
                track_sum, track_n = {}, {}
                for f, g, p in zip(frames, g_cnt, p_cnt):
                    if not (g and p):
                        continue
                    gt = xyz_t[f, k][act_t[f, k]]
                    pr = xyz_p[f, k][act_p[f, k]]
                    cost = np.array([[_angular_distance_deg(a, b) for b in pr] for a in gt])
                    rows, cols = linear_sum_assignment(cost)
                    for r, col in zip(rows, cols):
                        track_sum[r] = track_sum.get(r, 0) + cost[r, col]
                        track_n[r] = track_n.get(r, 0) + 1
                if not track_sum:
                    loc_fn += nb_pred
                    c["FN"] += nb_pred
                    c["DE_FN"] += nb_pred
                else:
                    for r in track_sum:
                        avg = track_sum[r] / track_n[r]
                        total_de += avg
                        c["DE_TP"] += 1
                        if avg <= doa_threshold:
                            c["TP"] += 1
                        else:
                            loc_fp += 1
                            c["FP"] += 1
                    if nb_pred > nb_gt:
                        loc_fp += nb_pred - nb_gt
                        c["FP"] += nb_pred - nb_gt
                        c["DE_FP"] += nb_pred - nb_gt
                    elif nb_pred < nb_gt:
                        loc_fn += nb_gt - nb_pred
                        c["FN"] += nb_gt - nb_pred
                        c["DE_FN"] += nb_gt - nb_pred
            elif nb_gt:
                loc_fn += nb_gt
                c["FN"] += nb_gt
                c["DE_FN"] += nb_gt
            elif nb_pred:
                loc_fp += nb_pred
                c["FP"] += nb_pred
                c["DE_FP"] += nb_pred
        c["S"] += min(loc_fp, loc_fn)
        c["D"] += max(0, loc_fn - loc_fp)
        c["I"] += max(0, loc_fp - loc_fn)
    return c, total_de


def test_results(counts: dict, total_de: float, epoch=0):
    """The 16 numbers evaluate_test returns (train.py:131-150) from the accumulated counters
    (`compute_seld_scores`, Dcase21_metrics.py:33-49, inlined)."""
    import sys
    eps_f, eps = sys.float_info.epsilon, np.finfo(float).eps
    TP, FP, FN = counts["TP"], counts["FP"], counts["FN"]
    precision = TP / (TP + FP + eps_f)
    recall = TP / (TP + FN + eps_f)
    F_score = 2 * ((precision * recall) / (precision + recall + eps_f))
    Nref, Nsys = TP + FN, TP + FP
    ER_score = (max(Nref, Nsys) - TP) / (Nref + 0.0)
    ER = (counts["dc_S"] + counts["dc_D"] + counts["dc_I"]) / float(counts["dc_Nref"] + eps)
    F = counts["dc_TP"] / (eps + counts["dc_TP"] + 0.5 * (counts["dc_FP"] + counts["dc_FN"]))
    LE = total_de / float(counts["dc_DE_TP"] + eps) if counts["dc_DE_TP"] else 180
    LR = counts["dc_DE_TP"] / (eps + counts["dc_DE_TP"] + counts["dc_DE_FN"])
    SELD_dcase21 = np.mean([ER, 1 - F, LE / 180, 1 - LR])
    SELD_L3DAS21_LRLE = np.mean([ER_score, 1 - F_score, LE / 180, 1 - LR])
    CSL_score = np.mean([LE / 180, 1 - LR])
    LSD_score = np.mean([1 - F_score, ER_score])
    return [epoch, F_score, ER_score, precision, recall, TP, FP, FN, CSL_score, LSD_score, SELD_L3DAS21_LRLE, SELD_dcase21,
            ER, F, LE, LR]


def evaluate_clips(sed, doa, target, num_frames=600, max_loc_value=2.0, spatial_threshold=2.0, doa_threshold=20,
                   num_classes=14, max_overlaps=3, epoch=0):
    """train.py:84-150 after the model call: sed (clips, T, 42), doa (clips, T, 126), target (clips, T, 168)."""
    counts = {k: 0 for k in METRIC_COUNTERS}
    total_de = 0.0
    n = num_classes * max_overlaps
    for s, d, t in zip(sed, doa, target):
        ap, xp = decode_events(s, d, max_loc_value, num_classes, max_overlaps)
        at, xt = decode_events(t[:, :n], t[:, n:], max_loc_value, num_classes, max_overlaps)
        tp, fp, fn = lsd_counts(ap, xp, at, xt, num_frames, spatial_threshold)
        counts["TP"] += tp
        counts["FP"] += fp
        counts["FN"] += fn
        c, de = dcase_counts(ap, xp, at, xt, num_frames, doa_threshold)
        for k, v in c.items():
            counts["dc_" + k] += v
        total_de += de
    return test_results(counts, total_de, epoch), counts, total_de


# --------------------------------------------------------------------------------------
# Deterministic fills shared by the fixture generator, the tests and the HIP model
# --------------------------------------------------------------------------------------
def closed_form_fill_(named_tensors, amp=0.3):
    """Fill parameters/buffers in place with the closed forms of SURVEY App. C (amplitude raised
    to `amp` per the App. C caveat).  `named_tensors` = ordered (name, tensor) pairs of a
    state dict.  Parameter index t counts only weight-like tensors."""
    t = 0
    with torch.no_grad():
        for name, ten in named_tensors:
            if name.endswith("num_batches_tracked"):
                ten.zero_()
                continue
            n = torch.arange(ten.numel(), dtype=torch.float64)
            leaf = name.rsplit(".", 1)[-1]
            is_bn = (".batch_" in name) or (".cnn." in name and name.rsplit(".", 2)[-2] == "1")
            if leaf == "running_mean":
                val = 0.05 * torch.sin(2 * n)
            elif leaf == "running_var":
                val = 1 + 0.2 * torch.cos(n) ** 2
            elif is_bn and leaf == "weight":
                val = 1 + 0.1 * torch.sin(n)
            elif is_bn and leaf == "bias":
                val = 0.1 * torch.cos(n)
            else:
                val = amp * torch.sin(0.37 * n + 1.3 * t)
                fan = max(1, ten.numel() // max(1, ten.shape[0])) if ten.dim() > 1 else 1
                val = val / math.sqrt(fan) * 2.0
                t += 1
            ten.copy_(val.view(ten.shape).to(ten.dtype))


def closed_form_input(shape, dtype=torch.float32):
    n = torch.arange(int(np.prod(shape)), dtype=torch.float64)
    return (torch.sin(0.011 * n) + 0.5 * torch.cos(0.0037 * n)).view(shape).to(dtype)
