"""(Dual-)quaternion SELD-TCN on the gfx950 kernels.

Drop-in for the reference's model.py: same class names, constructor signatures, sub-module names
(so state dicts / checkpoints are interchangeable), same `forward(x) -> (sed, doa)`.
What differs is underneath: every tensor op of the forward/backward is a HIP kernel behind
include/seld_hip.h, and the graph is fused where the reference runs op by op:

  * BatchNorm + ReLU / tanh are one pass (model.py:114-116, 279-280 in the reference),
  * the gate tanh(BN(.)) * sigmoid(BN(.)) * Dropout1d mask is one pass (:121-128),
  * `x + conv2_residual(y)` and the running sum of the skip connections ride in the epilogue of
    the 1x1 convolutions (:130-132, :210-212),
  * the dead `conv2_residual` of the last residual block (its output is discarded, :207) is skipped,
  * attention is flash-style on the (N, E, T) layout the 1x1 projections produce: none of the
    permutes of :30-37 / :220-222 and no T x T energy tensor exist.

Unlike the reference this module has no import-time side effects (the reference disables cuDNN /
MIOpen globally at model.py:10).
"""
import numpy as np
import torch
import torch.nn as nn

from . import _lib as L
from .dp import cut_backward_here
from . import hip_nn as hnn
from . import hip_ops as H
from .dual_quaternion.dual_quaternion_layers import *   # noqa: F401,F403
from .quaternion.quaternion_layers import *             # noqa: F401,F403
from .dual_quaternion.dual_quaternion_layers import DualQuaternionConv, DualQuaternionLinear
from .quaternion.quaternion_layers import QuaternionConv, QuaternionLinear

_TCN_BN = {'BN', 'BN_on_TCN', 'BNonTCN'}
_CNN_BN = {'BN', 'BN_on_CNN', 'BNonCNN'}
_TWO_STREAM = {'2Parallel', '2BParallel', '2ParallelBranches', '2PB'}


def _components(conv):
    """Component weight tensors of a real / quaternion / dual-quaternion conv module."""
    if isinstance(conv, DualQuaternionConv):
        return conv.components()
    if isinstance(conv, QuaternionConv):
        return (conv.r_weight, conv.i_weight, conv.j_weight, conv.k_weight)
    return (conv.weight,)


def _geom(conv):
    if isinstance(conv, (DualQuaternionConv, QuaternionConv)):
        return conv.stride, conv.padding, conv.dilatation
    return conv.stride, conv.padding, conv.dilation


def _make_conv(domain, nd, cin, cout, k, stride=1, padding=0, dilation=1, bias=True):
    op = 'convolution1d' if nd == 1 else 'convolution2d'
    if domain == 'Q':
        return QuaternionConv(cin, cout, kernel_size=k, stride=stride, padding=padding, dilatation=dilation,
                              bias=bias, operation=op)
    if domain == 'DQ':
        return DualQuaternionConv(cin, cout, kernel_size=k, stride=stride, padding=padding, dilatation=dilation,
                                  bias=bias, operation=op)
    cls = hnn.Conv1d if nd == 1 else hnn.Conv2d
    return cls(cin, cout, kernel_size=k, stride=stride, padding=padding, dilation=dilation, bias=bias)


class MultiHeadAttention(nn.Module):
    """model.py:12-51.  `forward(v, k, q)` takes (N, T, E) like the reference; `forward_nct(x)` is the
    self-attention fast path on (N, E, T) used by TC_Block."""

    def __init__(self, embed_size, num_heads):
        super().__init__()
        assert embed_size % num_heads == 0, "Embedding size must be divisible by number of heads"
        self.num_heads = num_heads
        self.head_dim = embed_size // num_heads
        self.values = hnn.Conv1d(embed_size, embed_size, kernel_size=1, bias=False)
        self.keys = hnn.Conv1d(embed_size, embed_size, kernel_size=1, bias=False)
        self.queries = hnn.Conv1d(embed_size, embed_size, kernel_size=1, bias=False)
        self.fc_out = hnn.Linear(embed_size, embed_size)

    def _attend(self, v, k, q):
        v, k, q = self.values(v), self.keys(k), self.queries(q)
        out = H.mha_core(q, k, v, self.num_heads)
        # fc_out on the (N, E, T) layout = 1x1 convolution with the Linear's (E, E) weight
        w = self.fc_out.weight
        return H.hyper_conv(out, (H.as_conv_weight(w, (*w.shape, 1)),), self.fc_out.bias, 1, 0, 1)

    def forward_nct(self, x):
        y = self._attend_stacked(x)
        return y if y is not None else self._attend(x, x, x)

    def _attend_stacked(self, x):
        """Self-attention with the three projections as ONE convolution (weights stacked in place: hip_ops.
        stacked_conv_weight) and the attention core on the packed result; None when the layout does not allow it."""
        convs = (self.values, self.keys, self.queries)
        if x.dim() != 3 or any(c.bias is not None or c.kernel_size != (1,) or c.stride != (1,) or c.padding != (0,) or
                               c.dilation != (1,) or c.groups != 1 for c in convs):
            return None
        if not H.mha_packed_ok(x.shape[2], self.head_dim):
            return None
        w = H.stacked_conv_weight(tuple(c.weight for c in convs))
        if w is None:
            return None
        out = H.mha_core_packed(H.hyper_conv(x, (w,), None, 1, 0, 1), self.num_heads)
        wo = self.fc_out.weight
        return H.hyper_conv(out, (H.as_conv_weight(wo, (*wo.shape, 1)),), self.fc_out.bias, 1, 0, 1)

    def forward(self, v, k, q, mask=None):
        if mask is not None:
            raise L.SeldHipError("MultiHeadAttention: attention masks are not supported (the reference always passes None)")
        if v is k and k is q:
            return H.transpose12(self.forward_nct(H.transpose12(q)))
        return H.transpose12(self._attend(H.transpose12(v), H.transpose12(k), H.transpose12(q)))


class ResBlock(nn.Module):
    """Pre-activation gated dilated residual block (model.py:53-132)."""

    def __init__(self, in_channels, domain='DQ', G=128, U=128, kernel_size_dilated_conv=3, dilation=1, stride=1,
                 spatial_dropout_rate=0.5, use_bias_conv=True, batch_norm='BN', verbose=False):
        super().__init__()
        self.verbose, self.batch_norm, self.domain = verbose, batch_norm, domain
        self.spatial_dropout_rate = spatial_dropout_rate
        padding = int(((kernel_size_dilated_conv - 1) * dilation) / 2)
        Lc = in_channels
        self.conv1_filter = _make_conv(domain, 1, Lc, G, kernel_size_dilated_conv, stride, padding, dilation, use_bias_conv)
        self.conv1_gate = _make_conv(domain, 1, Lc, G, kernel_size_dilated_conv, stride, padding, dilation, use_bias_conv)
        if batch_norm in _TCN_BN:
            self.batch_filter1 = hnn.BatchNorm1d(Lc)
            self.batch_gate1 = hnn.BatchNorm1d(Lc)      # allocated, never used: same in the reference (:90)
            self.batch_filter2 = hnn.BatchNorm1d(G)
            self.batch_gate2 = hnn.BatchNorm1d(G)
        self.tanh = hnn.Tanh()
        self.sigmoid = hnn.Sigmoid()
        if not spatial_dropout_rate == 0:
            self.dropout = hnn.Dropout1d(p=spatial_dropout_rate)
        self.conv2_skip = _make_conv(domain, 1, G, U, 1, 1, 0, 1, use_bias_conv)
        self.conv2_residual = _make_conv(domain, 1, G, Lc, 1, 1, 0, 1, use_bias_conv)

    def _conv(self, conv, x, addend=None):
        s, p, d = _geom(conv)
        if addend is None:
            return H.hyper_conv(x, _components(conv), conv.bias, s, p, d)
        return H.hyper_conv_add(x, _components(conv), conv.bias, addend, s, p, d)

    def _conv_pair(self, conv_a, conv_b, x, add_a=None, add_b=None, stats_a=None, stats_b=None):
        """Two convolutions of the same input: one call when they share the geometry (hip_ops.hyper_conv_pair)."""
        ga, gb = _geom(conv_a), _geom(conv_b)
        s, p, d = ga
        if ga != gb:
            ya, yb = self._conv(conv_a, x, add_a), self._conv(conv_b, x, add_b)
            for y_, st_ in ((ya, stats_a), (yb, stats_b)):
                if st_ is not None:
                    H.channel_stats(y_, st_)
            return ya, yb
        return H.hyper_conv_pair(x, _components(conv_a), conv_a.bias, _components(conv_b), conv_b.bias, s, p, d,
                                 add_a, add_b, stats_a, stats_b)

    def fused(self, x, skip_sum=None, need_residual=True, x_stats=None, want_res_stats=False, mask=None):
        """Returns (x_hat + conv2_residual(y)  or None, skip_sum + conv2_skip(y), statistics of the residual or None).
        The BatchNorm batch statistics of a convolution's result are gathered by that convolution's epilogue:
        x_stats are those of this block's input (from the previous block), the returned ones go to the next block."""
        bn = self.batch_norm in _TCN_BN
        train_stats = bn and self.training
        x_res = x
        if bn:
            # x_hat feeds the dilated convolutions AND the residual sum: two handles on one tensor (hip_ops.BnActFn)
            x, x_res = H.bn_act(x, self.batch_filter1, L.SELD_ACT_TANH,
                                x_stats if self.batch_filter1.training else None, twin=True)
        st_f = H.new_stats(self.batch_filter2.num_features, x.device) if train_stats else None
        st_g = H.new_stats(self.batch_gate2.num_features, x.device) if train_stats else None
        yf, yg = self._conv_pair(self.conv1_filter, self.conv1_gate, x, None, None, st_f, st_g)
        if mask is None and self.training and not self.spatial_dropout_rate == 0:
            mask = H.channel_dropout_mask(yf.shape[0], yf.shape[1], self.spatial_dropout_rate, yf.device)
        if bn:
            y = H.gate(yf, yg, self.batch_filter2, self.batch_gate2, mask, st_f, st_g)
        else:
            y = H.gate_plain(yf, yg, mask)
        res_stats = None
        if need_residual:
            if want_res_stats and train_stats:
                res_stats = H.new_stats(self.batch_filter1.num_features, x.device)
            skip, res = self._conv_pair(self.conv2_skip, self.conv2_residual, y, skip_sum, x_res, None, res_stats)
        else:
            skip, res = self._conv(self.conv2_skip, y, skip_sum), None
        return res, skip, res_stats

    def forward(self, x):
        res, skip, _ = self.fused(x, None, True)
        return res, skip


class TC_Block(nn.Module):
    """Stack of residual blocks + attention + pooled output convolutions (model.py:134-232)."""

    def __init__(self, in_channels, domain='DQ', G=128, U=128, V=[128, 128], V_kernel_size=3,
                 pool_size=[[8, 2], [8, 2], [2, 2]], D=[10], spatial_dropout_rate=0.5, use_bias_conv=True,
                 dilation_mode='fibonacci', pool_time='TCN', batch_norm='BN', kernel_size_dilated_conv=3,
                 verbose=False, attention_type=None, key_size=None, value_size=None):
        super().__init__()
        self.verbose, self.D, self.pool_time, self.domain = verbose, D, pool_time, domain
        self._gate_channels = G
        self.ResBlocks = nn.ModuleList()
        for d in self.dilation_schedule(D, dilation_mode):
            self.ResBlocks.append(ResBlock(in_channels=in_channels, domain=domain, G=G, U=U,
                                           kernel_size_dilated_conv=kernel_size_dilated_conv, dilation=d,
                                           spatial_dropout_rate=spatial_dropout_rate, use_bias_conv=use_bias_conv,
                                           batch_norm=batch_norm, verbose=verbose))
        self.relu1 = hnn.ReLU()
        if self.pool_time == 'TCN':
            self.maxpool1 = hnn.MaxPool1d(pool_size[0][1])
        self.conv1 = _make_conv(domain, 1, in_channels, V[0], V_kernel_size, 1, 1, 1, use_bias_conv)
        self.attention = MultiHeadAttention(embed_size=V[0], num_heads=8)
        self.relu2 = hnn.ReLU()
        if self.pool_time == 'TCN':
            self.maxpool2 = hnn.MaxPool1d(pool_size[1][1])
        self.conv2 = _make_conv(domain, 1, V[0], V[1], V_kernel_size, 1, 1, 1, use_bias_conv)
        self.tanh = hnn.Tanh()
        if self.pool_time == 'TCN':
            self.maxpool3 = hnn.MaxPool1d(pool_size[2][1])

    @staticmethod
    def dilation_schedule(D, dilation_mode):
        """model.py:146-174: explicit lists, Fibonacci 1,1,2,3,5,... or powers of two, per stack."""
        out = []
        for stack in D:
            if type(stack) == list:
                out.extend(stack)
                continue
            a, b = 1, 0
            for d in range(stack):
                if dilation_mode == 'fibonacci':
                    dil = 1 if d == 0 else a + b
                    if d > 0:
                        b, a = a, dil
                else:
                    dil = 2 ** d
                out.append(dil)
        return out

    def _dropout_masks(self, x):
        """The Dropout1d channel masks of ALL residual blocks from one launch.  Nothing else draws random numbers between
        the blocks, so block i's rows are exactly what its own launch would have produced (consecutive Philox groups,
        rows % 4 == 0); otherwise every block draws for itself."""
        blocks = list(self.ResBlocks)
        if not (self.training and x.is_cuda and blocks):
            return None
        rate = blocks[0].spatial_dropout_rate
        G = self._gate_channels
        if rate == 0 or any(b.spatial_dropout_rate != rate for b in blocks):
            return None
        rows = x.shape[0] * G
        if rows % 4:
            return None
        return H.channel_dropout_mask(x.shape[0] * len(blocks), G, rate, x.device).view(len(blocks), rows)

    def forward(self, residual):
        skip = None
        last = len(self.ResBlocks) - 1
        stats = None                     # batch statistics of `residual`, gathered by the convolution that wrote it
        masks = self._dropout_masks(residual)
        for i, blk in enumerate(self.ResBlocks):
            residual, skip, stats = blk.fused(residual, skip, need_residual=i < last, x_stats=stats,
                                              want_res_stats=i < last, mask=None if masks is None else masks[i])
        out = self.relu1(skip)
        if self.pool_time == 'TCN':
            out = self.maxpool1(out)
        out = self.conv1(out)
        out = self.attention.forward_nct(out)
        out = self.relu2(out)
        if self.pool_time == 'TCN':
            out = self.maxpool2(out)
        out = self.conv2(out)
        out = self.tanh(out)
        if self.pool_time == 'TCN':
            out = self.maxpool3(out)
        return out


class _CnnStage(nn.Sequential):
    """[conv3x3, BatchNorm2d, ReLU, MaxPool2d, Dropout] with the reference's sub-module indices (model.py:269-283);
    `forward` runs the fused path: conv (+ batch statistics in its epilogue) -> BN+ReLU+MaxPool in one pass -> dropout."""

    def forward(self, x):
        mods = list(self)
        conv = mods[0]
        if isinstance(mods[1], hnn.BatchNorm2d):
            bn, pool = mods[1], mods[3]
            s, p, d = _geom(conv)
            ph, pw = hnn._window(pool.kernel_size, 2)
            drop = mods[4]
            if isinstance(drop, hnn.Dropout) and drop.training == bn.training:
                # the stage's Dropout rides in the pooled-size pass of the fused first stage (same mask, same draw order)
                return H.conv_bn_relu_pool(x, _components(conv), conv.bias, bn, ph, pw, s, p, d,
                                           drop_p=drop.p if drop.training else 0.0)
            x = H.conv_bn_relu_pool(x, _components(conv), conv.bias, bn, ph, pw, s, p, d)
            return drop(x)
        x = conv(x)
        for m in mods[1:]:
            x = m(x)
        return x


class ConvTC_Block(nn.Module):
    """3 x [conv3x3 -> BN -> ReLU -> MaxPool(f, 1) -> Dropout] then the TCN (model.py:234-322)."""

    def __init__(self, time_dim, freq_dim=256, input_channels=4, domain='DQ', cnn_filters=[64, 64, 64],
                 kernel_size_cnn_blocks=3, pool_size=[[8, 2], [8, 2], [2, 2]], pool_time='TCN', D=[10],
                 dilation_mode='fibonacci', G=128, U=128, kernel_size_dilated_conv=3, spatial_dropout_rate=0.5,
                 V=[128, 128], V_kernel_size=3, dropout_perc=0.3, use_bias_conv=True, batch_norm='noBN',
                 attention_type=None, key_size=None, value_size=None, verbose=False):
        super().__init__()
        self.time_dim, self.freq_dim, self.domain, self.verbose = time_dim, freq_dim, domain, verbose
        self.D, self.kernel_size_dilated_conv, self.dilation_mode = D, kernel_size_dilated_conv, dilation_mode
        self.attenyion_type = attention_type          # (sic) attribute name of the reference, model.py:250
        self.batch_norm = batch_norm
        if pool_time == 'CNN':
            self.time_pooled_size = int(time_dim / np.prod(np.array(pool_size), axis=0)[-1])
        else:
            self.time_pooled_size = time_dim
        stages = []
        in_chans = input_channels
        for p, c in zip(pool_size, np.array(cnn_filters)):
            c = int(c)
            pool = [p[0], p[1]] if pool_time == 'CNN' else [p[0], 1]
            layers = [_make_conv(domain, 2, in_chans, c, kernel_size_cnn_blocks, 1, 1, 1, use_bias_conv)]
            if batch_norm in _CNN_BN:
                layers.append(hnn.BatchNorm2d(c))
            layers += [hnn.ReLU(), hnn.MaxPool2d(pool), hnn.Dropout(dropout_perc)]
            stages.append(_CnnStage(*layers))
            in_chans = c
        self.cnn = nn.Sequential(*stages)
        Lc = int(freq_dim / np.prod(np.array(pool_size), axis=0)[0] * cnn_filters[-1])
        self.tcn = TC_Block(in_channels=Lc, domain=domain, G=G, U=U, V=V, V_kernel_size=V_kernel_size,
                            pool_size=pool_size, D=D, spatial_dropout_rate=spatial_dropout_rate,
                            use_bias_conv=use_bias_conv, dilation_mode=dilation_mode, pool_time=pool_time,
                            batch_norm=batch_norm, kernel_size_dilated_conv=kernel_size_dilated_conv, verbose=verbose,
                            attention_type=attention_type, key_size=key_size, value_size=value_size)

    def forward(self, x):
        x = self.cnn(x)
        # (B, C, F', T) -> (B, C*F', T): the permute/reshape/permute of model.py:302-310 is a pure
        # relabelling of a contiguous NCHW tensor (channel index c*F' + f)
        B, C, Fp, T = x.shape
        x = x.reshape(B, C * Fp, T)
        # data-parallel training (dp.BackwardCut): the backward pass can stop here so that the gradients of everything
        # behind this point are exchanged while the front end's backward pass still runs; a no-op otherwise
        x = cut_backward_here(self, x)
        x = self.tcn(x)
        return H.transpose12(x)                             # (B, T', V), model.py:318


class SELD_Model(nn.Module):
    """model.py:324-480."""

    def __init__(self, time_dim, freq_dim=256, input_channels=4, output_classes=14, domain='DQ',
                 domain_classifier='same', cnn_filters=[64, 64, 64], kernel_size_cnn_blocks=3,
                 pool_size=[[8, 2], [8, 2], [2, 2]], pool_time='TCN', D=[10], dilation_mode='fibonacci', G=128, U=128,
                 kernel_size_dilated_conv=3, spatial_dropout_rate=0.5, V=[128, 128], V_kernel_size=3, fc_layers=[128],
                 fc_activations='Linear', fc_dropout='all', dropout_perc=0.3, class_overlaps=3., use_bias_conv=False,
                 use_bias_linear=True, batch_norm='BN', parallel_ConvTC_block='False', parallel_magphase=False,
                 extra_name='', attention_type=None, key_size=None, value_size=None, verbose=False):
        super().__init__()
        self.input_channels, self.time_dim, self.freq_dim = input_channels, time_dim, freq_dim
        self.domain, self.verbose, self.D = domain, verbose, D
        self.kernel_size_dilated_conv, self.dilation_mode = kernel_size_dilated_conv, dilation_mode
        self.parallel_magphase = parallel_magphase
        self.domain_classifier = domain if domain_classifier == 'same' else domain_classifier
        self.receptive_field, self.total_n_resblocks = self.calculate_receptive_field()
        self.parallel_ConvTC_block = parallel_ConvTC_block
        self.model_name = self._name(domain, dilation_mode, D, parallel_ConvTC_block, batch_norm, pool_time, extra_name)

        sed_output_size = int(output_classes * class_overlaps)
        doa_output_size = sed_output_size * 3
        block_kw = dict(time_dim=time_dim, freq_dim=freq_dim, domain=domain, cnn_filters=cnn_filters,
                        kernel_size_cnn_blocks=kernel_size_cnn_blocks, pool_size=pool_size, pool_time=pool_time, D=D,
                        dilation_mode=dilation_mode, G=G, U=U, kernel_size_dilated_conv=kernel_size_dilated_conv,
                        spatial_dropout_rate=spatial_dropout_rate, V=V, V_kernel_size=V_kernel_size,
                        dropout_perc=dropout_perc, use_bias_conv=use_bias_conv, batch_norm=batch_norm, verbose=False)
        if parallel_ConvTC_block in _TWO_STREAM:
            self.branch_A = ConvTC_Block(input_channels=input_channels // 2, **block_kw)
            self.branch_B = ConvTC_Block(input_channels=input_channels // 2, **block_kw)
            fc_input_size = V[-1] * 2
        else:
            self.seld_block = ConvTC_Block(input_channels=input_channels, attention_type=attention_type,
                                           key_size=key_size, value_size=value_size, **block_kw)
            fc_input_size = V[-1]

        sed_layers, doa_layers = [], []
        for width in fc_layers:
            for layers in (sed_layers, doa_layers):
                if self.domain_classifier == 'Q':
                    layers.append(QuaternionLinear(fc_input_size, width, bias=use_bias_linear))
                elif self.domain_classifier == 'DQ':
                    layers.append(DualQuaternionLinear(fc_input_size, width, bias=use_bias_linear))
                else:
                    layers.append(hnn.Linear(fc_input_size, width, bias=use_bias_linear))
            if fc_activations in {'relu', 'ReLU', 'RELU'}:
                sed_layers.append(hnn.ReLU())
                doa_layers.append(hnn.ReLU())
            if fc_dropout in {'all', 'ALL', 'True'}:
                sed_layers.append(hnn.Dropout(dropout_perc))
                doa_layers.append(hnn.Dropout(dropout_perc))
            fc_input_size = width
        if fc_dropout in {'last', 'Last', 'LAST'}:
            sed_layers.append(hnn.Dropout(dropout_perc))
            doa_layers.append(hnn.Dropout(dropout_perc))
        self.sed = nn.Sequential(*sed_layers, hnn.Linear(fc_layers[-1], sed_output_size, bias=use_bias_linear),
                                 hnn.Sigmoid())
        self.doa = nn.Sequential(*doa_layers, hnn.Linear(fc_layers[-1], doa_output_size, bias=use_bias_linear),
                                 hnn.Tanh())

    def _name(self, domain, dilation_mode, D, parallel, batch_norm, pool_time, extra_name):
        """Checkpoint / log name, model.py:347-372."""
        if domain in {'q', 'Q', 'quaternion', 'Quaternion'}:
            name = 'Q'
        elif domain in {'dq', 'dQ', 'DQ', 'dual_quaternion', 'Dual_Quaternion'}:
            name = 'DualQ'
        else:
            name = ''
        name += 'SELD-TCN'
        if dilation_mode == 'fibonacci':
            name += '-PHI'
        name += '-'
        if len(D) > 1 and D[0] < D[1]:
            name += 'I'
        name += 'S' + str(len(D))
        if parallel not in {'False', 'false', 'None', 'none'}:
            name += '_' + parallel
        name += '_' + batch_norm
        if pool_time == 'CNN':
            name += '_pooltCNN'
        name += '_RF{}_{}RB'.format(self.receptive_field, self.total_n_resblocks)
        return name + extra_name

    def forward(self, x):
        if self.parallel_ConvTC_block in _TWO_STREAM:
            if self.parallel_magphase:
                x_A = torch.cat((x[:, :4], x[:, 8:12]), 1)      # mic A magnitude + phase
                x_B = torch.cat((x[:, 4:8], x[:, 12:]), 1)      # mic B magnitude + phase
            else:
                h = self.input_channels // 2
                x_A, x_B = x[:, :h].contiguous(), x[:, h:].contiguous()
            y_A, y_B = H.run_branches(self.branch_A, x_A, self.branch_B, x_B)
            x = torch.cat((y_A, y_B), 2)
        else:
            x = self.seld_block(x)
        # the two classifier heads are independent chains of small-grid kernels: two queues (hip_ops.run_branches)
        xa, xb = H.fan_out2(x)
        return H.run_branches(self.sed, xa, self.doa, xb)

    def calculate_receptive_field(self, verbose=0):
        """model.py:482-517."""
        k = self.kernel_size_dilated_conv
        dil = TC_Block.dilation_schedule(self.D, self.dilation_mode)
        rf = 1 + sum((k - 1) * d for d in dil)
        if verbose:
            print(self.D, '  Receptive field:', rf, ', Total number of Resblocks:', len(dil))
        return rf, len(dil)
