"""torch.nn-shaped modules whose forward runs on the HIP kernels.

Each class subclasses the torch.nn module it stands in for, so constructor signatures, parameter /
buffer names, default initialisation (and therefore torch RNG consumption and state-dict layout)
are exactly those the reference gets from torch.nn (model.py:20-23, 81-107, 175-202, 276-282,
439-459).  Only `forward` is replaced.
"""
import torch
import torch.nn as nn

from . import _lib as L
from . import hip_ops as H


class Conv1d(nn.Conv1d):
    def forward(self, x):
        if self.groups != 1 or self.padding_mode != "zeros" or isinstance(self.padding, str):
            raise L.SeldHipError("Conv1d: only groups=1, zero padding given as integers is supported")
        return H.hyper_conv(x, (self.weight,), self.bias, self.stride, self.padding, self.dilation)


class Conv2d(nn.Conv2d):
    def forward(self, x):
        if self.groups != 1 or self.padding_mode != "zeros" or isinstance(self.padding, str):
            raise L.SeldHipError("Conv2d: only groups=1, zero padding given as integers is supported")
        return H.hyper_conv(x, (self.weight,), self.bias, self.stride, self.padding, self.dilation)


class Linear(nn.Linear):
    def forward(self, x):
        return H.hyper_linear(x, (self.weight,), self.bias, L.SELD_LIN_REAL)


class BatchNorm1d(nn.BatchNorm1d):
    def forward(self, x):
        return H.bn_act(x, self, L.SELD_ACT_NONE)


class BatchNorm2d(nn.BatchNorm2d):
    def forward(self, x):
        return H.bn_act(x, self, L.SELD_ACT_NONE)


class ReLU(nn.ReLU):
    def forward(self, x):
        return H.act(x, L.SELD_ACT_RELU)


class Tanh(nn.Tanh):
    def forward(self, x):
        return H.act(x, L.SELD_ACT_TANH)


class Sigmoid(nn.Sigmoid):
    def forward(self, x):
        return H.act(x, L.SELD_ACT_SIGMOID)


def _window(v, nd):
    if isinstance(v, (tuple, list)):
        return tuple(int(a) for a in v)
    return (int(v),) * nd


class MaxPool1d(nn.MaxPool1d):
    def forward(self, x):
        k = _window(self.kernel_size, 1)[0]
        s = _window(self.stride, 1)[0]
        if s != k or self.padding not in (0, (0,)) or self.ceil_mode:
            raise L.SeldHipError("MaxPool1d: only stride == kernel_size, no padding, floor mode")
        return H.maxpool(x, 1, k)


class MaxPool2d(nn.MaxPool2d):
    def forward(self, x):
        k = _window(self.kernel_size, 2)
        s = _window(self.stride, 2)
        if s != k or self.padding not in (0, (0, 0)) or self.ceil_mode:
            raise L.SeldHipError("MaxPool2d: only stride == kernel_size, no padding, floor mode")
        return H.maxpool(x, k[0], k[1])


class Dropout(nn.Dropout):
    def forward(self, x):
        return H.dropout(x, self.p, self.training)


class Dropout1d(nn.Dropout1d):
    """Zeroes whole channels of a (N, C, T) tensor (model.py:96-97)."""

    def forward(self, x):
        if not self.training or self.p == 0.0:
            return x
        mask = H.channel_dropout_mask(x.shape[0], x.shape[1], self.p, x.device)
        return H.RowScaleFn.apply(x, mask)
