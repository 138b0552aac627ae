"""Quaternion layers: class names, constructor signatures and parameter names of the reference's
quaternion/quaternion_layers.py (checkpoints are interchangeable); forwards run on HIP."""
import numpy as np
import torch
from numpy.random import RandomState
from torch.nn import Module
from torch.nn.parameter import Parameter

from .quaternion_ops import *          # noqa: F401,F403  (the reference star-imports its ops too)
from . import quaternion_ops as _ops

_INITS = {'quaternion': _ops.quaternion_init, 'unitary': _ops.unitary_init, 'random': _ops.random_init}


class _QuaternionConvBase(Module):
    _transposed = False

    def __init__(self, in_channels, out_channels, kernel_size, stride, dilatation, padding, groups, bias,
                 init_criterion, weight_init, seed, operation, rotation, quaternion_format):
        super().__init__()
        self.in_channels = in_channels // 4
        self.out_channels = out_channels // 4
        self.stride, self.padding, self.groups, self.dilatation = stride, padding, groups, dilatation
        self.init_criterion, self.weight_init = init_criterion, weight_init
        self.seed = seed if seed is not None else np.random.randint(0, 1234)
        self.rng = RandomState(self.seed)
        self.operation, self.rotation, self.quaternion_format = operation, rotation, quaternion_format
        self.winit = _INITS[self.weight_init]
        a, b = (self.out_channels, self.in_channels) if self._transposed else (self.in_channels, self.out_channels)
        self.kernel_size, self.w_shape = _ops.get_kernel_and_weight_shape(self.operation, a, b, kernel_size)
        for name in ('r_weight', 'i_weight', 'j_weight', 'k_weight'):
            setattr(self, name, Parameter(torch.Tensor(*self.w_shape)))
        if bias:
            self.bias = Parameter(torch.Tensor(out_channels))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        _ops.affect_init_conv(self.r_weight, self.i_weight, self.j_weight, self.k_weight, self.kernel_size, self.winit,
                              self.rng, self.init_criterion)
        if self.bias is not None:
            self.bias.data.zero_()

    def extra_repr(self):
        return (f"in_channels={self.in_channels}, out_channels={self.out_channels}, bias={self.bias is not None}, "
                f"kernel_size={self.kernel_size}, stride={self.stride}, padding={self.padding}, "
                f"dilatation={self.dilatation}, init_criterion={self.init_criterion}, weight_init={self.weight_init}, "
                f"seed={self.seed}, operation={self.operation}")


class QuaternionConv(_QuaternionConvBase):
    """y = W (x) x over component-major channels [r|i|j|k] (quaternion_layers.py:100-172)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, dilatation=1, padding=0, groups=1, bias=True,
                 init_criterion='glorot', weight_init='quaternion', seed=None, operation='convolution2d',
                 rotation=False, quaternion_format=False):
        super().__init__(in_channels, out_channels, kernel_size, stride, dilatation, padding, groups, bias,
                         init_criterion, weight_init, seed, operation, rotation, quaternion_format)

    def forward(self, input):
        if self.rotation:
            return _ops.quaternion_conv_rotation(input)
        return _ops.quaternion_conv(input, self.r_weight, self.i_weight, self.j_weight, self.k_weight, self.bias,
                                    self.stride, self.padding, self.groups, self.dilatation)


class QuaternionTransposeConv(_QuaternionConvBase):
    """API surface only (quaternion_layers.py:19-98): parameters and state dict are provided, the SELD
    models never call it and there is no HIP kernel for it."""
    _transposed = True

    def __init__(self, in_channels, out_channels, kernel_size, stride, dilatation=1, padding=0, output_padding=0,
                 groups=1, bias=True, init_criterion='glorot', weight_init='quaternion', seed=None,
                 operation='convolution2d', rotation=False, quaternion_format=False):
        self.output_padding = output_padding
        super().__init__(in_channels, out_channels, kernel_size, stride, dilatation, padding, groups, bias,
                         init_criterion, weight_init, seed, operation, rotation, quaternion_format)

    def forward(self, input):
        return _ops.quaternion_transpose_conv(input)


class _QuaternionLinearBase(Module):
    def __init__(self, in_features, out_features, bias, init_criterion, weight_init, seed):
        super().__init__()
        self.in_features = in_features // 4
        self.out_features = out_features // 4
        for name in ('r_weight', 'i_weight', 'j_weight', 'k_weight'):
            setattr(self, name, Parameter(torch.Tensor(self.in_features, self.out_features)))
        if bias:
            self.bias = Parameter(torch.Tensor(self.out_features * 4))
        else:
            self.register_parameter('bias', None)
        self.init_criterion, self.weight_init = init_criterion, weight_init
        self.seed = seed if seed is not None else np.random.randint(0, 1234)
        self.rng = RandomState(self.seed)
        self.reset_parameters()

    def reset_parameters(self):
        if self.bias is not None:
            self.bias.data.fill_(0)
        _ops.affect_init(self.r_weight, self.i_weight, self.j_weight, self.k_weight, self._winit(), self.rng,
                         self.init_criterion)

    def extra_repr(self):
        return (f"in_features={self.in_features}, out_features={self.out_features}, bias={self.bias is not None}, "
                f"init_criterion={self.init_criterion}, weight_init={self.weight_init}, seed={self.seed}")


class QuaternionLinear(_QuaternionLinearBase):
    """quaternion_layers.py:227-286; 2-D or 3-D input (3-D is flattened over the two leading dims)."""

    def __init__(self, in_features, out_features, bias=True, init_criterion='glorot', weight_init='quaternion',
                 seed=None):
        super().__init__(in_features, out_features, bias, init_criterion, weight_init, seed)

    def _winit(self):
        return {'quaternion': _ops.quaternion_init, 'unitary': _ops.unitary_init}[self.weight_init]

    def forward(self, input):
        if input.dim() not in (2, 3):
            raise NotImplementedError
        return _ops.QuaternionLinearFunction.apply(input, self.r_weight, self.i_weight, self.j_weight, self.k_weight,
                                                   self.bias)


class QuaternionLinearAutograd(_QuaternionLinearBase):
    """quaternion_layers.py:174-225."""

    def __init__(self, in_features, out_features, bias=True, init_criterion='glorot', weight_init='quaternion',
                 seed=None, rotation=False, quaternion_format=False):
        self.rotation, self.quaternion_format = rotation, quaternion_format
        super().__init__(in_features, out_features, bias, init_criterion, weight_init, seed)

    def _winit(self):
        return _INITS[self.weight_init]

    def forward(self, input):
        if self.rotation:
            return _ops.quaternion_linear_rotation(input)
        return _ops.quaternion_linear(input, self.r_weight, self.i_weight, self.j_weight, self.k_weight, self.bias)
