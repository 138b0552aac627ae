"""Dual-quaternion layers: class names, constructor signatures and parameter names
(r_weight ... k_weight_2, bias) of the reference's dual_quaternion/dual_quaternion_layers.py."""
import numpy as np
import torch
from numpy.random import RandomState
from torch.nn import Module
from torch.nn.parameter import Parameter

from .dual_quaternion_ops import *     # noqa: F401,F403
from . import dual_quaternion_ops as _ops

_NAMES = ('r_weight', 'i_weight', 'j_weight', 'k_weight', 'r_weight_2', 'i_weight_2', 'j_weight_2', 'k_weight_2')


class DualQuaternionConv(Module):
    """Channels hold C/8 dual quaternions, component-major [p_r|p_i|p_j|p_k|d_r|d_i|d_j|d_k]
    (dual_quaternion_layers.py:49-135).  `scale` / `rotation` create their parameters as in the
    reference, whose forward ignores them (:115-119); so does this one."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, dilatation=1, padding=0, groups=1, bias=True,
                 init_criterion='glorot', weight_init='quaternion', seed=None, operation='convolution2d',
                 rotation=False, quaternion_format=True, scale=False):
        super().__init__()
        self.in_channels = in_channels // 8
        self.out_channels = out_channels // 8
        self.stride, self.padding, self.groups, self.dilatation = stride, padding, groups, dilatation
        self.init_criterion, self.weight_init = init_criterion, weight_init
        self.seed = seed if seed is not None else np.random.randint(0, 1234)
        self.rng = RandomState(self.seed)
        self.operation, self.rotation, self.quaternion_format = operation, rotation, quaternion_format
        self.winit = {'quaternion': _ops.quaternion_init, 'unitary': _ops.unitary_init,
                      'random': _ops.random_init}[self.weight_init]
        self.scale = scale
        self.kernel_size, self.w_shape = _ops.get_kernel_and_weight_shape(self.operation, self.in_channels,
                                                                          self.out_channels, kernel_size)
        for name in _NAMES:
            setattr(self, name, Parameter(torch.Tensor(*self.w_shape)))
        self.scale_param = Parameter(torch.Tensor(self.r_weight.shape)) if self.scale else None
        if self.rotation:
            self.zero_kernel = Parameter(torch.zeros(self.r_weight.shape), requires_grad=False)
        if bias:
            self.bias = Parameter(torch.Tensor(out_channels))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        _ops.affect_init_conv(self.r_weight, self.i_weight, self.j_weight, self.k_weight, self.kernel_size, self.winit,
                              self.rng, self.init_criterion, self.r_weight_2, self.i_weight_2, self.j_weight_2,
                              self.k_weight_2)
        if self.scale_param is not None:
            torch.nn.init.xavier_uniform_(self.scale_param.data)
        if self.bias is not None:
            self.bias.data.zero_()

    def components(self):
        return tuple(getattr(self, n) for n in _NAMES)

    def forward(self, input):
        return _ops.dual_quaternion_conv(input, *self.components(), self.bias, self.stride, self.padding, self.groups,
                                         self.dilatation)

    def extra_repr(self):
        return (f"in_channels={self.in_channels}, out_channels={self.out_channels}, bias={self.bias is not None}, "
                f"kernel_size={self.kernel_size}, stride={self.stride}, padding={self.padding}, "
                f"init_criterion={self.init_criterion}, weight_init={self.weight_init}, seed={self.seed}, "
                f"operation={self.operation}")


class DualQuaternionLinear(Module):
    """dual_quaternion_layers.py:138-206; weights are (in/8, out/8); 2-D or 3-D input."""

    def __init__(self, in_features, out_features, bias=True, init_criterion='he', weight_init='quaternion', seed=None):
        super().__init__()
        self.in_features = in_features // 8
        self.out_features = out_features // 8
        for name in _NAMES:
            setattr(self, name, Parameter(torch.Tensor(self.in_features, self.out_features)))
        if bias:
            self.bias = Parameter(torch.Tensor(self.out_features * 8))
        else:
            self.register_parameter('bias', None)
        self.init_criterion, self.weight_init = init_criterion, weight_init
        self.seed = seed if seed is not None else np.random.randint(0, 1234)
        self.rng = RandomState(self.seed)
        self.reset_parameters()

    def reset_parameters(self):
        winit = {'quaternion': _ops.quaternion_init, 'unitary': _ops.unitary_init}[self.weight_init]
        if self.bias is not None:
            self.bias.data.fill_(0)
        _ops.affect_init(*(getattr(self, n) for n in _NAMES), winit, self.rng, self.init_criterion)

    def components(self):
        return tuple(getattr(self, n) for n in _NAMES)

    def forward(self, input):
        if input.dim() not in (2, 3):
            raise NotImplementedError
        return _ops.dual_quaternion_linear(input, *self.components(), bias=self.bias)

    def extra_repr(self):
        return (f"in_features={self.in_features}, out_features={self.out_features}, bias={self.bias is not None}, "
                f"init_criterion={self.init_criterion}, weight_init={self.weight_init}, seed={self.seed}")
