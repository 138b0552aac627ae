"""Functional quaternion API with the names and argument meaning of the reference's
quaternion/quaternion_ops.py, computed by the gfx950 kernels (include/seld_hip.h).

The Hamilton block matrix (quaternion_ops.py:131-135 / :310-314) is never assembled; the kernels
read the four component tensors directly.  Weight initialisers are host-side numpy and reproduce
the reference's random-number draws call for call so that equal seeds give equal weights.
"""
import numpy as np
import torch
from numpy.random import RandomState

from .. import _lib as L
from .. import hip_ops as H


# ---- input checks / component views (quaternion_ops.py:52-98) ------------------------------
def check_input(input):
    if input.dim() not in {2, 3, 4, 5}:
        raise RuntimeError("Quaternion linear accepts only input of dimension 2 or 3. Quaternion conv accepts up to 5 dim "
                           " input.dim = " + str(input.dim()))
    nb_hidden = input.size()[-1] if input.dim() < 4 else input.size()[1]
    if nb_hidden % 4 != 0:
        raise RuntimeError("Quaternion Tensors must be divisible by 4. input.size()[1] = " + str(nb_hidden))


def _component(input, idx):
    check_input(input)
    axis = input.dim() - 1 if input.dim() < 4 else 1
    n = input.size(axis) // 4
    return input.narrow(axis, idx * n, n)


def get_r(input):
    return _component(input, 0)


def get_i(input):
    return _component(input, 1)


def get_j(input):
    return _component(input, 2)


def get_k(input):
    return _component(input, 3)


# ---- ops ---------------------------------------------------------------------------------
def _conv_guard(input, groups):
    if input.dim() not in (3, 4):
        if input.dim() == 5:
            raise L.SeldHipError("convolution3d has no HIP kernel (the SELD models use 1-D and 2-D only)")
        raise Exception("The convolutional input is either 3, 4 or 5 dimensions. input.dim = " + str(input.dim()))
    if groups != 1:
        raise L.SeldHipError("groups != 1 is not supported")


def quaternion_conv(input, r_weight, i_weight, j_weight, k_weight, bias, stride, padding, groups, dilatation):
    """y = W (x) x, left Hamilton product as one implicit GEMM (replaces quaternion_ops.py:125-147)."""
    _conv_guard(input, groups)
    return H.hyper_conv(input, (r_weight, i_weight, j_weight, k_weight), bias, stride, padding, dilatation)


def quaternion_linear(input, r_weight, i_weight, j_weight, k_weight, bias=None):
    """y = x @ W_hamilton + b (replaces quaternion_ops.py:299-327); any leading dims."""
    return H.hyper_linear(input, (r_weight, i_weight, j_weight, k_weight), bias, L.SELD_LIN_QUAT)


class QuaternionLinearFunction:
    """Name-compatible stand-in for the reference's custom autograd Function (quaternion_ops.py:392-464):
    `QuaternionLinearFunction.apply(input, r, i, j, k, bias)`."""

    @staticmethod
    def apply(input, r_weight, i_weight, j_weight, k_weight, bias=None):
        check_input(input)
        return quaternion_linear(input, r_weight, i_weight, j_weight, k_weight, bias)


def _no_kernel(name):
    def fn(*a, **k):
        raise L.SeldHipError(f"{name}: not on the DualQ-SELD-TCN hot path, no HIP kernel in this build")
    fn.__name__ = name
    return fn


quaternion_transpose_conv = _no_kernel("quaternion_transpose_conv")
quaternion_conv_rotation = _no_kernel("quaternion_conv_rotation")
quaternion_transpose_conv_rotation = _no_kernel("quaternion_transpose_conv_rotation")
quaternion_linear_rotation = _no_kernel("quaternion_linear_rotation")


# ---- initialisers (host side, numpy float64; quaternion_ops.py:509-645) --------------------
def _fans(in_features, out_features, kernel_size):
    if kernel_size is not None:
        rf = np.prod(kernel_size)
        return in_features * rf, out_features * rf
    return in_features, out_features


def _scale(fan_in, fan_out, criterion):
    if criterion == 'glorot':
        return 1. / np.sqrt(2 * (fan_in + fan_out))
    if criterion == 'he':
        return 1. / np.sqrt(2 * fan_in)
    raise ValueError('Invalid criterion: ' + criterion)


def _kernel_shape(in_features, out_features, kernel_size):
    if kernel_size is None:
        return (in_features, out_features)
    if type(kernel_size) is int:
        return (out_features, in_features, kernel_size)
    return (out_features, in_features) + tuple(kernel_size)


def unitary_init(in_features, out_features, rng, kernel_size=None, criterion='he'):
    s = _scale(*_fans(in_features, out_features, kernel_size), criterion)
    shape = _kernel_shape(in_features, out_features, kernel_size)
    n = int(np.prod(shape))
    v = [np.random.normal(0.0, s, n) for _ in range(4)]
    norm = np.sqrt(v[0] ** 2 + v[1] ** 2 + v[2] ** 2 + v[3] ** 2) + 0.0001
    return tuple((c / norm).reshape(shape) for c in v)


def random_init(in_features, out_features, rng, kernel_size=None, criterion='glorot'):
    s = _scale(*_fans(in_features, out_features, kernel_size), criterion)
    shape = _kernel_shape(in_features, out_features, kernel_size)
    n = int(np.prod(shape))
    return tuple(np.random.uniform(0.0, 1.0, n).reshape(shape) * s for _ in range(4))


def quaternion_init(in_features, out_features, rng, kernel_size=None, criterion='glorot'):
    """Polar-form init: modulus/phase from the fixed RandomState(123) of quaternion_ops.py:611, the unit
    imaginary axis from the GLOBAL numpy generator (:623-625)."""
    s = _scale(*_fans(in_features, out_features, kernel_size), criterion)
    fixed = RandomState(123)
    shape = _kernel_shape(in_features, out_features, kernel_size)
    n = int(np.prod(shape))
    v_i = np.random.normal(0.0, s, n)
    v_j = np.random.normal(0.0, s, n)
    v_k = np.random.normal(0.0, s, n)
    norm = np.sqrt(v_i ** 2 + v_j ** 2 + v_k ** 2) + 0.0001
    v_i, v_j, v_k = (v_i / norm).reshape(shape), (v_j / norm).reshape(shape), (v_k / norm).reshape(shape)
    modulus = fixed.uniform(low=-s, high=s, size=shape)
    phase = fixed.uniform(low=-np.pi, high=np.pi, size=shape)
    return (modulus * np.cos(phase), modulus * v_i * np.sin(phase), modulus * v_j * np.sin(phase),
            modulus * v_k * np.sin(phase))


def _same_sizes(r, i, j, k):
    if not (r.size() == i.size() == j.size() == k.size()):
        raise ValueError('The real and imaginary weights should have the same size. Found:'
                         + ' r:' + str(r.size()) + ' i:' + str(i.size()) + ' j:' + str(j.size()) + ' k:' + str(k.size()))


def _assign(params, arrays):
    for p, a in zip(params, arrays):
        p.data = torch.from_numpy(np.ascontiguousarray(a)).type_as(p.data)


def affect_init(r_weight, i_weight, j_weight, k_weight, init_func, rng, init_criterion):
    _same_sizes(r_weight, i_weight, j_weight, k_weight)
    if r_weight.dim() != 2:
        raise Exception('affect_init accepts only matrices. Found dimension = ' + str(r_weight.dim()))
    _assign((r_weight, i_weight, j_weight, k_weight),
            init_func(r_weight.size(0), r_weight.size(1), rng, None, init_criterion))


def affect_init_conv(r_weight, i_weight, j_weight, k_weight, kernel_size, init_func, rng, init_criterion):
    _same_sizes(r_weight, i_weight, j_weight, k_weight)
    if r_weight.dim() <= 2:
        raise Exception('affect_conv_init accepts only tensors that have more than 2 dimensions. Found dimension = '
                        + str(r_weight.dim()))
    _assign((r_weight, i_weight, j_weight, k_weight),
            init_func(r_weight.size(1), r_weight.size(0), rng=rng, kernel_size=kernel_size, criterion=init_criterion))


def get_kernel_and_weight_shape(operation, in_channels, out_channels, kernel_size):
    """quaternion_ops.py:706-735."""
    if operation == 'convolution1d':
        if type(kernel_size) is not int:
            raise ValueError("An invalid kernel_size was supplied for a 1d convolution. The kernel size "
                             "must be integer in the case. Found kernel_size = " + str(kernel_size))
        return kernel_size, (out_channels, in_channels, kernel_size)
    nd = {'convolution2d': 2, 'convolution3d': 3}.get(operation)
    if type(kernel_size) is int:
        ks = (kernel_size,) * (nd or 2)
    else:
        if nd is not None and len(kernel_size) != nd:
            raise ValueError(f"An invalid kernel_size was supplied for a {nd}d convolution. The kernel size must be "
                             f"either an integer or a tuple of {nd}. Found kernel_size = " + str(kernel_size))
        ks = kernel_size
    return ks, (out_channels, in_channels) + tuple(ks)
