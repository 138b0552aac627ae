"""Functional dual-quaternion API with the names / argument meaning of the reference's
dual_quaternion/dual_quaternion_ops.py, computed by the gfx950 kernels.

Convolution:  y_p = Q (x) x_p ,  y_d = Q2 (x) x_p + Q (x) x_d   -- block matrix [[Q, 0], [Q2, Q]]
(dual_quaternion_ops.py:111-153).  The linear layer uses the TRANSPOSED arrangement of
dual_quaternion_ops.py:170-188; both are table lookups inside the kernels.
"""
import numpy as np
import torch
from numpy.random import RandomState
from scipy.stats import chi

from .. import _lib as L
from .. import hip_ops as H
from ..quaternion.quaternion_ops import (_assign, _conv_guard, _fans, _kernel_shape, _same_sizes, _scale,  # noqa: F401
                                         get_kernel_and_weight_shape, random_init as _q_random_init)


def check_input(input):
    """dual_quaternion_ops.py:14-31 (like the reference, only divisibility by 4 is checked here;
    the kernels additionally require divisibility by 8 and report SELD_EINVAL otherwise)."""
    if input.dim() not in {2, 3, 4, 5}:
        raise RuntimeError("Quaternion linear accepts only input of dimension 2 or 3. Quaternion conv accepts up to 5 dim "
                           " input.dim = " + str(input.dim()))
    nb_hidden = input.size()[-1] if input.dim() < 4 else input.size()[1]
    if nb_hidden % 4 != 0:
        raise RuntimeError("Quaternion Tensors must be divisible by 4. input.size()[1] = " + str(nb_hidden))


def dual_quaternion_conv(input, r_weight, i_weight, j_weight, k_weight, r_weight_2, i_weight_2, j_weight_2,
                         k_weight_2, bias, stride, padding, groups, dilatation):
    _conv_guard(input, groups)
    return H.hyper_conv(input, (r_weight, i_weight, j_weight, k_weight, r_weight_2, i_weight_2, j_weight_2,
                                k_weight_2), bias, stride, padding, dilatation)


def dual_quaternion_linear(input, r_weight, i_weight, j_weight, k_weight, r_weight_2, i_weight_2, j_weight_2,
                           k_weight_2, bias=True):
    b = None if (bias is None or bias is True or bias is False) else bias
    return H.hyper_linear(input, (r_weight, i_weight, j_weight, k_weight, r_weight_2, i_weight_2, j_weight_2,
                                  k_weight_2), b, L.SELD_LIN_DUALQ)


# ---- initialisers (dual_quaternion_ops.py:416-552) -----------------------------------------
def unitary_init(in_features, out_features, rng, kernel_size=None, criterion='he'):
    shape = _kernel_shape(in_features, out_features, kernel_size)
    n = int(np.prod(shape))
    v = [np.random.uniform(-1.0, 1.0, n) for _ in range(4)]
    norm = np.sqrt(v[0] ** 2 + v[1] ** 2 + v[2] ** 2 + v[3] ** 2) + 0.0001
    return tuple((c / norm).reshape(shape) for c in v)


def random_init(in_features, out_features, rng, kernel_size=None, criterion='glorot'):
    _scale(*_fans(in_features, out_features, kernel_size), criterion)      # validates the criterion only
    shape = _kernel_shape(in_features, out_features, kernel_size)
    n = int(np.prod(shape))
    return tuple(np.random.uniform(-1.0, 1.0, n).reshape(shape) for _ in range(4))


def quaternion_init(in_features, out_features, rng, kernel_size=None, criterion='glorot'):
    """Modulus ~ chi(4, scale=s) (dual_quaternion_ops.py:529), axis uniform on the sphere from the global
    numpy generator, phase from a RandomState seeded by one global draw (:518)."""
    s = _scale(*_fans(in_features, out_features, kernel_size), criterion)
    local = RandomState(np.random.randint(1, 1234))
    shape = _kernel_shape(in_features, out_features, kernel_size)
    modulus = chi.rvs(4, loc=0, scale=s, size=shape)
    n = int(np.prod(shape))
    v_i = np.random.uniform(-1.0, 1.0, n)
    v_j = np.random.uniform(-1.0, 1.0, n)
    v_k = np.random.uniform(-1.0, 1.0, n)
    norm = np.sqrt(v_i ** 2 + v_j ** 2 + v_k ** 2 + 0.0001)
    v_i, v_j, v_k = (v_i / norm).reshape(shape), (v_j / norm).reshape(shape), (v_k / norm).reshape(shape)
    phase = local.uniform(low=-np.pi, high=np.pi, size=shape)
    return (modulus * np.cos(phase), modulus * v_i * np.sin(phase), modulus * v_j * np.sin(phase),
            modulus * v_k * np.sin(phase))


def affect_init(r_weight, i_weight, j_weight, k_weight, r_weight_2, i_weight_2, j_weight_2, k_weight_2, init_func,
                rng, init_criterion):
    _same_sizes(r_weight, i_weight, j_weight, k_weight)
    if r_weight.dim() != 2:
        raise Exception('affect_init accepts only matrices. Found dimension = ' + str(r_weight.dim()))
    _assign((r_weight, i_weight, j_weight, k_weight),
            init_func(r_weight.size(0), r_weight.size(1), rng, None, init_criterion))
    _assign((r_weight_2, i_weight_2, j_weight_2, k_weight_2),
            init_func(r_weight_2.size(0), r_weight_2.size(1), rng, None, init_criterion))


def affect_init_conv(r_weight, i_weight, j_weight, k_weight, kernel_size, init_func, rng, init_criterion,
                     r_weight_2=None, i_weight_2=None, j_weight_2=None, k_weight_2=None):
    _same_sizes(r_weight, i_weight, j_weight, k_weight)
    if r_weight.dim() <= 2:
        raise Exception('affect_conv_init accepts only tensors that have more than 2 dimensions. Found dimension = '
                        + str(r_weight.dim()))
    _assign((r_weight, i_weight, j_weight, k_weight),
            init_func(r_weight.size(1), r_weight.size(0), rng=rng, kernel_size=kernel_size, criterion=init_criterion))
    if r_weight_2 is not None:
        _assign((r_weight_2, i_weight_2, j_weight_2, k_weight_2),
                init_func(r_weight_2.size(1), r_weight_2.size(0), rng=rng, kernel_size=kernel_size,
                          criterion=init_criterion))
