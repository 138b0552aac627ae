"""ctypes binding of libseld_hip.so (the C ABI declared in include/seld_hip.h).

There is deliberately NO fallback: if the shared library is missing or a launch fails the
call raises.  The library is built in-tree by `__graft_entry__.build()` (hipcc, gfx950).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SELD_HIP_LIB: another build of the same library (tools/hcq_ablate.sh timing variants); there is still no fallback
LIB_PATH = os.environ.get("SELD_HIP_LIB") or os.path.join(_HERE, "csrc", "libseld_hip.so")

SELD_OK = 0
_ERRORS = {-1: "SELD_EINVAL", -2: "SELD_EWORKSPACE", -3: "SELD_ELAUNCH", -4: "SELD_EUNSUPPORTED"}

SELD_EPI_NONE, SELD_EPI_ACCUMULATE, SELD_EPI_ADD, SELD_EPI_STATS = 0, 1, 2, 4
SELD_ACT_NONE, SELD_ACT_RELU, SELD_ACT_TANH, SELD_ACT_SIGMOID = 0, 1, 2, 3
SELD_LIN_REAL, SELD_LIN_QUAT, SELD_LIN_DUALQ = 1, 4, 8


class SeldHipError(RuntimeError):
    pass


class ConvDesc(ctypes.Structure):
    _fields_ = [("algebra", ctypes.c_int32), ("ndim", ctypes.c_int32), ("N", ctypes.c_int32),
                ("Cin", ctypes.c_int32), ("Cout", ctypes.c_int32), ("in_", ctypes.c_int32 * 2),
                ("k", ctypes.c_int32 * 2), ("stride", ctypes.c_int32 * 2), ("pad", ctypes.c_int32 * 2),
                ("dil", ctypes.c_int32 * 2), ("groups", ctypes.c_int32)]


class WgradJob(ctypes.Structure):
    """seld_wgrad_job (include/seld_hip.h): one convolution of a grouped weight-gradient call."""
    _fields_ = [("desc", ConvDesc), ("x", ctypes.c_void_p), ("dy", ctypes.c_void_p), ("dw", ctypes.c_void_p * 8)]


_lib = None


def lib():
    """Load (once) and return the ctypes handle.  Raises if the HIP library is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SeldHipError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'`. "
                "This package has no CPU/eager fallback by design.")
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.seld_build_arch.restype = ctypes.c_char_p
        _lib.seld_hc_conv_bwd_weight_workspace.restype = ctypes.c_size_t
        _lib.seld_hcq_wgrad_group_workspace.restype = ctypes.c_size_t
    return _lib


_reload_hooks = []


def reload_env():
    """Make the library re-read its SELD_* environment switches (it reads them once, at first use); host-side caches
    of kernel choices (hip_ops) are dropped with it."""
    check(lib().seld_env_reload(), "seld_env_reload")
    for fn in _reload_hooks:
        fn()


def check(rc, what):
    if rc != SELD_OK:
        extra = ""
        if rc == -3:
            extra = f" (hipError {lib().seld_last_hip_error()})"
        raise SeldHipError(f"{what} failed: {_ERRORS.get(rc, rc)}{extra}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def ptr_array8(tensors):
    arr = (ctypes.c_void_p * 8)()
    for i in range(8):
        arr[i] = tensors[i].data_ptr() if i < len(tensors) and tensors[i] is not None else 0
    return arr


def current_stream():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
