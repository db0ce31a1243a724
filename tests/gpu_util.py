"""Helpers for the -m gpu tests: device buffers via torch (plumbing only), every compute call goes through the C ABI."""
import ctypes

import numpy as np
import torch

from recursivefactorization.jl_amd import _ffi


def handle():
    h = _ffi.default_handle(0)
    h.set_stream(None)
    return h


def ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def to_dev_rm(A):
    """numpy (any layout) -> row-major device tensor with the same logical shape."""
    return torch.from_numpy(np.ascontiguousarray(A)).to("cuda:0")


def to_dev_cm(A):
    """numpy -> column-major device view (stride(0) == 1), like a Julia Matrix."""
    return torch.from_numpy(np.ascontiguousarray(np.asarray(A).T)).to("cuda:0").T


def sfx(dtype):
    return "f64" if np.dtype(dtype) == np.float64 else "f32"


def tdtype(dtype):
    return torch.float64 if np.dtype(dtype) == np.float64 else torch.float32
