"""Argument checking that mirrors the reference's Cython typed-ndarray signatures.

`np.ndarray[float, ndim=2, mode='c']` rejects wrong dtype / rank / layout with
ValueError (Cython's "Buffer dtype mismatch" / "ndarray is not C-contiguous"), and
the function bodies validate shapes with `assert` (AssertionError); see e.g.
transient_rendering_cython/smoothed_transient/renderer.pyx:94-111.
"""
import ctypes

import numpy as np


def buf(a, dtype, ndim, name):
    if not isinstance(a, np.ndarray):
        raise TypeError("Argument '%s' has incorrect type (expected numpy.ndarray, got %s)"
                        % (name, type(a).__name__))
    if a.ndim != ndim:
        raise ValueError("Buffer has wrong number of dimensions (expected %d, got %d)" % (ndim, a.ndim))
    if a.dtype != np.dtype(dtype):
        raise ValueError("Buffer dtype mismatch, expected '%s' but got '%s'"
                         % (np.dtype(dtype).name, a.dtype.name))
    if not a.flags["C_CONTIGUOUS"]:
        raise ValueError("ndarray is not C-contiguous")
    return a


def f32(a, ndim, name):
    return buf(a, np.float32, ndim, name)


def i32(a, ndim, name):
    return buf(a, np.int32, ndim, name)


def f64(a, ndim, name):
    return buf(a, np.float64, ndim, name)


def ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)
