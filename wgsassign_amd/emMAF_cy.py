"""Drop-in for the reference's Cython module `emMAF_cy` (emMAF_cy.pyx), backed by HIP.

Same signatures, in-place semantics and buffer contract as the typed memoryviews
`float[:,::1]` / `float[::1]`: float32, C-contiguous, writable -- violations raise the same
ValueError texts Cython produces.  `t` (OpenMP threads) is accepted and ignored.
"""
import ctypes

import numpy as np

from . import _lib
from .device import default_em_mode, get_context

_CNAMES = {"float64": "double", "float32": "float", "int32": "int", "int64": "long", "int16": "short",
           "int8": "signed char", "uint8": "unsigned char", "uint16": "unsigned short", "uint32": "unsigned int",
           "uint64": "unsigned long", "float16": "half", "bool": "bool"}


def _memview(a, ndim):
    """Checks Cython performs when binding `float[:, ::1]` / `float[::1]`."""
    if not isinstance(a, np.ndarray):
        a = np.asarray(a)
    if a.ndim != ndim:
        raise ValueError("Buffer has wrong number of dimensions (expected %d, got %d)" % (ndim, a.ndim))
    if a.dtype != np.float32:
        raise ValueError("Buffer dtype mismatch, expected 'float' but got '%s'"
                         % _CNAMES.get(a.dtype.name, a.dtype.name))
    if not a.flags.c_contiguous:
        raise ValueError("ndarray is not C-contiguous")
    if not a.flags.writeable:
        raise ValueError("buffer source array is read-only")
    return a


def emMAF_update(L, f, t=1):
    """emMAF_cy.pyx:10-23: one EM step of every SNP's frequency; f is updated in place."""
    L = _memview(L, 2)
    f = _memview(f, 1)
    m, n = L.shape[0], L.shape[1] // 2
    if m == 0:
        return None
    if L.shape[1] != 2 * n:
        L = np.ascontiguousarray(L[:, :2 * n])
    _lib.check(_lib.load().wgs_emmaf_update(get_context().handle, _lib.f32p(L), m, n, _lib.f32p(f), default_em_mode()))
    return None


def rmse1d(v1, v2):
    """emMAF_cy.pyx:26-33: sqrt of the serially float32-accumulated mean squared difference."""
    v1 = _memview(v1, 1)
    v2 = _memview(v2, 1)
    out = ctypes.c_double()
    _lib.check(_lib.load().wgs_rmse1d(get_context().handle, _lib.f32p(v1), _lib.f32p(v2), v1.shape[0], ctypes.byref(out)))
    return out.value
