"""Drop-in for the reference's Cython module `glassy_cy` (glassy_cy.pyx), backed by HIP."""
import numpy as np

from . import _lib
from .device import default_mode, get_context
from .emMAF_cy import _memview


def loglike(L, A, loglike_vec, t, i, k):
    """glassy_cy.pyx:12-21: per-site log-likelihood of individual i under population k,
    ACCUMULATED into loglike_vec in place.  The reference does no bounds checks on i, k
    (boundscheck=False); here out-of-range indices raise ValueError instead of reading wild."""
    L = _memview(L, 2)
    A = _memview(A, 2)
    vec = _memview(loglike_vec, 1)
    m, n = L.shape[0], L.shape[1] // 2
    if m == 0:
        return None
    if A.shape[0] < m or vec.shape[0] < m:
        raise ValueError("A and loglike_vec must cover the %d sites of L" % m)
    if L.shape[1] != 2 * n:   # odd trailing column: the reference ignores it (n = L.shape[1]//2)
        L = np.ascontiguousarray(L[:, :2 * n])
    _lib.check(_lib.load().wgs_loglike(get_context().handle, _lib.f32p(L), m, n,
                                       _lib.f32p(A), A.shape[1], _lib.f32p(vec), int(i), int(k), default_mode()))
    return None
