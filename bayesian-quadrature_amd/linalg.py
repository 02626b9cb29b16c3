"""Drop-in for the reference's ``linalg_c`` module (linalg_c.pyx:55-408).

Same function names, argument order, in-place/aliasing semantics and error
behaviour; the Cholesky factor, the solves and the log-determinant run on the
GPU through the C ABI.  The tiny dot-product helpers stay on the host: the
reference itself special-cases n in {1, 2} (linalg_c.pyx:236-243) because they
are only ever called on d-vectors.
"""
import numpy as np

from .engine import get_engine


def _mat(a, name):
    if not isinstance(a, np.ndarray) or a.dtype != np.float64 or a.ndim != 2:
        raise ValueError("%s must be a 2-D float64 array" % name)
    if not a.flags.f_contiguous:
        # the Cython memoryview float64_t[::1, :] rejects C-order input
        raise ValueError("ndarray is not Fortran contiguous")
    return a


def _vec(a, name):
    if not isinstance(a, np.ndarray) or a.dtype != np.float64 or a.ndim != 1:
        raise ValueError("%s must be a 1-D float64 array" % name)
    if not a.flags.c_contiguous:
        raise ValueError("ndarray is not contiguous")
    return a


def cho_factor(C, L):
    """Cholesky factor of C into L (lower; strict upper = whatever C had there).
    ``L`` may be ``C`` (in place).  linalg_c.pyx:55-93."""
    C, L = _mat(C, "C"), _mat(L, "L")
    n = C.shape[0]
    if C.shape[1] != n:
        raise ValueError("C is not square")
    if L.shape[0] != n or L.shape[1] != n:
        raise ValueError("invalid shape for L")
    get_engine().cho_factor(C, L)  # raises LinAlgError when not positive definite
    return 0


def cho_solve_vec(L, b, x):
    """Solve (L L^T) x = b; ``x`` may be ``b``.  linalg_c.pyx:96-136."""
    L, b, x = _mat(L, "L"), _vec(b, "b"), _vec(x, "x")
    n = L.shape[0]
    if L.shape[1] != n:
        raise ValueError("L is not square")
    if b.shape[0] != n:
        raise ValueError("b has invalid size")
    if x.shape[0] != n:
        raise ValueError("x has invalid size")
    get_engine().cho_solve(L, b, x, 1)
    return 0


def cho_solve_mat(L, B, X):
    """Solve (L L^T) X = B for square B; ``X`` may be ``B``.  linalg_c.pyx:139-179."""
    L, B, X = _mat(L, "L"), _mat(B, "B"), _mat(X, "X")
    n = L.shape[0]
    if L.shape[1] != n:
        raise ValueError("L is not square")
    if B.shape[0] != n or B.shape[1] != n:
        raise ValueError("B has invalid shape")
    if X.shape[0] != n or X.shape[1] != n:
        raise ValueError("X has invalid shape")
    get_engine().cho_solve(L, B, X, n)
    return 0


def logdet(L):
    """2 sum log L_ii.  linalg_c.pyx:182-210."""
    L = _mat(L, "L")
    if L.shape[1] != L.shape[0]:
        raise ValueError("L is not square")
    return get_engine().logdet(L)


def dot11(x, y):
    x, y = _vec(x, "x"), _vec(y, "y")
    if y.shape[0] != x.shape[0]:
        raise ValueError("shape mismatch")
    return float(np.dot(x, y))


def dot12(x, Y, xY):
    x, Y, xY = _vec(x, "x"), _mat(Y, "Y"), _vec(xY, "xY")
    if Y.shape[0] != x.shape[0] or xY.shape[0] != Y.shape[1]:
        raise ValueError("shape mismatch")
    xY[:] = x.dot(Y)
    return 0


def dot21(X, y, Xy):
    X, y, Xy = _mat(X, "X"), _vec(y, "y"), _vec(Xy, "Xy")
    if y.shape[0] != X.shape[1] or Xy.shape[0] != X.shape[0]:
        raise ValueError("shape mismatch")
    Xy[:] = X.dot(y)
    return 0


def dot22(X, Y, XY):
    X, Y, XY = _mat(X, "X"), _mat(Y, "Y"), _mat(XY, "XY")
    if Y.shape[0] != X.shape[1] or XY.shape != (X.shape[0], Y.shape[1]):
        raise ValueError("shape mismatch")
    XY[:, :] = X.dot(Y)
    return 0


def vecdiff(x, y):
    """Euclidean distance.  linalg_c.pyx:373-408."""
    x, y = _vec(x, "x"), _vec(y, "y")
    if y.shape[0] != x.shape[0]:
        raise ValueError("shape mismatch")
    return float(np.sqrt(np.sum((x - y) ** 2)))
