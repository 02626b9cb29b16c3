"""Closed-form Gaussian-kernel integrals against a Gaussian prior.

Mirror of the exact half of the reference's ``gauss_c`` module
(gauss_c.pyx:20-164,235-339,416-531,617-713,796-855): same names, same argument
order (caller-allocated output first), points ``d x n``, ``w``/``mu`` length-d,
``cov`` d x d.  Every result is "scale x exp(log-pdf of a small Gaussian)" over n or
n^2 points: the d x d algebra is done once on the host side of the C ABI, the
per-point / per-pair work runs in HIP kernels.  The scalar
helpers (``mvn_logpdf``, ``int_exp_norm``, ``int_int_K``) stay on the host.

The trapezoid ``approx_*`` twins are out of scope (non-Gaussian kernels only).
"""
import numpy as np

LOG_2PI = float(np.log(2.0 * np.pi))
#: largest argument for which exp() is evaluated, gauss_c.pyx:16
MAX = float(np.log(np.exp2(np.float64(np.finfo(np.float64).maxexp - 4))))


def _pts(x, name):
    if not isinstance(x, np.ndarray) or x.ndim != 2:
        raise ValueError("%s has invalid shape" % name)
    return x


def _chol(C):
    try:
        return np.linalg.cholesky(C)
    except np.linalg.LinAlgError:
        raise np.linalg.LinAlgError("matrix is not positive definite")


def _logpdf_cols(D, L):
    """log N(0 | ., L L^T) for every column of the d x n difference matrix D."""
    d = D.shape[0]
    if d == 1:
        z = D[0] / L[0, 0]
        maha = z * z
        logdet = 2.0 * np.log(L[0, 0])
    else:
        from scipy.linalg import solve_triangular
        Z = solve_triangular(L, D, lower=True)
        maha = np.einsum("ij,ij->j", Z, Z)
        logdet = 2.0 * np.sum(np.log(np.diag(L)))
    return -0.5 * (d * LOG_2PI + logdet + maha)


def _check(d, w, mu, cov, wname="w"):
    if w.shape[0] != d:
        raise ValueError("%s has invalid shape" % wname)
    if mu.shape[0] != d:
        raise ValueError("mu has invalid shape")
    if cov.shape[0] != d or cov.shape[1] != d:
        raise ValueError("cov has invalid shape")


def mvn_logpdf(x, m, L, logdet):
    """log N(x | m, L L^T) for one point; ``logdet`` = log|L L^T| (gauss_c.pyx:20-62)."""
    x, m = np.asarray(x, dtype=np.float64), np.asarray(m, dtype=np.float64)
    d = x.shape[0]
    if m.shape[0] != d:
        raise ValueError("m has invalid size")
    if L.shape[0] != d or L.shape[1] != d:
        raise ValueError("C has invalid size")
    diff = x - m
    if d == 1:
        maha = diff[0] * diff[0] / (L[0, 0] * L[0, 0])
    else:
        from scipy.linalg import cho_solve
        maha = float(diff.dot(cho_solve((np.tril(L), True), diff)))
    return float(-0.5 * (LOG_2PI * d + logdet + maha))


def int_exp_norm(c, m, S):
    """int exp(c x) N(x | m, S) dx = exp(c m + c^2 S / 2), saturating to inf above
    MAX (gauss_c.pyx:65-92)."""
    out = (c * m) + (0.5 * c ** 2 * S)
    if out > MAX:
        return float("inf")
    return float(np.exp(out))


def _eng():
    from .engine import get_engine
    return get_engine()


def int_K(out, x, h, w, mu, cov):
    """out_i = h^2 N(x_i | mu, diag(w^2) + cov)   (gauss_c.pyx:95-164)."""
    x = _pts(x, "x")
    d, n = x.shape
    if out.shape[0] != n:
        raise ValueError("out has invalid shape")
    _check(d, w, mu, cov)
    out[:] = _eng().int_K(x, h, w, mu, cov)
    return 0


def int_K1_K2(out, x1, x2, h1, w1, h2, w2, mu, cov):
    """out_ij = h1^2 h2^2 N([x1_i, x2_j] | [mu, mu], [[W1+S, S], [S, W2+S]])
    (gauss_c.pyx:235-339)."""
    x1, x2 = _pts(x1, "x1"), _pts(x2, "x2")
    d, n1 = x1.shape
    n2 = x2.shape[1]
    if out.shape[0] != n1 or out.shape[1] != n2:
        raise ValueError("out has invalid shape")
    if x2.shape[0] != d:
        raise ValueError("x2 has invalid shape")
    _check(d, w1, mu, cov, "w1")
    if w2.shape[0] != d:
        raise ValueError("w2 has invalid shape")
    out[:, :] = _eng().int_K1_K2(x1, x2, h1, w1, h2, w2, mu, cov)
    return 0


def int_int_K1_K2_K1(out, x, h1, w1, h2, w2, mu, cov):
    """Double integral of K1 K2 K1 against the prior twice (gauss_c.pyx:416-531):
    out_ij = h1^4 h2^2 N(x_i|mu,W1+S) N(x_j|mu,W1+S) N(G x_i | G x_j, W2 + 2S - 2 G S),
    G = S (W1 + S)^-1."""
    x = _pts(x, "x")
    d, n = x.shape
    if out.shape[0] != n or out.shape[1] != n:
        raise ValueError("out has invalid shape")
    _check(d, w1, mu, cov, "w1")
    if w2.shape[0] != d:
        raise ValueError("w2 has invalid shape")
    out[:, :] = _eng().int_int_K1_K2_K1(x, h1, w1, h2, w2, mu, cov)
    return 0


def int_int_K1_K2(out, x, h1, w1, h2, w2, mu, cov):
    """out_i = h1^2 h2^2 N(0|0, W1+2S) N(x_i | mu, W2 + S - S (W1+2S)^-1 S)
    (gauss_c.pyx:617-713)."""
    x = _pts(x, "x")
    d, n = x.shape
    if out.shape[0] != n:
        raise ValueError("out has invalid shape")
    _check(d, w1, mu, cov, "w1")
    if w2.shape[0] != d:
        raise ValueError("w2 has invalid shape")
    out[:] = _eng().int_int_K1_K2(x, h1, w1, h2, w2, mu, cov)
    return 0


def int_int_K(d, h, w, mu, cov):
    """h^2 N(0 | 0, diag(w^2) + 2 S)   (gauss_c.pyx:796-855)."""
    w, mu, cov = np.asarray(w), np.asarray(mu), np.asarray(cov)
    _check(d, w, mu, cov)
    W = 2 * cov + np.diag(w ** 2)
    return float((h ** 2) * np.exp(_logpdf_cols(np.zeros((d, 1)), _chol(W))[0]))
