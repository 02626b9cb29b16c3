"""Bayesian-quadrature moments from GP outputs.

Mirror of the exact half of the reference's ``bq_c`` module
(bq_c.pyx:63-97,127-213,264-355,425-535,601-649): same names and argument order.
The heavy inputs -- ``alpha = K^-1 y`` and the Cholesky factor ``L`` -- come from the
device engine; the linear solves here go through the ``linalg`` drop-ins, i.e. the
GPU.  The trapezoid ``approx_*`` functions and the von Mises helpers are out of
scope (periodic / non-Gaussian kernels only, bq.py:125,1014-1018).
"""
from warnings import warn

import numpy as np

from . import gauss as ga
from . import linalg as la

EPS = float(np.finfo(np.float64).eps)
MIN = float(np.log(np.exp2(np.float64(np.finfo(np.float64).minexp + 4))))


def p_x_gaussian(p_x, x, mu, cov):
    """p_x_i = N(x_i | mu, cov); x is d x n (bq_c.pyx:63-97)."""
    d, n = x.shape
    if p_x.shape[0] != n:
        raise ValueError("p_x has invalid shape")
    if mu.shape[0] != d:
        raise ValueError("mu has invalid shape")
    if cov.shape[0] != d or cov.shape[1] != d:
        raise ValueError("cov has invalid shape")
    L = ga._chol(np.asarray(cov, dtype=np.float64))
    p_x[:] = np.exp(ga._logpdf_cols(x - np.asarray(mu)[:, None], L))


def improve_covariance_conditioning(M, jitters, idx):
    """Add ``max(eps, max(M)) * 1e-4`` to M[i, i] and jitters[i] for i in idx, in
    place (bq_c.pyx:127-140)."""
    sqd_jitter = max(EPS, float(np.max(M))) * 1e-4
    for i in idx:
        jitters[i] += sqd_jitter
        M[i, i] += sqd_jitter


def remove_jitter(M, jitters, idx):
    """Undo improve_covariance_conditioning for the indices idx (bq_c.pyx:143-154)."""
    for i in idx:
        M[i, i] -= jitters[i]
        jitters[i] = 0


def Z_mean(x_sc, alpha_l, h_l, w_l, mu, cov):
    """E[Z] = (int K_l(x, x_sc) p(x) dx) . alpha_l   (bq_c.pyx:157-213)."""
    d, nc = x_sc.shape
    if alpha_l.shape[0] != nc:
        raise ValueError("alpha_l has invalid shape")
    int_K_l = np.empty(nc)
    ga.int_K(int_K_l, x_sc, h_l, w_l, mu, cov)
    m_Z = float(np.dot(int_K_l, alpha_l))
    if m_Z <= 0:
        warn("m_Z = %s" % m_Z)
    return m_Z


def Z_var(x_s, x_sc, alpha_l, L_tl, h_l, w_l, h_tl, w_tl, mu, cov):
    """V(Z) = alpha' (int int K_l K_tl K_l) alpha - beta' K_tl^-1 beta with
    beta = (int K_tl K_l) alpha   (bq_c.pyx:264-355)."""
    d, ns = x_s.shape
    nc = x_sc.shape[1]
    if x_sc.shape[0] != d:
        raise ValueError("x_s has invalid shape")
    if alpha_l.shape[0] != nc:
        raise ValueError("alpha_l has invalid shape")
    if L_tl.shape[0] != ns or L_tl.shape[1] != ns:
        raise ValueError("L_tl has invalid shape")
    I3 = np.empty((nc, nc), order="F")
    ga.int_int_K1_K2_K1(I3, x_sc, h_l, w_l, h_tl, w_tl, mu, cov)
    alpha_int_alpha = float(alpha_l.dot(I3).dot(alpha_l))
    I2 = np.empty((ns, nc), order="F")
    ga.int_K1_K2(I2, x_s, x_sc, h_tl, w_tl, h_l, w_l, mu, cov)
    beta = np.ascontiguousarray(I2.dot(alpha_l))
    L_tl_beta = np.empty(ns)
    la.cho_solve_vec(np.asfortranarray(L_tl), beta, L_tl_beta)  # full K^-1 beta (bq_c.pyx:348)
    V_Z = alpha_int_alpha - float(beta.dot(L_tl_beta))
    if V_Z <= 0:
        warn("V_Z = %s" % V_Z)
    return V_Z


def expected_squared_mean_and_mean(l_sc, K_l, tm_a, tC_a, x_sca, h_l, w_l, mu, cov):
    """(E[m(Z)^2 | x_a], E[m(Z) | x_a]); ``K_l`` is the Cholesky factor of the
    bordered, jittered Gram on x_sca (the reference passes ``L`` under this name,
    bq.py:507-512; bq_c.pyx:425-535)."""
    n = x_sca.shape[1]
    if K_l.shape[0] != n or K_l.shape[1] != n:
        raise ValueError("L_l is not square")
    if l_sc.shape[0] != n - 1:
        raise ValueError("l_sc has invalid shape")
    int_K_l = np.empty(n)
    ga.int_K(int_K_l, x_sca, h_l, w_l, mu, cov)
    A_sca = np.empty(n)
    la.cho_solve_vec(np.asfortranarray(K_l), int_K_l, A_sca)
    A_a = A_sca[n - 1]
    A_sc_l = float(np.dot(A_sca[:n - 1], l_sc))
    tm_a, tC_a = float(np.ravel(tm_a)[0]), float(np.ravel(tC_a)[0])
    e1 = ga.int_exp_norm(1, tm_a, tC_a)
    if np.isinf(e1):
        return (float("inf"), float("inf"))
    E_m = A_sc_l + A_a * e1
    e2 = ga.int_exp_norm(2, tm_a, tC_a)
    if np.isinf(e2):
        return (float("inf"), E_m)
    E_m2 = (A_sc_l ** 2) + (2 * A_sc_l * A_a * e1) + (A_a ** 2 * e2)
    return (E_m2, E_m)


def filter_candidates(x_c, x_s, thresh):
    """Merge candidates closer than ``thresh`` to each other (replace by their
    average, repeat until stable) and drop those closer than ``thresh`` to an
    observation; rejected entries become NaN, in place (bq_c.pyx:601-649)."""
    nc = x_c.shape[0]
    done = False
    while not done:
        done = True
        for i in range(nc):
            if np.isnan(x_c[i]):
                continue
            for j in range(i + 1, nc):
                if np.isnan(x_c[j]):
                    continue
                if abs(x_c[i] - x_c[j]) < thresh:
                    x_c[i] = (x_c[i] + x_c[j]) / 2.0
                    x_c[j] = np.nan
                    done = False
    if nc and x_s.shape[0]:
        close = (np.abs(x_c[:, None] - x_s[None, :]) < thresh).any(axis=1)
        x_c[close & ~np.isnan(x_c)] = np.nan
