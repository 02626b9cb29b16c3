"""ctypes binding of oracle/libbq_oracle.so (TEST INFRASTRUCTURE ONLY).

The C file restates, on the CPU, the algorithms of the reference hot path
(see the header of bq_oracle.c for the file:line map).  This module only
marshals numpy arrays; there is no arithmetic here.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libbq_oracle.so")

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int64)


def build(force=False):
    """Compile libbq_oracle.so with the committed Makefile (gcc)."""
    src = os.path.join(_HERE, "bq_oracle.c")
    if (not force and os.path.exists(_SO)
            and os.path.getmtime(_SO) >= os.path.getmtime(src)):
        return _SO
    subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libbq_oracle.so"])
    return _SO


def _f(a):
    """float64, Fortran-contiguous view/copy."""
    return np.asfortranarray(np.asarray(a, dtype=np.float64))


def _p(a):
    return a.ctypes.data_as(_dp)


def _pts(x):
    """Points as the reference stores them: d x n, column-major."""
    x = np.asarray(x, dtype=np.float64)
    if x.ndim == 1:
        x = x[None, :]
    return np.asfortranarray(x)


def _vec(w, d=None):
    w = np.atleast_1d(np.asarray(w, dtype=np.float64)).copy()
    if d is not None and w.shape[0] != d:
        raise ValueError("length-d vector expected")
    return w


class Oracle(object):
    def __init__(self, path=None):
        self.lib = lib = C.CDLL(path or build())
        d, i = C.c_double, C.c_int
        sig = {
            "bqo_set_threads": (None, [i]),
            "bqo_get_threads": (i, []),
            "bqo_max_threads": (i, []),
            "bqo_potf2": (i, [_dp, i, i]),
            "bqo_potrf": (i, [_dp, i, i, i]),
            "bqo_cho_factor": (i, [_dp, _dp, i]),
            "bqo_potrs": (None, [_dp, i, i, _dp, i, i]),
            "bqo_trsm_lower": (None, [_dp, i, i, _dp, i, i]),
            "bqo_logdet": (d, [_dp, i]),
            "bqo_dot11": (d, [_dp, _dp, i]),
            "bqo_vecdiff": (d, [_dp, _dp, i]),
            "bqo_kernel_scale": (d, [i, d, _dp]),
            "bqo_gram_gauss_cross": (None, [_dp, i, _dp, i, i, d, _dp, _dp]),
            "bqo_gram_gauss": (None, [_dp, i, i, d, _dp, d, _dp]),
            "bqo_gp_fit": (i, [_dp, _dp, i, i, d, _dp, d, _dp, _dp, _dp]),
            "bqo_gp_predict": (None, [_dp, i, i, d, _dp, _dp, _dp, _dp, i, _dp, _dp, _dp]),
            "bqo_gp_cov": (None, [_dp, i, i, d, _dp, _dp, _dp, i, _dp, _dp]),
            "bqo_mvn_logpdf": (d, [_dp, _dp, _dp, d, i]),
            "bqo_int_exp_norm": (d, [d, d, d]),
            "bqo_int_K": (i, [_dp, _dp, i, i, d, _dp, _dp, _dp]),
            "bqo_int_K1_K2": (i, [_dp, _dp, i, _dp, i, i, d, _dp, d, _dp, _dp, _dp]),
            "bqo_int_int_K1_K2_K1": (i, [_dp, _dp, i, i, d, _dp, d, _dp, _dp, _dp]),
            "bqo_int_int_K1_K2": (i, [_dp, _dp, i, i, d, _dp, d, _dp, _dp, _dp]),
            "bqo_int_int_K": (d, [i, d, _dp, _dp, _dp]),
            "bqo_p_x_gaussian": (i, [_dp, _dp, i, i, _dp, _dp]),
            "bqo_Z_mean": (d, [_dp, i, i, _dp, d, _dp, _dp, _dp]),
            "bqo_Z_var": (d, [_dp, i, _dp, i, i, _dp, _dp, d, _dp, d, _dp, _dp, _dp]),
            "bqo_esm_and_em": (i, [_dp, _dp, _dp, d, d, _dp, i, i, d, _dp, _dp, _dp]),
            "bqo_filter_candidates": (None, [_dp, i, _dp, i, d]),
            "bqo_improve_covariance_conditioning": (None, [_dp, i, _dp, _ip, i]),
            "bqo_max_exp_arg": (d, []),
        }
        for name, (res, args) in sig.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args

    # -- threads --------------------------------------------------------
    def set_threads(self, t):
        self.lib.bqo_set_threads(int(t))

    def max_threads(self):
        return int(self.lib.bqo_max_threads())

    # -- linalg_c -------------------------------------------------------
    def cho_factor(self, A, nb=64, unblocked=False):
        """Lower Cholesky factor (strict upper zeroed for convenience).
        Raises numpy.linalg.LinAlgError like linalg_c.pyx:90-91."""
        L = _f(A).copy(order="F")
        n = L.shape[0]
        if L.ndim != 2 or L.shape[1] != n:
            raise ValueError("C is not square")
        if unblocked:
            info = self.lib.bqo_potf2(_p(L), n, n)
        else:
            info = self.lib.bqo_potrf(_p(L), n, n, nb)
        if info > 0:
            raise np.linalg.LinAlgError("matrix is not positive definite")
        return np.asfortranarray(np.tril(L))

    def cho_solve(self, L, B):
        L = _f(L)
        X = _f(B).copy(order="F")
        n = L.shape[0]
        nrhs = 1 if X.ndim == 1 else X.shape[1]
        if X.shape[0] != n:
            raise ValueError("b has invalid size")
        self.lib.bqo_potrs(_p(L), n, n, _p(X), nrhs, n)
        return X

    def trsm_lower(self, L, B):
        L = _f(L)
        X = _f(B).copy(order="F")
        n = L.shape[0]
        nrhs = 1 if X.ndim == 1 else X.shape[1]
        self.lib.bqo_trsm_lower(_p(L), n, n, _p(X), nrhs, n)
        return X

    def logdet(self, L):
        L = _f(L)
        return float(self.lib.bqo_logdet(_p(L), L.shape[0]))

    def dot11(self, x, y):
        x, y = _vec(x), _vec(y)
        return float(self.lib.bqo_dot11(_p(x), _p(y), x.shape[0]))

    def vecdiff(self, x, y):
        x, y = _vec(x), _vec(y)
        return float(self.lib.bqo_vecdiff(_p(x), _p(y), x.shape[0]))

    # -- gp restatement -------------------------------------------------
    def kernel_scale(self, d, h, w):
        w = _vec(w, d)
        return float(self.lib.bqo_kernel_scale(d, float(h), _p(w)))

    def gram_cross(self, x1, x2, h, w):
        x1, x2 = _pts(x1), _pts(x2)
        d = x1.shape[0]
        w = _vec(w, d)
        K = np.empty((x1.shape[1], x2.shape[1]), order="F")
        self.lib.bqo_gram_gauss_cross(_p(x1), x1.shape[1], _p(x2), x2.shape[1], d,
                                      float(h), _p(w), _p(K))
        return K

    def gram(self, x, h, w, s=0.0):
        x = _pts(x)
        d, n = x.shape
        w = _vec(w, d)
        K = np.empty((n, n), order="F")
        self.lib.bqo_gram_gauss(_p(x), n, d, float(h), _p(w), float(s), _p(K))
        return K

    def gp_fit(self, x, y, h, w, s=0.0):
        """Returns (L, alpha, logml)."""
        x = _pts(x)
        d, n = x.shape
        w = _vec(w, d)
        y = _vec(y, n)
        L = np.empty((n, n), order="F")
        alpha = np.empty(n)
        logml = C.c_double()
        info = self.lib.bqo_gp_fit(_p(x), _p(y), d, n, float(h), _p(w), float(s), _p(L),
                                   _p(alpha), C.cast(C.byref(logml), _dp))
        if info > 0:
            raise np.linalg.LinAlgError("matrix is not positive definite")
        return np.asfortranarray(np.tril(L)), alpha, float(logml.value)

    def gp_predict(self, x, h, w, L, alpha, xo, want_var=True):
        x, xo = _pts(x), _pts(xo)
        d, n = x.shape
        M = xo.shape[1]
        w = _vec(w, d)
        L = _f(L)
        alpha = _vec(alpha, n)
        mean = np.empty(M)
        var = np.empty(M) if want_var else None
        work = np.empty((n, M), order="F")
        self.lib.bqo_gp_predict(_p(x), d, n, float(h), _p(w), _p(L), _p(alpha), _p(xo), M,
                                _p(mean), _p(var) if want_var else None, _p(work))
        return (mean, var) if want_var else mean

    def gp_cov(self, x, h, w, L, xo):
        """Full posterior covariance K(xo,xo) - K(xo,x) Kxx^-1 K(x,xo), M x M (gp.GP.cov)."""
        x, xo = _pts(x), _pts(xo)
        d, n = x.shape
        M = xo.shape[1]
        w = _vec(w, d)
        L = _f(L)
        cov = np.empty((M, M), order="F")
        work = np.empty((n, M), order="F")
        self.lib.bqo_gp_cov(_p(x), d, n, float(h), _p(w), _p(L), _p(xo), M, _p(cov), _p(work))
        return cov

    # -- gauss_c --------------------------------------------------------
    def mvn_logpdf(self, x, m, L, logdet):
        x, m, L = _vec(x), _vec(m), _f(L)
        return float(self.lib.bqo_mvn_logpdf(_p(x), _p(m), _p(L), float(logdet), x.shape[0]))

    def int_exp_norm(self, c, m, S):
        return float(self.lib.bqo_int_exp_norm(float(c), float(m), float(S)))

    def _mc(self, d, w, mu, cov):
        return _vec(w, d), _vec(mu, d), _f(np.atleast_2d(cov))

    def int_K(self, x, h, w, mu, cov):
        x = _pts(x)
        d, n = x.shape
        w, mu, cov = self._mc(d, w, mu, cov)
        out = np.empty(n)
        info = self.lib.bqo_int_K(_p(out), _p(x), d, n, float(h), _p(w), _p(mu), _p(cov))
        if info > 0:
            raise np.linalg.LinAlgError("matrix is not positive definite")
        return out

    def int_K1_K2(self, x1, x2, h1, w1, h2, w2, mu, cov):
        x1, x2 = _pts(x1), _pts(x2)
        d = x1.shape[0]
        w1, mu, cov = self._mc(d, w1, mu, cov)
        w2 = _vec(w2, d)
        out = np.empty((x1.shape[1], x2.shape[1]), order="F")
        info = self.lib.bqo_int_K1_K2(_p(out), _p(x1), x1.shape[1], _p(x2), x2.shape[1], d,
                                      float(h1), _p(w1), float(h2), _p(w2), _p(mu), _p(cov))
        if info > 0:
            raise np.linalg.LinAlgError("matrix is not positive definite")
        return out

    def int_int_K1_K2_K1(self, x, h1, w1, h2, w2, mu, cov):
        x = _pts(x)
        d, n = x.shape
        w1, mu, cov = self._mc(d, w1, mu, cov)
        w2 = _vec(w2, d)
        out = np.empty((n, n), order="F")
        info = self.lib.bqo_int_int_K1_K2_K1(_p(out), _p(x), d, n, float(h1), _p(w1), float(h2),
                                             _p(w2), _p(mu), _p(cov))
        if info > 0:
            raise np.linalg.LinAlgError("matrix is not positive definite")
        return out

    def int_int_K1_K2(self, x, h1, w1, h2, w2, mu, cov):
        x = _pts(x)
        d, n = x.shape
        w1, mu, cov = self._mc(d, w1, mu, cov)
        w2 = _vec(w2, d)
        out = np.empty(n)
        info = self.lib.bqo_int_int_K1_K2(_p(out), _p(x), d, n, float(h1), _p(w1), float(h2),
                                          _p(w2), _p(mu), _p(cov))
        if info > 0:
            raise np.linalg.LinAlgError("matrix is not positive definite")
        return out

    def int_int_K(self, d, h, w, mu, cov):
        w, mu, cov = self._mc(d, w, mu, cov)
        return float(self.lib.bqo_int_int_K(d, float(h), _p(w), _p(mu), _p(cov)))

    # -- bq_c -----------------------------------------------------------
    def p_x_gaussian(self, x, mu, cov):
        x = _pts(x)
        d, n = x.shape
        mu, cov = _vec(mu, d), _f(np.atleast_2d(cov))
        p = np.empty(n)
        self.lib.bqo_p_x_gaussian(_p(p), _p(x), d, n, _p(mu), _p(cov))
        return p

    def Z_mean(self, x_sc, alpha_l, h_l, w_l, mu, cov):
        x_sc = _pts(x_sc)
        d, n = x_sc.shape
        w_l, mu, cov = self._mc(d, w_l, mu, cov)
        a = _vec(alpha_l, n)
        return float(self.lib.bqo_Z_mean(_p(x_sc), d, n, _p(a), float(h_l), _p(w_l), _p(mu),
                                         _p(cov)))

    def Z_var(self, x_s, x_sc, alpha_l, L_tl, h_l, w_l, h_tl, w_tl, mu, cov):
        x_s, x_sc = _pts(x_s), _pts(x_sc)
        d, ns = x_s.shape
        nsc = x_sc.shape[1]
        w_l, mu, cov = self._mc(d, w_l, mu, cov)
        w_tl = _vec(w_tl, d)
        a = _vec(alpha_l, nsc)
        L_tl = _f(L_tl)
        return float(self.lib.bqo_Z_var(_p(x_s), ns, _p(x_sc), nsc, d, _p(a), _p(L_tl),
                                        float(h_l), _p(w_l), float(h_tl), _p(w_tl), _p(mu),
                                        _p(cov)))

    def esm_and_em(self, l_sc, L_l, tm_a, tC_a, x_sca, h_l, w_l, mu, cov):
        x_sca = _pts(x_sca)
        d, nca = x_sca.shape
        w_l, mu, cov = self._mc(d, w_l, mu, cov)
        l_sc = _vec(l_sc, nca - 1)
        L_l = _f(L_l)
        out = np.empty(2)
        self.lib.bqo_esm_and_em(_p(out), _p(l_sc), _p(L_l), float(tm_a), float(tC_a),
                                _p(x_sca), d, nca, float(h_l), _p(w_l), _p(mu), _p(cov))
        return float(out[0]), float(out[1])

    def filter_candidates(self, x_c, x_s, thresh):
        """In place on x_c (must be a contiguous float64 array)."""
        if x_c.dtype != np.float64 or not x_c.flags.c_contiguous:
            raise ValueError("x_c must be contiguous float64")
        x_s = _vec(x_s)
        self.lib.bqo_filter_candidates(_p(x_c), x_c.shape[0], _p(x_s), x_s.shape[0],
                                       float(thresh))

    def improve_covariance_conditioning(self, M, jitters, idx):
        idx = np.ascontiguousarray(idx, dtype=np.int64)
        self.lib.bqo_improve_covariance_conditioning(
            _p(M), M.shape[0], _p(jitters), idx.ctypes.data_as(_ip), idx.shape[0])


_singleton = None


def load():
    global _singleton
    if _singleton is None:
        _singleton = Oracle()
    return _singleton
