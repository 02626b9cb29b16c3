"""CPU test double of ``bayesian_quadrature_amd.engine.Engine`` backed by the oracle.

TEST INFRASTRUCTURE ONLY.  It exists so that the CPU-only suite (``-m "not gpu"``)
can exercise the *host logic* of the product -- the ``BQ`` class, the ``gp`` memoisation
protocol, the ``linalg`` argument checking, pickling, the sharding helpers -- on a
machine without a GPU.  It is installed with ``engine.set_engine`` from tests only; the
product has no way to select it and never imports this file.  GPU parity is established
by ``test_gpu_parity.py`` / the ``gpu`` parameter of ``test_bq_object.py``, which run the
same assertions against the real HIP engine.
"""
import numpy as np


class FitDouble(object):
    def __init__(self, o, x, y, h, w, s):
        self.o = o
        self.x = np.asarray(x, dtype=np.float64)
        self.y = np.ascontiguousarray(y, dtype=np.float64)
        self.n = self.y.shape[0]
        self.d = 1 if self.x.ndim == 1 else self.x.shape[0]
        self.refit(h, w, s)

    def refit(self, h, w, s):
        self.h, self.w, self.s = float(h), np.atleast_1d(np.asarray(w, dtype=np.float64)), float(s)
        self._L, self._alpha, self.logml = self.o.gp_fit(self.x, self.y, self.h, self.w, self.s)

    def set_y(self, y):
        self.y = np.ascontiguousarray(y, dtype=np.float64)
        self._L = self._alpha = self.logml = None   # invalid until the next refit

    def refit_predict(self, h, w, s, xo):
        """The device engine's one-sweep route (bq_gp_refit_predict), here as its two halves."""
        self.refit(h, w, s)
        self.refit_predicts = getattr(self, "refit_predicts", 0) + 1
        m, v, _ = self.predict(xo)
        return m, v

    def close(self):
        pass

    def L(self):
        return self._L.copy(order="F")

    def alpha(self):
        return self._alpha.copy()

    def z(self):
        return self.o.trsm_lower(self._L, self.y)

    def solve(self, B):
        return self.o.cho_solve(self._L, B)

    def K(self):
        return self.o.gram(self.x, self.h, self.w, self.s)

    def predict(self, xo, want_mean=True, want_var=True, want_cov=False):
        xo = np.atleast_1d(np.asarray(xo, dtype=np.float64))
        mean, var = self.o.gp_predict(self.x, self.h, self.w, self._L, self._alpha, xo)
        cov = None
        if want_cov:
            cov = self.o.gp_cov(self.x, self.h, self.w, self._L, xo)
        return (mean if want_mean else None), (var if want_var else None), cov


MAX_LOG = float(np.log(np.exp2(np.float64(np.finfo(np.float64).maxexp - 4))))


class PairDouble(object):
    """The batched pair evaluators (engine.Pair) as plain loops over the oracle: one parameter
    set after the other, exactly the reference's order of operations (bq.py:933-957)."""

    def __init__(self, eng, x_s, tl_s, l_s, x_c, x_a, S):
        self.eng, self.o = eng, eng.o
        def f(a):
            if a is None:
                return np.empty(0)
            return np.ascontiguousarray(np.atleast_1d(a), dtype=np.float64)

        self.x_s, self.tl_s, self.l_s, self.x_c, self.x_a = f(x_s), f(tl_s), f(l_s), f(x_c), f(x_a)
        self.ns, self.nc, self.ma, self.S = len(self.x_s), len(self.x_c), len(self.x_a), int(S)
        self.x_sc = np.concatenate([self.x_s, self.x_c])

    def close(self):
        pass

    def _stage1(self, p):
        """GP1 under (h, w, s): (log-ML, l_c, mean / var at x_a, status)."""
        try:
            f1 = FitDouble(self.o, self.x_s, self.tl_s, p[0], p[1], p[2])
        except np.linalg.LinAlgError:
            return None, None, None, None, 1
        xo = np.concatenate([self.x_c, self.x_a])
        m, v, _ = f1.predict(xo) if xo.size else (np.empty(0), np.empty(0), None)
        mc, vc = m[:self.nc], np.maximum(v[:self.nc], 0.0)
        if ((mc + 2 * np.sqrt(vc)) > MAX_LOG).any():
            return None, None, None, None, 2
        return f1.logml, np.exp(mc), m[self.nc:], v[self.nc:], 0

    def llh(self, p_tl, p_l):
        llh = np.full(self.S, -np.inf)
        l_c = np.zeros((self.S, self.nc))
        status = np.zeros(self.S, dtype=np.int32)
        for b in range(self.S):
            lm1, lc, _, _, st = self._stage1(p_tl[b])
            if st == 0:
                l_c[b] = lc
                try:
                    f2 = FitDouble(self.o, self.x_sc, np.concatenate([self.l_s, lc]), *p_l[b])
                    llh[b] = lm1 + f2.logml
                except np.linalg.LinAlgError:
                    st = 3
            status[b] = st
        return llh, l_c, status

    def esm(self, p_tl, p_l, thresh, mu, cov):
        S, ma = self.S, self.ma
        out = {k: np.zeros((S, ma)) for k in ("A_a", "A_sc_l", "tm_a", "tC_a")}
        out["status"] = np.zeros((S, ma), dtype=np.int32)
        out["l_c"] = np.zeros((S, self.nc))
        out["sstatus"] = np.zeros(S, dtype=np.int32)
        for b in range(S):
            _, lc, tm, tC, st = self._stage1(p_tl[b])
            out["sstatus"][b] = st
            if st:
                continue
            out["l_c"][b], out["tm_a"][b], out["tC_a"][b] = lc, tm, tC
            A_a, A_sc_l, status = self.eng.esm_batch(
                self.x_sc, np.concatenate([self.l_s, lc]), self.ns, self.x_a, p_l[b][0],
                p_l[b][1], thresh, mu, cov)
            out["A_a"][b], out["A_sc_l"][b], out["status"][b] = A_a, A_sc_l, status
        return out


class EngineDouble(object):
    device = 0

    def pair(self, x_s, tl_s, l_s, x_c, x_a, S):
        return PairDouble(self, x_s, tl_s, l_s, x_c, x_a, S)

    def __init__(self, oracle):
        self.o = oracle

    def cho_factor(self, Cm, Lm):
        L = self.o.cho_factor(Cm)           # raises LinAlgError like the device engine
        upper = np.triu(Cm, 1)
        Lm[:, :] = L + upper

    def cho_solve(self, Lm, B, X, nrhs):
        X[...] = self.o.cho_solve(np.tril(Lm), B)

    def logdet(self, Lm):
        return self.o.logdet(Lm)

    def gram(self, x, h, w, s=0.0):
        w = np.atleast_1d(np.asarray(w, dtype=np.float64))
        if (w <= 0).any():
            raise ValueError("w must be positive and finite")
        return self.o.gram(x, h, w, s)

    def gram_cross(self, x1, x2, h, w):
        return self.o.gram_cross(x1, x2, h, w)

    def gp_fit(self, x, y, h, w, s=0.0):
        return FitDouble(self.o, x, y, h, w, s)

    def int_K(self, x, h, w, mu, cov):
        return self.o.int_K(x, h, w, mu, cov)

    def int_K1_K2(self, x1, x2, h1, w1, h2, w2, mu, cov):
        return self.o.int_K1_K2(x1, x2, h1, w1, h2, w2, mu, cov)

    def int_int_K1_K2_K1(self, x, h1, w1, h2, w2, mu, cov):
        return self.o.int_int_K1_K2_K1(x, h1, w1, h2, w2, mu, cov)

    def int_int_K1_K2(self, x, h1, w1, h2, w2, mu, cov):
        return self.o.int_int_K1_K2(x, h1, w1, h2, w2, mu, cov)

    def esm_batch(self, x_sc, l_sc, ns, x_a, h, w, thresh, mu, cov):
        """The reference's per-candidate recipe (bq.py:462-480, bq_c.pyx:425-455)."""
        x_sc = np.asarray(x_sc, dtype=np.float64)
        nsc = x_sc.shape[0]
        M = len(x_a)
        A_a, A_sc_l = np.empty(M), np.empty(M)
        status = np.zeros(M, dtype=np.int32)
        for k in range(M):
            x_sca = np.concatenate([x_sc, [x_a[k]]])
            K = np.ascontiguousarray(self.o.gram_cross(x_sca, x_sca, h, w))
            jitter = np.zeros(nsc + 1)
            close = np.abs(x_sc[ns:] - x_a[k]) < thresh
            if close.any():
                self.o.improve_covariance_conditioning(K, jitter, np.nonzero(close)[0] + ns)
            self.o.improve_covariance_conditioning(K, jitter, np.array([nsc]))
            try:
                L = self.o.cho_factor(K)
            except np.linalg.LinAlgError:
                status[k] = 1
                A_a[k] = A_sc_l[k] = np.nan
                continue
            A = self.o.cho_solve(L, self.o.int_K(x_sca, h, w, mu, cov))
            A_a[k] = A[nsc]
            A_sc_l[k] = A[:nsc].dot(l_sc)
        return A_a, A_sc_l, status

    def esm_border(self, fit_l, ns, x_a, thresh, mu, cov):
        """Same interface as the device engine's bordered update of gp_l's resident factor;
        the double just applies the reference's recipe to the fit's own data."""
        if fit_l.s != 0.0:
            raise ValueError("esm_border: gp_l carries a noise term")
        return self.esm_batch(fit_l.x, fit_l.y, ns, x_a, fit_l.h, float(fit_l.w[0]), thresh, mu,
                              cov)

    def Z_mean(self, fit_l, mu, cov):
        return self.o.Z_mean(fit_l.x, fit_l._alpha, fit_l.h, fit_l.w, mu, cov)

    def Z_var(self, fit_tl, fit_l, mu, cov):
        return self.o.Z_var(fit_tl.x, fit_l.x, fit_l._alpha, fit_tl._L, fit_l.h, fit_l.w,
                            fit_tl.h, fit_tl.w, mu, cov)

    def batch_fit_predict(self, x, y, h, w, s, xo):
        P = len(x)
        M = np.asarray(xo).shape[-1]
        mean, var = np.empty((P, M)), np.empty((P, M))
        logml, status = np.empty(P), np.zeros(P, dtype=np.int32)
        for p in range(P):
            try:
                f = FitDouble(self.o, x[p], y[p], h, w, s)
                mean[p], var[p], _ = f.predict(xo[p])
                logml[p] = f.logml
            except np.linalg.LinAlgError:
                status[p], logml[p] = 1, -np.inf
        return mean, var, logml, status

    def logml_grid(self, x, y, h, w, s=0.0, chunk=0):
        out = np.empty(len(h))
        for g in range(len(h)):
            try:
                out[g] = self.o.gp_fit(x, y, h[g], np.atleast_1d(w[g]), s)[2]
            except np.linalg.LinAlgError:
                out[g] = -np.inf
        return out
