"""``BQ``: Bayesian quadrature of Z = int l(x) N(x | mu, sigma^2) dx with two stacked
GPs (one over log l, one over exp(log l)), after Osborne et al. 2012.

Drop-in for ``bayesian_quadrature.BQ`` (reference bq.py:19-1026): same constructor,
options, method names, return values and exceptions.  Python 3; the GP fits, posterior
evaluations, log marginal likelihoods and Cholesky solves run on the MI355X through the
C ABI (``gp.py`` / ``linalg.py``).  Only the exact (Gaussian-kernel) branch exists: the
trapezoid ``approx`` branch and the periodic kernel are outside the accelerated path
 and raise ``NotImplementedError``; plotting helpers
are likewise not part of this package.
"""
import logging
from copy import copy, deepcopy
from warnings import warn

import numpy as np

from . import bq_c
from . import linalg as la
from . import util
from .engine import get_engine
from .gp import GP, GaussianKernel, PeriodicKernel

logger = logging.getLogger("bayesian_quadrature")
DTYPE = np.dtype("float64")
MIN = float(np.log(np.exp2(np.float64(np.finfo(np.float64).minexp + 4))))
MAX = float(np.log(np.exp2(np.float64(np.finfo(np.float64).maxexp - 4))))

_STATE_ALWAYS = ("x_s", "l_s", "tl_s", "options", "initialized")


def _row(x):
    """1-D points as the d x n Fortran array the integral helpers expect (bq.py:281)."""
    return np.array(np.asarray(x, dtype=DTYPE)[None], order="F")


class BQ(object):
    """Estimate Z = int l(x) N(x | x_mean, x_var) dx from samples (x, l)."""

    # ------------------------------------------------------------------ setup
    def __init__(self, x, l, **options):
        self.x_s = np.array(x, dtype=DTYPE)
        self.l_s = np.array(l, dtype=DTYPE)
        if (self.l_s <= 0).any():
            raise ValueError("l_s contains zero or negative values")
        if self.x_s.ndim > 1:
            raise ValueError("invalid number of dimensions for x")
        if self.l_s.ndim > 1:
            raise ValueError("invalid number of dimensions for l")
        if self.x_s.shape != self.l_s.shape:
            raise ValueError("shape mismatch for x and l")
        self.tl_s = np.log(self.l_s)
        self.ns = self.x_s.shape[0]
        self.load_options(**options)
        self.initialized = False
        self._clear_fit_state()

    def _clear_fit_state(self):
        self.gp_log_l = None   # GP over log(l)
        self.gp_l = None       # GP over exp(log(l))
        self.x_c = self.l_c = self.nc = None      # candidate points
        self.x_sc = self.l_sc = self.nsc = None   # samples followed by candidates
        self._approx_x = None
        self._approx_px = None
        self._drop_pairs()

    def _drop_pairs(self):
        """The resident batched evaluators (engine.Pair) belong to one set of samples and
        candidates: dropped whenever those change."""
        for p in getattr(self, "_pairs", {}).values():
            try:
                p.close()
            except Exception:
                pass
        self._pairs = {}

    def _pair(self, S, x_a=None):
        x_a = np.empty(0) if x_a is None else np.ascontiguousarray(x_a, dtype=DTYPE)
        key = (int(S), x_a.tobytes())
        if key not in self._pairs:
            if len(self._pairs) >= 4:   # a few shapes at most stay resident
                self._drop_pairs()
            self._pairs[key] = get_engine().pair(self.x_s, self.tl_s, self.l_s, self.x_c, x_a, S)
        return self._pairs[key]

    def load_options(self, kernel, n_candidate, candidate_thresh, x_mean, x_var, optim_method):
        """All six options are required (bq.py:94).  ``kernel`` is the kernel class,
        ``x_mean`` / ``x_var`` the Gaussian prior over x."""
        self.options = {
            "kernel": kernel,
            "n_candidate": int(n_candidate),
            "candidate_thresh": float(candidate_thresh),
            "x_mean": np.array([x_mean], dtype=DTYPE, order="F"),
            "x_cov": np.array([[x_var]], dtype=DTYPE, order="F"),
            "use_approx": not (kernel is GaussianKernel),
            "wrapped": kernel is PeriodicKernel,
            "optim_method": optim_method,
        }

    def _require_exact(self):
        if self.options["use_approx"]:
            raise NotImplementedError(
                "only the Gaussian-kernel (exact) branch is implemented on the MI355X engine")

    def init(self, params_tl, params_l):
        """Create both GPs.  ``params_*`` = (h, w, s): kernel parameters then noise."""
        self._require_exact()
        kernel = self.options["kernel"]
        self._drop_pairs()
        self.gp_log_l = GP(kernel(*params_tl[:-1]), self.x_s, self.tl_s, s=params_tl[-1])
        self.gp_log_l.jitter = np.zeros(self.ns, dtype=DTYPE)
        self._choose_candidates()
        self.gp_l = GP(kernel(*params_l[:-1]), self.x_sc, self.l_sc, s=params_l[-1])
        self.gp_l.jitter = np.zeros(self.nsc, dtype=DTYPE)
        self._approx_x = self._make_approx_x()
        self._approx_px = self._make_approx_px()
        self.initialized = True

    # ------------------------------------------------------- posterior over l
    def l_mean(self, x):
        """Mean of the final approximation: the posterior mean of the second GP."""
        return self.gp_l.mean(x)

    def l_var(self, x):
        """Marginal variance: var of the log-GP times the squared mean of the
        second GP, negatives clamped to zero (bq.py:227-231).  Only the diagonal of
        the covariance is computed."""
        v_log_l = self.gp_log_l.var(x)
        m_l = self.gp_l.mean(x)
        l_var = v_log_l * m_l ** 2
        l_var[l_var < 0] = 0
        return l_var

    # ---------------------------------------------------------------- moments
    def Z_mean(self):
        self._require_exact()
        return self._exact_Z_mean()

    def _exact_Z_mean(self):
        """E[Z] = (int K_l(x, x_sc) p(x) dx) . alpha_l, fused on the device with the
        resident fit of the second GP (bq_c.pyx:157-213)."""
        m_Z = get_engine().Z_mean(self.gp_l._device_fit(), self.options["x_mean"],
                                  self.options["x_cov"])
        if m_Z <= 0:
            warn("m_Z = %s" % m_Z)
        return m_Z

    def Z_var(self):
        self._require_exact()
        return self._exact_Z_var()

    def _exact_Z_var(self):
        """V(Z) from the two resident fits (bq_c.pyx:264-355); no n x n matrix leaves
        the device."""
        V_Z = get_engine().Z_var(self.gp_log_l._device_fit(), self.gp_l._device_fit(),
                                 self.options["x_mean"], self.options["x_cov"])
        if V_Z <= 0:
            warn("V_Z = %s" % V_Z)
        return V_Z

    # ------------------------------------------------- expected moments given x_a
    def expected_Z_var(self, x_a):
        """E[V(Z) | new observation at x_a] for every entry of x_a."""
        second_moment = self.Z_mean() ** 2 + self.Z_var()
        return second_moment - self.expected_squared_mean(x_a)

    def expected_squared_mean(self, x_a):
        return self._esm_and_em_batch(x_a)[:, 0].copy()

    def expected_mean(self, x_a):
        return self._esm_and_em_batch(x_a)[:, 1].copy()

    def expected_squared_mean_and_mean(self, x_a):
        return self._esm_and_em_batch(x_a)

    def _esm_and_em(self, x_a):
        """(E[m(Z)^2], E[m(Z)]) after a hypothetical observation at the single
        point x_a (bq.py:447-527)."""
        if x_a is None:
            raise ValueError("invalid value for x_a: %s" % x_a)
        esm, em = self._esm_and_em_batch(np.atleast_1d(np.asarray(x_a, dtype=DTYPE)))[0]
        return esm, em

    def _esm_and_em_batch(self, x_a):
        """All candidates of x_a at once.  The reference loops over them and
        re-factors an (nsc+1)^2 Gram for each (bq.py:399-402,447-527); here all of them are
        one bordered update of gp_l's resident factor (bq_esm_border; with a noise term in
        gp_l: one batched factorisation, bq_esm_batch) and the closed forms of
        bq_c.pyx:425-490 are applied to the results.  Same short-circuit, jitter and
        fallback rules."""
        self._require_exact()
        x_a = np.atleast_1d(np.asarray(x_a, dtype=DTYPE))
        if x_a.ndim != 1 or np.isnan(x_a).any() or np.isinf(x_a).any():
            raise ValueError("invalid value for x_a: %s" % x_a)
        M = x_a.shape[0]
        out = np.empty((M, 2))
        current = None  # (Z_mean^2, Z_mean), computed only if some candidate needs it

        def fallback(rows):
            nonlocal current
            if current is None:
                em = self.Z_mean()
                current = (em ** 2, em)
            out[rows] = current

        # a point we (almost) already have cannot move the mean (bq.py:456-459)
        near = np.isclose(x_a[:, None], self.x_s[None, :], atol=1e-4).any(axis=1)
        if near.any():
            fallback(near)
        idx = np.nonzero(~near)[0]
        if idx.size == 0:
            return out
        xa = np.ascontiguousarray(x_a[idx])
        eng = get_engine()
        if float(self.gp_l.s) == 0.0:
            # bordered update of gp_l's resident factor: one multi-right-hand-side solve for
            # all candidates instead of a factorisation each
            A_a, A_sc_l, status = eng.esm_border(
                self.gp_l._device_fit(), self.ns, xa, self.options["candidate_thresh"],
                self.options["x_mean"], self.options["x_cov"])
        else:
            A_a, A_sc_l, status = eng.esm_batch(
                self.x_sc, self.l_sc, self.ns, xa, self.gp_l.K.h, self.gp_l.K.w,
                self.options["candidate_thresh"], self.options["x_mean"],
                self.options["x_cov"])
        tm_a, tC_a = self.gp_log_l.mean_var(xa)
        # int exp(c x) N(x | m, S) dx = exp(c m + c^2 S / 2), saturating (gauss_c.pyx:65-92)
        arg1 = tm_a + 0.5 * tC_a
        arg2 = 2.0 * tm_a + 2.0 * tC_a
        with np.errstate(over="ignore", invalid="ignore"):
            e1 = np.where(arg1 > MAX, np.inf, np.exp(np.minimum(arg1, MAX)))
            e2 = np.where(arg2 > MAX, np.inf, np.exp(np.minimum(arg2, MAX)))
            em = A_sc_l + A_a * e1
            esm = (A_sc_l ** 2) + (2 * A_sc_l * A_a * e1) + (A_a ** 2 * e2)
        esm = np.where(np.isinf(e1) | np.isinf(e2), np.inf, esm)
        em = np.where(np.isinf(e1), np.inf, em)
        ok = status == 0
        for k in np.nonzero(ok)[0]:
            if np.isnan(esm[k]) or esm[k] < 0:
                raise RuntimeError(
                    "invalid expected squared mean for x_a=%s: %s" % (xa[k], esm[k]))
            if np.isnan(em[k]):
                raise RuntimeError("invalid expected mean for x_a=%s: %s" % (xa[k], em[k]))
            if np.isinf(esm[k]):
                logger.warning("expected squared mean for x_a=%s is infinity!", xa[k])
            if np.isinf(em[k]):
                logger.warning("expected mean for x_a=%s is infinity!", xa[k])
        out[idx[ok], 0] = esm[ok]
        out[idx[ok], 1] = em[ok]
        if (~ok).any():
            # singular system: x_a duplicates information we have (bq.py:481-490)
            fallback(idx[~ok])
        return out

    # ------------------------------------------------------- hyper-parameters
    def _make_llh_params(self, params):
        """Closure x -> log_lh(GP1) + log_lh(GP2) for parameter vector
        [params of GP1..., params of GP2...]; any failure is -inf (bq.py:536-550)."""
        nparam = len(params)

        def f(x):
            if x is None or np.isnan(x).any():
                return -np.inf
            try:
                self._set_gp_log_l_params(dict(zip(params, x[:nparam])))
                self._set_gp_l_params(dict(zip(params, x[nparam:])))
                return self.gp_log_l.log_lh + self.gp_l.log_lh
            except (ValueError, np.linalg.LinAlgError):
                return -np.inf

        return f

    def _current_params(self, params):
        return np.array([self.gp_log_l.get_param(p) for p in params] +
                        [self.gp_l.get_param(p) for p in params])

    def _param_sets(self, params, X):
        """S parameter vectors [params of GP1..., params of GP2...] as the two S x 3 arrays
        (h, w, s) of the batched evaluators, and the mask of the sets the reference's closure
        would reject before touching a GP (NaN, or a value set_param refuses: bq.py:539-543)."""
        X = np.atleast_2d(np.asarray(X, dtype=DTYPE))
        S, nparam = X.shape[0], len(params)
        order = {"h": 0, "w": 1, "s": 2}
        p_tl = np.repeat(np.asarray(self.gp_log_l.params, dtype=DTYPE)[None, :], S, axis=0)
        p_l = np.repeat(np.asarray(self.gp_l.params, dtype=DTYPE)[None, :], S, axis=0)
        for j, name in enumerate(params):
            p_tl[:, order[name]] = X[:, j]
            p_l[:, order[name]] = X[:, nparam + j]
        ok = np.isfinite(p_tl).all(axis=1) & np.isfinite(p_l).all(axis=1)
        for p in (p_tl, p_l):
            ok &= (p[:, 0] > 0) & (p[:, 1] > 0) & (p[:, 2] >= 0)
        # rejected sets still ride through the batch, with harmless parameters
        p_tl[~ok] = np.asarray(self.gp_log_l.params, dtype=DTYPE)
        p_l[~ok] = np.asarray(self.gp_l.params, dtype=DTYPE)
        return np.ascontiguousarray(p_tl), np.ascontiguousarray(p_l), ok

    def _make_llh_batch(self, params):
        """X (S x 2 len(params)) -> the S values of the closure of ``_make_llh_params``, all sets
        in ONE batched device pass (bq_pair_llh) instead of S sequential refits of both GPs.
        Unlike the closure it leaves the GPs' parameters alone."""
        order = {"h": 0, "w": 1, "s": 2}
        cols = [order[name] for name in params]
        nparam = len(params)
        base_tl = np.asarray(self.gp_log_l.params, dtype=DTYPE).copy()
        base_l = np.asarray(self.gp_l.params, dtype=DTYPE).copy()
        buf = {}

        def fb(X):
            # (the loops call this thousands of times on tiny systems: the S x 3 parameter arrays
            # are kept per S and filled in place -- _param_sets, the general form, costs 14 us)
            X = np.asarray(X, dtype=DTYPE)
            if X.ndim != 2:
                X = np.atleast_2d(X)
            S = X.shape[0]
            if S not in buf:
                buf[S] = (np.empty((S, 3)), np.empty((S, 3)))
            p_tl, p_l = buf[S]
            p_tl[:] = base_tl
            p_l[:] = base_l
            p_tl[:, cols] = X[:, :nparam]
            p_l[:, cols] = X[:, nparam:]
            ok = (np.isfinite(X).all(axis=1) & (p_tl[:, :2] > 0).all(axis=1) & (p_tl[:, 2] >= 0) &
                  (p_l[:, :2] > 0).all(axis=1) & (p_l[:, 2] >= 0))
            if not ok.all():
                # rejected sets still ride through the batch, with harmless parameters
                p_tl[~ok] = base_tl
                p_l[~ok] = base_l
            llh, _, _ = self._pair(S).llh(p_tl, p_l)
            return llh if ok.all() else np.where(ok, llh, -np.inf)

        return fb

    def fit_hypers(self, params):
        f = self._make_llh_params(params)
        p0 = util.find_good_parameters(f, self._current_params(params),
                                       self.options["optim_method"],
                                       logpdf_batch=self._make_llh_batch(params))
        if p0 is None:
            raise RuntimeError("couldn't find good parameters")
        f(p0)  # leave the GPs at the optimum (the optimiser's last call may be elsewhere)

    def sample_hypers(self, params, n=1, nburn=10):
        """Slice-sample new hyper-parameters for both GPs; returns the samples for
        GP1 and GP2 (bq.py:565-598).  The GPs are left at the last evaluated point.  The chain
        is the reference's, draw for draw; its log-pdf requests are evaluated several at a time
        in batched device passes (util._slice_sample_batched)."""
        nparam = len(params)
        window = 2 * nparam
        p0 = self._current_params(params)
        f = self._make_llh_params(params)
        if f(p0) < MIN:
            pn = util.find_good_parameters(f, p0, self.options["optim_method"],
                                           logpdf_batch=self._make_llh_batch(params))
            if pn is None:
                raise RuntimeError("couldn't find good starting parameters")
            p0 = pn
        # points per device pass: sixteen stacked systems of a few dozen points cost what six do
        # (a pass is launch-bound there), and a window that steps out far is walked six
        # positions per end and pass instead of two; large systems keep the narrow pass
        spec = 14 if self.ns + self.nc <= 128 else 4
        hypers = util.slice_sample(f, nburn + n, window, p0, nburn=nburn, freq=1,
                                   logpdf_batch=self._make_llh_batch(params), spec=spec)
        f(hypers[-1])  # the chain's last evaluated point, as the sequential sampler leaves it
        return hypers[:, :nparam], hypers[:, nparam:]

    # --------------------------------------------------------- active sampling
    def marginalize(self, funs, n, params):
        """Evaluate each function of ``funs`` under ``n`` sampled hyper-parameter
        settings; the object's state is restored afterwards (bq.py:604-657)."""
        state = deepcopy(self.__getstate__())
        values = []
        for fun in funs:
            shape = getattr(fun(), "shape", ())
            values.append(np.empty((n,) + tuple(shape)))
        hypers_tl, hypers_l = self.sample_hypers(params, n=n, nburn=1)
        for i in range(n):
            params_tl = dict(zip(params, hypers_tl[i]))
            params_l = dict(zip(params, hypers_l[i]))
            self._set_gp_log_l_params(params_tl)
            self._set_gp_l_params(params_l)
            for j, fun in enumerate(funs):
                try:
                    values[j][i] = fun()
                except Exception:
                    logger.error("error with parameters %s and %s", params_tl, params_l)
                    raise
        self.__setstate__(state)
        return values

    def choose_next(self, x_a, n, params, plot=False):
        """The entry of x_a with the smallest marginal loss -E[m(Z)^2] under ``n`` sampled
        hyper-parameter settings (bq.py:659-662).  The reference -- and ``marginalize`` -- visit
        the settings one after the other; here all of them are one batched device pass
        (``_esm_marginal``).  Same random draws, same state restored afterwards."""
        if plot:
            raise NotImplementedError("plotting is not part of the MI355X engine")
        x_a = np.atleast_1d(np.asarray(x_a, dtype=DTYPE))
        state = deepcopy(self.__getstate__())
        hypers_tl, hypers_l = self.sample_hypers(params, n=n, nburn=1)
        esm = self._esm_marginal(x_a, params, hypers_tl, hypers_l)
        self.__setstate__(state)
        loss = (-esm).mean(axis=0)
        ties = np.nonzero(np.isclose(loss, np.min(loss)))[0]
        return x_a[np.random.choice(ties)]

    def _esm_marginal(self, x_a, params, hypers_tl, hypers_l):
        """E[m(Z)^2 | x_a] for every entry of x_a under every hyper-parameter setting
        (rows of hypers_tl / hypers_l over ``params``): n x len(x_a).  One batched pass for
        GP1's refits and posteriors and the n x len(x_a) bordered systems of bq.py:447-527;
        the closed forms of bq_c.pyx:425-490 and the rules of ``_esm_and_em_batch`` (near-sample
        short-circuit, singular-system fallback, overflow) are applied per setting."""
        self._require_exact()
        if x_a.ndim != 1 or np.isnan(x_a).any() or np.isinf(x_a).any():
            raise ValueError("invalid value for x_a: %s" % x_a)
        n, M = hypers_tl.shape[0], x_a.shape[0]
        X = np.concatenate([hypers_tl, hypers_l], axis=1)
        p_tl, p_l, ok = self._param_sets(params, X)
        if not ok.all():
            raise ValueError("invalid hyper-parameter sample")
        out = np.empty((n, M))
        near = np.isclose(x_a[:, None], self.x_s[None, :], atol=1e-4).any(axis=1)
        idx = np.nonzero(~near)[0]
        r = None
        if idx.size:
            xa = np.ascontiguousarray(x_a[idx])
            r = self._pair(n, xa).esm(p_tl, p_l, self.options["candidate_thresh"],
                                      self.options["x_mean"], self.options["x_cov"])
            bad = np.nonzero(r["sstatus"] != 0)[0]
            if bad.size:
                # what _set_gp_log_l_params raises for this setting inside marginalize
                logger.error("error with parameters %s and %s", p_tl[bad[0]], p_l[bad[0]])
                raise np.linalg.LinAlgError(
                    "GP mean is too large" if r["sstatus"][bad[0]] == 2
                    else "matrix is not positive definite")
            arg1 = r["tm_a"] + 0.5 * r["tC_a"]
            arg2 = 2.0 * r["tm_a"] + 2.0 * r["tC_a"]
            A_a, A_sc_l = r["A_a"], r["A_sc_l"]
            with np.errstate(over="ignore", invalid="ignore"):
                e1 = np.where(arg1 > MAX, np.inf, np.exp(np.minimum(arg1, MAX)))
                e2 = np.where(arg2 > MAX, np.inf, np.exp(np.minimum(arg2, MAX)))
                esm = (A_sc_l ** 2) + (2 * A_sc_l * A_a * e1) + (A_a ** 2 * e2)
            esm = np.where(np.isinf(e1) | np.isinf(e2), np.inf, esm)
            good = r["status"] == 0
            if (good & (np.isnan(esm) | (esm < 0))).any():
                b, k = np.argwhere(good & (np.isnan(esm) | (esm < 0)))[0]
                raise RuntimeError(
                    "invalid expected squared mean for x_a=%s: %s" % (xa[k], esm[b, k]))
            out[:, idx] = esm
        # settings with a point that cannot move the mean, or a singular system: the current
        # squared mean under THAT setting (bq.py:456-459, 481-490), through the object path
        need = near[None, :].repeat(n, axis=0)
        if r is not None:
            need[:, idx] |= r["status"] != 0
        for b in np.nonzero(need.any(axis=1))[0]:
            self._set_gp_log_l_params(dict(zip(("h", "w", "s"), p_tl[b])))
            self._set_gp_l_params(dict(zip(("h", "w", "s"), p_l[b])))
            out[b, need[b]] = self.Z_mean() ** 2
        return out

    def add_observation(self, x_a, l_a):
        """Add (x_a, l_a); an x_a within ``candidate_thresh`` of a sample is averaged
        into it instead.  Re-initialises both GPs with their current parameters."""
        diffs = np.abs(x_a - self.x_s)
        if diffs.min() < self.options["candidate_thresh"]:
            c = diffs.argmin()
            self.x_s[c] = (self.x_s[c] + x_a) / 2.0
            self.l_s[c] = (self.l_s[c] + l_a) / 2.0
            self.tl_s[c] = np.log(float(self.l_s[c]))
        else:
            self.x_s = np.append(self.x_s, float(x_a))
            self.l_s = np.append(self.l_s, float(l_a))
            self.tl_s = np.append(self.tl_s, np.log(float(l_a)))
            self.ns += 1
        self.init(self.gp_log_l.params, self.gp_l.params)

    # ------------------------------------------------------- pickling, copying
    def __getstate__(self):
        state = {k: getattr(self, k) for k in _STATE_ALWAYS}
        if self.initialized:
            state["gp_log_l"] = self.gp_log_l
            state["gp_log_l_jitter"] = self.gp_log_l.jitter
            state["gp_l"] = self.gp_l
            state["gp_l_jitter"] = self.gp_l.jitter
            state["_approx_x"] = self._approx_x
            state["_approx_px"] = self._approx_px
        return state

    def __setstate__(self, state):
        same = (getattr(self, "_pairs", None) and getattr(self, "x_s", None) is not None
                and np.array_equal(self.x_s, state["x_s"])
                and np.array_equal(self.l_s, state["l_s"]) and state.get("gp_l") is not None
                and self.x_c is not None
                and np.array_equal(self.x_c, state["gp_l"]._x[state["x_s"].shape[0]:]))
        if not same:
            self._drop_pairs()
        for k in _STATE_ALWAYS:
            setattr(self, k, state[k])
        self.ns = self.x_s.shape[0]
        if not self.initialized:
            self._clear_fit_state()
            return
        self.gp_log_l = state["gp_log_l"]
        self.gp_log_l.jitter = state["gp_log_l_jitter"]
        self.gp_l = state["gp_l"]
        self.gp_l.jitter = state["gp_l_jitter"]
        self.x_sc = self.gp_l._x
        self.l_sc = self.gp_l._y
        self.nsc = self.x_sc.shape[0]
        self.x_c = self.x_sc[self.ns:]
        self.l_c = self.l_sc[self.ns:]
        self.nc = self.nsc - self.ns
        self._approx_x = state["_approx_x"]
        self._approx_px = state["_approx_px"]

    def __copy__(self):
        new = type(self).__new__(type(self))
        new.__setstate__(self.__getstate__())
        return new

    def __deepcopy__(self, memo):
        new = type(self).__new__(type(self))
        new.__setstate__(deepcopy(self.__getstate__(), memo))
        return new

    def copy(self, deep=True):
        return deepcopy(self) if deep else copy(self)

    # ---------------------------------------------------------------- helpers
    def _set_gp_log_l_params(self, params):
        """New hyper-parameters for GP1: refit, re-evaluate the candidates
        (l_c = exp(mean), with the overflow guard of bq.py:945-947) and hand the
        new targets to GP2 (bq.py:933-957)."""
        for p, v in params.items():
            self.gp_log_l.set_param(p, v)
        self.gp_log_l.jitter.fill(0)
        m, V = self.gp_log_l.mean_var(self.x_c) if self.nc else (np.empty(0), np.empty(0))
        V = V.copy()
        V[V < 0] = 0
        if ((m + 2 * np.sqrt(V)) > MAX).any():
            raise np.linalg.LinAlgError("GP mean is too large")
        self.l_c = np.exp(m)
        self.l_sc = np.array(np.concatenate([self.l_s, self.l_c]))
        self.gp_l.x = self.x_sc
        self.gp_l.y = self.l_sc
        self.gp_l.jitter.fill(0)

    def _set_gp_l_params(self, params):
        for p, v in params.items():
            self.gp_l.set_param(p, v)
        self.gp_l.jitter.fill(0)

    def _choose_candidates(self):
        """Uniform draws over the sample range widened by w, filtered for spacing
        (bq.py:967-991); their values are exp(mean of GP1)."""
        if self.options["wrapped"]:
            raise NotImplementedError("periodic kernels are out of scope")
        w = self.gp_log_l.K.w
        xc = np.random.uniform(self.x_s.min() - w, self.x_s.max() + w,
                               self.options["n_candidate"])
        bq_c.filter_candidates(xc, self.x_s, self.options["candidate_thresh"])
        self.x_c = np.sort(xc[~np.isnan(xc)])
        self.nc = self.x_c.shape[0]
        self.l_c = np.exp(self.gp_log_l.mean(self.x_c)) if self.nc else np.empty(0, dtype=DTYPE)
        self.x_sc = np.array(np.concatenate([self.x_s, self.x_c]))
        self.l_sc = np.array(np.concatenate([self.l_s, self.l_c]))
        self.nsc = self.ns + self.nc

    def _make_approx_x(self, xmin=None, xmax=None, n=1000):
        w = self.gp_log_l.K.w
        if xmin is None:
            xmin = self.x_sc.min() - w
        if xmax is None:
            xmax = self.x_sc.max() + w
        return np.linspace(xmin, xmax, n)

    def _make_approx_px(self, x=None):
        if x is None:
            x = self._approx_x
        p = np.empty(x.size, order="F")
        bq_c.p_x_gaussian(p, _row(x), self.options["x_mean"], self.options["x_cov"])
        return p
