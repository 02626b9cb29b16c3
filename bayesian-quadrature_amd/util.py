"""Host-side drivers of the hyper-parameter loop (callers of the hot path).

Mirror of the parts of the reference's ``util.py`` / ``util_c.pyx`` that ``BQ``
uses: ``find_good_parameters`` (util.py:151-169) and ``slice_sample``
(util.py:45-75 over util_c.pyx:25-148).  The slice sampler is a sequential MCMC
driver around a Python log-pdf; it stays on the host.  Unlike the reference it
draws from numpy's generator only (the reference mixes in libc ``rand()``,
util_c.pyx:21-33), so runs are reproducible under ``np.random.seed``.
"""
import logging

import numpy as np
import scipy.optimize as optim

logger = logging.getLogger("bayesian_quadrature.util")

MIN = float(np.log(np.exp2(np.float64(np.finfo(np.float64).minexp + 4))))


# absolute forward-difference steps scipy.optimize.minimize uses when no gradient is supplied
# (scipy/optimize/_lbfgsb_py.py: eps = 1e-8; _optimize.py: _epsilon = sqrt(machine epsilon))
_FD_STEP = {"L-BFGS-B": 1e-8, "BFGS": 1.4901161193847656e-08, "CG": 1.4901161193847656e-08}


def fd_points(x0, step):
    """The p + 1 points of scipy's 2-point gradient at x0 (approx_derivative with an absolute
    step: x0 and x0 + step e_i) and the exactly representable steps dx_i."""
    x0 = np.asarray(x0, dtype=np.float64)
    h = np.full(x0.shape, float(step))
    dx = (x0 + h) - x0
    sign = (x0 >= 0).astype(np.float64) * 2 - 1
    h = np.where(dx == 0, np.finfo(np.float64).eps ** 0.5 * sign * np.maximum(1.0, np.abs(x0)), h)
    X = np.repeat(x0[None, :], x0.size + 1, axis=0)
    for i in range(x0.size):
        X[i + 1, i] = x0[i] + h[i]
    return X, np.array([X[i + 1, i] - x0[i] for i in range(x0.size)])


def cd_points(x0):
    """The 2p + 1 points of a central-difference gradient at x0 -- x0, x0 + h_i e_i, x0 - h_i e_i
    -- with scipy's '3-point' relative step h_i = eps^(1/3) max(1, |x0_i|), and the exactly
    representable spans (x0 + h_i) - (x0 - h_i)."""
    x0 = np.asarray(x0, dtype=np.float64)
    h = np.finfo(np.float64).eps ** (1.0 / 3.0) * np.maximum(1.0, np.abs(x0))
    p = x0.size
    X = np.repeat(x0[None, :], 2 * p + 1, axis=0)
    for i in range(p):
        X[1 + i, i] = x0[i] + h[i]
        X[1 + p + i, i] = x0[i] - h[i]
    return X, np.array([X[1 + i, i] - X[1 + p + i, i] for i in range(p)])


# what the last find_good_parameters did: wall times of fit_hypers are trajectory-dependent, so
# timing tools print the iterations, evaluations and the final value beside them
LAST_OPT = {}


def find_good_parameters(logpdf, x0, method, ntry=10, logpdf_batch=None):
    """Up to ``ntry`` restarts of scipy.optimize.minimize on -logpdf; returns the
    first optimum whose log-pdf exceeds MIN, else None.

    Without ``logpdf_batch`` this is the reference's call (util.py:151-169): scipy differences
    the objective itself, p + 1 sequential evaluations per gradient with an absolute forward step
    of 1e-8 -- on an fp64 objective whose value carries ~1e-13 of rounding noise that gradient is
    good to ~1e-5, and L-BFGS-B stops wherever on the flat top the noise sends it.

    ``logpdf_batch`` (S x p array -> S values) evaluates the 2p + 1 points of a CENTRAL difference
    in ONE batched device pass (a pass costs the same for 2p + 1 stacked systems as for p + 1) and
    hands scipy the value and that gradient together (``jac=True``): step eps^(1/3), truncation
    and noise both ~1e-8.  NOT the reference's trajectory -- a better-conditioned one over the
    same objective: it ends at least as high (round 5's forward-difference batch ended at -1.114
    where the sequential run reached -0.940 on the reference's fixture; tests/test_bq_object.py
    bounds the gap now)."""
    batched = logpdf_batch is not None

    def fun_and_grad(x):
        X, span = cd_points(x)
        p = x.size
        with np.errstate(invalid="ignore"):
            f = -np.asarray(logpdf_batch(X), dtype=np.float64)
            g = (f[1:1 + p] - f[1 + p:]) / span
            # a side that left the domain (-inf log-pdf): the one-sided difference on the other
            bad = ~np.isfinite(g)
            if bad.any():
                fw = (f[1:1 + p] - f[0]) / (X[np.arange(1, 1 + p), np.arange(p)] - x)
                bw = (f[0] - f[1 + p:]) / (x - X[np.arange(1 + p, 1 + 2 * p), np.arange(p)])
                g = np.where(bad, np.where(np.isfinite(fw), fw, bw), g)
        return f[0], g

    for i in range(ntry):
        logger.debug("Attempt #%d with %s", i + 1, method)
        if batched:
            res = optim.minimize(fun=fun_and_grad, x0=x0, method=method, jac=True)
        else:
            res = optim.minimize(fun=lambda x: -logpdf(x), x0=x0, method=method)
        p = logpdf(res["x"])
        LAST_OPT.clear()
        LAST_OPT.update({"attempts": i + 1, "nit": int(res.get("nit", -1)),
                         "nfev": int(res.get("nfev", -1)), "logpdf": float(p)})
        if p > MIN:
            return res["x"]
        if logpdf(x0) < p:
            x0 = res["x"]
    return None


def _slice_sample(samples, logpdf, xval, w, verbose=False):
    """Univariate slice sampling along random directions; fills samples[1:]."""
    n, d = samples.shape
    i = 0
    while i < n - 1:
        xpr = logpdf(samples[i])
        if xpr == -np.inf:
            raise RuntimeError("zero probability encountered")
        yval = np.random.uniform(0, np.exp(xpr))
        logyval = np.log(yval) if yval > 0 else -np.inf
        direction = np.random.rand(d) - 0.5
        direction /= np.linalg.norm(direction)
        left, right = -w, w
        # step the window out until both ends leave the slice (at most 100 steps)
        for _ in range(101):
            if logpdf(samples[i] + left * direction) < logyval:
                break
            left -= w
        for _ in range(101):
            if logpdf(samples[i] + right * direction) < logyval:
                break
            right += w
        # shrink until a point inside the slice is drawn
        while True:
            if (right - left) < 1e-9:
                break  # window collapsed: redraw the slice height at the same point
            loc = np.random.uniform(left, right)
            samples[i + 1] = samples[i] + loc * direction
            if logpdf(samples[i + 1]) > logyval:
                i += 1
                break
            if loc < 0:
                left = loc
            else:
                right = loc


_PEEK_RS = np.random.RandomState(0)


class _Draws(object):
    """The chain's random numbers as raw uniform doubles, in the order the sequential sampler
    draws them, from a private copy of numpy's global generator -- so that looking AHEAD costs
    nothing (copying the generator's state for every look, 624 words each way, was a third of a
    small system's pass).  ``uniform(lo, hi)`` is numpy's own ``lo + (hi - lo) * u`` and
    ``rand(d)`` its next d doubles; ``close`` leaves the global generator where the sequential
    sampler would have: at the start state advanced by exactly the doubles consumed."""

    def __init__(self):
        self._state0 = np.random.get_state()
        self._rs = _PEEK_RS
        self._rs.set_state(self._state0)
        self._buf = np.empty(0)
        self._pos = 0
        self._used = 0

    def _need(self, k):
        if self._pos + k > self._buf.size:
            fresh = self._rs.random_sample(max(512, k))
            self._buf = np.concatenate([self._buf[self._pos:], fresh])
            self._pos = 0

    def take(self, k):
        self._need(k)
        out = self._buf[self._pos:self._pos + k]
        self._pos += k
        self._used += k
        return out

    def peek(self, k):
        self._need(k)
        return self._buf[self._pos:self._pos + k]

    def uniform(self, lo, hi):
        return lo + (hi - lo) * float(self.take(1)[0])

    def close(self):
        np.random.set_state(self._state0)
        if self._used:
            np.random.random_sample(self._used)


def _slice_sample_batched(samples, logpdf_batch, xval, w, spec=4):
    """The same chain as ``_slice_sample`` -- the same random draws in the same order, the same
    comparisons -- with the log-pdf evaluated in BATCHES of ``spec + 2`` points per device pass.

    The sequential sampler spends at least four dependent evaluations per state: the current
    point, the two ends of the window, one proposal or more.  What it will ask for next is
    largely known in advance: the current point's value is the accepted proposal's; both ends are
    always needed; and the shrinkage proposals depend on the random draws and on the SIGN of the
    rejected ones only, not on their values -- so the next ``spec`` proposals can be read off a
    copy of the generator without disturbing it.  One pass evaluates the ends and those proposals
    together; the chain then runs as written and finds (almost) every value it asks for in the
    cache.  A window that has to step out, or more than ``spec`` rejections, costs further
    passes; never a different result."""
    n, d = samples.shape
    S = spec + 2
    cache = {}

    def fetch(x0, direction, ts):
        """one pass: the values at x0 + t direction for the listed t (padded to S points)"""
        ts = [t for t in ts if t not in cache][:S]
        if not ts:
            return
        pad = ts + [ts[0]] * (S - len(ts))
        X = np.array([x0 + t * direction for t in pad])
        vals = np.asarray(logpdf_batch(X), dtype=np.float64)
        for t, v in zip(ts, vals):
            cache[t] = float(v)

    draws = _Draws()

    def peek(left, right, k):
        """the next k shrinkage proposals if every one of them is rejected"""
        out = []
        for u in draws.peek(k):
            if (right - left) < 1e-9:
                break
            loc = left + (right - left) * float(u)
            out.append(loc)
            if loc < 0:
                left = loc
            else:
                right = loc
        return out

    try:
        _slice_chain(samples, logpdf_batch, w, spec, S, cache, fetch, peek, draws)
    finally:
        draws.close()


def _slice_chain(samples, logpdf_batch, w, spec, S, cache, fetch, peek, draws):
    n, d = samples.shape
    xpr = float(np.asarray(logpdf_batch(np.repeat(samples[0][None, :], S, axis=0)))[0])
    i = 0
    while i < n - 1:
        if xpr == -np.inf:
            raise RuntimeError("zero probability encountered")
        x0 = samples[i]
        yval = draws.uniform(0, np.exp(xpr))
        logyval = np.log(yval) if yval > 0 else -np.inf
        direction = draws.take(d) - 0.5
        direction /= np.linalg.norm(direction)
        left, right = -w, w
        cache.clear()
        cache[0.0] = xpr

        def val(t, ahead=()):
            if t not in cache:
                fetch(x0, direction, [t] + list(ahead))
            return cache[t]

        # both ends and, should neither have to move, the first proposals: one pass
        fetch(x0, direction, [left, right] + peek(left, right, spec))
        def steps(t, dw, k):
            """the next k positions of a window end that keeps stepping out (as it will add them)"""
            out = []
            for _ in range(k):
                t = t + dw
                out.append(t)
            return out

        # (a window that keeps stepping out: half of a pass for each end's next positions)
        half = max(2, (S - 2) // 2)
        for _ in range(101):
            if val(left, steps(left, -w, half) + [right] + steps(right, w, half)) < logyval:
                break
            left -= w
        for _ in range(101):
            if val(right, steps(right, w, spec + 1)) < logyval:
                break
            right += w
        while True:
            if (right - left) < 1e-9:
                break  # window collapsed: redraw the slice height at the same point
            loc = draws.uniform(left, right)
            samples[i + 1] = x0 + loc * direction
            nl, nr = (loc, right) if loc < 0 else (left, loc)
            v = val(loc, peek(nl, nr, spec + 1) if loc not in cache else ())
            if v > logyval:
                i += 1
                xpr = v
                break
            left, right = nl, nr


def slice_sample(logpdf, niter, w, xval, nburn=1, freq=1, logpdf_batch=None, spec=4):
    """Draw ``niter`` states starting at ``xval``; drops the first ``nburn`` and
    keeps every ``freq``-th of the rest.  ``logpdf_batch`` (S x d array -> S values): evaluate
    the chain's requests in batched device passes of ``spec + 2`` points
    (``_slice_sample_batched``)."""
    xval = np.asarray(xval, dtype=np.float64)
    samples = np.empty((niter, xval.size))
    samples[0] = xval
    verbose = (logger.level != 0) and (logger.level < 10)
    if logpdf_batch is not None:
        _slice_sample_batched(samples, logpdf_batch, xval, float(w), spec)
    else:
        _slice_sample(samples, logpdf, xval, float(w), verbose)
    return samples[nburn:][::freq]
