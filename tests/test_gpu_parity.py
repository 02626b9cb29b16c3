"""GPU parity: every hot-path entry point of the C ABI against the CPU oracle on
the same seeded inputs (SURVEY.md section 8d bars), plus size-independent
properties at the full BASELINE sizes.  All calls go through libbqhip.so."""
import numpy as np
import pytest

from conftest import rand_spd
from bayesian_quadrature_amd import workloads as wl

pytestmark = pytest.mark.gpu

RTOL = 1e-10  # north_star: posterior mean/variance and log-ML within 1e-10 relative fp64


def relmax(a, b, scale=None):
    a, b = np.asarray(a), np.asarray(b)
    s = np.max(np.abs(b)) if scale is None else scale
    return np.max(np.abs(a - b)) / s


# ---- hardware facts the kernels rely on ----------------------------------------
def test_mfma_f64_layout(engine):
    """D register r of lane l holds D[(l>>4) + 4r][l&15] (gemm.h gemm_sub_kernel)."""
    lay = engine.probe_mfma_layout()
    for l in range(64):
        for r in range(4):
            row, col = (l >> 4) + 4 * r, l & 15
            assert lay[l, r] == 16 * row + col, (l, r, lay[l, r])


def test_exp_gauss_accuracy(engine):
    """The Gram kernels' hand-written exp (csrc/common.h: ln 2 hi/lo reduction, degree-13
    polynomial, v_ldexp_f64) per element: within 1 ulp of the true exp -- taken in x87
    extended precision -- over 10^6 arguments of [-745.2, 0], the subnormal results
    (x < -708.4) and the flush to zero included.  Gram parity is max|dK| / max|K| and says
    nothing about entries 300 orders of magnitude below the diagonal; this does."""
    from bayesian_quadrature_amd import _lib as L
    rs = np.random.RandomState(42)
    x = np.concatenate([rs.uniform(-745.2, 0.0, 600000), rs.uniform(-745.2, -708.0, 150000),
                        -np.exp(rs.uniform(-40, 3, 249000)),
                        np.array([0.0, -0.0, -1e-300, -708.396418532264, -745.1332191019411,
                                  -745.1332191019412, -746.0, -800.0, -1e4])])
    out = np.empty_like(x)
    engine._check(engine._lib.bq_probe_exp(engine._ctx, L.dptr(x), x.size, L.dptr(out)))
    truth = np.exp(x.astype(np.longdouble))
    ref = truth.astype(np.float64)
    ulp = np.spacing(np.maximum(np.abs(ref), np.finfo(np.float64).tiny * 2.0 ** -52))
    err = np.abs(out.astype(np.longdouble) - truth) / ulp
    assert np.isfinite(out).all() and (out >= 0).all()
    assert float(err.max()) <= 1.0, (float(err.max()), x[np.argmax(err)])
    assert (out[-9:-6] == 1.0).all() and (out[-3:] == 0.0).all()
    # no glitch at the range-reduction seams: on a sorted sample the result never falls by
    # more than the one ulp the accuracy bound allows
    xs = np.sort(x[:200000])
    o2 = np.empty_like(xs)
    engine._check(engine._lib.bq_probe_exp(engine._ctx, L.dptr(xs), xs.size, L.dptr(o2)))
    assert (np.diff(o2) >= -np.spacing(o2[1:])).all()


# ---- Gram ------------------------------------------------------------------------
@pytest.mark.parametrize("n", [1, 2, 9, 63, 64, 129, 1000])
def test_gram_1d(engine, oracle, n):
    rs = np.random.RandomState(n)
    x = np.sort(rs.uniform(-5, 5, n))
    h, w, s = 1.7, 0.31, 0.05
    K = engine.gram(x, h, w, s)
    Ko = oracle.gram(x, h, w, s)
    assert relmax(K, Ko) < 1e-14
    assert (K == K.T).all()


@pytest.mark.parametrize("d", [2, 3, 8])
def test_gram_nd(engine, oracle, d):
    rs = np.random.RandomState(d)
    pts = rs.uniform(-2, 2, size=(d, 301))
    w = rs.uniform(0.3, 1.0, d)
    assert relmax(engine.gram(pts, 0.9, w, 0.1), oracle.gram(pts, 0.9, w, 0.1)) < 1e-14


def test_gram_cross(engine, oracle):
    rs = np.random.RandomState(0)
    x1, x2 = rs.uniform(-5, 5, 77), rs.uniform(-5, 5, 130)
    assert relmax(engine.gram_cross(x1, x2, 1.2, 0.4), oracle.gram_cross(x1, x2, 1.2, 0.4)) < 1e-14


def test_gram_bad_args(engine):
    with pytest.raises(ValueError):
        engine.gram(np.zeros(4), 1.0, -1.0)
    with pytest.raises(ValueError):
        engine.gram(np.zeros((9, 4)), 1.0, np.ones(9))


# ---- linalg_c drop-ins (reference tests/test_linalg_c.py) --------------------------
@pytest.mark.parametrize("n", list(range(1, 11)) + [33, 64, 65, 128, 200, 513])
def test_cho_factor(engine, oracle, n):
    from bayesian_quadrature_amd import la
    rs = np.random.RandomState(100 + n)
    A = rand_spd(rs, n)
    L = np.empty_like(A, order="F")
    la.cho_factor(A, L)
    Lo = oracle.cho_factor(A)
    assert relmax(np.tril(L), Lo) < 1e-13
    # upper triangle is C's (linalg_c.pyx:58-59 "could be anything"; here: untouched copy)
    assert (np.triu(L, 1) == np.triu(A, 1)).all()
    # in place
    A2 = A.copy(order="F")
    la.cho_factor(A2, A2)
    assert (np.tril(A2) == np.tril(L)).all()


def test_cho_factor_errors(engine):
    from bayesian_quadrature_amd import la
    A = np.asfortranarray(np.array([[1.0, 2.0], [2.0, 1.0]]))
    with pytest.raises(np.linalg.LinAlgError):
        la.cho_factor(A, np.empty_like(A, order="F"))
    with pytest.raises(ValueError):
        la.cho_factor(np.asfortranarray(np.ones((2, 3))), np.asfortranarray(np.ones((2, 3))))
    with pytest.raises(ValueError):
        la.cho_factor(np.ascontiguousarray(np.eye(3) + 1), np.empty((3, 3), order="F"))
    big = rand_spd(np.random.RandomState(0), 300)
    big[250, 250] = -5.0  # fails deep inside the blocked sweep
    with pytest.raises(np.linalg.LinAlgError):
        la.cho_factor(big, np.empty_like(big, order="F"))


@pytest.mark.parametrize("n", [1, 2, 5, 10, 64, 100, 300])
def test_cho_solve(engine, oracle, n):
    from bayesian_quadrature_amd import la
    rs = np.random.RandomState(n)
    A = rand_spd(rs, n)
    L = oracle.cho_factor(A)
    b = rs.rand(n)
    x = np.empty(n)
    la.cho_solve_vec(L, b, x)
    assert relmax(x, oracle.cho_solve(L, b)) < 1e-12
    la.cho_solve_vec(L, b, b)  # aliased
    assert (b == x).all()
    B = np.asfortranarray(rs.rand(n, n))
    X = np.empty_like(B, order="F")
    la.cho_solve_mat(L, B, X)
    assert relmax(X, oracle.cho_solve(L, B)) < 1e-12
    assert np.allclose(A.dot(X), B)


@pytest.mark.parametrize("n,M", [(9, 3), (100, 10), (300, 63), (1024, 20)])
def test_refit_predict_one_sweep(engine, oracle, n, M):
    """bq_gp_refit_predict -- the hyper-parameter loop's body (bq.py:933-947): new parameters
    and the candidates' posterior in ONE sweep, the points as border rows of the fit's own
    system -- against the oracle and against the two-call route; the fit it leaves behind
    (log-ML, alpha with the y row shifted by the border, a later plain refit) is the same."""
    x, y, h, w, s = _problem(n, 70 + n)
    xo = np.linspace(-5.5, 5.5, M) + 0.013
    fit = engine.gp_fit(x, y, h, w, s)
    h2, w2, s2 = 1.1 * h, 0.9 * w, 2.0 * s
    m, v = fit.refit_predict(h2, w2, s2, xo)
    Lo, ao, lmo = oracle.gp_fit(x, y, h2, w2, s2)
    mo, vo = oracle.gp_predict(x, h2, w2, Lo, ao, xo)[:2]
    k0 = h2 * h2 / (np.sqrt(2 * np.pi) * w2)
    assert relmax(m, mo) < 1e-10
    assert np.abs(v - vo).max() / k0 < 1e-10
    assert abs(fit.logml - lmo) <= 1e-10 * abs(lmo)
    assert relmax(fit.alpha(), ao) < 1e-9
    m2, v2, _ = fit.predict(xo)                     # the resident route on the same factor
    assert relmax(m2, mo) < 1e-10 and np.abs(v2 - vo).max() / k0 < 1e-10
    b = np.random.RandomState(n).randn(n)
    assert relmax(fit.solve(b), oracle.cho_solve(Lo, b)) < 1e-9
    fit.refit(h, w, s)                               # back to a fit without border points
    L1, a1, lm1 = oracle.gp_fit(x, y, h, w, s)
    assert abs(fit.logml - lm1) <= 1e-10 * abs(lm1)
    assert relmax(fit.alpha(), a1) < 1e-9
    assert relmax(fit.z(), np.linalg.solve(L1, y)) < 1e-9
    with pytest.raises(np.linalg.LinAlgError):       # failure leaves an invalid fit behind
        fit.refit_predict(h, 60.0 * w, 0.0, xo)
    with pytest.raises(np.linalg.LinAlgError):
        fit.predict(xo)
    m3, v3 = fit.refit_predict(h2, w2, s2, xo)
    assert relmax(m3, mo) < 1e-10
    fit.close()


@pytest.mark.parametrize("n", [11, 300])
def test_set_y_keeps_the_fit(engine, oracle, n):
    """bq_gp_set_y: same points, new targets (what the hyper-parameter loop does to GP2 on every
    evaluation, bq.py:948-954).  The fit refuses its consumers until it is refitted, then
    equals a fresh fit on the new data."""
    x, y, h, w, s = _problem(n, 5 + n)
    fit = engine.gp_fit(x, y, h, w, s)
    y2 = np.cos(x) + 0.1 * y
    fit.set_y(y2)
    # "refit required" is a usage error (ValueError), not a failed factorisation
    for use in (fit.alpha, lambda: fit.logml, lambda: fit.predict(x[:3]), lambda: fit.solve(y)):
        with pytest.raises(ValueError, match="refit required"):
            use()
    fit.refit(h, 1.05 * w, s)
    Lo, ao, lmo = oracle.gp_fit(x, y2, h, 1.05 * w, s)
    assert abs(fit.logml - lmo) <= 1e-10 * abs(lmo)
    assert relmax(fit.alpha(), ao) < 1e-9
    xo = np.linspace(-4, 4, 9)
    mo, vo = oracle.gp_predict(x, h, 1.05 * w, Lo, ao, xo)
    m, v, _ = fit.predict(xo)
    assert relmax(m, mo) < 1e-9
    fit.set_y(y)
    m3, v3 = fit.refit_predict(h, w, s, xo)         # new targets straight into the one-sweep route
    L1, a1, _ = oracle.gp_fit(x, y, h, w, s)
    assert relmax(m3, oracle.gp_predict(x, h, w, L1, a1, xo)[0]) < 1e-9
    with pytest.raises(ValueError):
        fit.set_y(y[:-1])
    fit.close()


@pytest.mark.parametrize("n", [513, 1100, 2048, 2500, 4097])
def test_cho_solve_vec_multi_block(engine, n):
    """One right-hand side through several B-wide steps of the GEMV sweeps (trsv.h): full and
    partial last blocks at B = 256 (npad < 2048) and B = 512, against LAPACK-style
    substitution in numpy on the same factor (residual and forward error), and the same
    answer from the row-form sweeps (the matrix entry point with two columns)."""
    import scipy.linalg as sla
    from bayesian_quadrature_amd import la
    rs = np.random.RandomState(n)
    G = rs.randn(n, 24)
    A = np.asfortranarray(G.dot(G.T) / 24 + np.eye(n))
    L = np.asfortranarray(np.linalg.cholesky(A))
    b = rs.randn(n)
    x = np.empty(n)
    la.cho_solve_vec(L, b, x)
    ref = sla.cho_solve((L, True), b)
    assert relmax(x, ref) < 1e-12
    assert np.abs(A.dot(x) - b).max() < 1e-12 * np.abs(b).max() * n
    B2 = np.asfortranarray(np.stack([b, -2.0 * b], axis=1))
    X2 = np.empty_like(B2, order="F")
    engine.cho_solve(L, B2, X2, 2)
    assert relmax(X2[:, 0], x) < 1e-12
    assert relmax(X2[:, 1], -2.0 * x) < 1e-12


@pytest.mark.parametrize("n", [1, 3, 10, 77, 640])
def test_logdet(engine, oracle, n):
    from bayesian_quadrature_amd import la
    rs = np.random.RandomState(n)
    L = oracle.cho_factor(rand_spd(rs, n))
    assert abs(la.logdet(L) - oracle.logdet(L)) <= 1e-13 * max(1.0, abs(oracle.logdet(L)))


# ---- GP fit / predict ----------------------------------------------------------------
def _problem(n, seed, w_scale=1.0):
    """Jittered grid: the conditioning regime of the BASELINE configs (w ~ dx,
    cond(K) of order 1e2..1e3; SURVEY.md section 7 "hard parts").  The 1e-10
    bar is a forward-error statement and only means something there: beyond
    cond ~ 1e6 the oracle's own rounding error exceeds it."""
    rs = np.random.RandomState(seed)
    dx = 10.0 / max(n - 1, 1)
    x = np.linspace(-5, 5, n) + rs.uniform(-dx / 4, dx / 4, n)
    y = wl.norm_logpdf(x) + 0.01 * rs.randn(n)
    return x, y, 1.3, w_scale * dx, 1e-2


@pytest.mark.parametrize("n", [1, 9, 32, 64, 100, 257, 1024])
def test_gp_fit(engine, oracle, n):
    x, y, h, w, s = _problem(n, n)
    fit = engine.gp_fit(x, y, h, w, s)
    Lo, ao, lmo = oracle.gp_fit(x, y, h, w, s)
    Lg = fit.L()
    K = oracle.gram(x, h, w, s)
    # factor: backward error at rounding level, forward error bounded by cond(K) eps
    assert np.linalg.norm(Lg.dot(Lg.T) - K) / np.linalg.norm(K) < 1e-14 * max(n, 64)
    assert relmax(Lg, Lo) < 1e-10
    assert relmax(fit.alpha(), ao) < RTOL
    assert relmax(fit.z(), oracle.trsm_lower(Lo, y)) < RTOL
    assert abs(fit.logml - lmo) <= RTOL * abs(lmo)
    assert relmax(fit.K(), oracle.gram(x, h, w, s)) < 1e-14
    fit.close()


@pytest.mark.parametrize("n,M", [(9, 5), (100, 1), (100, 64), (300, 257), (1024, 256)])
def test_gp_predict(engine, oracle, n, M):
    x, y, h, w, s = _problem(n, 7 * n + M)
    xo = np.linspace(-5.2, 5.2, M) + 0.013
    fit = engine.gp_fit(x, y, h, w, s)
    Lo, ao, _ = oracle.gp_fit(x, y, h, w, s)
    mo, vo = oracle.gp_predict(x, h, w, Lo, ao, xo)
    k0 = oracle.kernel_scale(1, h, [w])
    m, v, c = fit.predict(xo, want_cov=True)
    assert relmax(m, mo) < RTOL
    assert relmax(v, vo, scale=k0) < RTOL       # variance relative to the prior scale
    assert relmax(np.diag(c), vo, scale=k0) < RTOL
    assert np.allclose(c, c.T, rtol=0, atol=1e-12 * k0)
    # every one of the M^2 entries of gp.GP.cov (bq.py:325,496) against the oracle's
    # K(xo,xo) - V'V, relative to the prior scale like the variance
    co = oracle.gp_cov(x, h, w, Lo, xo)
    assert c.shape == (M, M)
    assert float(np.max(np.abs(c - co))) / k0 < RTOL
    m2 = fit.predict(xo, want_var=False)[0]    # fused mean-only path through alpha
    assert relmax(m2, mo) < RTOL
    fit.close()


def test_gp_refit_and_not_pd(engine, oracle):
    x, y, h, w, s = _problem(200, 3)
    fit = engine.gp_fit(x, y, h, w, s)
    fit.refit(0.7, 2 * w, 0.1)
    _, _, lmo = oracle.gp_fit(x, y, 0.7, 2 * w, 0.1)
    assert abs(fit.logml - lmo) <= RTOL * abs(lmo)
    with pytest.raises(np.linalg.LinAlgError):
        fit.refit(1.0, 50 * w, 0.0)  # numerically singular Gaussian Gram
    # the handle now holds a half-overwritten factor: every consumer refuses it ...
    for use in (lambda: fit.logml, fit.alpha, fit.L, lambda: fit.predict(x[:3]),
                lambda: fit.solve(y)):
        with pytest.raises(np.linalg.LinAlgError):
            use()
    # ... until a refit succeeds again
    fit.refit(0.7, 2 * w, 0.1)
    assert abs(fit.logml - lmo) <= RTOL * abs(lmo)
    fit.close()
    with pytest.raises(ValueError, match="closed"):
        fit.alpha()


def test_gp_2d(engine, oracle):
    rs = np.random.RandomState(11)
    pts = rs.uniform(-3, 3, size=(2, 400))
    y = wl.norm_logpdf(pts[0]) + wl.norm_logpdf(pts[1])
    w = np.array([0.35, 0.5])
    fit = engine.gp_fit(pts, y, 1.0, w, 0.1)
    Lo, ao, lmo = oracle.gp_fit(pts, y, 1.0, w, 0.1)
    assert abs(fit.logml - lmo) <= RTOL * abs(lmo)
    xo = rs.uniform(-3, 3, size=(2, 50))
    mo, vo = oracle.gp_predict(pts, 1.0, w, Lo, ao, xo)
    m, v, _ = fit.predict(xo)
    assert relmax(m, mo) < RTOL
    assert relmax(v, vo, scale=oracle.kernel_scale(2, 1.0, w)) < RTOL
    fit.close()


# ---- the bordered one-pass pipeline (what bench.py times) ---------------------------
def test_fit_predict_c2(engine, oracle):
    c = wl.c2()
    mean, var, logml = engine.fit_predict(c["x"], c["y"], c["h"], c["w"], c["s"], c["xo"])
    Lo, ao, lmo = oracle.gp_fit(c["x"], c["y"], c["h"], c["w"], c["s"])
    mo, vo = oracle.gp_predict(c["x"], c["h"], c["w"], Lo, ao, c["xo"])
    k0 = oracle.kernel_scale(1, c["h"], c["w"])
    assert relmax(mean, mo) < RTOL
    assert relmax(var, vo, scale=k0) < RTOL
    assert abs(logml - lmo) <= RTOL * abs(lmo)


@pytest.mark.parametrize("n,M", [(1, 1), (5, 3), (64, 64), (65, 7), (191, 130)])
def test_fit_predict_ragged(engine, oracle, n, M):
    x, y, h, w, s = _problem(n, n + M)
    xo = np.linspace(-5, 5, M) + 0.01
    mean, var, logml = engine.fit_predict(x, y, h, w, s, xo)
    Lo, ao, lmo = oracle.gp_fit(x, y, h, w, s)
    mo, vo = oracle.gp_predict(x, h, w, Lo, ao, xo)
    assert relmax(mean, mo, scale=max(1e-300, np.abs(mo).max())) < RTOL
    assert relmax(var, vo, scale=oracle.kernel_scale(1, h, [w])) < RTOL
    assert abs(logml - lmo) <= RTOL * abs(lmo)


def test_logml_grid(engine, oracle):
    c = wl.c3(side=16, gh=3, gw=3)
    out = engine.logml_grid(c["x"], c["y"], c["h"], c["w"], c["s"], chunk=4)
    for g in range(9):
        _, _, lmo = oracle.gp_fit(c["x"], c["y"], c["h"][g], c["w"][g], c["s"])
        assert abs(out[g] - lmo) <= RTOL * abs(lmo)
    # a hopeless point yields -inf, the others survive (bq.py:542-548)
    x = np.linspace(-5, 5, 128)
    y = wl.norm_logpdf(x)
    h = np.array([1.0, 1.0, 1.0])
    w = np.array([0.08, 4.0, 0.1])
    out = engine.logml_grid(x, y, h, w, 0.0)
    assert np.isfinite(out[0]) and np.isfinite(out[2]) and out[1] == -np.inf


def test_logml_grid_shares_factorisations_without_noise(engine, oracle):
    """s = 0: all output scales h of one length scale w come from ONE factorisation
    (chol(h^2 G) = h chol(G)); every point still equals its own fit."""
    rs = np.random.RandomState(5)
    n = 200
    dx = 10.0 / (n - 1)
    x = np.linspace(-5, 5, n) + rs.uniform(-dx / 4, dx / 4, n)
    y = wl.norm_logpdf(x)
    hs = np.array([0.3, 1.0, 2.5, 7.0])
    ws = np.array([0.8 * dx, 1.0 * dx, 1.15 * dx])
    H, W = np.meshgrid(hs, ws, indexing="ij")
    out = engine.logml_grid(x, y, H.ravel(), W.ravel(), 0.0)
    for g, (hh, ww) in enumerate(zip(H.ravel(), W.ravel())):
        _, _, lmo = oracle.gp_fit(x, y, hh, ww, 0.0)
        # no noise floor here: the quadratic form carries cond(K) eps in either evaluation
        tol = max(RTOL, 1e-15 * np.linalg.cond(oracle.gram(x, hh, ww, 0.0)))
        assert abs(out[g] - lmo) <= tol * abs(lmo)
    # a hopeless length scale fails for every h, the others survive
    W2 = W.copy()
    W2[:, 1] = 60 * dx
    out = engine.logml_grid(x, y, H.ravel(), W2.ravel(), 0.0)
    bad = np.isinf(out).reshape(H.shape)
    assert bad[:, 1].all() and not bad[:, [0, 2]].any()
    # 2-D points, vector w
    c = wl.c3(side=12, gh=3, gw=2)
    out = engine.logml_grid(c["x"], c["y"], c["h"], c["w"], 0.0)
    for g in range(len(out)):
        _, _, lmo = oracle.gp_fit(c["x"], c["y"], c["h"][g], c["w"][g], 0.0)
        tol = max(RTOL, 1e-15 * np.linalg.cond(oracle.gram(c["x"], c["h"][g], c["w"][g], 0.0)))
        assert abs(out[g] - lmo) <= tol * abs(lmo)


def test_batch_fit_predict(engine, oracle):
    probs = [3, 4, 5, 6, 7]
    c = wl.c5(probs, n=200, m=33)
    mean, var, logml, status = engine.batch_fit_predict(c["x"], c["y"], c["h"], c["w"] * 10,
                                                        c["s"], c["xo"])
    assert (status == 0).all()
    for i in range(len(probs)):
        Lo, ao, lmo = oracle.gp_fit(c["x"][i], c["y"][i], c["h"], c["w"] * 10, c["s"])
        mo, vo = oracle.gp_predict(c["x"][i], c["h"], c["w"] * 10, Lo, ao, c["xo"][i])
        assert relmax(mean[i], mo) < RTOL
        assert relmax(var[i], vo, scale=oracle.kernel_scale(1, c["h"], c["w"] * 10)) < RTOL
        assert abs(logml[i] - lmo) <= RTOL * abs(lmo)


@pytest.mark.parametrize("n", [600, 900])
def test_batch_fit_predict_mid_block(engine, oracle, n):
    """24 x (N = 900, M = 40): the batch regime with outer block 128 (recursive panels, results
    read off the border rows, most updates too small for the kernel that skips the border
    block) -- between the one-launch steps of small systems and the C5-sized batches; 24 x
    (N = 600): since round 3 still on the one-launch steps (a step's 24 x 66 workgroups fit
    the chip a few times over: auto_nb)."""
    probs = list(range(24))
    c = wl.c5(probs, n=n, m=40)
    mean, var, logml, status = engine.batch_fit_predict(c["x"], c["y"], c["h"], c["w"] * 3,
                                                        c["s"], c["xo"])
    assert (status == 0).all()
    for i in (0, 11, 23):
        Lo, ao, lmo = oracle.gp_fit(c["x"][i], c["y"][i], c["h"], c["w"] * 3, c["s"])
        mo, vo = oracle.gp_predict(c["x"][i], c["h"], c["w"] * 3, Lo, ao, c["xo"][i])
        assert relmax(mean[i], mo) < RTOL
        assert relmax(var[i], vo, scale=oracle.kernel_scale(1, c["h"], c["w"] * 3)) < RTOL
        assert abs(logml[i] - lmo) <= RTOL * abs(lmo)


def test_batched_workspace_is_reused_and_trimmed(engine):
    """The batched entry points keep their workspace in the context: the same call again
    (same shapes, other hyper-parameters), a call of another shape, and a call after
    trim() all give the results of a fresh evaluation."""
    c = wl.c3(side=16, gh=3, gw=3)
    a = engine.logml_grid(c["x"], c["y"], c["h"], c["w"], c["s"], chunk=4)
    b = engine.logml_grid(c["x"], c["y"], c["h"][::-1], c["w"][::-1], c["s"], chunk=4)
    assert np.array_equal(a, b[::-1])
    c5 = wl.c5([3, 4, 5], n=200, m=33)
    m1 = engine.batch_fit_predict(c5["x"], c5["y"], c5["h"], c5["w"] * 10, c5["s"], c5["xo"])
    a2 = engine.logml_grid(c["x"], c["y"], c["h"], c["w"], c["s"], chunk=4)
    assert np.array_equal(a, a2)
    engine.trim()
    m2 = engine.batch_fit_predict(c5["x"], c5["y"], c5["h"], c5["w"] * 10, c5["s"], c5["xo"])
    for u, v in zip(m1, m2):
        assert np.array_equal(u, v)
    engine.trim()


def test_plan_rerun_is_deterministic(engine):
    c = wl.c5([0, 1, 2], n=300, m=40)
    plan = engine.plan(3, 1, 300, 40)
    plan.set_inputs(c["x"], c["y"], c["xo"], c["h"], c["w"] * 10, c["s"])
    plan.run()
    a = plan.results()
    plan.run()
    b = plan.results()
    for u, v in zip(a, b):
        assert (u == v).all()
    plan.close()


# ---- golden fixtures ----------------------------------------------------------------
def test_golden_c1(engine):
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "c1_n32.npz"))
    mean, var, logml = engine.fit_predict(g["x"], g["y"], float(g["h"]), g["w"], float(g["s"]),
                                          g["xo"])
    assert relmax(mean, g["mean"]) < RTOL
    assert relmax(var, g["var"], scale=float(g["k0"])) < RTOL
    assert abs(logml - float(g["logml"])) <= RTOL * abs(float(g["logml"]))
    fit = engine.gp_fit(g["x"], g["y"], float(g["h"]), g["w"], float(g["s"]))
    assert relmax(fit.L(), g["L"]) < 1e-10
    assert relmax(fit.alpha(), g["alpha"]) < RTOL
    fit.close()


# ---- full BASELINE sizes through size-independent properties --------------------------
def _resid(K, L, rs, nvec=4):
    """|| L (L^T v) - K v || / || K v || on random vectors: O(n^2) per vector."""
    worst = 0.0
    for _ in range(nvec):
        v = rs.randn(K.shape[0])
        Kv = K.dot(v)
        worst = max(worst, np.linalg.norm(L.dot(L.T.dot(v)) - Kv) / np.linalg.norm(Kv))
    return worst


def test_c3_size_factorisation_property(engine):
    c = wl.c3()  # N = 4096, d = 2
    g = 200      # a mid-grid hyper-parameter point
    fit = engine.gp_fit(c["x"], c["y"], c["h"][g], c["w"][g], c["s"])
    K, L = fit.K(), fit.L()
    assert (K == K.T).all()
    assert _resid(K, L, np.random.RandomState(0)) < 1e-14 * 4096
    # log-ML identity from the factor: -1/2 |z|^2 - sum log L_ii - n/2 log 2pi
    z = fit.z()
    ref = -0.5 * z.dot(z) - np.log(np.diag(L)).sum() - 0.5 * 4096 * np.log(2 * np.pi)
    assert abs(fit.logml - ref) <= 1e-12 * abs(ref)
    # alpha solves K alpha = y
    assert np.linalg.norm(K.dot(fit.alpha()) - c["y"]) / np.linalg.norm(c["y"]) < 1e-9
    fit.close()


def test_c5_full_size_shard(engine, oracle):
    """BASELINE config 5 at full size: one rank's shard, 64 x (N = 2048, M = 256), through
    the resident batched plan (outer block 256, batch in blockIdx.z) -- what bench.py
    --workload c5 times.  Three problems against the oracle at the 1e-10 bar, all 64
    through the single-problem entry point's invariants."""
    B = 64
    c = wl.c5(list(range(B)))
    plan = engine.plan(B, 1, 2048, 256)
    plan.set_inputs(c["x"], c["y"], c["xo"], c["h"], c["w"], c["s"])
    plan.run()
    mean, var, logml, status = plan.results()
    plan.close()
    assert (status == 0).all() and np.isfinite(logml).all()
    k0 = oracle.kernel_scale(1, c["h"], c["w"])
    assert (var > -1e-10 * k0).all() and (var <= k0 * (1 + 1e-12)).all()
    for i in (0, 31, 63):
        Lo, ao, lmo = oracle.gp_fit(c["x"][i], c["y"][i], c["h"], c["w"], c["s"])
        mo, vo = oracle.gp_predict(c["x"][i], c["h"], c["w"], Lo, ao, c["xo"][i])
        assert relmax(mean[i], mo) < RTOL
        assert relmax(var[i], vo, scale=k0) < RTOL
        assert abs(logml[i] - lmo) <= RTOL * abs(lmo)
    # the same problems one at a time (another blocking, another launch sequence)
    for i in (7, 40):
        m1, v1, l1 = engine.fit_predict(c["x"][i], c["y"][i], c["h"], c["w"], c["s"], c["xo"][i])
        assert relmax(mean[i], m1) < RTOL and relmax(var[i], v1, scale=k0) < RTOL
        assert abs(logml[i] - l1) <= RTOL * abs(l1)


def test_c3_full_size_grid(engine, oracle):
    """BASELINE config 3 at full size: the 20 x 20 (h, w) log-ML grid at N = 4096, d = 2 in
    chunks of 100 batched factorisations.  Corners and centre against the oracle and
    against the single-fit path with its log-ML identity; every point finite."""
    c = wl.c3()
    lm = engine.logml_grid(c["x"], c["y"], c["h"], c["w"], c["s"], chunk=100)
    assert lm.shape == (400,) and np.isfinite(lm).all()
    for g in (0, 19, 210, 380, 399):
        _, _, lmo = oracle.gp_fit(c["x"], c["y"], c["h"][g], c["w"][g], c["s"])
        assert abs(lm[g] - lmo) <= RTOL * abs(lmo), (g, lm[g], lmo)
        fit = engine.gp_fit(c["x"], c["y"], c["h"][g], c["w"][g], c["s"])
        assert abs(lm[g] - fit.logml) <= RTOL * abs(fit.logml)
        z, L = fit.z(), fit.L()
        ref = -0.5 * z.dot(z) - np.log(np.diag(L)).sum() - 0.5 * 4096 * np.log(2 * np.pi)
        assert abs(lm[g] - ref) <= RTOL * abs(ref)
        fit.close()
    # h enters only through K = h^2 G + s^2 I: monotone pieces are not asserted, but the
    # grid must vary smoothly -- no point is a copy of its neighbour
    assert len(np.unique(lm)) == 400
    engine.trim()


def test_c4_size_cholesky_property(engine):
    """N = 16384 (the MFMA roofline size) on device-resident data."""
    import ctypes as C
    c = wl.c4()
    n = 16384
    x = np.ascontiguousarray(c["x"])
    xd = engine.alloc(8 * n)
    Kd = engine.alloc(8 * n * n)
    info = engine.alloc(64)
    engine.upload(xd, x)
    lib, ctx = engine._lib, engine._ctx
    w = np.ascontiguousarray(c["w"])
    from bayesian_quadrature_amd import _lib as L_
    engine._check(lib.bq_gram_gauss_dev(ctx, xd, 1, n, c["h"], L_.dptr(w), c["s"], Kd, n))
    K = np.empty((n, n), order="F")
    engine.download(K, Kd)
    engine._check(lib.bq_potrf_dev(ctx, Kd, n, n, info))
    Lf = np.empty((n, n), order="F")
    engine.download(Lf, Kd)
    hinfo = np.zeros(1, dtype=np.int32)
    engine.download(hinfo, info)
    engine.free(xd), engine.free(Kd), engine.free(info)
    assert hinfo[0] == 0
    Lf = np.tril(Lf)
    assert _resid(K, Lf, np.random.RandomState(1), nvec=2) < 1e-14 * n
    # spot-check Gram entries against the closed form
    rs = np.random.RandomState(2)
    i, j = rs.randint(0, n, 1000), rs.randint(0, n, 1000)
    ref = c["h"] ** 2 / (np.sqrt(2 * np.pi) * w[0]) * np.exp(-(x[i] - x[j]) ** 2 / (2 * w[0] ** 2)) \
        + (i == j) * c["s"] ** 2
    assert np.allclose(K[i, j], ref, rtol=1e-13, atol=0)


def _potrf_dev(eng, x, c, n, nb=0, la=True):
    """Gram + potrf of the C4-style system of size n on device-resident data -> (K, L)."""
    from bayesian_quadrature_amd import _lib as L_
    lib, ctx = eng._lib, eng._ctx
    w = np.ascontiguousarray(c["w"])
    xd, Kd, info = eng.alloc(8 * n), eng.alloc(8 * n * n), eng.alloc(64)
    try:
        eng.upload(xd, x)
        eng.set_block(nb)
        eng.set_lookahead(la)
        eng._check(lib.bq_gram_gauss_dev(ctx, xd, 1, n, c["h"], L_.dptr(w), c["s"], Kd, n))
        K = np.empty((n, n), order="F")
        eng.download(K, Kd)
        eng._check(lib.bq_potrf_dev(ctx, Kd, n, n, info))
        Lf = np.empty((n, n), order="F")
        eng.download(Lf, Kd)
        hinfo = np.zeros(1, dtype=np.int32)
        eng.download(hinfo, info)
        assert hinfo[0] == 0
    finally:
        eng.set_block(0)
        eng.set_lookahead(True)
        eng.free(xd), eng.free(Kd), eng.free(info)
    return K, np.tril(Lf)


@pytest.mark.parametrize("env,nb,la", [
    ({}, 0, True),                       # shipped: LDS-staged update, MFMA panel solve
    ({}, 128, False),                    # k = 128 chunks, sequential launches
    ({}, 512, True),                     # k = 512
    ({}, 192, True),                     # k = 192: not a multiple of 32 x 4 -> both kernels
    ({"BQ_GEMM_LDS": "0"}, 0, True),     # the reference variant: register-streaming kernel
    ({"BQ_GEMM_LDS": "0"}, 256, False),
    ({"BQ_LA_MIN": "0"}, 0, True),       # look-ahead to the last panel (no hand-over)
    ({"BQ_LA_MIN": "0"}, 128, True),
    ({"BQ_GEMM_TILE": "64"}, 0, True),   # the 64 x 64 LDS tile wherever it can run
    ({"BQ_GEMM_TILE": "64"}, 320, False),
    ({"BQ_GEMM_TILE": "128"}, 0, False),  # the 128 x 128 tile wherever it can run
])
def test_trailing_update_variants_agree(engine, env, nb, la):
    """Every blocking / scheduling of the factorisation, and the one reference kernel variant
    kept beside the shipped trailing update (BQ_GEMM_LDS=0, read when a context is created),
    gives the same factor of an N = 4480 system -- a
    size that is a multiple of 64 but not of 128, large enough for the 128 x 128 tiles."""
    import os
    from bayesian_quadrature_amd import Engine
    n = 4480
    c = wl.c4(n)
    x = np.ascontiguousarray(c["x"])
    K, Lref = _potrf_dev(engine, x, c, n)
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        eng = Engine(0)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    try:
        K2, L2 = _potrf_dev(eng, x, c, n, nb, la)
    finally:
        eng.close()
    assert np.array_equal(K, K2)
    assert _resid(K, L2, np.random.RandomState(3), nvec=2) < 1e-14 * n
    # the variants differ only in summation order
    assert np.max(np.abs(L2 - Lref)) <= 1e-11 * np.max(np.abs(Lref))


# ---- closed-form integrals and BQ moments on the device ------
MU1, COV1 = np.array([0.3]), np.array([[10.0]])


@pytest.mark.parametrize("n", [1, 9, 63, 300, 1000])
def test_device_integrals_1d(engine, oracle, n):
    rs = np.random.RandomState(n)
    x = np.sort(rs.uniform(-5, 5, n))
    x2 = np.sort(rs.uniform(-6, 4, max(1, n // 2 + 3)))
    w1, w2 = np.array([1.3]), np.array([2.0])
    assert relmax(engine.int_K(x, 0.2, w1, MU1, COV1), oracle.int_K(x, 0.2, w1, MU1, COV1)) < 1e-13
    assert relmax(engine.int_K1_K2(x, x2, 0.2, w1, 15.0, w2, MU1, COV1),
                  oracle.int_K1_K2(x, x2, 0.2, w1, 15.0, w2, MU1, COV1)) < 1e-12
    assert relmax(engine.int_int_K1_K2_K1(x, 0.2, w1, 15.0, w2, MU1, COV1),
                  oracle.int_int_K1_K2_K1(x, 0.2, w1, 15.0, w2, MU1, COV1)) < 1e-12
    assert relmax(engine.int_int_K1_K2(x, 0.2, w1, 15.0, w2, MU1, COV1),
                  oracle.int_int_K1_K2(x, 0.2, w1, 15.0, w2, MU1, COV1)) < 1e-13


@pytest.mark.parametrize("d", [2, 3, 8])
def test_device_integrals_nd(engine, oracle, d):
    rs = np.random.RandomState(d)
    x = rs.uniform(-2, 2, (d, 70))
    x2 = rs.uniform(-2, 2, (d, 41))
    w1, w2 = rs.uniform(0.5, 1.5, d), rs.uniform(0.5, 1.5, d)
    A = rs.rand(d, d)
    cov = A.dot(A.T) + d * np.eye(d)
    mu = rs.uniform(-0.5, 0.5, d)
    assert relmax(engine.int_K(x, 0.7, w1, mu, cov), oracle.int_K(x, 0.7, w1, mu, cov)) < 1e-12
    if d <= 4:   # the oracle's small-matrix buffers stop at 2d = 8
        assert relmax(engine.int_K1_K2(x, x2, 0.7, w1, 1.2, w2, mu, cov),
                      oracle.int_K1_K2(x, x2, 0.7, w1, 1.2, w2, mu, cov)) < 1e-11
    assert relmax(engine.int_int_K1_K2_K1(x, 0.7, w1, 1.2, w2, mu, cov),
                  oracle.int_int_K1_K2_K1(x, 0.7, w1, 1.2, w2, mu, cov)) < 1e-11
    assert relmax(engine.int_int_K1_K2(x, 0.7, w1, 1.2, w2, mu, cov),
                  oracle.int_int_K1_K2(x, 0.7, w1, 1.2, w2, mu, cov)) < 1e-12


@pytest.mark.parametrize("n,d", [(9, 1), (300, 1), (1000, 1), (70, 2)])
def test_device_integrals_same(engine, n, d):
    """The reference's "_same" contracts (tests/test_gauss_c.py:40-56,107-143,173-209,237-272,
    300-335: twenty calls on the same inputs, compared with ==) for the device integrals and
    for the fused V(Z): no atomics, no launch-order-dependent summation."""
    rs = np.random.RandomState(100 * n + d)
    if d == 1:
        x, x2 = np.sort(rs.uniform(-5, 5, n)), np.sort(rs.uniform(-6, 4, n // 2 + 3))
        mu, cov = MU1, COV1
    else:
        x, x2 = rs.uniform(-2, 2, (d, n)), rs.uniform(-2, 2, (d, n // 2 + 3))
        A = rs.rand(d, d)
        mu, cov = rs.uniform(-0.5, 0.5, d), A.dot(A.T) + d * np.eye(d)
    w1, w2 = rs.uniform(0.9, 1.5, d), rs.uniform(1.6, 2.2, d)
    calls = {
        "int_K": lambda: engine.int_K(x, 0.2, w1, mu, cov),
        "int_K1_K2": lambda: engine.int_K1_K2(x, x2, 0.2, w1, 15.0, w2, mu, cov),
        "int_int_K1_K2_K1": lambda: engine.int_int_K1_K2_K1(x, 0.2, w1, 15.0, w2, mu, cov),
        "int_int_K1_K2": lambda: engine.int_int_K1_K2(x, 0.2, w1, 15.0, w2, mu, cov),
    }
    for name, f in calls.items():
        first = np.array(f(), copy=True)
        assert np.all(np.isfinite(first)), name
        for _ in range(19):
            assert np.array_equal(np.asarray(f()), first), name


@pytest.mark.parametrize("ns,nc", [(9, 2), (200, 17), (700, 64)])
def test_device_Z_var_same(engine, ns, nc):
    """bq_bq_Z_var (and Z_mean) twenty times on the same resident fits: == (the reference's
    test_bq_object.py:136-141 asserts == for Z_mean; its gauss_c "_same" tests for the terms)."""
    rs = np.random.RandomState(ns)
    dx = 10.0 / (ns - 1)
    xs = np.linspace(-5, 5, ns) + rs.uniform(-dx / 4, dx / 4, ns)
    ls = np.exp(wl.norm_logpdf(xs))
    xc = np.sort(rs.uniform(-6, 6, nc))
    xsc = np.concatenate([xs, xc])
    lsc = np.concatenate([ls, np.exp(wl.norm_logpdf(xc))])
    f1 = engine.gp_fit(xs, np.log(ls), 15.0, 1.5 * dx, 1e-3)
    f2 = engine.gp_fit(xsc, lsc, 0.2, 1.5 * dx, 1e-3)
    zv = [engine.Z_var(f1, f2, MU1, COV1) for _ in range(20)]
    zm = [engine.Z_mean(f2, MU1, COV1) for _ in range(20)]
    assert np.isfinite(zv[0]) and np.isfinite(zm[0])
    assert all(v == zv[0] for v in zv)
    assert all(m == zm[0] for m in zm)
    f1.close(), f2.close()


def test_device_moments_known_answers(engine, oracle):
    """E[Z] and V(Z) of the reference's notebook fixture from device-resident fits."""
    from fixture_chain import build_chain, known_answers
    c = build_chain(lambda x, y, h, w, s: oracle.gp_fit(x, y, h, w, s),
                    lambda x, h, w, L, a, xo: oracle.gp_predict(x, h, w, L, a, xo, want_var=False),
                    oracle.filter_candidates)
    f1 = engine.gp_fit(c["xs"], np.log(c["ls"]), c["h1"], c["w1"], 0.0)
    f2 = engine.gp_fit(c["xsc"], c["lsc"], c["h2"], c["w2"], 0.0)
    exp = known_answers()["expected"]
    assert abs(engine.Z_mean(f2, c["mu"], c["cov"]) - exp["Z_mean"]["value"]) < 1e-12
    zv = engine.Z_var(f1, f2, c["mu"], c["cov"])
    assert abs(zv - exp["Z_var"]["value"]) / exp["Z_var"]["value"] < 1e-7
    f1.close(), f2.close()


@pytest.mark.parametrize("ns,nc", [(30, 5), (200, 17), (700, 64)])
def test_device_moments_vs_oracle(engine, oracle, ns, nc):
    rs = np.random.RandomState(ns)
    dx = 10.0 / (ns - 1)
    xs = np.linspace(-5, 5, ns) + rs.uniform(-dx / 4, dx / 4, ns)
    ls = np.exp(wl.norm_logpdf(xs))
    xc = np.sort(rs.uniform(-6, 6, nc))
    h1, w1, s1 = 15.0, 1.5 * dx, 1e-3
    h2, w2, s2 = 0.2, 1.5 * dx, 1e-3
    L1, a1, _ = oracle.gp_fit(xs, np.log(ls), h1, w1, s1)
    lc = np.exp(oracle.gp_predict(xs, h1, w1, L1, a1, xc, want_var=False))
    xsc, lsc = np.concatenate([xs, xc]), np.concatenate([ls, lc])
    L2, a2, _ = oracle.gp_fit(xsc, lsc, h2, w2, s2)
    f1 = engine.gp_fit(xs, np.log(ls), h1, w1, s1)
    f2 = engine.gp_fit(xsc, lsc, h2, w2, s2)
    zm = oracle.Z_mean(xsc, a2, h2, w2, MU1, COV1)
    assert abs(engine.Z_mean(f2, MU1, COV1) - zm) <= 1e-10 * abs(zm)
    # V(Z) is a difference of two nearly equal terms: compare at the scale of the terms
    I3 = oracle.int_int_K1_K2_K1(xsc, h2, w2, h1, w1, MU1, COV1)
    t1 = a2.dot(I3).dot(a2)
    zv = oracle.Z_var(xs, xsc, a2, L1, h2, w2, h1, w1, MU1, COV1)
    assert abs(engine.Z_var(f1, f2, MU1, COV1) - zv) <= 1e-10 * abs(t1)
    # resident solve
    b = rs.randn(ns)
    assert relmax(f1.solve(b), oracle.cho_solve(L1, b)) < 1e-9
    B = rs.randn(ns, 3)
    assert relmax(f1.solve(B), oracle.cho_solve(L1, B)) < 1e-9
    # the single-vector sweeps replay a captured launch chain: a second call and a call after
    # a refit (same buffers, new factor) must follow the data, not the capture
    assert relmax(f1.solve(2.0 * b), 2.0 * oracle.cho_solve(L1, b)) < 1e-9
    f1.refit(h1, 0.8 * w1, s1)
    L1b, _, _ = oracle.gp_fit(xs, np.log(ls), h1, 0.8 * w1, s1)
    assert relmax(f1.solve(b), oracle.cho_solve(L1b, b)) < 1e-9
    f1.close(), f2.close()


# ---- batched active-sampling systems ---------------------------
@pytest.mark.parametrize("ns,nc,M", [(9, 2, 25), (60, 8, 40), (300, 20, 64)])
def test_esm_batch_vs_reference_recipe(engine, oracle, ns, nc, M):
    from engine_double import EngineDouble
    rs = np.random.RandomState(ns + M)
    from bayesian_quadrature_amd import bq_c
    xs = np.linspace(-5, 5, ns)
    dx = 10.0 / (ns - 1)
    # candidates as BQ._choose_candidates makes them: spaced from the samples and
    # from each other (here by 0.4 dx so that some survive on a fine grid)
    xc = rs.uniform(-6, 6, 4 * nc)
    bq_c.filter_candidates(xc, xs, 0.4 * dx)
    xc = np.sort(xc[~np.isnan(xc)])[:nc]
    x_sc = np.concatenate([xs, xc])
    l_sc = np.exp(wl.norm_logpdf(x_sc))
    # candidates: random, plus some within candidate_thresh of a candidate point and one
    # nearly on top of a sample (a near-singular bordered system)
    x_a = np.concatenate([rs.uniform(-7, 7, M - 3), xc[:2] + 0.05, [xs[3] + 2e-4]])
    h, w, thresh = 0.2, 1.04 * dx, 0.5     # w ~ dx: the conditioning regime of the configs
    ref = EngineDouble(oracle).esm_batch(x_sc, l_sc, ns, x_a, h, w, thresh, MU1, COV1)
    got = engine.esm_batch(x_sc, l_sc, ns, x_a, h, w, thresh, MU1, COV1)
    assert (got[2] == 0).all() and (ref[2] == 0).all()
    # the solved coefficients are conditioned like K_l^-1: compare at cond * eps
    K = oracle.gram_cross(x_sc, x_sc, h, w)
    tol = max(1e-10, 50 * np.linalg.cond(K + 1e-4 * K.max() * np.eye(len(x_sc))) * 2.2e-16)
    scale = max(np.abs(ref[0]).max(), np.abs(ref[1]).max())
    assert np.max(np.abs(got[0] - ref[0])) <= tol * scale
    assert np.max(np.abs(got[1] - ref[1])) <= tol * scale


@pytest.mark.parametrize("ns,nc,M", [(9, 2, 25), (60, 8, 40), (300, 20, 64), (1000, 24, 300)])
def test_esm_border_vs_refactorisation(engine, oracle, ns, nc, M):
    """bq_esm_border -- the bordered update of gp_l's resident factor, O(n^2) per candidate
     -- against the reference's recipe, which re-factors the jittered
    (nsc + 1)^2 matrix for every candidate (bq.py:463-480): the oracle for the smaller
    sizes, the batched device refactorisation bq_esm_batch for all.  Same inputs as
    test_esm_batch_vs_reference_recipe, including candidates inside the jitter radius of
    one or two candidate points and one nearly on top of a sample."""
    from engine_double import EngineDouble
    from bayesian_quadrature_amd import bq_c
    rs = np.random.RandomState(ns + M)
    xs = np.linspace(-5, 5, ns)
    dx = 10.0 / (ns - 1)
    xc = rs.uniform(-6, 6, 4 * nc)
    bq_c.filter_candidates(xc, xs, 0.4 * dx)
    xc = np.sort(xc[~np.isnan(xc)])[:nc]
    x_sc = np.concatenate([xs, xc])
    l_sc = np.exp(wl.norm_logpdf(x_sc))
    x_a = np.concatenate([rs.uniform(-7, 7, M - 4), xc[:2] + 0.05, [0.5 * (xc[0] + xc[-1])],
                          [xs[3] + 2e-4]])
    h, w, thresh = 0.2, 1.04 * dx, 0.5
    fit = engine.gp_fit(x_sc, l_sc, h, w, 0.0)
    got = engine.esm_border(fit, ns, x_a, thresh, MU1, COV1)
    chk = engine.esm_batch(x_sc, l_sc, ns, x_a, h, w, thresh, MU1, COV1)
    assert (got[2] == 0).all() and (chk[2] == 0).all()
    K = oracle.gram_cross(x_sc, x_sc, h, w)
    tol = max(1e-10, 50 * np.linalg.cond(K) * 2.2e-16)
    scale = max(np.abs(chk[0]).max(), np.abs(chk[1]).max())
    assert np.max(np.abs(got[0] - chk[0])) <= tol * scale
    assert np.max(np.abs(got[1] - chk[1])) <= tol * scale
    if ns <= 300:
        ref = EngineDouble(oracle).esm_batch(x_sc, l_sc, ns, x_a, h, w, thresh, MU1, COV1)
        assert np.max(np.abs(got[0] - ref[0])) <= tol * scale
        assert np.max(np.abs(got[1] - ref[1])) <= tol * scale
    # a noisy gp_l has no Kxoxo factor to update: refused, the caller re-factors
    noisy = engine.gp_fit(x_sc, l_sc, h, w, 1e-3)
    with pytest.raises(ValueError):
        engine.esm_border(noisy, ns, x_a, thresh, MU1, COV1)
    noisy.close()
    fit.close()


def test_esm_border_bounded_host_tail(engine, oracle):
    """The trailing block of bq_esm_border lives on the host (one nt^3 / 3 Cholesky per distinct
    set of jittered candidates).  Beyond nt = 320 rows or 64 distinct sets the call hands all
    candidates to the batched device refactorisation instead: same answers either way."""
    rs = np.random.RandomState(77)
    for ns, nc0, M, thresh in ((40, 360, 30, 0.02), (64, 215, 150, 0.06)):
        xs = np.linspace(-5, 5, ns)
        # candidates on a jittered grid (spacing ~ w: the conditioning regime of the configs),
        # none closer than half a spacing to a sample
        dc = 11.0 / nc0
        xc = -5.5 + dc * (np.arange(nc0) + 0.5 + rs.uniform(-0.2, 0.2, nc0))
        xc = xc[np.abs(xc[:, None] - xs[None, :]).min(axis=1) > 0.5 * dc]
        nc = xc.shape[0]
        assert (ns == 40 and ns + nc - ns // 64 * 64 > 320) or (ns == 64 and nc > 150)
        x_sc = np.concatenate([xs, xc])
        l_sc = np.exp(wl.norm_logpdf(x_sc))
        x_a = np.sort(rs.uniform(-6, 6, M))
        h, w = 0.2, 1.1 * dc
        fit = engine.gp_fit(x_sc, l_sc, h, w, 0.0)
        got = engine.esm_border(fit, ns, x_a, thresh, MU1, COV1)
        chk = engine.esm_batch(x_sc, l_sc, ns, x_a, h, w, thresh, MU1, COV1)
        if ns == 64:   # many candidates within the radius: more than 64 distinct close sets
            sets = {tuple(np.nonzero(np.abs(xc - a) < thresh)[0]) for a in x_a}
            assert len(sets) > 64
        assert (got[2] == chk[2]).all()
        ok = chk[2] == 0
        assert ok.sum() > M // 2
        assert np.array_equal(got[0][ok], chk[0][ok]) and np.array_equal(got[1][ok], chk[1][ok])
        fit.close()


def _ld_chol_solve(K, B):
    """Cholesky solve in x87 extended precision (numpy longdouble), K: (n, n), B: (n, m)."""
    n = K.shape[0]
    L = np.zeros_like(K)
    for j in range(n):
        L[j, j] = np.sqrt(K[j, j] - np.dot(L[j, :j], L[j, :j]))
        L[j + 1:, j] = (K[j + 1:, j] - L[j + 1:, :j].dot(L[j, :j])) / L[j, j]
    Y = np.array(B, dtype=np.longdouble, copy=True)
    for j in range(n):
        Y[j] = (Y[j] - L[j, :j].dot(Y[:j])) / L[j, j]
    for j in range(n - 1, -1, -1):
        Y[j] = (Y[j] - L[j + 1:, j].dot(Y[j + 1:])) / L[j, j]
    return Y


@pytest.mark.parametrize("ns,nc,wfac", [(9, 2, 1.04), (60, 8, 1.04), (60, 8, 1.6)])
def test_acquisition_and_posterior_vs_extended_precision(engine, oracle, ns, nc, wfac):
    """The loose bars of the acquisition tests are conditioning, not the HIP path: against a
    truth computed in x87 extended precision (64-bit mantissa) -- the reference's recipe,
    bq.py:463-480 and bq_c.pyx:455-470, for every candidate; the posterior mean / variance
    formulas of SURVEY appendix B -- the device results err no more than a small multiple of
    what the CPU oracle errs in fp64 (25x for the acquisition coefficients, whose Schur
    complement cancels seven digits in the third case; 10x for the posterior).  This test is
    what showed that explicit inverses of 64- to 512-wide diagonal blocks are NOT good enough
    for the acquisition borders (400x the oracle's error): bq_esm_border sweeps with the
    16 x 16 block inverses."""
    from engine_double import EngineDouble
    from bayesian_quadrature_amd import bq_c
    ld = np.longdouble
    rs = np.random.RandomState(7 * ns + nc)
    xs = np.linspace(-5, 5, ns)
    dx = 10.0 / (ns - 1)
    xc = rs.uniform(-6, 6, 6 * nc)
    bq_c.filter_candidates(xc, xs, 0.4 * dx)
    xc = np.sort(xc[~np.isnan(xc)])[:nc]
    x_sc = np.concatenate([xs, xc])
    l_sc = np.exp(wl.norm_logpdf(x_sc))
    x_a = np.concatenate([rs.uniform(-7, 7, 20), xc[:1] + 0.05, [xs[3] + 2e-3]])
    h, w, thresh = 0.2, wfac * dx, 0.5
    mu, cov = MU1, COV1

    def kern(a, b):
        a, b = np.asarray(a, dtype=ld), np.asarray(b, dtype=ld)
        return ld(h) ** 2 / (np.sqrt(2 * ld(np.pi)) * ld(w)) * np.exp(
            -(a[:, None] - b[None]) ** 2 / (2 * ld(w) ** 2))

    def intk(a):  # int K(a, x) N(x | mu, cov) dx = h^2 N(a | mu, w^2 + cov)
        v = ld(w) ** 2 + ld(cov[0, 0])
        a = np.asarray(a, dtype=ld)
        return ld(h) ** 2 / np.sqrt(2 * ld(np.pi) * v) * np.exp(-(a - ld(mu[0])) ** 2 / (2 * v))

    k0 = float(kern([0.0], [0.0])[0, 0])
    eps = np.finfo(np.float64).eps
    tA, tB = [], []
    for xa in x_a:
        x_sca = np.concatenate([x_sc, [xa]])
        K = kern(x_sca, x_sca)
        close = np.abs(xc - xa) < thresh
        j1 = max(eps, k0) * 1e-4 if close.any() else 0.0
        idx = np.nonzero(close)[0] + ns
        K[idx, idx] += ld(j1)
        K[-1, -1] += ld(max(eps, k0 + j1) * 1e-4)
        A = _ld_chol_solve(K, intk(x_sca)[:, None])[:, 0]
        tA.append(A[-1])
        tB.append(A[:-1].dot(l_sc.astype(ld)))
    tA, tB = np.array(tA), np.array(tB)
    fit = engine.gp_fit(x_sc, l_sc, h, w, 0.0)
    got = {"border": engine.esm_border(fit, ns, x_a, thresh, mu, cov),
           "batch": engine.esm_batch(x_sc, l_sc, ns, x_a, h, w, thresh, mu, cov)}
    ref = EngineDouble(oracle).esm_batch(x_sc, l_sc, ns, x_a, h, w, thresh, mu, cov)
    scale = float(max(np.abs(tA).max(), np.abs(tB).max()))

    def err(r):
        return float(max(np.abs(r[0].astype(ld) - tA).max(), np.abs(r[1].astype(ld) - tB).max()))

    e_ref = err(ref)
    for name, r in got.items():
        assert (r[2] == 0).all()
        assert err(r) <= 25 * e_ref + 1e-13 * scale, (name, err(r), e_ref, scale)
    # posterior mean / variance of gp_l at the candidates
    Kss = kern(x_sc, x_sc)
    kx = kern(x_sc, x_a)
    sol = _ld_chol_solve(Kss, np.concatenate([l_sc.astype(ld)[:, None], kx], axis=1))
    t_mean = kx.T.dot(sol[:, 0])
    t_var = ld(k0) - np.einsum("ij,ij->j", kx, sol[:, 1:])
    m, v, _ = fit.predict(x_a)
    Lo, ao, _ = oracle.gp_fit(x_sc, l_sc, h, w, 0.0)
    mo, vo = oracle.gp_predict(x_sc, h, w, Lo, ao, x_a)
    em, eo = np.abs(m.astype(ld) - t_mean).max(), np.abs(mo.astype(ld) - t_mean).max()
    ev, evo = np.abs(v.astype(ld) - t_var).max(), np.abs(vo.astype(ld) - t_var).max()
    assert float(em) <= 10 * float(eo) + 1e-14 * float(np.abs(t_mean).max())
    assert float(ev) <= 10 * float(evo) + 1e-14 * k0
    fit.close()


def test_bq_expected_moments_gpu_vs_double(engine, oracle):
    """The whole BQ acquisition path: HIP engine against the oracle-backed double."""
    import bayesian_quadrature_amd as pkg
    from bayesian_quadrature_amd import engine as eng_mod
    from engine_double import EngineDouble
    import scipy.stats
    x = np.linspace(-5, 5, 9)
    opts = dict(n_candidate=10, x_mean=0.0, x_var=10.0, candidate_thresh=0.5,
                optim_method="L-BFGS-B", kernel=pkg.GaussianKernel)
    x_a = np.concatenate([np.linspace(-8, 8, 37), x[:2], [np.nextafter(x[4], 9)]])
    res = {}
    saved = dict(eng_mod._engines)
    try:
        for name, eng in (("double", EngineDouble(oracle)), ("gpu", engine)):
            eng_mod._engines.clear()
            eng_mod.set_engine(eng, 0)
            np.random.seed(8728)
            bq = pkg.BQ(x, scipy.stats.norm.pdf(x), **opts)
            bq.init(params_tl=(15, 2, 0), params_l=(0.2, 1.3, 0))
            res[name] = (bq.Z_mean(), bq.Z_var(), bq.expected_squared_mean_and_mean(x_a),
                         bq.expected_Z_var(x_a), bq.l_mean(x_a), bq.l_var(x_a))
    finally:
        eng_mod._engines.clear()
        eng_mod._engines.update(saved)
    d, g = res["double"], res["gpu"]
    assert abs(d[0] - g[0]) <= 1e-10 * abs(d[0])
    assert abs(d[1] - g[1]) <= 1e-6 * abs(d[1])     # cancellation-limited, see DESIGN.md
    assert np.allclose(d[2], g[2], rtol=1e-8, atol=1e-14)
    assert np.allclose(d[3], g[3], rtol=1e-8, atol=1e-9 * d[0] ** 2)
    assert np.allclose(d[4], g[4], rtol=1e-9, atol=1e-13)
    assert np.allclose(d[5], g[5], rtol=1e-6, atol=1e-13)


@pytest.mark.parametrize("n,M", [(6144, 200), (16384, 256)])
def test_one_shot_plan_vs_resident_fit_large(engine, n, M):
    """N = 6144, M = 200 and N = 16384, M = 256 (the largest size of bench.py's
    fit_posterior_ms_at_n) through the one-shot bordered plan -- the two-stream look-ahead sweep
    whose bulk updates skip the border x border block, results read off the border rows -- and
    through a resident fit + bq_gp_predict (full updates of its own system, row sweeps with
    the explicit block inverses): two different routes to the same posterior and log-ML."""
    c = wl.c4(n)
    rs = np.random.RandomState(11)
    y = wl.norm_logpdf(c["x"]) + 0.01 * rs.randn(n)
    xo = np.sort(rs.uniform(-5.2, 5.2, M))
    m1, v1, lm1 = engine.fit_predict(c["x"], y, c["h"], c["w"], c["s"], xo)
    fit = engine.gp_fit(c["x"], y, c["h"], c["w"], c["s"])
    m2, v2, _ = fit.predict(xo)
    k0 = c["h"] ** 2 / (np.sqrt(2 * np.pi) * float(np.atleast_1d(c["w"])[0]))
    assert relmax(m1, m2) < 1e-10
    assert np.abs(v1 - v2).max() / k0 < 1e-10
    assert abs(lm1 - fit.logml) <= 1e-12 * abs(fit.logml)
    fit.close()


def test_large_fit_properties(engine):
    """N = 8192 (wide outer block, look-ahead) through the GP object path: the factor,
    z, alpha and the log-ML must satisfy their defining identities."""
    n = 8192
    c = wl.c4(n)
    rs = np.random.RandomState(4)
    y = wl.norm_logpdf(c["x"]) + 0.01 * rs.randn(n)
    fit = engine.gp_fit(c["x"], y, c["h"], c["w"], c["s"])
    L, z, alpha = fit.L(), fit.z(), fit.alpha()
    K = fit.K()
    assert _resid(K, L, rs, nvec=2) < 1e-14 * n
    assert np.linalg.norm(L.dot(z) - y) / np.linalg.norm(y) < 1e-12
    assert np.linalg.norm(K.dot(alpha) - y) / np.linalg.norm(y) < 1e-10
    ref = -0.5 * z.dot(z) - np.log(np.diag(L)).sum() - 0.5 * n * np.log(2 * np.pi)
    assert abs(fit.logml - ref) <= 1e-12 * abs(ref)
    # posterior at the samples reproduces y up to the noise level; variance is tiny there
    m, v, _ = fit.predict(c["x"][::64])
    assert np.max(np.abs(m - y[::64])) < 1e-3
    assert (v > -1e-9).all() and v.max() < 1e-3 * fit.K()[0, 0]
    fit.close()


# ---- the sizes bench.py prints rooflines for ---------------------
@pytest.mark.parametrize("n", [4096, 16384])
def test_fit_solve_at_benched_sizes(engine, n):
    """``rooflines.cho_solve_n{4096,16384}_rhs{1,256}`` of bench.py: ``fit.solve`` on the
    resident factor of the C4-style problem with ONE right-hand side (GEMV sweeps, 512-column
    steps from a hipGraph) and with 256 (wide row sweeps on the MFMA kernels), against LAPACK's
    ``dpotrs`` on the downloaded factor (linalg_c.pyx:96-179 semantics): forward error <= 1e-10
    relative, and the residual ``K X - B`` with the device's own Gram matrix."""
    import scipy.linalg as sla
    c = wl.c4(n)
    y = wl.norm_logpdf(c["x"])
    fit = engine.gp_fit(c["x"], y, c["h"], c["w"], c["s"])
    rs = np.random.RandomState(n)
    L = fit.L()
    K = fit.K()
    b1 = rs.randn(n)
    B = np.asfortranarray(rs.randn(n, 256))
    x1 = fit.solve(b1)
    X = fit.solve(B)
    fit.close()
    Lt = np.tril(L)
    r1 = sla.cho_solve((Lt, True), b1)
    R = sla.cho_solve((Lt, True), B)
    assert relmax(x1, r1) < 1e-10
    assert relmax(X, R) < 1e-10
    # column by column: no column may hide behind the largest one
    colerr = np.abs(X - R).max(axis=0) / np.abs(R).max(axis=0)
    assert colerr.max() < 1e-10
    assert np.abs(K.dot(x1) - b1).max() < 1e-11 * np.abs(K).max() * np.abs(x1).max() * np.sqrt(n)
    assert np.abs(K.dot(X) - B).max() < 1e-11 * np.abs(K).max() * np.abs(X).max() * np.sqrt(n)


def test_predict_m1000_at_benched_size(engine, oracle):
    """``rooflines.predict_mean_var_n1024_m{256,1000}``: ``fit.predict`` at C2's N = 1024
    with M = 1000 points (the reference's own plotting grid, bq.py:993) against the oracle."""
    c = wl.c2()
    fit = engine.gp_fit(c["x"], c["y"], c["h"], c["w"], c["s"])
    Lo, ao, _ = oracle.gp_fit(c["x"], c["y"], c["h"], c["w"], c["s"])
    k0 = oracle.kernel_scale(1, c["h"], c["w"])
    for M in (256, 1000):
        xo = np.linspace(-5.0, 5.0, M) + 1e-3       # bench.py's points
        mo, vo = oracle.gp_predict(c["x"], c["h"], c["w"], Lo, ao, xo)
        m, v, _ = fit.predict(xo)
        assert relmax(m, mo) < RTOL
        assert relmax(v, vo, scale=k0) < RTOL
        m2 = fit.predict(xo, want_var=False)[0]
        assert relmax(m2, mo) < RTOL
    fit.close()


@pytest.mark.parametrize("d,n,M,route", [(1, 2048, 100, "direct"), (1, 2500, 333, "direct"),
                                         (2, 4096, 4200, "staged")])
def test_predict_point_routes_at_large_npad(engine, oracle, d, n, M, route):
    """Mean + variance on a resident fit with npad >= 2048 and M NOT a multiple of 64: the cross
    Gram reads the prediction points straight out of the mapped staging buffer and writes its own
    padding (`direct`, fit.hip) while the npad / 64 uncached passes over the points stay below
    4 MiB, and takes them through one device copy above (`staged`: d M 8 B x 64 passes = 4.3 MB
    here) -- ADVICE r05.  Both against the oracle at 1e-10; the covariance route (always staged)
    gives the same mean and diagonal."""
    rs = np.random.RandomState(n + M)
    if d == 1:
        c = wl.c4(n)
        x, h, w, s = c["x"], c["h"], c["w"], c["s"]
        y = wl.norm_logpdf(x) + 0.01 * rs.randn(n)
        xo = np.sort(rs.uniform(-5.2, 5.2, M))
    else:
        c = wl.c3()
        x, h, w, s = c["x"], float(c["h"][210]), c["w"][210], c["s"]
        y = c["y"]
        xo = rs.uniform(-5.0, 5.0, (2, M))
    npad = -(-n // 64) * 64
    assert (d * M * 8 * (npad // 64) <= (4 << 20)) == (route == "direct")
    fit = engine.gp_fit(x, y, h, w, s)
    Lo, ao, _ = oracle.gp_fit(x, y, h, w, s)
    mo, vo = oracle.gp_predict(x, h, w, Lo, ao, xo)
    m, v, _ = fit.predict(xo)
    k0 = oracle.kernel_scale(d, h, w)
    assert relmax(m, mo) < RTOL and relmax(v, vo, scale=k0) < RTOL
    if M <= 512:
        m2, v2, cov = fit.predict(xo, want_cov=True)
        assert relmax(m2, m) < 1e-12 and relmax(np.diag(cov), v, scale=k0) < 1e-11
    fit.close()


def test_kernel_copies_match_the_copy_engine_route(engine):
    """BQ_SOLVE_KCOPY (default on): the small transfers of the latency-bound calls go through kernels
    on mapped pinned staging buffers instead of copy-engine operations and memsets, and the smallest
    linalg_c drop-ins are one launch.  The same calls with BQ_SOLVE_KCOPY=0 give the same bits where
    the same kernels do the arithmetic, and agree to rounding where the one-launch forms replace
    them (cho_solve up to 64 rows)."""
    import os
    from bayesian_quadrature_amd import Engine
    old = os.environ.get("BQ_SOLVE_KCOPY")
    os.environ["BQ_SOLVE_KCOPY"] = "0"
    try:
        eng0 = Engine(0)
    finally:
        if old is None:
            os.environ.pop("BQ_SOLVE_KCOPY", None)
        else:
            os.environ["BQ_SOLVE_KCOPY"] = old
    try:
        rs = np.random.RandomState(4)
        # a resident fit: one-vector solves (one-launch sweeps and the per-block route), posterior
        for n in (700, 1500, 2100):
            c = wl.c4(n)
            xo = rs.uniform(-5, 5, 100)
            b = rs.randn(n)
            res = []
            for eng in (engine, eng0):
                fit = eng.gp_fit(c["x"], wl.norm_logpdf(c["x"]), c["h"], c["w"] * 3.0, c["s"])
                m, v, _ = fit.predict(xo)
                res.append((fit.solve(b), m, v, fit.logml, fit.refit_predict(c["h"], c["w"] * 2.5,
                                                                          c["s"], xo[:10])))
                fit.close()
            assert np.array_equal(res[0][0], res[1][0])
            assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])
            assert res[0][3] == res[1][3]
            assert np.array_equal(res[0][4][0], res[1][4][0])
            assert np.array_equal(res[0][4][1], res[1][4][1])
        # the host-buffer plan call (scatter / gather kernels against four copies each way)
        c2 = wl.c2()
        a = engine.fit_predict(c2["x"], c2["y"], c2["h"], c2["w"], c2["s"], c2["xo"])
        b0 = eng0.fit_predict(c2["x"], c2["y"], c2["h"], c2["w"], c2["s"], c2["xo"])
        for u, v in zip(a, b0):
            assert np.array_equal(np.asarray(u), np.asarray(v))
        # linalg drop-ins: small (one launch), middle (staged), large (copies)
        for n in (7, 64, 65, 150, 300):
            A = rand_spd(rs, n)
            La, Lb = np.zeros_like(A), np.zeros_like(A)
            engine.cho_factor(A, La)
            eng0.cho_factor(A, Lb)
            assert np.array_equal(np.tril(La), np.tril(Lb))
            bvec = rs.randn(n)
            xa, xb = np.empty(n), np.empty(n)
            engine.cho_solve(La, bvec, xa, 1)
            eng0.cho_solve(La, bvec, xb, 1)
            assert np.max(np.abs(xa - xb)) <= 1e-12 * np.max(np.abs(xb))
            if n > 64:
                assert np.array_equal(xa, xb)
            assert engine.logdet(La) == eng0.logdet(La)
    finally:
        eng0.close()


def test_diagonal_factor_block_inverses_and_pivots(engine):
    """The 64 x 64 diagonal factor alone (bq_probe_potf2: four and eight waves, the block read from
    global memory and handed over through LDS): L against LAPACK, the reciprocal pivots, and the four
    16 x 16 block inverses every panel solve multiplies by -- W_b L_bb = I to rounding --; the four
    forms agree bit for bit (the epilogue's slices over the lane groups apply the substitution's
    operations in the substitution's order); a non-positive pivot is reported at its column."""
    import ctypes as C
    from bayesian_quadrature_amd import _lib as L

    def probe(A, flags):
        A = np.asfortranarray(A, dtype=np.float64)
        Lo = np.zeros((64, 64), order="F")
        dv = np.zeros(64 + 4 * 256)
        info = C.c_int32(0)
        us = C.c_double(0)
        st = (C.c_int64 * 136)()
        engine._check(engine._lib.bq_probe_potf2(engine._ctx, L.dptr(A), flags, 2, L.dptr(Lo),
                                                 L.dptr(dv), C.byref(info),
                                                 C.cast(C.byref(us), L._dp), st))
        return np.tril(Lo), dv, info.value

    rs = np.random.RandomState(11)
    x = np.linspace(-5, 5, 1024)[:64]
    dx = 10.0 / 1023
    K = np.exp(-0.5 * (x[:, None] - x[None]) ** 2 / dx ** 2) / (np.sqrt(2 * np.pi) * dx)
    K += 1e-6 * np.eye(64)
    S = rand_spd(rs, 64)
    for A in (K, S):
        Lr = np.linalg.cholesky(A)
        outs = [probe(A, fl) for fl in (0, 1, 2, 3)]
        for Lo, dv, info in outs:
            assert info == 0
            assert np.max(np.abs(Lo - Lr)) <= 1e-13 * np.max(np.abs(Lr))
            assert np.max(np.abs(dv[:64] * np.diag(Lr) - 1.0)) < 1e-13
            for b in range(4):
                W = np.tril(dv[64 + 256 * b:64 + 256 * (b + 1)].reshape(16, 16, order="F"))
                Lb = Lr[16 * b:16 * b + 16, 16 * b:16 * b + 16]
                assert np.max(np.abs(W.dot(Lb) - np.eye(16))) < 1e-12
        for Lo, dv, info in outs[1:]:
            assert np.array_equal(Lo, outs[0][0])
            assert np.array_equal(dv[:64], outs[0][1][:64])
            for b in range(4):
                Wa = dv[64 + 256 * b:64 + 256 * (b + 1)].reshape(16, 16, order="F")
                Wr = outs[0][1][64 + 256 * b:64 + 256 * (b + 1)].reshape(16, 16, order="F")
                assert np.array_equal(np.tril(Wa), np.tril(Wr))
    B = S.copy()
    Lr = np.linalg.cholesky(S)
    B[37, 37] = Lr[37, :37].dot(Lr[37, :37]) - 1e-3
    for fl in (0, 1, 2, 3):
        Lo, dv, info = probe(B, fl)
        assert info == 38
        assert np.max(np.abs(Lo[:, :37] - Lr[:, :37])) < 1e-12


def test_potf2_eight_waves_is_the_same_factor(engine):
    """The one-launch steps' diagonal factor on eight waves (slab_step_kernel<., 8>,
    potf2f_body<8>; BQ_POTF2_8W, read when a context is created) applies the same updates to
    every column in the same order as the four-wave form: a C2 pass and a small failing
    system give bit-identical results."""
    import os
    from bayesian_quadrature_amd import Engine
    old = os.environ.get("BQ_POTF2_8W")
    os.environ["BQ_POTF2_8W"] = "0"
    try:
        eng4 = Engine(0)
    finally:
        if old is None:
            os.environ.pop("BQ_POTF2_8W", None)
        else:
            os.environ["BQ_POTF2_8W"] = old
    try:
        c = wl.c2()
        res = []
        for eng in (engine, eng4):
            plan = eng.plan(1, 1, 1024, 256)
            plan.set_inputs(c["x"][None], c["y"][None], c["xo"][None], c["h"], c["w"], c["s"])
            plan.run()
            res.append(plan.results())
            plan.close()
        for u, v in zip(res[0], res[1]):
            assert np.array_equal(u, v)
        # a system that stops being positive definite deep in the sweep: the same first failing
        # column from both forms
        n = 700
        x = np.linspace(-5, 5, n)
        x[500] = x[499]
        y = np.sin(x)
        for eng in (engine, eng4):
            with pytest.raises(np.linalg.LinAlgError) as ei:
                eng.gp_fit(x, y, 1.0, 0.2, 0.0)
            res.append(str(ei.value))
        assert res[-1] == res[-2]
    finally:
        eng4.close()


def test_folded_readout_matches_finalize(engine, oracle):
    """A sweep of one-launch steps carries its read-out (slab.h, SlabOut: log|K| from the diagonal
    factors as they go, mean / variance / log-ML from the last step's tiles) instead of a
    finalize launch; BQ_FOLD_READOUT=0 (read when a context is created) gives the stand-alone
    form.  Plans (one and several problems, with and without prediction points) and the
    resident fit agree between the two and with the oracle."""
    import os
    from bayesian_quadrature_amd import Engine
    old = os.environ.get("BQ_FOLD_READOUT")
    os.environ["BQ_FOLD_READOUT"] = "0"
    try:
        eng0 = Engine(0)
    finally:
        if old is None:
            os.environ.pop("BQ_FOLD_READOUT", None)
        else:
            os.environ["BQ_FOLD_READOUT"] = old
    try:
        rs = np.random.RandomState(11)
        for B, n, M in ((1, 1024, 256), (3, 200, 33), (5, 40, 0), (2, 64, 64), (1, 63, 1)):
            dx = 10.0 / n
            x = np.linspace(-5, 5, n)[None, :] + 0.2 * dx * rs.uniform(-1, 1, (B, n))
            y = np.sin(x) + 0.1 * rs.randn(B, n)
            xo = rs.uniform(-5, 5, (B, max(M, 1)))[:, :M]
            wv = np.array([1.3 * dx])
            res = []
            for eng in (engine, eng0):
                plan = eng.plan(B, 1, n, M)
                plan.set_inputs(x, y, xo if M else None, 1.3, wv, 1e-3)
                plan.run()
                res.append(plan.results())
                plan.close()
            mean, var, logml, status = res[0]
            assert (status == 0).all() and (res[1][3] == 0).all()
            assert np.abs(logml - res[1][2]).max() <= 1e-13 * np.abs(logml).max()
            if M:
                assert relmax(mean, res[1][0]) < 1e-13
                assert relmax(var, res[1][1], scale=oracle.kernel_scale(1, 1.3, wv)) < 1e-13
            Lo, ao, lmo = oracle.gp_fit(x[0], y[0], 1.3, wv, 1e-3)
            assert abs(logml[0] - lmo) <= RTOL * abs(lmo)
            if M:
                k0 = oracle.kernel_scale(1, 1.3, wv)
                mo, vo = oracle.gp_predict(x[0], 1.3, wv, Lo, ao, xo[0])
                assert relmax(mean[0], mo) < RTOL and relmax(var[0], vo, scale=k0) < RTOL
        c = wl.c2()
        lm = []
        for eng in (engine, eng0):
            fit = eng.gp_fit(c["x"], c["y"], c["h"], c["w"], c["s"])
            lm.append(fit.logml)
            fit.refit(c["h"] * 1.1, c["w"], c["s"])
            lm.append(fit.logml)
            fit.close()
        assert abs(lm[0] - lm[2]) <= 1e-13 * abs(lm[0]) and abs(lm[1] - lm[3]) <= 1e-13 * abs(lm[1])
    finally:
        eng0.close()


def test_ksplit_variants_agree(engine):
    """The eight-wave forms of the sweeps' step kernels (rows_step_kernel<8>,
    rows_fused_kernel<., 2>; BQ_GEMM_KSPLIT, read when a context is created) against the
    four-wave forms: a posterior variance at C2's size and a 256-column solve on an N = 4096
    factor differ only in summation order."""
    import os
    from bayesian_quadrature_amd import Engine
    old = os.environ.get("BQ_GEMM_KSPLIT")
    os.environ["BQ_GEMM_KSPLIT"] = "0"
    try:
        eng4 = Engine(0)
    finally:
        if old is None:
            os.environ.pop("BQ_GEMM_KSPLIT", None)
        else:
            os.environ["BQ_GEMM_KSPLIT"] = old
    try:
        c = wl.c2()
        xo = np.linspace(-5.0, 5.0, 256) + 1e-3
        res = []
        for eng in (engine, eng4):
            fit = eng.gp_fit(c["x"], c["y"], c["h"], c["w"], c["s"])
            res.append(fit.predict(xo))
            fit.close()
        k0 = float(c["h"]) ** 2
        assert relmax(res[0][0], res[1][0]) < 1e-12
        assert relmax(res[0][1], res[1][1], scale=k0) < 1e-11
        n = 4096
        c4 = wl.c4(n)
        B = np.asfortranarray(np.random.RandomState(5).randn(n, 256))
        X = []
        for eng in (engine, eng4):
            fit = eng.gp_fit(c4["x"], wl.norm_logpdf(c4["x"]), c4["h"], c4["w"], c4["s"])
            X.append(fit.solve(B))
            fit.close()
        assert relmax(X[0], X[1]) < 1e-9  # (cond(K) ~ 1e6: both are within it of the solution)
    finally:
        eng4.close()


def test_cho_solve_mat_square_multi_block(engine, oracle):
    """``la.cho_solve_mat`` as the reference uses it -- B square (linalg_c.pyx:166) -- at
    n = 1100: several 256-column steps of the row sweeps AND many right-hand sides (a partial
    last block both ways), against the oracle's substitution and the residual."""
    from bayesian_quadrature_amd import la
    n = 1100
    rs = np.random.RandomState(n)
    A = rand_spd(rs, n)
    L = oracle.cho_factor(A)
    B = np.asfortranarray(rs.rand(n, n))
    X = np.empty_like(B, order="F")
    la.cho_solve_mat(L, B, X)
    assert relmax(X, oracle.cho_solve(L, B)) < 1e-12
    assert np.abs(A.dot(X) - B).max() < 1e-12 * n * np.abs(A).max() * np.abs(X).max()
    la.cho_solve_mat(L, B, B)   # aliased (linalg_c.pyx:171)
    assert (B == X).all()


# ---- the stacked pair at many hyper-parameter sets (bq_pair_*) ----------------------------
def _pair_problem(ns, nc, seed):
    rs = np.random.RandomState(seed)
    xs = np.linspace(-5, 5, ns)
    dx = 10.0 / (ns - 1)
    ls = np.exp(wl.norm_logpdf(xs))
    xc = np.sort(rs.uniform(-6, 6, 6 * nc))
    from bayesian_quadrature_amd import bq_c
    bq_c.filter_candidates(xc, xs, 0.4 * dx)
    xc = xc[~np.isnan(xc)][:nc]
    return xs, ls, np.sort(xc), dx, rs


@pytest.mark.parametrize("ns,nc,S", [(9, 3, 5), (60, 8, 7), (400, 12, 6),
                                     (60, 0, 5), (300, 0, 6)])   # nc = 0: ONE plan of 2 S systems
def test_pair_llh_vs_oracle(engine, oracle, ns, nc, S):
    """bq_pair_llh: the hyper-parameter objective (bq.py:536-550, 933-965) at S parameter sets
    in one batched pass, against the oracle evaluated set by set; a set whose GP1 is singular
    and one whose GP2 is singular come back as -inf with their status."""
    from engine_double import EngineDouble
    xs, ls, xc, dx, rs = _pair_problem(ns, nc, ns + S)
    nc = xc.shape[0]
    p_tl = np.column_stack([rs.uniform(8, 20, S), rs.uniform(1.0, 1.6, S) * dx, np.full(S, 1e-4)])
    p_l = np.column_stack([rs.uniform(0.1, 0.4, S), rs.uniform(0.9, 1.3, S) * dx, np.zeros(S)])
    p_tl[1, 1], p_tl[1, 2] = 400 * dx, 0.0      # GP1 numerically singular
    p_l[2, 1] = 400 * dx                        # GP2 numerically singular
    pair = engine.pair(xs, np.log(ls), ls, xc, None, S)
    llh, l_c, status = pair.llh(p_tl, p_l)
    ref = EngineDouble(oracle).pair(xs, np.log(ls), ls, xc, None, S).llh(p_tl, p_l)
    assert status[1] == 1 and status[2] == 3 and (status[[0] + list(range(3, S))] == 0).all()
    assert (status == ref[2]).all()
    ok = status == 0
    assert np.isinf(llh[~ok]).all() and (llh[~ok] < 0).all()
    assert np.abs(llh[ok] - ref[0][ok]).max() <= RTOL * np.abs(ref[0][ok]).max()
    good1 = status != 1
    if nc:
        assert relmax(l_c[good1], ref[1][good1]) < 1e-9
    # a second call with other parameters on the same object
    llh2, _, st2 = pair.llh(p_tl[::-1].copy(), p_l[::-1].copy())
    assert (st2 == status[::-1]).all()
    assert np.abs(llh2[::-1][ok] - llh[ok]).max() <= 1e-12 * np.abs(llh[ok]).max()
    pair.close()


@pytest.mark.parametrize("ns,nc,S,M", [(9, 3, 4, 10), (60, 8, 5, 16), (70, 8, 5, 16),
                                       (300, 10, 3, 12), (1030, 12, 4, 20)])
def test_pair_esm_vs_per_set_route(engine, oracle, ns, nc, S, M):
    """bq_pair_esm: GP1's posterior at the candidates and acquisition points, GP2's targets and
    the S x M bordered systems of the acquisition in one batched pass, against the per-set route
    (a resident fit per set + bq_esm_border / bq_gp_predict) and, for the small sizes, the
    oracle's recipe.  From ns = 64 on the pass is S factorisations + border rows (round 4: the
    first 64 floor(ns / 64) columns of a set's M systems eliminated once, then S M small systems);
    below, and with BQ_PAIR_BORDER=0, the S M full systems -- test_pair_esm_routes_agree."""
    from engine_double import EngineDouble
    xs, ls, xc, dx, rs = _pair_problem(ns, nc, 3 * ns + M)
    nc = xc.shape[0]
    assert nc >= 1
    x_a = np.sort(np.concatenate([rs.uniform(-7, 7, M - 2), xc[:1] + 0.05, [0.5 * (xc[0] + xc[-1])]]))
    p_tl = np.column_stack([rs.uniform(8, 20, S), rs.uniform(1.0, 1.3, S) * dx, np.full(S, 1e-4)])
    p_l = np.column_stack([rs.uniform(0.1, 0.4, S), rs.uniform(0.9, 1.1, S) * dx, np.zeros(S)])
    thresh = 0.5
    pair = engine.pair(xs, np.log(ls), ls, xc, x_a, S)
    r = pair.esm(p_tl, p_l, thresh, MU1, COV1)
    assert (r["sstatus"] == 0).all() and (r["status"] == 0).all()
    x_sc = np.concatenate([xs, xc])
    for b in range(S):
        f1 = engine.gp_fit(xs, np.log(ls), *p_tl[b])
        m, v, _ = f1.predict(np.concatenate([xc, x_a]))
        f1.close()
        k0 = oracle.kernel_scale(1, p_tl[b, 0], [p_tl[b, 1]])
        assert relmax(r["tm_a"][b], m[nc:]) < RTOL
        assert np.abs(r["tC_a"][b] - v[nc:]).max() / k0 < RTOL
        assert relmax(r["l_c"][b], np.exp(m[:nc])) < 1e-9
        l_sc = np.concatenate([ls, r["l_c"][b]])
        f2 = engine.gp_fit(x_sc, l_sc, p_l[b, 0], p_l[b, 1], 0.0)
        A_a, A_sc_l, st = engine.esm_border(f2, ns, x_a, thresh, MU1, COV1)
        f2.close()
        K = oracle.gram_cross(x_sc, x_sc, p_l[b, 0], p_l[b, 1])
        tol = max(1e-10, 50 * np.linalg.cond(K) * 2.2e-16)
        scale = max(np.abs(A_a).max(), np.abs(A_sc_l).max())
        assert (st == 0).all()
        assert np.abs(r["A_a"][b] - A_a).max() <= tol * scale
        assert np.abs(r["A_sc_l"][b] - A_sc_l).max() <= tol * scale
        if ns <= 60:
            ref = EngineDouble(oracle).esm_batch(x_sc, l_sc, ns, x_a, p_l[b, 0], p_l[b, 1], thresh,
                                                 MU1, COV1)
            assert np.abs(r["A_a"][b] - ref[0]).max() <= tol * scale
            assert np.abs(r["A_sc_l"][b] - ref[1]).max() <= tol * scale
    pair.close()


def test_pair_esm_routes_agree(engine):
    """The two routes of bq_pair_esm on the same inputs: S factorisations + border rows (default)
    against the S M full bordered systems (BQ_PAIR_BORDER=0), incl. a candidate on top of a
    candidate point (both jitters) and a set whose GP2 is numerically singular (its elements'
    status non-zero on both routes)."""
    import os
    from bayesian_quadrature_amd import Engine
    ns, S, M = 200, 6, 14
    xs, ls, xc, dx, rs = _pair_problem(ns, 9, 3 * ns + M)
    x_a = np.sort(np.concatenate([rs.uniform(-7, 7, M - 2), xc[:1] + 0.05, xc[-1:]]))
    p_tl = np.column_stack([rs.uniform(8, 20, S), rs.uniform(1.0, 1.3, S) * dx, np.full(S, 1e-4)])
    p_l = np.column_stack([rs.uniform(0.1, 0.4, S), rs.uniform(0.9, 1.1, S) * dx, np.zeros(S)])
    p_l[3, 1] = 300 * dx
    pair = engine.pair(xs, np.log(ls), ls, xc, x_a, S)
    r = pair.esm(p_tl, p_l, 0.5, MU1, COV1)
    pair.close()
    os.environ["BQ_PAIR_BORDER"] = "0"
    try:
        e2 = Engine(0)
    finally:
        del os.environ["BQ_PAIR_BORDER"]
    try:
        pair2 = e2.pair(xs, np.log(ls), ls, xc, x_a, S)
        r2 = pair2.esm(p_tl, p_l, 0.5, MU1, COV1)
        pair2.close()
    finally:
        e2.close()
    assert ((r["status"] != 0) == (r2["status"] != 0)).all() and (r["status"][3] != 0).all()
    ok = r["status"] == 0
    assert ok.sum() >= (S - 1) * M - 2
    for k in ("A_a", "A_sc_l"):
        assert np.abs(r[k][ok] - r2[k][ok]).max() <= 1e-9 * np.abs(r2[k][ok]).max(), k
    for k in ("tm_a", "tC_a", "l_c"):
        assert np.array_equal(r[k], r2[k]), k


# ---- the batched factorisation, diagonal block first (round 4) --------------------
@pytest.mark.parametrize("m,kb,batch", [(64, 64, 1), (128, 192, 2), (256, 448, 3), (1472, 448, 4),
                                        (3712, 448, 2), (320, 256, 9)])
def test_panel_solve_vs_lapack(engine, m, kb, batch):
    """The batched panel solve X <- X L^-T of one outer block (potrf.hip, enqueue_panel_solve) on
    DENSE random factors against LAPACK's dtrtrs, slab by slab: the recursive products + solves
    (gemm_trsm64_kernel) and the one-launch sweep (trsm_sweep_kernel).  The 1-D sorted inputs of
    C2 / C5 give banded matrices whose panels are zeros below the band -- a product over them
    tests nothing, which is how a missing LDS-DMA wait passed every config test but C3's."""
    import scipy.linalg as sla
    rs = np.random.RandomState(m + kb)
    Ls, Xs, refs = [], [], []
    for _ in range(batch):
        G = rs.standard_normal((kb, kb))
        Lf = np.linalg.cholesky(G @ G.T + kb * np.eye(kb))
        X = rs.standard_normal((m, kb))
        Ls.append(Lf)
        Xs.append(X)
        refs.append(sla.solve_triangular(Lf, X.T, lower=True).T)
    Ls, Xs, refs = np.array(Ls), np.array(Xs), np.array(refs)
    for mode in (1, 2):
        out = engine.probe_panel_solve(Ls, Xs, mode)
        for s in range(kb // 64):
            err = relmax(out[:, :, 64 * s:64 * s + 64], refs[:, :, 64 * s:64 * s + 64],
                         scale=np.max(np.abs(refs)))
            assert err < 1e-13, (mode, s, err)
        # and twice the same bits
        assert np.array_equal(out, engine.probe_panel_solve(Ls, Xs, mode))


def _dense_batch(batch, n, m, d=2, seed=11):
    rs = np.random.RandomState(seed)
    x = rs.uniform(-3, 3, (batch, d, n))
    xo = rs.uniform(-3, 3, (batch, d, m))
    y = wl.norm_logpdf(x[:, 0]) + wl.norm_logpdf(x[:, 1])
    return x, y, xo, 1.3, np.full(d, 6.0 / np.sqrt(n) * 1.5), 0.05


@pytest.mark.parametrize("batch,n,m", [(12, 1100, 70), (100, 700, 40)])
def test_batch_dense_vs_oracle_and_variants(engine, oracle, batch, n, m):
    """A batch of DENSE 2-D problems (outer block 128: diagonal block first -- a workgroup per
    matrix from 96 matrices on, the one-launch steps below) against the oracle, and every switch of
    the batched sweep against the default on the same inputs: the recursive panels of rounds 1-3
    (BQ_DIAG_FIRST=0), either form of the diagonal factor, the recursive panel solve, the fork
    after instead of before the panel solve, no second stream."""
    import os
    from bayesian_quadrature_amd import Engine
    x, y, xo, h, w, s = _dense_batch(batch, n, m)
    mean, var, logml, status = engine.batch_fit_predict(x, y, h, w, s, xo)
    assert (status == 0).all()
    for i in (0, batch // 2, batch - 1):
        Lo, ao, lmo = oracle.gp_fit(x[i], y[i], h, w, s)
        mo, vo = oracle.gp_predict(x[i], h, w, Lo, ao, xo[i])
        assert relmax(mean[i], mo) < RTOL
        assert relmax(var[i], vo, scale=oracle.kernel_scale(2, h, w)) < RTOL
        assert abs(logml[i] - lmo) <= RTOL * abs(lmo)
    for env in ({"BQ_DIAG_FIRST": "0"}, {"BQ_DF_WG": "0"}, {"BQ_DF_WG": "1"}, {"BQ_DF_SWEEP": "0"},
                {"BQ_DF_EARLY": "0"}, {"BQ_DF_WG_ROWS": "0"}, {"BQ_DF_WG_ROWS": "300"},
                {"BQ_LOOKAHEAD": "0"}):
        os.environ.update(env)
        try:
            e2 = Engine(0)
        finally:
            for k in env:
                del os.environ[k]
        try:
            m2, v2, l2, st2 = e2.batch_fit_predict(x, y, h, w, s, xo)
        finally:
            e2.close()
        assert (st2 == 0).all(), env
        assert relmax(m2, mean) < 1e-12 and relmax(l2, logml) < 1e-12, env
        assert relmax(v2, var, scale=oracle.kernel_scale(2, h, w)) < 1e-12, env


@pytest.mark.parametrize("n", [700, 1100, 2048])
@pytest.mark.parametrize("batch", [63, 64, 65, 95, 96, 97, 191, 192, 256])
def test_batch_rule_boundaries(engine, oracle, batch, n):
    """The batched sweep picks its launch sequence from the batch size and the system size: the
    outer block (448, 384 from 96 matrices), the diagonal factor (one-launch steps below 96
    matrices, a workgroup per matrix from 96 on -- and below, for blocks with >= 1000 rows under
    them), 128- or 64-tiles per product.  Either side of every boundary, on DENSE 2-D problems:
    two problems against the oracle at 1e-10, and every workspace of the plan behind a 4 KiB
    sentinel band that a correct launch sequence leaves alone (round 5 sized the block-inverse
    records for the wrong block width at 96-191 matrices and nothing noticed).  The reference's
    contract: a factor is correct however it is blocked (tests/test_linalg_c.py:22-37)."""
    m = 40
    x, y, xo, h, w, s = _dense_batch(batch, n, m, seed=batch + n)
    engine.set_guard(True)
    try:
        plan = engine.plan(batch, 2, n, m)
    finally:
        engine.set_guard(False)
    try:
        plan.set_inputs(x, y, xo, h, w, s)
        plan.run()
        mean, var, logml, status = plan.results()
        guarded, damaged = plan.check_guards()
        # and a second pass over the same plan (the graph replay) leaves the bands alone too
        plan.run()
        mean2, var2, logml2, _ = plan.results()
        guarded2, damaged2 = plan.check_guards()
    finally:
        plan.close()
    assert guarded >= 8 and damaged == 0 and damaged2 == 0, (guarded, damaged, damaged2)
    assert (status == 0).all()
    assert np.array_equal(mean, mean2) and np.array_equal(var, var2) and np.array_equal(logml, logml2)
    for i in (0, batch - 1):
        Lo, ao, lmo = oracle.gp_fit(x[i], y[i], h, w, s)
        mo, vo = oracle.gp_predict(x[i], h, w, Lo, ao, xo[i])
        assert relmax(mean[i], mo) < RTOL
        assert relmax(var[i], vo, scale=oracle.kernel_scale(2, h, w)) < RTOL
        # (relative to the size of the log-ML's own terms: on these random problems the three
        # terms, each of order n, can cancel to a value of order one -- [95-2048]: 1.33 -- and no
        # two summation orders agree to 1e-10 of THAT)
        assert abs(logml[i] - lmo) <= RTOL * max(abs(lmo), 0.5 * n * np.log(2 * np.pi))


def test_guard_bands_are_reported(engine):
    """The sentinel switch itself: a plan created under it reports its guarded buffers (all ten
    workspaces) with no damage before any pass; a plan created without it reports none."""
    engine.set_guard(True)
    try:
        plan = engine.plan(3, 1, 200, 8)
    finally:
        engine.set_guard(False)
    try:
        guarded, damaged = plan.check_guards()
        assert guarded == 10 and damaged == 0
    finally:
        plan.close()
    plain = engine.plan(3, 1, 200, 8)
    try:
        assert plain.check_guards() == (0, 0)
    finally:
        plain.close()


def test_c2_batch_256_vs_oracle(engine, oracle):
    """The benched `c2_batch_256x1024` (256 copies of C2 per pass: bench.py, batched_configs) --
    a batch size no other test reaches -- problems 0 / 128 / 255 against the oracle at 1e-10, all
    256 the same bits (the copies are identical problems)."""
    c = wl.c2()
    B = 256
    plan = engine.plan(B, 1, 1024, 256)
    try:
        plan.set_inputs(np.repeat(c["x"][None], B, axis=0), np.repeat(c["y"][None], B, axis=0),
                        np.repeat(c["xo"][None], B, axis=0), c["h"], c["w"], c["s"])
        plan.run()
        mean, var, logml, status = plan.results()
    finally:
        plan.close()
    assert (status == 0).all()
    Lo, ao, lmo = oracle.gp_fit(c["x"], c["y"], c["h"], c["w"], c["s"])
    mo, vo = oracle.gp_predict(c["x"], c["h"], c["w"], Lo, ao, c["xo"])
    for i in (0, 128, 255):
        assert relmax(mean[i], mo) < RTOL
        assert relmax(var[i], vo, scale=oracle.kernel_scale(1, c["h"], c["w"])) < RTOL
        assert abs(logml[i] - lmo) <= RTOL * abs(lmo)
    assert (mean == mean[0]).all() and (var == var[0]).all() and (logml == logml[0]).all()


@pytest.mark.parametrize("batch,n,m,d", [(3, 700, 20, 2), (12, 1100, 70, 2), (7, 900, 33, 3),
                                         (100, 700, 40, 2), (64, 2048, 40, 1), (5, 3000, 33, 2)])
def test_fused_assembly_same_bits(engine, oracle, batch, n, m, d):
    """A batched plan assembles only its first outer block's columns; block 0's three products
    compute the rest of the bordered system in their accumulators instead of loading it
    (gram_seed_neg: the assembly's own arithmetic) -- or, where a product cannot (d > 2, a shape
    that goes to a register-streaming kernel), its region is assembled right in front of it.
    Either way the results carry the bits of the full assembly (BQ_ASM_FUSE=0), and problem 0
    agrees with the oracle."""
    import os
    from bayesian_quadrature_amd import Engine
    rs = np.random.RandomState(batch + n + d)
    x = rs.uniform(-3, 3, (batch, d, n))
    xo = rs.uniform(-3, 3, (batch, d, m))
    y = sum(wl.norm_logpdf(x[:, k]) for k in range(d))
    h, s = 1.3, 0.05
    w = np.full(d, 6.0 / n ** (1.0 / d) * 1.5)
    mean, var, logml, status = engine.batch_fit_predict(x, y, h, w, s, xo)
    assert (status == 0).all()
    os.environ["BQ_ASM_FUSE"] = "0"
    try:
        e2 = Engine(0)
    finally:
        del os.environ["BQ_ASM_FUSE"]
    try:
        m2, v2, l2, st2 = e2.batch_fit_predict(x, y, h, w, s, xo)
    finally:
        e2.close()
    assert np.array_equal(mean, m2) and np.array_equal(var, v2) and np.array_equal(logml, l2)
    assert np.array_equal(status, st2)
    Lo, ao, lmo = oracle.gp_fit(x[0], y[0], h, w, s)
    mo, vo = oracle.gp_predict(x[0], h, w, Lo, ao, xo[0])
    assert relmax(mean[0], mo) < RTOL
    assert relmax(var[0], vo, scale=oracle.kernel_scale(d, h, w)) < RTOL
    assert abs(logml[0] - lmo) <= RTOL * max(abs(lmo), 0.5 * n * np.log(2 * np.pi))


def test_batch_diag_first_reports_not_pd(engine):
    """A hopeless matrix in the batch (length scale far beyond the spacing, no noise: not positive
    definite in fp64) is reported with a non-zero status, its neighbours are untouched."""
    x, y, xo, h, w, s = _dense_batch(10, 900, 16)
    plan = engine.plan(10, 2, 900, 16)
    ww = np.tile(w[None], (10, 1))
    ww[4] = 5.0
    ss = np.full(10, s)
    ss[4] = 0.0
    plan.set_inputs(x, y, xo, h, ww, ss)
    plan.run()
    mean, var, logml, status = plan.results()
    plan.close()
    assert status[4] != 0 and (np.delete(status, 4) == 0).all()
    assert np.isfinite(np.delete(logml, 4)).all()
    m1, v1, l1, st1 = engine.batch_fit_predict(np.delete(x, 4, 0), np.delete(y, 4, 0), h, w, s,
                                               np.delete(xo, 4, 0))
    assert relmax(np.delete(mean, 4, 0), m1) < 1e-12 and relmax(np.delete(logml, 4), l1) < 1e-12


# ---- the single-vector sweeps as one launch each (round 4) --------------------------
@pytest.mark.parametrize("n", [2048, 2500, 4096, 5000])
def test_flow_sweeps_same_bits_as_per_block_launches(engine, n):
    """fit.solve(b) and alpha through the one-launch sweeps (trsvflow.h: every step's workgroups in
    one grid, values handed over through sentinel-filled slots) against the one-launch-per-block
    sweeps (BQ_TRSV_FLOW=0): the same arithmetic in the same order, so the same bits; and the
    residual K x - b."""
    import os
    from bayesian_quadrature_amd import Engine
    c = wl.c4(n)
    y = wl.norm_logpdf(c["x"])
    b = np.random.RandomState(n).randn(n)
    fit = engine.gp_fit(c["x"], y, c["h"], c["w"] * 3.0, c["s"])
    x1, a1 = fit.solve(b), fit.alpha()
    K = fit.K()
    fit.close()
    os.environ["BQ_TRSV_FLOW"] = "0"
    try:
        e2 = Engine(0)
    finally:
        del os.environ["BQ_TRSV_FLOW"]
    try:
        f2 = e2.gp_fit(c["x"], y, c["h"], c["w"] * 3.0, c["s"])
        x0, a0 = f2.solve(b), f2.alpha()
        f2.close()
    finally:
        e2.close()
    assert np.array_equal(x1, x0) and np.array_equal(a1, a0)
    # (w = 3 dx: a wider band than C4's, cond(K) ~ 1e8 -- the bar is the residual with the device's
    # own Gram matrix, as in test_fit_solve_at_benched_sizes)
    assert np.abs(K.dot(x1) - b).max() < 1e-11 * np.abs(K).max() * np.abs(x1).max() * np.sqrt(n)
    assert np.abs(K.dot(a1) - y).max() < 1e-11 * np.abs(K).max() * np.abs(a1).max() * np.sqrt(n)


def test_flow_sweep_time_out_falls_back_and_returns_the_right_answer(engine):
    """A lost hand-off (BQ_FLOW_FAULT=1: step 1 of the forward sweep never publishes its block)
    ends in a time-out: every spinning workgroup leaves, and the SAME call re-issues the solve on
    the per-block sweeps -- BQ_OK, the same bits as a healthy solve, one more fall-back in the
    context's statistics (degrade, do not fail).  Every entry point that
    launches a one-launch sweep does so: bq_gp_solve, alpha, V(Z), bq_cho_solve."""
    import os
    import time
    n = 4096
    c = wl.c4(n)
    y = wl.norm_logpdf(c["x"])
    fit = engine.gp_fit(c["x"], y, c["h"], c["w"] * 3.0, c["s"])
    b = np.random.RandomState(1).randn(n)
    good = fit.solve(b)
    L = fit.L()
    from bayesian_quadrature_amd import la
    xg = np.empty(n)
    la.cho_solve_vec(L, b, xg)
    fit2 = engine.gp_fit(c["x"], y, c["h"], c["w"] * 3.0, c["s"])
    alpha_good = fit2.alpha()
    fit2.close()
    before = engine.stats()["flow_fallbacks"]
    os.environ["BQ_FLOW_FAULT"] = "1"
    t0 = time.time()
    try:
        got = fit.solve(b)
        mid = engine.stats()["flow_fallbacks"]
        fit3 = engine.gp_fit(c["x"], y, c["h"], c["w"] * 3.0, c["s"])
        alpha_got = fit3.alpha()          # backward sweep only: the fault sits in the forward one
        fit3.close()
        xf = b.copy()
        la.cho_solve_vec(L, xf, xf)       # aliased, as linalg_c.pyx:128 allows
    finally:
        del os.environ["BQ_FLOW_FAULT"]
    assert time.time() - t0 < 60.0
    assert mid == before + 1
    assert engine.stats()["flow_fallbacks"] >= before + 2
    assert np.array_equal(got, good)
    assert np.array_equal(alpha_got, alpha_good)
    assert np.array_equal(xf, xg)
    # and without the fault nothing falls back
    now = engine.stats()["flow_fallbacks"]
    assert np.array_equal(fit.solve(b), good)
    assert engine.stats()["flow_fallbacks"] == now
    fit.close()
