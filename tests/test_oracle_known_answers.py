"""Pins the CPU oracle (oracle/bq_oracle.c) before anything is compared with it:
  * the seven printed known answers of the reference's visual-tests notebook
  * the property contracts of the reference's own unit tests
    (tests/test_linalg_c.py, tests/test_gauss_c.py)
No GPU involved."""
import numpy as np
import pytest
import scipy.stats

from conftest import rand_spd
from fixture_chain import build_chain, known_answers


@pytest.fixture(scope="module")
def chain(oracle):
    o = oracle
    return build_chain(
        lambda x, y, h, w, s: o.gp_fit(x, y, h, w, s),
        lambda x, h, w, L, a, xo: o.gp_predict(x, h, w, L, a, xo, want_var=False),
        o.filter_candidates)


def _digits_ok(got, ref, digits):
    return abs(got - ref) <= 0.6 * 10.0 ** (np.floor(np.log10(abs(ref))) - digits + 1)


def test_fixture_candidates(chain):
    # nc = 2 after filtering
    assert chain["xc"].shape == (2,)
    assert chain["xsc"].shape == (11,)


@pytest.mark.parametrize("key", ["sum_int_K", "sum_int_K1_K2", "sum_int_int_K1_K2_K1",
                                 "sum_int_int_K1_K2", "int_int_K", "Z_mean"])
def test_known_answers(oracle, chain, key):
    o, c = oracle, chain
    exp = known_answers()["expected"][key]
    got = {
        "sum_int_K": lambda: o.int_K(c["xsc"], c["h2"], c["w2"], c["mu"], c["cov"]).sum(),
        "sum_int_K1_K2": lambda: o.int_K1_K2(c["xsc"], c["xs"], c["h2"], c["w2"], c["h1"],
                                             c["w1"], c["mu"], c["cov"]).sum(),
        "sum_int_int_K1_K2_K1": lambda: o.int_int_K1_K2_K1(c["xsc"], c["h2"], c["w2"], c["h1"],
                                                           c["w1"], c["mu"], c["cov"]).sum(),
        "sum_int_int_K1_K2": lambda: o.int_int_K1_K2(c["xs"], c["h2"], c["w2"], c["h1"], c["w1"],
                                                     c["mu"], c["cov"]).sum(),
        "int_int_K": lambda: o.int_int_K(1, c["h2"], c["w2"], c["mu"], c["cov"]),
        "Z_mean": lambda: o.Z_mean(c["xsc"], c["a2"], c["h2"], c["w2"], c["mu"], c["cov"]),
    }[key]()
    assert _digits_ok(got, exp["value"], exp["digits"]), (key, got, exp["value"])


def test_known_answer_Z_var(oracle, chain):
    # V(Z) = alpha' I alpha - beta' K^-1 beta: two ~1e-2 terms cancel to 6e-7,
    # so agreement is limited to ~1e-8 relative by rounding, not by the method.
    o, c = oracle, chain
    exp = known_answers()["expected"]["Z_var"]["value"]
    got = o.Z_var(c["xs"], c["xsc"], c["a2"], c["L1"], c["h2"], c["w2"], c["h1"], c["w1"],
                  c["mu"], c["cov"])
    assert abs(got - exp) / exp < 5e-8


# ---- reference tests/test_linalg_c.py contracts --------------------------------
@pytest.mark.parametrize("n", list(range(1, 11)) + [32, 64, 65, 130, 257])
def test_cho_factor_vs_numpy(oracle, n):
    rs = np.random.RandomState(8728 + n)
    A = rand_spd(rs, n)
    for unblocked in (True, False):
        L = oracle.cho_factor(A, unblocked=unblocked)
        assert np.allclose(L, np.linalg.cholesky(A), rtol=1e-13, atol=1e-13)


def test_blocked_equals_unblocked_large(oracle):
    rs = np.random.RandomState(1)
    A = rand_spd(rs, 500)
    oracle.set_threads(1)
    L1 = oracle.cho_factor(A, unblocked=True)
    L2 = oracle.cho_factor(A, nb=64)
    oracle.set_threads(4)
    L3 = oracle.cho_factor(A, nb=32)
    assert np.allclose(L1, L2, rtol=1e-12, atol=1e-12)
    assert np.allclose(L1, L3, rtol=1e-12, atol=1e-12)


def test_cho_factor_not_pd(oracle):
    A = np.asfortranarray(np.array([[1.0, 2.0], [2.0, 1.0]]))
    with pytest.raises(np.linalg.LinAlgError):
        oracle.cho_factor(A)


@pytest.mark.parametrize("n", [1, 2, 5, 10, 100])
def test_cho_solve(oracle, n):
    rs = np.random.RandomState(n)
    A = rand_spd(rs, n)
    L = oracle.cho_factor(A)
    b = rs.rand(n)
    B = np.asfortranarray(rs.rand(n, n))
    assert np.allclose(A.dot(oracle.cho_solve(L, b)), b)
    assert np.allclose(A.dot(oracle.cho_solve(L, B)), B)


@pytest.mark.parametrize("n", [1, 3, 10, 77])
def test_logdet(oracle, n):
    rs = np.random.RandomState(n)
    A = rand_spd(rs, n)
    L = oracle.cho_factor(A)
    assert np.allclose(oracle.logdet(L), np.linalg.slogdet(A)[1])


def test_dots(oracle):
    rs = np.random.RandomState(0)
    for n in (1, 2, 7):
        x, y = rs.rand(n), rs.rand(n)
        assert np.allclose(oracle.dot11(x, y), x.dot(y))
        assert np.allclose(oracle.vecdiff(x, y), np.sqrt(((x - y) ** 2).sum()))


# ---- reference tests/test_gauss_c.py contracts ---------------------------------
def test_mvn_logpdf_vs_scipy(oracle):
    rs = np.random.RandomState(3)
    x = rs.uniform(-10, 10, 20)
    m = rs.uniform(-10, 10, 20)
    C = np.exp(rs.uniform(-10, 0, 20))
    for xi, mi, Ci in zip(x, m, C):
        L = np.array([[np.sqrt(Ci)]], order="F")
        got = oracle.mvn_logpdf([xi], [mi], L, np.log(Ci))
        assert np.allclose(got, scipy.stats.norm.logpdf(xi, mi, np.sqrt(Ci)))


def test_int_exp_norm(oracle):
    def approx(c, m, S):
        xo = np.linspace(m - 20 * np.sqrt(S), m + 20 * np.sqrt(S), 200001)
        return np.trapezoid(np.exp(xo * c) * scipy.stats.norm.pdf(xo, m, np.sqrt(S)), xo)
    for c, m, S in [(2, 0, 1), (1, 0, 1), (2, 1, 1), (2, 4, 2), (1.5, -0.3, 0.4)]:
        assert np.allclose(oracle.int_exp_norm(c, m, S), approx(c, m, S), rtol=1e-6)
    assert np.isinf(oracle.int_exp_norm(2, 400, 1))


def _K(h, w, a, b):
    return h * h / (np.sqrt(2 * np.pi) * w) * np.exp(-(a[:, None] - b[None, :]) ** 2 / (2 * w * w))


def test_int_K_vs_trapezoid(oracle):
    # closed form vs quadrature, as tests/test_gauss_c.py:84-104 does
    x = np.linspace(-4, 4, 7)
    xo = np.linspace(-60, 60, 60001)
    p = scipy.stats.norm.pdf(xo, 0.3, np.sqrt(10.0))
    approx = np.trapezoid(_K(0.7, 1.1, x, xo) * p[None, :], xo, axis=1)
    got = oracle.int_K(x, 0.7, 1.1, [0.3], [[10.0]])
    assert np.allclose(got, approx, atol=1e-7)


def test_int_K1_K2_vs_trapezoid(oracle):
    x1 = np.linspace(-3, 3, 5)
    x2 = np.linspace(-2, 4, 4)
    xo = np.linspace(-60, 60, 60001)
    p = scipy.stats.norm.pdf(xo, 0.0, np.sqrt(10.0))
    K1 = _K(0.7, 1.1, x1, xo)
    K2 = _K(1.3, 0.6, xo, x2)
    approx = np.trapezoid(K1[:, None, :] * K2.T[None, :, :] * p[None, None, :], xo, axis=2)
    got = oracle.int_K1_K2(x1, x2, 0.7, 1.1, 1.3, 0.6, [0.0], [[10.0]])
    assert np.allclose(got, approx, atol=1e-7)


def test_int_int_K_vs_trapezoid(oracle):
    xo = np.linspace(-40, 40, 2001)
    p = scipy.stats.norm.pdf(xo, 0.0, np.sqrt(10.0))
    K = _K(0.7, 1.1, xo, xo)
    approx = np.trapezoid(np.trapezoid(K * p[None, :], xo, axis=1) * p, xo)
    assert np.allclose(oracle.int_int_K(1, 0.7, 1.1, [0.0], [[10.0]]), approx, atol=1e-7)


def test_determinism(oracle):
    # the reference's "_same" tests: bit-identical repeats
    x = np.linspace(-4, 4, 9)
    a = oracle.int_int_K1_K2_K1(x, 0.2, 1.3, 15, 2.0, [0.0], [[10.0]])
    for _ in range(5):
        assert (oracle.int_int_K1_K2_K1(x, 0.2, 1.3, 15, 2.0, [0.0], [[10.0]]) == a).all()


# ---- gp restatement self-consistency -------------------------------------------
def test_gp_fit_consistency(oracle):
    rs = np.random.RandomState(5)
    n = 200
    x = np.sort(rs.uniform(-5, 5, n))
    y = rs.randn(n)
    h, w, s = 1.3, 0.08, 0.05
    L, alpha, logml = oracle.gp_fit(x, y, h, w, s)
    K = _K(h, w, x, x) + s * s * np.eye(n)
    assert np.allclose(oracle.gram(x, h, w, s), K, rtol=1e-13, atol=1e-15)
    assert np.allclose(L.dot(L.T), K, rtol=1e-12, atol=1e-12)
    assert np.allclose(K.dot(alpha), y, atol=1e-8)
    ref = -0.5 * y.dot(np.linalg.solve(K, y)) - 0.5 * np.linalg.slogdet(K)[1] \
        - 0.5 * n * np.log(2 * np.pi)
    assert np.allclose(logml, ref, rtol=1e-11)
    xo = np.linspace(-5, 5, 37)
    mean, var = oracle.gp_predict(x, h, w, L, alpha, xo)
    Ks = _K(h, w, xo, x)
    assert np.allclose(mean, Ks.dot(alpha), atol=1e-9)
    cov = _K(h, w, xo, xo) - Ks.dot(np.linalg.solve(K, Ks.T))
    assert np.allclose(var, np.diag(cov), atol=1e-9)
    # the full posterior covariance (gp.GP.cov, bq.py:325,496): every entry against the
    # textbook formula, its diagonal against the marginal variance, symmetric to the bit
    co = oracle.gp_cov(x, h, w, L, xo)
    assert co.shape == (37, 37)
    assert np.allclose(co, cov, rtol=0, atol=1e-9)
    assert np.allclose(np.diag(co), var, rtol=0, atol=1e-14)
    assert (co == co.T).all()


def test_gp_2d(oracle):
    rs = np.random.RandomState(6)
    pts = rs.uniform(-2, 2, size=(2, 60))
    w = np.array([0.7, 0.4])
    K = oracle.gram(pts, 1.1, w, 0.1)
    d2 = ((pts[:, :, None] - pts[:, None, :]) ** 2 / (2 * w[:, None, None] ** 2)).sum(0)
    ref = 1.1 ** 2 / (2 * np.pi * w[0] * w[1]) * np.exp(-d2) + 0.01 * np.eye(60)
    assert np.allclose(K, ref, rtol=1e-13)
