"""Generates the golden vectors of tests/golden/ (SURVEY.md section 8c's list) with
the CPU oracle (oracle/bq_oracle.c), which is itself pinned to the reference's printed
known answers by tests/test_oracle_known_answers.py:

  c1_n32.npz     the C1 plumbing configuration of BASELINE.json (N=32 reference-style fixture)
  e2e_n256.npz   the same end-to-end chain at N=256
  c2_n1024.npz   BASELINE config C2 (N=1024, M=256, s=1e-3): the headline config
  c5_p0.npz      problem 0 of BASELINE config C5 (N=2048, M=256, s=1e-2)
  la_ga.npz      per-function vectors: linalg_c (cho_factor / cho_solve_vec / cho_solve_mat /
                 logdet on the random SPD matrices of tests/test_linalg_c.py:16-19, n = 1..10,
                 32, 64) and gauss_c / bq_c closed forms (the 9-point fixture and an N=32, d=2
                 case)

Large factors are not stored whole: the C2 / C5 files keep diag(L), the last row of L and
alpha, mean, var, logml (a few tens of KB each).

The reference cannot be run in this pipeline (Python 2 + the absent `gp`
package, and its Cython needs ATLAS headers the image lacks), so these vectors
come from the pinned restatement, not from the reference itself.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
from scipy.stats import norm

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import load  # noqa: E402


def main():
    o = load()
    o.set_threads(1)
    n = 32
    x = np.linspace(-5, 5, n)
    y = np.log(norm.pdf(x, 0, 1))
    h, w, s = 15.0, np.array([0.4]), 0.0
    xo = np.linspace(-5.4, 5.4, 50)
    L, alpha, logml = o.gp_fit(x, y, h, w, s)
    mean, var = o.gp_predict(x, h, w, L, alpha, xo)
    np.savez(os.path.join(HERE, "c1_n32.npz"), x=x, y=y, h=h, w=w, s=s, xo=xo, L=L, alpha=alpha,
             logml=logml, mean=mean, var=var, k0=o.kernel_scale(1, h, w),
             cond=np.linalg.cond(o.gram(x, h, w, s)))
    print("cond(K) = %.3g, logml = %.15g" % (np.linalg.cond(o.gram(x, h, w, s)), logml))
    o.set_threads(8)
    end_to_end(o)
    la_ga(o)


def _e2e(o, name, x, y, h, w, s, xo, keep_L):
    L, alpha, logml = o.gp_fit(x, y, h, w, s)
    mean, var = o.gp_predict(x, h, w, L, alpha, xo)
    out = dict(x=x, y=y, h=h, w=np.atleast_1d(w), s=s, xo=xo, alpha=alpha, logml=logml, mean=mean,
               var=var, k0=o.kernel_scale(1, h, np.atleast_1d(w)), diagL=np.diag(L).copy(),
               lastL=L[-1].copy(), z=o.trsm_lower(L, y))
    if keep_L:
        out["L"] = L
    np.savez(os.path.join(HERE, name), **out)
    print("%s: logml = %.15g" % (name, logml))


def end_to_end(o):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)),
                                    "bayesian-quadrature_amd"))
    import workloads as wl
    n = 256
    x = np.linspace(-5, 5, n)
    dx = 10.0 / (n - 1)
    _e2e(o, "e2e_n256.npz", x, np.log(norm.pdf(x, 0, 1)), 15.0, np.array([1.3 * dx]), 0.0,
         np.linspace(-5.3, 5.3, 100), True)
    c = wl.c2()
    _e2e(o, "c2_n1024.npz", c["x"], c["y"], c["h"], c["w"], c["s"], c["xo"], False)
    c = wl.c5([0])
    _e2e(o, "c5_p0.npz", c["x"][0], c["y"][0], c["h"], c["w"], c["s"], c["xo"][0], False)


def la_ga(o):
    out = {}
    rs = np.random.RandomState(20251004)
    ns = list(range(1, 11)) + [32, 64]
    out["la_ns"] = np.array(ns)
    for n in ns:
        # the reference's rand_mat (tests/test_linalg_c.py:16-19): rand(n, n) + its transpose + n I
        A = rs.rand(n, n)
        C = np.asfortranarray(A + A.T + n * np.eye(n))
        b = rs.randn(n)
        B = np.asfortranarray(rs.randn(n, n))
        L = o.cho_factor(C)
        out["la_C_%d" % n], out["la_b_%d" % n], out["la_B_%d" % n] = C, b, B
        out["la_L_%d" % n] = np.tril(L)
        out["la_xv_%d" % n] = o.cho_solve(L, b)
        out["la_xm_%d" % n] = o.cho_solve(L, B)
        out["la_logdet_%d" % n] = o.logdet(L)
    # gauss_c / bq_c on the 9-point fixture (tests/util.py:12-59) and an N=32, d=2 case
    for tag, x, x2, w1, w2, mu, cov in (
            ("f9", np.linspace(-5, 5, 9), np.linspace(-4, 6, 11), np.array([1.3]),
             np.array([2.0]), np.array([0.0]), np.array([[10.0]])),
            ("n32d2", rs.uniform(-2, 2, (2, 32)), rs.uniform(-2, 2, (2, 19)),
             np.array([0.9, 1.2]), np.array([1.7, 2.1]), np.array([0.2, -0.1]),
             np.array([[4.0, 0.6], [0.6, 3.0]]))):
        h1, h2 = 0.2, 15.0
        out["ga_%s_x" % tag], out["ga_%s_x2" % tag] = x, x2
        out["ga_%s_w1" % tag], out["ga_%s_w2" % tag] = w1, w2
        out["ga_%s_mu" % tag], out["ga_%s_cov" % tag] = mu, cov
        out["ga_%s_int_K" % tag] = o.int_K(x, h1, w1, mu, cov)
        out["ga_%s_int_K1_K2" % tag] = o.int_K1_K2(x, x2, h1, w1, h2, w2, mu, cov)
        out["ga_%s_int_int_K1_K2_K1" % tag] = o.int_int_K1_K2_K1(x, h1, w1, h2, w2, mu, cov)
        out["ga_%s_int_int_K1_K2" % tag] = o.int_int_K1_K2(x, h1, w1, h2, w2, mu, cov)
        out["ga_%s_int_int_K" % tag] = o.int_int_K(len(mu), h1, w1, mu, cov)
    np.savez(os.path.join(HERE, "la_ga.npz"), **out)
    print("la_ga.npz: %d arrays" % len(out))


if __name__ == "__main__":
    main()
