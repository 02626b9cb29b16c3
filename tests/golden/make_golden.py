"""Generates tests/golden/c1_n32.npz: the C1 plumbing configuration of
BASELINE.json (N=32 reference-style fixture) evaluated by
the CPU oracle (oracle/bq_oracle.c), which is itself pinned to the reference's
printed known answers by tests/test_oracle_known_answers.py.

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


if __name__ == "__main__":
    main()
